#!/bin/bash
# tools/fx_pmc.sh NAME [engine] [quirk] [sigma] -- on the GPU box: SQ counter passes (separate --pmc runs, never combined with a trace) of tools/fx_prof.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fxpmc_$1
rm -rf $OUT; mkdir -p $OUT
ENG=${2:-fused}; Q=${3:-0}; SG=${4:-20}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/fx_prof.py $ENG $Q $SG > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/fx_prof.py $ENG $Q $SG > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $GRAFT_REPO_ROOT/tools/fx_prof.py $ENG $Q $SG > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "fx_" not in k and "mx_" not in k and "fw_" not in k: continue
        k = k.split("(")[0][-60:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for c in sorted(m): print("   %-28s %14.0f  %6.3f of wave cycles" % (c, m[c], m[c] / wc))
PY
