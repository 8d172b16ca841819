#!/bin/bash
# tools/pmc_any.sh NAME PATTERN python-script [args...] -- on the GPU box: SQ counter passes (separate --pmc runs, never combined with a
# trace) of any script; prints per kernel matching PATTERN the mean of each counter per dispatch and writes gpurun_out/pmc_NAME.json
NAME=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcany_$NAME
rm -rf $OUT; mkdir -p $OUT
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $SCRIPT "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 $SCRIPT "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/c -- python3 $SCRIPT "$@" > $OUT/c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $SCRIPT "$@" > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $SCRIPT "$@" > $OUT/w.log 2>&1
python3 - "$OUT" "$PAT" "$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME.json" <<'PY'
import csv, glob, collections, json, re, sys
out, pat, dst = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        m = re.search(pat, k)
        if not m: continue
        acc[m.group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    m = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in m: m[c.lower() + "_frac_of_wave_cycles"] = round(m[c] / wc, 4)
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES"): m["mfma_busy_frac_of_4x_wave_cycles"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc), 4)
    if "FETCH_SIZE" in m: m["fetch_bytes_x2_gfx950"] = round(m["FETCH_SIZE"] * 1024 * 2, 1)       # KB -> bytes, wide reads counted at half (MI355X_MICROARCH.md)
    if "WRITE_SIZE" in m: m["write_bytes"] = round(m["WRITE_SIZE"] * 1024, 1)
    res[k] = m
json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
for k, m in res.items():
    print(k, {c: v for c, v in m.items() if "frac" in c or "bytes" in c})
PY
