#!/bin/bash
# tools/fx_kstats.sh NAME [engine] [quirk] -- on the GPU box: rocprofv3 --kernel-trace --stats of tools/fx_prof.py; prints the per-kernel table
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fxstats_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/fx_prof.py ${2:-fused} ${3:-1} > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print("%-70s calls %5s  avg %10.1f us  total %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
