#!/bin/bash
# tools/one_shape_prof.sh "ROWS COLS SIGMA [FRAMES] [ENGINE]" ... -- on the GPU box: kernel stats of tools/one_shape.py per argument string
cd /tmp && export TMPDIR=/tmp
i=0
for a in "$@"; do
  i=$((i+1)); OUT=$GRAFT_REPO_ROOT/gpurun_out/shape_$i; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/one_shape.py $a > $OUT/run.log 2>&1
  grep "sigma" $OUT/run.log
  python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:6] if f else []:
    print("    %-70s calls %4s avg %9.1f us  %5s %%" % (row["Name"].split("(")[0][-70:], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
done
