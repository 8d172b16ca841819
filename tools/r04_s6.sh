#!/bin/bash
# kernel times of the run-time-planned FFT kernels on sweep sizes (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s6; rm -rf $O; mkdir -p $O
for sh in "11400 7600 106.77" "6000 4000 77.46" "3300 2200 57.45"; do
  set -- $sh
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$1 -- python3 $GRAFT_REPO_ROOT/tools/one_shape.py $1 $2 $3 1 > $O/k_$1.log 2>&1
  tail -1 $O/k_$1.log
  f=$(find $O/k_$1 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-60s calls %s avg %.1f us  pct %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
