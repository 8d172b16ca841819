#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s7; rm -rf $O; mkdir -p $O
for sh in "1500 1000 38.73" "2400 1600 48.99" "4425 2950 66.52"; do
  set -- $sh
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$1 -- python3 $GRAFT_REPO_ROOT/tools/one_shape.py $1 $2 $3 1 > $O/k_$1.log 2>&1
  tail -1 $O/k_$1.log
  f=$(find $O/k_$1 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-70s calls %s avg %.1f us  total %.0f us pct %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/7, r["Percentage"]))
PY
  t=$(find $O/k_$1 -name "*kernel_trace.csv" | head -1); python3 - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if "blur_amd" in r["Kernel_Name"] or "anonymous" in r["Kernel_Name"] and "at::native" not in r["Kernel_Name"]]
n=len(rows)//7
last=rows[-n:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print("      %8.1f us +%7.1f  %s grid %s wg %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:50], r.get("Grid_Size","?"), r.get("Workgroup_Size","?")))
PY
done
