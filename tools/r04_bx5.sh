#!/bin/bash
# tools/r04_bx5.sh -- on the GPU box: vertical box kernel, strip width and segments (kernel stats at 8K)
for cfg in "4 0" "4 8" "4 9" "4 12" "8 0" "8 10" "8 15" "8 20"; do
  set -- $cfg
  BLUR_BX_VNT=$1 BLUR_BX_VSEG=$2 bash tools/bx_kstats.sh vv > /dev/null 2>&1
  python3 - "$cfg" <<'PY'
import csv, glob, os, sys
fs = glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/bxk_vv/**/*_kernel_stats.csv", recursive=True)
f = max(fs, key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if "bx_vert" in r["Name"]: print("VNT VSEG %-6s %-40s avg %.1f us min %.1f" % (sys.argv[1], r["Name"].split("(anonymous namespace)::")[1][:30], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
