#!/bin/bash
# tools/r04_bx3.sh -- on the GPU box: prefetch depth variants of the box kernels (kernel stats at 8K)
for spec in "base|" "pf8|-DBX_PF=8" "vpf4|-DBX_VPF=4" "both|-DBX_PF=8 -DBX_VPF=4"; do
  name=${spec%%|*}; flags=${spec#*|}
  bash tools/variant.sh "bx_$name" "bx_box.hip" "$flags" bash tools/bx_kstats.sh v_$name 2>&1 | grep -E "variant|fastboxblur 8K"
  python3 - $name <<'PY'
import csv, glob, os, sys
fs = glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/bxk_v_%s/**/*_kernel_stats.csv" % sys.argv[1], recursive=True)
f = max(fs, key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if "bx_" in r["Name"]: print("   %-40s avg %.1f us min %.1f" % (r["Name"].split("(anonymous namespace)::")[1][:40], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
