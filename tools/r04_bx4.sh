#!/bin/bash
# tools/r04_bx4.sh "name|flags" ... -- on the GPU box: fastboxblur bytes with the repo's build, the box tests, then kernel stats at 8K per variant of bx_box.hip
O=gpurun_out/bx; mkdir -p $O
timeout -k 10 300 python tools/bx_dev.py --shapes "640,480,3,41,3;333,517,3,41,3;335,200,3,41,3;1001,300,3,113,2;130,90,3,9,1;46,40,3,9,3;256,300,1,9,2;128,200,4,15,1;1920,1080,3,41,3;7680,4320,3,41,3" > $O/dev.log 2>&1 || { tail -20 $O/dev.log; exit 1; }
grep -v amdgpu.ids $O/dev.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "boxblur or cpp_surface" > $O/tests.log 2>&1; tail -2 $O/tests.log
for spec in "$@"; do
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
