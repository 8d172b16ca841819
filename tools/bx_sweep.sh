#!/bin/bash
# tools/bx_sweep.sh "name|ENV=V ENV=V" ... -- on the GPU box: kernel times of fastboxblur 8K k=41 P=3 over the launch knobs of csrc/bx_box.hip
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  name=${v%%|*}; envs=${v#*|}
  OUT=$R/gpurun_out/bxs_$name; rm -rf $OUT; mkdir -p $OUT
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bx_dev.py --time-only > $OUT/run.log 2>&1 )
  python3 - "$name" "$OUT" <<'PY'
import csv, glob, sys
name, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
ks = []
for row in csv.DictReader(open(f[0])) if f else []:
    n = row["Name"]
    if "bx_" in n or "box" in n:
        ks.append("%s %.1f" % ("horz" if "horz" in n else "vert" if "vert" in n else "marg" if "margins" in n else n.split("(")[0][-20:], float(row["AverageNs"]) / 1e3))
t = [l for l in open(out + "/run.log").read().splitlines() if "median" in l]
print(name, sorted(ks), t[-1][29:62] if t else "", flush=True)
PY
done
