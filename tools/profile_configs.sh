#!/bin/bash
# tools/profile_configs.sh ROUND -- on the GPU box: bench.py --config c2 / c3 / c5 under rocprofv3 --kernel-trace --stats, and the same
# three commands without the profiler for their JSON lines; raw output under gpurun_out/prof_ROUNDcfg/ (copy the summaries into profiles/)
R=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${R}cfg
rm -rf $OUT; mkdir -p $OUT
for c in c2 c3 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu --no-natural --no-copy --steps 10 --warmup 3 > $OUT/${c}_under_rocprof.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu > $OUT/$c.json 2> $OUT/$c.err || exit 1
  cut -c1-400 $OUT/$c.json
done
