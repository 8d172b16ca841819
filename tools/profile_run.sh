#!/bin/bash
# tools/profile_run.sh ROUND -- on the GPU box: kernel-trace stats and the HBM-traffic PMC passes of
# the default bench.py command; summaries land in gpurun_out/prof_ROUND/ (copy into profiles/).
# Each --pmc set is its own run, never combined with a trace (MI355X_MICROARCH.md, rocprofv3 section).
R=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-events > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-events > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-events > $OUT/pmc_l2.log 2>&1
find $OUT -name "*.csv" | head -20
