#!/bin/bash
# tools/mx_profile.sh NAME -- on the GPU box: rocprofv3 kernel-trace stats of tools/mx_bench.py -> gpurun_out/mxprof_NAME/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mxprof_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/tools/mx_bench.py > $OUT/bench.log 2>&1
cat $OUT/bench.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | head -20
