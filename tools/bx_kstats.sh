#!/bin/bash
# tools/bx_kstats.sh NAME -- on the GPU box: rocprofv3 kernel stats of tools/bx_dev.py --time-only (fastboxblur 8K k=41 P=3)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bxk_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bx_dev.py --time-only > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -c1-200 "$f" | sed -n 1,12p
tail -1 $OUT/run.log
