#!/bin/bash
# tools/fx_fixed_cost.sh -- on the GPU box: kernel trace of tools/fx_fixed_cost.py and the fit
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fx_fixed
rm -rf $OUT; mkdir -p $OUT
BLUR_AMD_LIB=$BLUR_AMD_LIB rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/fx_fixed_cost.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/fx_fixed_cost.py --parse $OUT
