#!/bin/bash
# round-4 GPU session 1: quick parity of the fused engine, default bench, staging-map A/B with phase stamps, the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/s1; mkdir -p $O
timeout -k 10 300 python tools/fx_dev.py --quirk 8 > $O/fx_dev_quirk.log 2>&1 || { tail -20 $O/fx_dev_quirk.log; exit 1; }
tail -4 $O/fx_dev_quirk.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-1500 $O/bench.json
export BLUR_FX_STAMPS=1 FX_DEV_ARGS="--no-check --fused-only 8"
tools/fx_variants.sh "stamps_new|-DFX_STAMPS" "stamps_linear|-DFX_STAMPS -DFX_STAGE_LINEAR" > $O/variants.log 2>&1 || { tail -20 $O/variants.log; exit 1; }
unset BLUR_FX_STAMPS
export FX_DEV_ARGS="--no-check --fused-only --quirk 8"
tools/fx_variants.sh "linear|-DFX_STAGE_LINEAR" "new|-DFX_DUMMY" >> $O/variants.log 2>&1
grep -E "variant|stamps|engine" $O/variants.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
