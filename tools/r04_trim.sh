#!/bin/bash
# tools/r04_trim.sh -- on the GPU box: the fused kernel with the column products of non-existent tiles skipped (-DFX_TRIM) against the build as it is
for spec in "base|" "trim|-DFX_TRIM"; do
  name=${spec%%|*}; flags=${spec#*|}
  bash tools/variant.sh "fx_$name" "fx_conv_11.hip" "$flags" bash -c "timeout -k 10 300 python tools/fx_dev.py --quirk 8 2>&1 | grep -v amdgpu.ids | grep -E 'engine fused|max diff [2-9]|bad' ; bash tools/fx_fixed_cost.sh 2>&1 | grep -E 'quirk 1|duration'"
done
