#!/bin/bash
# tools/fx_variants.sh "name|flags" ... -- on the GPU box: the fused kernel (fx_conv_11.hip) rebuilt with extra flags as a scratch
# library under variants/ (tools/variant.sh; the repo's own build is not touched), timed with tools/fx_dev.py $FX_DEV_ARGS
for spec in "$@"; do
    name=${spec%%|*}; flags=${spec#*|}
    "$(dirname "$0")"/variant.sh "fx_$name" "fx_conv_11.hip" "$flags" bash -c "timeout -k 10 200 python tools/fx_dev.py $FX_DEV_ARGS 2>&1 | grep -v amdgpu.ids" || exit 1
done
