#!/bin/bash
# tools/fw_variants.sh "name|flags" ... -- on the GPU box: the wide fused kernels (fw_conv_*.hip) rebuilt with extra flags as a scratch
# library under variants/ (tools/variant.sh; the repo's own build is not touched), timed with tools/fw_dev.py --no-check $FW_DEV_ARGS
for spec in "$@"; do
    name=${spec%%|*}; flags=${spec#*|}
    "$(dirname "$0")"/variant.sh "fw_$name" "fw_conv_13.hip fw_conv_15.hip fw_conv_17.hip fw_conv_19.hip fw_conv_21.hip fw_conv_23.hip" "$flags" \
        bash -c "timeout -k 10 200 python tools/fw_dev.py --no-check $FW_DEV_ARGS 2>&1 | grep -v amdgpu.ids" || exit 1
done
