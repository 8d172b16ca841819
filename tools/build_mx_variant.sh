#!/bin/bash
# tools/build_mx_variant.sh NAME   [env: VFLAGS="-DMX_..."]
# Builds blur_algorithms_amd/variants/libblur_amd_NAME.so: the matrix-core translation unit of the metric's kernel
# (mx_conv_11.hip) recompiled with VFLAGS, everything else taken from csrc/build (run `make -C blur_algorithms_amd/csrc` first).
#   BLUR_AMD_LIB=blur_algorithms_amd/variants/libblur_amd_NAME.so python tools/mx_bench.py
set -e
NAME=$1
CS=/root/repo/blur_algorithms_amd/csrc
BD=$CS/build_$NAME
mkdir -p $BD /root/repo/blur_algorithms_amd/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -I$CS $VFLAGS"
/opt/rocm/bin/hipcc $FLAGS -c $CS/mx_conv_11.hip -o $BD/mx_conv_11.o
OTHERS=$(ls $CS/build/*.o | grep -v mx_conv_11.o)
/opt/rocm/bin/hipcc $FLAGS -shared -o /root/repo/blur_algorithms_amd/variants/libblur_amd_$NAME.so $BD/mx_conv_11.o $OTHERS
rm -rf $BD
echo built variants/libblur_amd_$NAME.so
