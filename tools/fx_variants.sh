#!/bin/bash
# tools/fx_variants.sh "name|flags" ... -- on the GPU box: rebuild fx_conv_11.o with extra flags, relink, time the fused engine.
# The last variant built stays in the (scratch) .so; the repo's own build is untouched.
cd $GRAFT_REPO_ROOT/blur_algorithms_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function"
for spec in "$@"; do
    name=${spec%%|*}; flags=${spec#*|}
    /opt/rocm/bin/hipcc $FLAGS $flags -c fx_conv_11.hip -o build/fx_conv_11.o || exit 1
    /opt/rocm/bin/hipcc $FLAGS -shared -o ../libblur_amd.so build/*.o || exit 1
    echo "=== variant $name ($flags)"
    (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/fx_dev.py $FX_DEV_ARGS 2>&1 | grep -v amdgpu.ids)
done
