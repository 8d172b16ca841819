#!/bin/bash
# tools/fw_variants.sh "name|flags" ... -- on the GPU box: rebuild the wide fused kernels (fw_conv_*.hip) with extra flags, relink, time them
# (tools/fw_dev.py --no-check).  The last variant built stays in the (scratch) .so; the repo's own build is untouched.
cd $GRAFT_REPO_ROOT/blur_algorithms_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function"
for spec in "$@"; do
    name=${spec%%|*}; flags=${spec#*|}
    for n in 13 15 17 19 21 23; do /opt/rocm/bin/hipcc $FLAGS $flags -c fw_conv_$n.hip -o build/fw_conv_$n.o & done; wait
    /opt/rocm/bin/hipcc $FLAGS -shared -o ../libblur_amd.so build/*.o || exit 1
    echo "=== variant $name ($flags)"
    (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/fw_dev.py --no-check $FW_DEV_ARGS 2>&1 | grep -v amdgpu.ids)
done
