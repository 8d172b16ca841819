#!/bin/bash
# tools/build_wr_variant.sh NAME   [env: VFLAGS="-DWR_STAMPS ..."]
# Builds blur_algorithms_amd/variants/libblur_amd_NAME.so: engine.hip and the wave-resident translation units (wr_*.hip)
# recompiled with VFLAGS, everything else taken from csrc/build (run `make -C blur_algorithms_amd/csrc` first).
#   BLUR_AMD_LIB=blur_algorithms_amd/variants/libblur_amd_NAME.so python tools/kbench.py
set -e
NAME=$1
CS=/root/repo/blur_algorithms_amd/csrc
BD=$CS/build_$NAME
mkdir -p $BD /root/repo/blur_algorithms_amd/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -I$CS $VFLAGS"
pids=""
for f in $CS/wr_*.hip $CS/engine.hip; do b=$(basename ${f%.hip}); /opt/rocm/bin/hipcc $FLAGS -c $f -o $BD/$b.o & pids="$pids $!"; done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc $FLAGS -shared -o /root/repo/blur_algorithms_amd/variants/libblur_amd_$NAME.so $BD/*.o $CS/build/fast_*.o $CS/build/host_math.o
rm -rf $BD
echo built variants/libblur_amd_$NAME.so
