#!/bin/bash
# tools/build_variant.sh NAME ["row N FLAGS T R,R,R" | "col N FLAGS T R,R,R" ...]   (FLAGS: StaticPlan bits, 1 = LDS padding, 2 = late radix-16 twiddles, 4 = pass-0 twiddles in LDS)   [env: VFLAGS="-D..."]   ROLE = row | col
# Builds blur_algorithms_amd/variants/libblur_amd_NAME.so with the given compile-time plans
# (any (role, length) not listed keeps the plan of csrc/fast_ROLE_N.hip).  For A/B runs on the GPU box:
#   BLUR_AMD_LIB=blur_algorithms_amd/variants/libblur_amd_NAME.so python tools/kbench.py
set -e
NAME=$1; shift
CS=/root/repo/blur_algorithms_amd/csrc
BD=$CS/build_$NAME
mkdir -p $BD /root/repo/blur_algorithms_amd/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -I$CS $VFLAGS"
for f in $CS/fast_*.hip; do cp $f $BD/; done
for spec in "$@"; do set -- $spec
U=$(echo $1 | tr a-z A-Z)
cat > $BD/fast_$1_$2.hip <<EOT
#include "fast_kernels.hpp"
BLUR_FAST_$U($2, $3, $4, $5)
EOT
done
pids=""
for f in $BD/fast_*.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o ${f%.hip}.o & pids="$pids $!"; done
/opt/rocm/bin/hipcc $FLAGS -c $CS/engine.hip -o $BD/engine.o & pids="$pids $!"
/opt/rocm/bin/hipcc $FLAGS -x c++ -c $CS/host_math.cpp -o $BD/host_math.o & pids="$pids $!"
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc $FLAGS -shared -o /root/repo/blur_algorithms_amd/variants/libblur_amd_$NAME.so $BD/*.o
rm -rf $BD
echo built variants/libblur_amd_$NAME.so
