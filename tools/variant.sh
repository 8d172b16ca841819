#!/bin/bash
# tools/variant.sh NAME "FILE.hip ..." "EXTRA FLAGS" COMMAND...   (on the GPU box)
# A/B builds that leave the repo's own build alone: the objects of blur_algorithms_amd/csrc/build are COPIED to variants/NAME/, the listed
# translation units are rebuilt there with the extra flags, variants/NAME/libblur_amd.so is linked, and COMMAND runs with
# BLUR_AMD_LIB pointing at it (blur_algorithms_amd/_lib.py).  variants/ is git-ignored.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; FILES=$2; EXTRA=$3; shift 3
SRC=$ROOT/blur_algorithms_amd/csrc
OUT=$ROOT/variants/$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function"
mkdir -p "$OUT"
cp "$SRC"/build/*.o "$OUT"/
for f in $FILES; do
    extra2=""
    [ "$f" = "bx_box.hip" ] && extra2="-mllvm -amdgpu-mfma-vgpr-form=1"
    case "$f" in fx_conv_*) extra2="-mllvm -pragma-unroll-threshold=100000";; esac
    (cd "$SRC" && /opt/rocm/bin/hipcc $FLAGS $extra2 $EXTRA -c "$f" -o "$OUT/${f%.hip}.o") &
done
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o "$OUT/libblur_amd.so" "$OUT"/*.o
echo "=== variant $NAME ($FILES: $EXTRA)"
if [ $# -gt 0 ]; then (cd "$ROOT" && BLUR_AMD_LIB="$OUT/libblur_amd.so" "$@"); fi
