#!/bin/bash
# tools/r04_band.sh -- on the GPU box: rows per band of the pre-pass (workgroups per CU threshold), kernel stats of 8 x 4K sigma 20 with the quirk
for v in 4 2 1; do
  bash tools/variant.sh "band$v" "engine.hip" "-DFX_BAND_WGS=$v" bash tools/fx_kstats.sh band$v fused 1 2>&1 | grep -E "variant|fx_blur|fx_prepass"
done
