#!/bin/bash
# tools/r04_strips.sh -- on the GPU box: narrow strips (as built) against full strips (-DFX_FULL_STRIPS), 8 x 4K sigma 20 with the quirk, and 8 x 1080p
for spec in "narrow|" "full|-DFX_FULL_STRIPS" "narrow2|" "full2|-DFX_FULL_STRIPS"; do
  name=${spec%%|*}; flags=${spec#*|}
  bash tools/variant.sh "st_$name" "engine.hip fx_conv_11.hip" "$flags" bash -c "timeout -k 10 200 python tools/fx_dev.py --no-check --fused-only --quirk 8 2>&1 | grep 'engine fused' | tail -1; timeout -k 10 200 python bench.py --config c2 --no-cpu --no-natural --no-copy 2>/dev/null | python -c \"import json,sys;d=json.loads(sys.stdin.read());print('c2',d['value'],d['roofline']['avg_launch_ms'])\""
done
