#!/bin/bash
# tools/r04_rb.sh -- on the GPU box: rows of loads in flight in the pre-pass (kernel stats, 8 x 4K and 8 x 1080p, sigma 20, quirk on)
for v in 4 8; do
  bash tools/variant.sh "rb$v" "engine.hip" "-DFX_PRE_RB=$v" bash -c "bash tools/fx_kstats.sh rb$v fused 1 2>&1 | grep -E 'fx_prepass'; timeout -k 10 200 python bench.py --config c2 --no-cpu --no-natural --no-copy 2>/dev/null | python -c \"import json,sys;d=json.loads(sys.stdin.read());print('c2',d['value'],d['roofline']['avg_launch_ms'])\"; timeout -k 10 200 python bench.py --no-cpu --no-natural --no-copy --no-configs 2>/dev/null | python -c \"import json,sys;d=json.loads(sys.stdin.read());print('metric',d['value'],d['roofline']['avg_launch_ms'])\""
done
