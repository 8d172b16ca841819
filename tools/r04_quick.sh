#!/bin/bash
# quick check: fused engine parity vs the two-kernel engine + timing (quirk on / off), C2 timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick; mkdir -p $O
timeout -k 10 300 python tools/fx_dev.py --quirk 8 > $O/q1.log 2>&1 || { tail -20 $O/q1.log; exit 1; }
grep -c "max diff [01]," $O/q1.log; grep -v "max diff [01]," $O/q1.log | tail -8
timeout -k 10 200 python tools/fx_dev.py --no-check --fused-only 8 2>&1 | grep engine
timeout -k 10 300 python bench.py --no-cpu --no-natural --no-copy --no-configs 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
timeout -k 10 300 python bench.py --config c2 --no-cpu --no-natural --no-copy 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c2', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
