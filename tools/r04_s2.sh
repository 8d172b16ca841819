#!/bin/bash
# round-4 GPU session 2: balanced phases (staging + stores in phase B) A/B with stamps; quirk prologue; fused-engine tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; mkdir -p $O
timeout -k 10 300 python tools/fx_dev.py --quirk 8 > $O/fx_dev_quirk.log 2>&1 || { tail -20 $O/fx_dev_quirk.log; exit 1; }
tail -24 $O/fx_dev_quirk.log
export BLUR_FX_STAMPS=1 FX_DEV_ARGS="--no-check --fused-only 8"
tools/fx_variants.sh "stamps_bal|-DFX_STAMPS" "stamps_nobal|-DFX_STAMPS -DFX_NO_BALANCE" > $O/variants.log 2>&1 || { tail -20 $O/variants.log; exit 1; }
unset BLUR_FX_STAMPS
export FX_DEV_ARGS="--no-check --fused-only --quirk 8"
tools/fx_variants.sh "nobal|-DFX_NO_BALANCE" >> $O/variants.log 2>&1
grep -E "variant|engine" $O/variants.log; grep stamps $O/variants.log | sort | uniq -c | sort -rn | awk 'NR<=1 || /A0/' | head -4
grep -n "stamps" $O/variants.log | awk -F: 'NR%25==1{print $0}' | cut -c1-200
timeout -k 10 400 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
r=json.load(open("gpurun_out/s2/bench.json"))
print({k:r.get(k) for k in ("value","ms_per_step","value_natural","value_cold","instrumented_pass")}); print(r["roofline"]["avg_launch_ms"]); print({k:(v["value"],v["ms_per_step"]) for k,v in r["configs"].items()})
PY
timeout -k 10 600 python -m pytest tests/test_gpu_fused_engine.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
