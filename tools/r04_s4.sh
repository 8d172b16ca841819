#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
timeout -k 10 300 python tools/fx_dev.py --quirk 8 > $O/fx_dev_quirk.log 2>&1 || { tail -20 $O/fx_dev_quirk.log; exit 1; }
tail -3 $O/fx_dev_quirk.log
timeout -k 10 400 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
r=json.load(open("gpurun_out/s4/bench.json"))
print({k:r.get(k) for k in ("value","ms_per_step","value_natural","value_cold","instrumented_pass")}); print(r["roofline"]["avg_launch_ms"]); print({k:(v["value"],v["ms_per_step"]) for k,v in r["configs"].items()})
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
