#!/bin/bash
# tools/power_probe.sh -- sample rocm-smi power / clocks while the bench loop runs (GPU box; diagnostic)
cd $GRAFT_REPO_ROOT
python bench.py --steps 20000 --warmup 20 --no-cpu > gpurun_out/power_bench.json 2>gpurun_out/power_bench.err &
BP=$!
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -e 's/.*: //' | tr '\n' ' '
  echo
  kill -0 $BP 2>/dev/null || break
  sleep 1
done
wait $BP
cat gpurun_out/power_bench.json | cut -c1-330
