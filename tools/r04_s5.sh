#!/bin/bash
# round-4 GPU session 5: any width / alignment on the fused kernels; wide fused kernel on the sweep's small sizes and C3
cd $GRAFT_REPO_ROOT
O=gpurun_out/s5; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/unaligned_probe tools/probes/unaligned_probe.hip 2>/dev/null && timeout -k 5 60 /tmp/unaligned_probe | tee $O/unaligned_probe.txt
timeout -k 10 900 python -m pytest tests/test_gpu_fused_engine.py tests/test_gpu_matrix_engine.py -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout -k 10 300 python bench.py --config c3 --engine fused --no-cpu --no-natural --no-copy 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c3 fused', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
timeout -k 10 300 python bench.py --config c3 --no-cpu --no-natural --no-copy 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c3 auto', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
timeout -k 10 300 python tools/one_shape.py 1500 1000 38.73 1 2>&1 | tail -8
for eng in fused matrix fft; do for sh in "1500 1000 38.73" "1950 1300 44.16" "2400 1600 48.99"; do timeout -k 10 100 python tools/one_shape.py $sh 1 $eng 2>&1 | tail -1; done; done
