cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_wave_resident.py -x -q -m gpu > gpurun_out/r2_wrtest.log 2>&1; tail -3 gpurun_out/r2_wrtest.log
python tools/kbench.py --frames 8 --iters 20 --tag default --check > gpurun_out/r2_abl.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_abl.log
