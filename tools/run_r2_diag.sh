cd $GRAFT_REPO_ROOT
V=blur_algorithms_amd/variants
python -m pytest tests/test_gpu_wave_resident.py -x -q -m gpu > gpurun_out/r2_wrtest.log 2>&1; tail -3 gpurun_out/r2_wrtest.log
python tools/kbench.py --frames 8 --iters 10 --tag default --check > gpurun_out/r2_abl.log 2>&1
BLUR_AMD_LIB=$V/libblur_amd_stamps.so python tools/wr_stamps.py 8 > gpurun_out/r2_stamps.log 2>&1
cat gpurun_out/r2_abl.log gpurun_out/r2_stamps.log
