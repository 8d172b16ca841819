cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_wave_resident.py -x -q -m gpu > gpurun_out/r2_wrtest.log 2>&1; tail -5 gpurun_out/r2_wrtest.log
for wr in on off; do
python tools/kbench.py --rows 1080 --cols 1920 --frames 8 --iters 20 --tag 1080p_wr_$wr --wave-resident $wr 2>&1 | grep -v amdgpu.ids
python tools/kbench.py --rows 1500 --cols 1000 --sigma 38.73 --frames 8 --iters 10 --tag sweep1_wr_$wr --wave-resident $wr 2>&1 | grep -v amdgpu.ids
python tools/kbench.py --rows 1950 --cols 1300 --sigma 44.16 --frames 8 --iters 10 --tag sweep3_wr_$wr --wave-resident $wr 2>&1 | grep -v amdgpu.ids
done
