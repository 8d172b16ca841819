cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fastboxblur" > gpurun_out/r2_boxtest.log 2>&1; tail -3 gpurun_out/r2_boxtest.log
python tools/boxbench.py 2>&1 | grep -v amdgpu.ids
BLUR_BOX_UNFUSED=1 python tools/boxbench.py 2>&1 | grep -v amdgpu.ids
python tools/boxbench.py 2>&1 | grep -v amdgpu.ids
