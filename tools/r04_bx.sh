#!/bin/bash
# tools/r04_bx.sh -- on the GPU box: fastboxblur bytes on a list of shapes (incl. widths that are not multiples of 4), the tests, kernel stats, row-segment sweep
O=gpurun_out/bx; mkdir -p $O
timeout -k 10 300 python tools/bx_dev.py --shapes "640,480,3,41,3;333,517,3,41,3;335,200,3,41,3;1001,300,3,113,2;130,90,3,9,1;1920,1080,3,41,3;1000,700,3,49,3;600,900,3,65,2;512,400,3,3,3;7680,4320,3,41,3" > $O/dev.log 2>&1 || { tail -20 $O/dev.log; exit 1; }
grep -v amdgpu.ids $O/dev.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "boxblur or cpp_surface" > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/bx_kstats.sh new | cut -c1-160
for w in 4 6 8 10 12 16 24; do echo "HWAVES $w"; BLUR_BX_HWAVES=$w timeout -k 10 100 python tools/bx_dev.py --time-only 2>&1 | tail -1; done
