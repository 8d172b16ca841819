#!/bin/bash
# tools/r04_bx2.sh -- on the GPU box: fastboxblur bytes, then kernel time of the horizontal kernel per row-segment setting
O=gpurun_out/bx; mkdir -p $O
timeout -k 10 300 python tools/bx_dev.py --shapes "640,480,3,41,3;333,517,3,41,3;335,200,3,41,3;1001,300,3,113,2;130,90,3,9,1;46,40,3,9,3;1920,1080,3,41,3;7680,4320,3,41,3" > $O/dev.log 2>&1 || { tail -20 $O/dev.log; exit 1; }
grep -v amdgpu.ids $O/dev.log
for w in 6 8 10 12 14 16 20 24; do echo "HWAVES $w"; BLUR_BX_HWAVES=$w bash tools/bx_kstats.sh hw$w 2>&1 | grep -E "horz|vert|fastboxblur" | cut -d, -f1,4 | sed 's/(unsigned char[^"]*"/"/'; done
