for v in "" gl w2; do
  if [ -z "$v" ]; then echo "== default"; python tools/mx_bench.py -m; else echo "== $v"; BLUR_AMD_LIB=blur_algorithms_amd/variants/libblur_amd_$v.so python tools/mx_bench.py -m; fi
done
