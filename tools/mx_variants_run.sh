# tools/mx_variants_run.sh NAME... -- on the GPU box: tools/mx_bench.py -m with the default library, then with each variant
echo "== default"; python tools/mx_bench.py -m
for v in "$@"; do echo "== $v"; BLUR_AMD_LIB=blur_algorithms_amd/variants/libblur_amd_$v.so python tools/mx_bench.py -m; done
