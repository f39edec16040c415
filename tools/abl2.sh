export M4D_GEMM_VARIANT=4
for a in 0 128 384 256; do echo "ABL=$a"; M4D_GEMM_ABL=$a python tools/bench_gemm.py 2>&1 | grep -E "^qkvo|^ffn_down" | sed "s/relerr.*//"; done
