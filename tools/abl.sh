for v in 1; do for a in 88 92 72; do echo VARIANT=$v ABL=$a; M4D_GEMM_VARIANT=$v M4D_GEMM_ABL=$a python tools/bench_gemm.py 2>&1 | grep -E "^qkvo|^ffn_down" | sed 's/relerr.*//'; done; done
