python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm 2>&1 | tail -2
python tools/bench_gemm.py 2>&1 | grep -E "^qkvo|^ffn_up|^ffn_down|^v_t" | sed 's/relerr.*//'
