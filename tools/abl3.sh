for a in 0 1 2 3; do echo "ATTN_ABL=$a"; M4D_ATTN_ABL=$a python tools/bench_attn.py 2>&1 | grep "^self" | sed "s/relerr.*//"; done
