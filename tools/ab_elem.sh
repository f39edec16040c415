# A/B of the HBM-bound DiT kernels at the bench shape: the LN-modulate variants (M4D_LN_VAR) and the two RMSNorm + RoPE kernels
for v in 0 5; do echo "M4D_LN_VAR=$v"; M4D_LN_VAR=$v python tools/bench_elem.py 2>&1 | grep ln_modulate; done
for v in 1 0; do echo "M4D_RMS_ROWS=$v"; M4D_RMS_ROWS=$v python tools/bench_elem.py 2>&1 | grep rmsnorm; done
