#!/bin/bash
# cycle budget of attn128q_kernel's steady state: phase stamps of timing-ablation builds (tools/gen_attn_q64.py --stamps --abl <letters>,
# side builds q64a<letters>; results are wrong by construction): what each class of instructions costs per 32-MFMA phase
for tag in "$@"; do
  printf "%-10s " $tag; M4D_LIB=q64$tag ATTN_ITERS=3 timeout 120 python tools/q64_stamps.py 2>&1 | grep -v amdgpu | awk '/^ 10[2-6]/{a+=$2; b+=$3; n++} END{if(n) printf "A %.0f  B %.0f  per MFMA %.2f\n", a/n, b/n, (a+b)/n/64; else print "no stamps"}'
done
