#!/bin/bash
# same-box A/B of attn128q_kernel schedule variants (side builds tools/side_lib.sh q64<tag> ... -DM4D_Q64_INC=...): tools/q64_ab.sh tag...
for rnd in 1 2; do for tag in "$@"; do
  printf "%-8s " $tag; M4D_LIB=q64$tag timeout 120 python tools/check_q64.py --child 1 time 2>&1 | grep "^mode"
done; done
