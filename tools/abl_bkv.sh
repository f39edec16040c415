#!/bin/bash
# Timing ablations of the fused dK/dV backward kernel (side builds -DBKV_ABL=n, tools/side_lib.sh): whole m4d_attention_bwd call at
# the train step's self-attention shape; subtract the dQ pass (printed by the kernel-trace run) to read the fused pass alone.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" abl kv1 kv2 kv4 kv6 kv7 kv8 kv16 kv32 kv63; do
  if [ -z "$v" ]; then unset M4D_LIB; else export M4D_LIB=$v; fi
  echo -n "lib=${v:-ship}  "
  AB_CHILD=1 python $R/tools/ab_attn_bwd.py 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms' % d['ms'])"
done
