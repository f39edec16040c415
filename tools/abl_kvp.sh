#!/bin/bash
# Timing ablations / tuning variants of the phased fused dK / dV kernel (attention_bwd_kvp.h): side builds (tools/side_lib.sh), whole
# m4d_attention_bwd call at the train step's self-attention shape on ONE box.
#   tools/abl_kvp.sh build   (here, no GPU)      tools/abl_kvp.sh run   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
VARIANTS="kp1:-DKVP_ABL=1 kp4:-DKVP_ABL=4 kp6:-DKVP_ABL=6 kp8:-DKVP_ABL=8 kp16:-DKVP_ABL=16 kp29:-DKVP_ABL=29 kp31:-DKVP_ABL=31 rd4:-DKVP_RD=4 prio0:-DKVP_PRIO=0 stamps:-DKVP_STAMPS=1"
if [ "$1" = "build" ]; then
  for v in $VARIANTS; do tag=${v%%:*}; fl=${v#*:}; $R/tools/side_lib.sh $tag attention_bwd.hip $fl 2>&1 | grep -E "built|error" ; done
  exit 0
fi
for v in ship $VARIANTS ship; do [ "${v%%:*}" = stamps ] && continue
  tag=${v%%:*}
  if [ "$tag" = "ship" ]; then unset M4D_LIB; else export M4D_LIB=$tag; fi
  echo -n "lib=$tag  "
  AB_CHILD=1 python $R/tools/ab_attn_bwd.py 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms' % d['ms'])"
done
