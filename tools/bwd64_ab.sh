#!/bin/bash
# Schedule variants of the one-wave-per-SIMD backward kernels as side builds (same-box A/B):
#   tools/bwd64_ab.sh build      (here: needs `python -m more4d_amd.build --ablations`)      -> lib/libmore4d_hip_b64<tag>.so
#   tools/bwd64_ab.sh run        (GPU box)  every variant: parity (tools/check_bwd64.py) + 2 x timing of the whole backward at L = 21 840
R=$(cd "$(dirname "$0")/.." && pwd)
VARS="base:: dqcap4:dq:--cap=4 dqcap6:dq:--cap=6 dqgreedy:dq:--greedy dqah6:dq:--ahead=6 kvcap4:kv:--cap=4 kvcap6:kv:--cap=6 kvah8:kv:--ahead=8 kvfg0:kv:--first-gap=0"
if [ "$1" = build ]; then
    for v in $VARS; do
        tag=${v%%:*}; rest=${v#*:}; which=${rest%%:*}; arg=${rest#*:}
        dq=$R/more4d_amd/csrc/attention_bwd64_dq_gen.inc; kv=$R/more4d_amd/csrc/attention_bwd64_kv_gen.inc
        if [ "$which" = dq ]; then dq=$R/more4d_amd/build/b64_$tag.inc; python $R/tools/gen_attn_bwd64.py $arg -o $dq 2>/dev/null; fi
        if [ "$which" = kv ]; then kv=$R/more4d_amd/build/b64_$tag.inc; python $R/tools/gen_attn_bwd64_kv.py $arg -o $kv 2>/dev/null; fi
        bash $R/tools/side_lib.sh b64$tag attention_bwd.hip "-DM4D_BWD64_DQ_INC=\"$dq\"" "-DM4D_BWD64_KV_INC=\"$kv\"" 2>&1 | grep "^built"
    done
else
    for rep in 1 2; do for v in $VARS; do
        tag=${v%%:*}
        if [ $rep = 1 ]; then M4D_LIB=b64$tag timeout 200 python $R/tools/check_bwd64.py --child 3 check 2>&1 | grep "RESULT" | sed "s/^/$tag /"; fi
        M4D_LIB=b64$tag timeout 200 python $R/tools/check_bwd64.py --child 3 time 2>&1 | grep "^mode" | sed "s/^/$tag /"
    done; done
fi
