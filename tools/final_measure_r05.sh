#!/bin/bash
# Round 5: everything profiles/r05_* is made from, in one GPU session.   tools/final_measure_r05.sh ; then tools/collect_profiles.py r05
# New this round: the same-box A/B of the round's headline change (attn128q_kernel vs the round-4 attention kernel, alternating inside
# bench.py), the clock / power of every bench line, the q64 phase stamps, the N-rank launch check in both layouts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['mfma_frac_whole_step'], d.get('effective_clock_mhz'), d['roofline_attention']['by_class']['self']['frac'])"
# same-box A/B of the round's headline change: M4D_ATTN_Q64=0 = the round-4 self-attention kernel (attn128p_kernel), everything else equal
for rep in 1 2; do for v in 0 1; do printf "M4D_ATTN_Q64=%s " $v; M4D_ATTN_Q64=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-traffic --steps 4 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('steps/s %.4f ms/step %.1f clock %.0f MHz power %.0f W self-attn %.3f gemm %.3f' % (d['value'], d['ms_per_step'], d['clock']['clock_mhz']['mean'], d['clock']['socket_power_w']['mean'], d['roofline_attention']['by_class']['self']['frac'], d['roofline']['frac']))"; done; done > $O/ab_q64_bench.log 2>&1; cat $O/ab_q64_bench.log
timeout 400 python tools/check_q64.py --time > $O/q64_check_time.log 2>&1; grep "^mode\|RESULT" $O/q64_check_time.log | tail -8
M4D_LIB=q64st timeout 200 python tools/q64_stamps.py 2>&1 | grep -v amdgpu.ids > $O/q64_stamps.log; tail -3 $O/q64_stamps.log
timeout 600 python bench.py --mode train --steps 2 --warmup 2 > $O/train.log 2>&1; tail -1 $O/train.log > $O/train_bench.json; tail -c 400 $O/train_bench.json
bash tools/prof.sh > $O/prof.log 2>&1
bash tools/prof_vae.sh > $O/prof_vae.log 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_ks -o p -- python $R/tools/bench_train.py --layers 4 --steps 2 --warmup 1 > $O/train_ks.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_train_ks -o p -- python $R/tools/bench_vae_train.py > $O/vae_train_ks.log 2>&1)
find $O -name "*kernel_trace.csv" -size +5M -delete
for m in cfg-sp sp; do for w in 2 4 8; do timeout 400 python tools/bench_shard.py --world $w --mode $m --steps 2 2>&1 | tail -1; done; done > $O/bench_shard.log 2>&1
M4D_SP_MODE=ulysses timeout 400 python tools/bench_shard.py --world 8 --mode cfg-sp --steps 2 2>&1 | tail -1 >> $O/bench_shard.log; cat $O/bench_shard.log
for p in cfg-sp sp; do timeout 300 python bench.py --gpus 8 --launch-check --parallelism $p 2>&1 | tail -1; done > $O/launch_check_8.log 2>&1; cat $O/launch_check_8.log | cut -c1-300
timeout 600 python tools/ab_attn_bwd.py 2 > $O/ab_attn_bwd.log 2>&1; tail -4 $O/ab_attn_bwd.log
timeout 300 python tools/bench_vae_train.py 2>&1 | tail -1 > $O/vae_train_bench.json
timeout 600 python tools/race_screen.py 30 > $O/race_screen.log 2>&1; tail -2 $O/race_screen.log
du -sh $O
