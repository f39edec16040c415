#!/bin/bash
# Round 6: everything profiles/r06_* is made from, in one GPU session.   tools/final_measure_r06.sh ; then tools/collect_profiles.py r06
# New this round: the same-box A/B of the round's headline change (one-wave-per-SIMD attention backward, M4D_ATTN_BWD64=0 vs default)
# inside bench.py --mode train and alone, the GEMM re-fetch A/B with clock and power (tools/ab_gemm_refetch.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 1200 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); s=d['secondary']; print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['mfma_frac_whole_step'], d.get('effective_clock_mhz'), d['roofline_attention']['by_class']['self']['frac'], 'vae', s['vae_roundtrip'].get('ms'), 'bwd', s['roofline_attention_bwd']['frac'], 'train', s['train_step'].get('s_per_step'))"
# same-box A/B of the round's headline change: M4D_ATTN_BWD64=0 = the two-waves-per-SIMD backward kernels of rounds 3-4
for rep in 1 2; do for v in 0 3; do printf "M4D_ATTN_BWD64=%s " $v; M4D_ATTN_BWD64=$v timeout 600 python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('s/step %.4f mfma %.3f' % (d['ms_per_step'] / 1e3, d['mfma_frac_whole_step']))"; done; done > $O/ab_bwd64_train.log 2>&1; cat $O/ab_bwd64_train.log
timeout 600 python tools/check_bwd64.py --time > $O/bwd64_check_time.log 2>&1; grep "^mode\|RESULT\|new vs old" $O/bwd64_check_time.log | tail -14
timeout 600 python tools/ab_gemm_refetch.py > $O/ab_gemm_refetch.log 2>&1; grep "^abl" $O/ab_gemm_refetch.log
timeout 600 python bench.py --mode train --steps 3 --warmup 2 > $O/train.log 2>&1; tail -1 $O/train.log > $O/train_bench.json; tail -c 300 $O/train_bench.json
bash tools/prof.sh > $O/prof.log 2>&1
bash tools/prof_vae.sh > $O/prof_vae.log 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_ks -o p -- python $R/tools/bench_train.py --layers 4 --steps 2 --warmup 1 > $O/train_ks.log 2>&1)
# MFMA-busy counters of the backward kernels (new and old) and of the VAE's conv kernels, each in its own PMC pass
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_bwd64 -o p -- python $R/tools/check_bwd64.py --child 3 time > $O/pmc_bwd64.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_bwd0 -o p -- python $R/tools/check_bwd64.py --child 0 time > $O/pmc_bwd0.log 2>&1)
{ echo "# one-wave-per-SIMD backward (M4D_ATTN_BWD64=3)"; python tools/pmc_busy.py $(find $O/pmc_bwd64 -name "*counter_collection.csv") attn_bwd_kv64 attn_bwd_dq64 attn128q attn_delta; echo; echo "# two-waves-per-SIMD backward (M4D_ATTN_BWD64=0)"; python tools/pmc_busy.py $(find $O/pmc_bwd0 -name "*counter_collection.csv") attn_bwd_kvp attn_bwd_dqp attn128q; } > $O/bwd64_pmc_summary.md 2>&1; cat $O/bwd64_pmc_summary.md
timeout 300 python -m pytest tests/test_round5_gpu.py -q -s -k "forty" 2>&1 | grep -E "stack40|depth40|passed|failed" > $O/depth40_distance.log; cat $O/depth40_distance.log
timeout 900 python tools/check_conv64.py --time > $O/conv64_check_time.log 2>&1; grep "^halo64=\|RESULT\|DIFFERENT\|differs" $O/conv64_check_time.log | tail -20; grep -c "bit-identical" $O/conv64_check_time.log; bash tools/ab_conv_narrow.sh > $O/ab_conv_narrow.log 2>&1; cut -c1-150 $O/ab_conv_narrow.log
bash tools/ab_conv_tiled.sh > $O/ab_conv_tiled.log 2>&1; grep "^==\|roundtrip" $O/ab_conv_tiled.log | cut -c1-200
bash tools/ab_sp_overlap.sh > $O/ab_sp_overlap.log 2>&1; cut -c1-200 $O/ab_sp_overlap.log
timeout 600 python tools/ab_gemm_tail.py 2>&1 | grep -v amdgpu.ids > $O/ab_gemm_tail.log; tail -15 $O/ab_gemm_tail.log
find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*counter_collection.csv" -size +5M -delete
{ timeout 400 python tools/bench_shard.py --world 1 --mode sp --steps 2 2>&1 | tail -1; for m in cfg-sp sp; do for w in 2 4 8; do timeout 400 python tools/bench_shard.py --world $w --mode $m --steps 2 2>&1 | tail -1; done; done; } > $O/bench_shard.log 2>&1
M4D_SP_MODE=ulysses timeout 400 python tools/bench_shard.py --world 8 --mode cfg-sp --steps 2 2>&1 | tail -1 >> $O/bench_shard.log; cat $O/bench_shard.log
for p in cfg-sp sp; do timeout 300 python bench.py --gpus 8 --launch-check --parallelism $p 2>&1 | tail -1; done > $O/launch_check_8.log 2>&1; cat $O/launch_check_8.log | cut -c1-300
timeout 300 python tools/bench_vae_train.py 2>&1 | tail -1 > $O/vae_train_bench.json
timeout 600 python tools/race_screen.py 20 > $O/race_screen.log 2>&1; tail -2 $O/race_screen.log
du -sh $O
