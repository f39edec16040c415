#!/bin/bash
# Round 4: everything profiles/r04_* is made from, in one GPU session (the GEMM / conv / forward-attention timelines, probes and the
# race screen of tools/final_measure.sh belong to kernels this round did not touch: profiles/r03_*).   tools/final_measure_r04.sh ;
# then tools/collect_profiles.py r04
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['mfma_frac_whole_step'])"
timeout 600 python bench.py --mode train --steps 2 --warmup 2 > $O/train.log 2>&1; tail -1 $O/train.log > $O/train_bench.json; tail -c 400 $O/train_bench.json
bash tools/prof.sh > $O/prof.log 2>&1
bash tools/prof_vae.sh > $O/prof_vae.log 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_ks -o p -- python $R/tools/bench_train.py --layers 4 --steps 2 --warmup 1 > $O/train_ks.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_train_ks -o p -- python $R/tools/bench_vae_train.py > $O/vae_train_ks.log 2>&1)
find $O -name "*kernel_trace.csv" -size +5M -delete
for w in 2 4 8; do timeout 400 python tools/bench_shard.py --world $w --mode cfg-sp --steps 2 2>&1 | tail -1; done > $O/bench_shard.log 2>&1; cat $O/bench_shard.log
timeout 600 python tools/ab_attn_bwd.py 2 > $O/ab_attn_bwd.log 2>&1; cat $O/ab_attn_bwd.log
timeout 300 python tools/bench_vae_train.py 2>&1 | tail -1 > $O/vae_train_bench.json
timeout 200 python tools/bench_gn_planar.py 17 > $O/bench_gn_planar.log 2>&1; timeout 200 python tools/bench_gn_planar.py 49 >> $O/bench_gn_planar.log 2>&1
du -sh $O
