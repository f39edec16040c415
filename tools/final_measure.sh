#!/bin/bash
# Everything profiles/<tag>_* is made from, in one GPU session: tools/final_measure.sh   (then tools/collect_profiles.py <tag>)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['mfma_frac_whole_step'])"
timeout 600 python bench.py --mode train --steps 2 --warmup 1 > $O/train.log 2>&1; tail -1 $O/train.log > $O/train_bench.json; tail -c 400 $O/train_bench.json
bash tools/prof.sh > $O/prof.log 2>&1
bash tools/prof_vae.sh > $O/prof_vae.log 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_ks -o p -- python $R/tools/bench_train.py --layers 4 --steps 2 --warmup 1 > $O/train_ks.log 2>&1)
find $O -name "*kernel_trace.csv" -size +5M -delete
for w in 2 4 8; do timeout 400 python tools/bench_shard.py --world $w --mode cfg-sp --steps 2 2>&1 | tail -1; done > $O/bench_shard.log 2>&1; cat $O/bench_shard.log
M4D_LIB=abl M4D_GEMM_ABL=64 timeout 200 python tools/gemm_timeline.py 43680 5120 5120 > $O/gemm_timeline.log 2>&1
M4D_LIB=abl M4D_GEMM_ABL=64 timeout 200 python tools/gemm_timeline.py 43680 5120 13824 >> $O/gemm_timeline.log 2>&1
for a in 65 66 71; do M4D_LIB=abl M4D_GEMM_ABL=$a timeout 200 python tools/gemm_timeline.py 43680 5120 5120 >> $O/gemm_timeline.log 2>&1; done
for m in plain planar norm normresid; do M4D_LIB=abl M4D_CONV_ABL=64 timeout 120 python tools/conv_timeline.py 0 $m >> $O/conv_timeline.log 2>&1; done
M4D_LIB=abl M4D_CONV_ABL=64 timeout 120 python tools/conv_timeline.py 5 planar >> $O/conv_timeline.log 2>&1
for a in 64 65 66; do M4D_LIB=abl M4D_ATTN_ABL=$a timeout 200 python tools/attn_clock.py 2>&1 | grep -v amdgpu >> $O/attn_clock.log; done
# per-phase stamps of the self-attention kernel (side build: tools/side_lib.sh stamps attention.hip -DM4D_ATTN_STAMPS=1, made before the session)
[ -f more4d_amd/lib/libmore4d_hip_stamps.so ] && for a in 192 193; do M4D_LIB=stamps M4D_ATTN_ABL=$a timeout 200 python tools/attn_clock.py 2>&1 | grep -v amdgpu >> $O/attn_phases.log; done
timeout 900 python tools/ab_gemm.py 4 5:0 5:1:0 5 --reps 2 --n 40 2>&1 | grep "^variant" > $O/ab_gemm.log
for cfg in "M4D_GEMM_PERSIST=0" "M4D_GEMM_PERSIST=1 M4D_GEMM_SYNC=0" "M4D_GEMM_PERSIST=1 M4D_GEMM_SYNC=1"; do env $cfg timeout 500 python tools/gemm_traffic.py 2>&1 | tail -1; done > $O/gemm_traffic_variants.log
timeout 120 tools/probes/mfma_rate.bin > $O/mfma_rate_probe.log 2>&1
timeout 300 python tools/probes/bw_torch.py 2>&1 | grep -v amdgpu > $O/bw_probe.log
timeout 300 tools/probes/atomic_dq.bin 171 > $O/atomic_probe.log 2>&1
timeout 300 python tools/race_screen.py 20 > $O/race_screen.log 2>&1; tail -1 $O/race_screen.log
du -sh $O
