#!/bin/bash
# Profile recipe used for profiles/ (run on the GPU box through gpurun): kernel trace + stats of the
# bench command, then separate PMC passes (never combined with trace domains other than kernel-trace).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/trace_bench.log 2>&1
tail -3 $OUT/trace_bench.log
PMC_ARGS="--steps 1 --warmup 0 --layers 2 --no-cpu-baseline --no-secondary --no-kernel-timers"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o bench -- python $R/bench.py $PMC_ARGS > $OUT/pmc_mfma.log 2>&1
tail -2 $OUT/pmc_mfma.log
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $R/bench.py $PMC_ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $R/bench.py $PMC_ARGS > $OUT/pmc_write.log 2>&1
cd $OUT && find . -type f -size +20M -delete; find . -type f | xargs ls -la; du -sh .
