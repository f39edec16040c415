#!/bin/bash
# VAE profile recipe (GPU box): kernel trace + stats of the 49x480x832 round trip, then separate PMC passes (MFMA busy, FETCH_SIZE,
# WRITE_SIZE) on a 17-frame round trip.  Output: gpurun_out/prof_vae; summarise with tools/summarize_prof_vae.py <tag>.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_vae
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o vae -- python $R/tools/bench_vae.py > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o vae -- python $R/tools/bench_vae.py 17 480 832 > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o vae -- python $R/tools/bench_vae.py 17 480 832 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o vae -- python $R/tools/bench_vae.py 17 480 832 > $OUT/pmc_write.log 2>&1
cd $OUT && find . -name "*kernel_trace.csv" -size +20M -delete; du -sh .
