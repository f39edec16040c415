O=gpurun_out/final; mkdir -p $O
{ timeout 400 python tools/bench_shard.py --world 1 --mode sp --steps 2 2>&1 | tail -1; for m in cfg-sp sp; do for w in 2 4 8; do timeout 400 python tools/bench_shard.py --world $w --mode $m --steps 2 2>&1 | tail -1; done; done; } > $O/bench_shard.log 2>&1
M4D_SP_MODE=ulysses timeout 400 python tools/bench_shard.py --world 8 --mode cfg-sp --steps 2 2>&1 | tail -1 >> $O/bench_shard.log; cat $O/bench_shard.log
