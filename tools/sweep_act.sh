for b in "$@"; do timeout 500 python tools/bench_train.py --layers 40 --steps 2 --warmup 1 --act-budget $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('budget', '$b', 's/step', round(d['value'],3), 'peakGB', round(d['max_mem_gb'],1), d['stored_blocks'])"; done
python -c "import torch; print('total GB', torch.cuda.get_device_properties(0).total_memory/2**30)"
