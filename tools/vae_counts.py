import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, bench_vae
from more4d_amd import ops
ops.launch_counts(reset=True)
r = bench_vae.run(49, 480, 832, iters=1, dev="cuda", verbose=False)
print({k: v for k, v in ops.launch_counts().items() if v})
