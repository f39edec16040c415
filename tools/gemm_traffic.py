#!/usr/bin/env python
"""Fabric-side traffic of the production GEMM at the bench's shapes (bench.measure_traffic: rocprofv3 PMC passes over bench.py --pmc-probe)
under the current environment, e.g.  M4D_GEMM_PERSIST=0 python tools/gemm_traffic.py ;  M4D_GEMM_SYNC=1 python tools/gemm_traffic.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tot, d = bench.measure_traffic()
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("M4D_GEMM"))
if tot is None:
    print(tag, "FAILED", d)
else:
    print(f"[{tag}] counter / algorithmic = {d['counter_over_algorithmic']:.2f}x  " +
          "  ".join(f"{k}: {v['counter_over_algorithmic']:.2f}x" for k, v in d["per_shape"].items()))
