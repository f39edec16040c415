#!/usr/bin/env python
"""Debug aid for the persistent wide GEMM (tool library: M4D_LIB=abl): per-tile error map against an fp32 reference for a few
(shape, grid) cases.  Usage: M4D_LIB=abl M4D_GEMM_PERSIST_GRID=<g> python tools/dbg_persist.py M N K [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops

M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
g = torch.Generator(device="cuda").manual_seed(7)
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16, generator=g) * K ** -0.5
b = torch.randn(N, device="cuda", dtype=torch.bfloat16, generator=g)
if os.environ.get("DBG_BIAS", "1") == "0":
    b = None
ref = torch.nn.functional.linear(a.float(), w.float(), None if b is None else b.float())
tm, tn = (M + 255) // 256, (N + 255) // 256
for r in range(reps):
    out = ops.gemm_bt(a, w, b).float()
    err = (out - ref).abs()
    pad = torch.zeros(tm * 256, tn * 256, device="cuda")
    pad[:M, :N] = err
    tile = pad.view(tm, 256, tn, 256).amax(dim=(1, 3))
    bad = (tile > 0.05).nonzero().tolist()
    print(f"rep {r}: max err {float(err.max()):.3g}, bad tiles {len(bad)}/{tm * tn}: {bad[:40]}")
    if bad and r == 0:
        i, j = bad[0]
        sub = pad[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]
        rows = (sub.amax(1) > 0.05).nonzero().flatten().tolist()
        cols = (sub.amax(0) > 0.05).nonzero().flatten().tolist()
        o = out[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]
        rf = ref[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]
        for (rr, cc) in (sub > 0.05).nonzero().tolist()[:12]:
            bb = float(b[j * 256 + cc]) if b is not None else 0.0
            print(f"    [{rr},{cc}] out {float(o[rr, cc]):.6g} ref {float(rf[rr, cc]):.6g} bias {bb:.6g} ref-bias {float(rf[rr, cc]) - bb:.6g} bits {int(o[rr, cc].bfloat16().view(torch.int16)) & 0xffff:#06x}")
        print(f"  tile ({i},{j}): bad rows {len(rows)} [{rows[:8]}..{rows[-4:]}], bad cols {len(cols)} [{cols[:8]}..{cols[-4:]}]")
abl = int(os.environ.get("M4D_GEMM_ABL", "0"))
if abl & 128:        # known accumulators: the output should be the tile-local n (or m) index everywhere
    out = ops.gemm_bt(a, w, b).float()[:256, :256]
    idx = torch.arange(256, device="cuda", dtype=torch.float32)
    want = idx[:, None].expand(256, 256) if abl & 256 else idx[None, :].expand(256, 256)
    if b is not None:
        want = (want + b.float()[None, :256]).bfloat16().float()
    badpos = (out != want).nonzero().tolist()
    print(f"pattern check: {len(badpos)} wrong positions in tile (0,0)")
    for (i, j) in badpos[:48]:
        print(f"  out[{i},{j}] = {float(out[i, j])!r} (want {float(want[i, j])})")
print(ops.launch_counts())
