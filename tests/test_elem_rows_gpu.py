"""The straight-line forms of the DiT's two HBM-bound row kernels at C = 5120 (more4d_amd/csrc/elementwise.hip:
ln_modulate_rows_kernel<..., FULL>, rmsnorm_rope_rows_kernel): same bits as the general kernels (run in a child process with the
A/B switch), and fp32 torch math on top — WanLayerNorm + modulation (reference wan_transformer4d.py:662-669), WanRMSNorm + rope_apply
(:391-394, :66-110)."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from more4d_amd import ops
g = torch.Generator(device="cuda").manual_seed(21)
C, d = 5120, 128
out = {}
for name, dt, B, L, fused, rope, norm in (("bf16_qk_fused", torch.bfloat16, 2, 2310, True, True, True), ("bf16_q_only", torch.bfloat16, 1, 4100, False, False, True),
                                         ("bf16_rope_only", torch.bfloat16, 1, 4099, False, True, False), ("f32_qk", torch.float32, 1, 4200, False, True, True)):
    rows = B * L
    buf = torch.randn(rows, 2 * C, generator=g, device="cuda").to(dt)
    if fused:
        q, k = buf[:, :C], buf[:, C:]            # column halves of one projection buffer (row stride 2C)
    else:
        q, k = buf[:, :C].contiguous(), None
    wq = (torch.rand(C, generator=g, device="cuda") + 0.5) if norm else None
    wk = (torch.rand(C, generator=g, device="cuda") + 0.5) if norm else None
    cos = torch.randn(L + 7, d // 2, generator=g, device="cuda") if rope else None
    sin = torch.randn(L + 7, d // 2, generator=g, device="cuda") if rope else None
    ops.rmsnorm_rope(q, wq, k, wk if k is not None else None, head_dim=d, cos=cos, sin=sin, rows_per_sample=L, rope_len=L - 5, pos_offset=3)
    out["rms_" + name] = buf.float().cpu()
for name, dt_in, dt_out, B, L, affine in (("f32_bf16_mod", torch.float32, torch.bfloat16, 2, 2100, False), ("f32_bf16_affine", torch.float32, torch.bfloat16, 1, 4097, True),
                                          ("bf16_bf16_mod", torch.bfloat16, torch.bfloat16, 1, 4200, False)):
    x = torch.randn(B, L, C, generator=g, device="cuda").to(dt_in)
    e = torch.randn(B, 6, C, generator=g, device="cuda")
    if affine:
        y = ops.ln_modulate(x, dt_out, ln_w=e[0, 0].contiguous(), ln_b=e[0, 1].contiguous())
    else:
        y = ops.ln_modulate(x, dt_out, shift=e[:, 0], scale=e[:, 1], mod_stride=6 * C, rows_per_sample=L)
    out["ln_" + name] = y.float().cpu()
torch.save(out, sys.argv[1])
'''


def run(env):
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "o.pt")
        r = subprocess.run([sys.executable, "-c", CODE, f], cwd=ROOT, env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return torch.load(f)


def test_row_kernels_same_bits_as_the_general_kernels():
    new = run({})
    old = run({"M4D_RMS_ROWS": "0", "M4D_LN_VAR": "5"})
    gen = run({"M4D_RMS_ROWS": "0", "M4D_LN_ROWS": "0"})
    assert set(new) == set(old) == set(gen) and len(new) == 7
    for k in new:
        assert torch.isfinite(new[k]).all(), k
        assert torch.equal(new[k], old[k]), k
        assert torch.equal(new[k], gen[k]), k
