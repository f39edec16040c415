"""attn128x_kernel (more4d_amd/csrc/attention_xp.h): the persistent (query tile, key tile) pipeline that takes the short-key-list
attention calls of the DiT — WanI2VCrossAttention's text + image branches (reference wan_transformer4d.py:533-552) and
WanT2VCrossAttention (:500-513).  Checked against fp32 torch math on the bf16-rounded operands and against the lock-step kernel
(M4D_ATTN_XP=0 in a child process); the launch-class counter proves the kernel under test ran."""
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def make_segs(g, B, C, lens, poison=True):
    """K [B, Lp, C] / V^T [C, B * Lp] per segment with Lp = len rounded up to 8; the padding rows / columns hold NaN: the kernel must
    neither read the K rows nor let the V^T columns reach the accumulators."""
    from more4d_amd.ops import KV
    segs, refs = [], []
    for L in lens:
        Lp = max(8, (L + 7) // 8 * 8)
        k = torch.randn(B, Lp, C, generator=g, device=DEV).to(BF)
        vt = torch.randn(C, B, Lp, generator=g, device=DEV).to(BF)
        if poison and Lp > L:
            k[:, L:] = float("nan")
            vt[:, :, L:] = float("nan")
        segs.append(KV(k.view(-1), vt.view(C, B * Lp), Lp * C, C, Lp, B * Lp, L))
        refs.append((k[:, :L], vt[:, :, :L]))
    return segs, refs


def torch_groups(q, refs, groups, B, Lq, n, d, rows):
    """fp32 softmax(q k^T / sqrt(d)) v per group of segments for the query rows `rows`, each group's result rounded to bf16 and added in
    bf16 like the reference's x + img_x."""
    qf = q.view(B, Lq, n, d)[:, rows].float()
    out = None
    for grp in groups:
        kf = torch.cat([refs[i][0] for i in grp], 1).reshape(B, -1, n, d).float()
        vf = torch.cat([refs[i][1] for i in grp], 2).permute(1, 2, 0).reshape(B, -1, n, d).float()
        s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) / math.sqrt(d)
        o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vf).reshape(B, len(rows), n * d).to(BF)
        out = o if out is None else (out.float() + o.float()).to(BF)
    return out


CASES = [
    # (B, heads, Lq, key segment lengths, new_softmax, accumulate)
    (2, 8, 2100, (512, 257), 0b10, False),        # the i2v cross-attention: text | image, ragged one-key tile, 9 query tiles
    (2, 5, 1300, (512, 257), 0b10, True),         # heads * B not a multiple of 8 (plain item order), accumulate on top of `out`
    (1, 8, 1500, (512,), 0, False),               # t2v cross-attention: one softmax, full tiles only
    (1, 8, 1100, (1,), 0, False),                 # one key: a single ragged tile per item
    (1, 16, 1280, (100,), 0, True),               # one ragged tile of 100 keys, accumulate, exactly 5 query tiles
    (1, 8, 1030, (130, 70, 64), 0, False),        # three segments, one softmax, ragged tiles in the middle of the list
    (1, 8, 1030, (130, 70, 64), 0b110, False),    # three softmaxes
    (1, 8, 1200, (64, 0, 200, 0, 0, 7, 0, 129), 0b100000, False),   # eight segments, empty ones in between, second softmax from segment 5
    (2, 40, 5000, (512, 257), 0b10, False),       # 1 600 items over 256 workgroups: runs of 6-7 items crossing (b, h) boundaries
]


@pytest.mark.parametrize("B,n,Lq,lens,new_softmax,accumulate", CASES)
def test_xp_kernel_vs_fp32(B, n, Lq, lens, new_softmax, accumulate):
    from more4d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    d = 128
    C = n * d
    q = torch.randn(B, Lq, C, generator=g, device=DEV).to(BF)
    segs, refs = make_segs(g, B, C, lens)
    base = torch.randn(B, Lq, C, generator=g, device=DEV).to(BF) if accumulate else None
    ops.launch_counts(reset=True)
    out = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=d, out=base.clone() if accumulate else None, accumulate=accumulate,
                        new_softmax=new_softmax)
    assert ops.launch_counts()["attn_xp"] == 1
    assert torch.isfinite(out.float()).all()
    groups, cur = [], []
    for i in range(len(lens)):
        if i > 0 and (new_softmax >> i) & 1:
            groups.append(cur)
            cur = []
        cur.append(i)
    groups.append(cur)
    rows = torch.unique(torch.cat([torch.arange(0, min(Lq, 300)), torch.arange(max(0, Lq - 300), Lq),
                                   torch.randint(0, Lq, (400,), generator=torch.Generator().manual_seed(1))])).to(DEV)
    want = torch_groups(q, refs, groups, B, Lq, n, d, rows)
    if accumulate:
        want = (want.float() + base[:, rows].float()).to(BF)
    # bf16 P and bf16 outputs: 3e-3 .. 6e-3 of the largest output; every group adds one more rounding
    assert rel_err(out[:, rows].float(), want.float()) < 1.2e-2
    # a second launch gives the same bits (no dependence on workgroup timing)
    again = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=d, out=base.clone() if accumulate else None, accumulate=accumulate,
                          new_softmax=new_softmax)
    assert torch.equal(out, again)


def test_xp_kernel_lse_and_full_shape():
    """The DiT's shape (B = 2, 40 heads, 21 840 queries): text + image in one launch against sampled fp32 rows, and the single-softmax
    call with the log-sum-exp the backward needs (512 keys)."""
    from more4d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(12)
    B, n, d, Lq = 2, 40, 128, 21840
    C = n * d
    q = torch.randn(B, Lq, C, generator=g, device=DEV).to(BF)
    segs, refs = make_segs(g, B, C, (512, 257))
    ops.launch_counts(reset=True)
    out = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=d, new_softmax=0b10)
    assert ops.launch_counts()["attn_xp"] == 1
    rows = torch.cat([torch.arange(0, 64), torch.arange(Lq - 100, Lq), torch.randint(0, Lq, (256,), generator=torch.Generator().manual_seed(2))]).to(DEV)
    want = torch_groups(q, refs, [[0], [1]], B, Lq, n, d, rows)
    assert rel_err(out[:, rows].float(), want.float()) < 1.2e-2
    # race screen: the interval-closing waits count what was issued behind the tile request (stores of a flush, the next item's Q
    # fragments) — a tile read before it has landed would show as run-to-run differences
    for _ in range(6):
        again = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=d, new_softmax=0b10)
        assert torch.equal(out, again)
    lse = torch.empty(B, n, Lq, device=DEV, dtype=torch.float32)
    o1 = ops.attention(q, [segs[0]], B=B, Lq=Lq, heads=n, head_dim=d, lse=lse)
    assert ops.launch_counts()["attn_xp"] == 8
    qf = q.view(B, Lq, n, d)[:, rows].float()
    kf = refs[0][0].reshape(B, -1, n, d).float()
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) / math.sqrt(d)
    want_lse = torch.logsumexp(s, -1) / math.log(2.0)          # log2 domain
    assert float((lse[:, :, rows] - want_lse).abs().max()) < 1e-3
    want1 = torch_groups(q, refs, [[0]], B, Lq, n, d, rows)
    assert rel_err(o1[:, rows].float(), want1.float()) < 8e-3


def test_xp_kernel_vs_lockstep_kernel_in_child_process():
    """M4D_ATTN_XP=0 routes the same call to the lock-step 4-wave kernel (exact running maximum instead of the lazy one): the two kernels
    agree to bf16 rounding on every output element."""
    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from more4d_amd import ops
from test_attn_xp_gpu import make_segs
g = torch.Generator(device="cuda").manual_seed(13)
B, n, d, Lq = 2, 8, 128, 3000
q = torch.randn(B, Lq, n * d, generator=g, device="cuda").bfloat16()
segs, _ = make_segs(g, B, n * d, (512, 257))
out = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=d, new_softmax=0b10)
torch.save(out.cpu(), sys.argv[1])
print(ops.launch_counts()["attn_xp"])
'''
    import tempfile
    outs, counts = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for xp in ("1", "0"):
            f = os.path.join(tmp, f"o{xp}.pt")
            r = subprocess.run([sys.executable, "-c", code, f], cwd=ROOT, env={**os.environ, "M4D_ATTN_XP": xp}, capture_output=True, text=True,
                               timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            counts.append(int(r.stdout.strip().splitlines()[-1]))
            outs.append(torch.load(f))
    assert counts == [1, 0]
    assert rel_err(outs[0].float(), outs[1].float()) < 8e-3


def test_phased_kernel_ragged_tail_with_poisoned_padding():
    """attn128p_kernel's peeled ragged tile reads whole 16-byte chunks from clamped addresses and masks them: NaN in the K rows / V^T
    columns beyond the segment must not reach the output (2 048 + 37 keys, two segments with ragged tails)."""
    from more4d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(14)
    B, n, d, Lq = 1, 8, 128, 1300
    C = n * d
    q = torch.randn(B, Lq, C, generator=g, device=DEV).to(BF)
    for lens in ((2048 + 37,), (1024 + 5, 1100)):
        segs, refs = make_segs(g, B, C, lens)
        if len(lens) > 1:      # the phased kernel shares one V^T row stride across segments: both segments as column ranges of one buffer
            from more4d_amd.ops import KV
            Lp = [max(8, (L + 7) // 8 * 8) for L in lens]
            k = torch.cat([s.k.view(B, lp, C) for s, lp in zip(segs, Lp)], 1).contiguous()
            vt = torch.cat([s.vt.view(C, B, lp) for s, lp in zip(segs, Lp)], 2).contiguous()
            tot = sum(Lp)
            segs = [KV(k.view(-1)[off * C:], vt.view(C, B * tot)[:, off:], tot * C, C, tot, B * tot, L)
                    for off, L in zip((0, Lp[0]), lens)]
        ops.launch_counts(reset=True)
        out = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=d)
        c = ops.launch_counts()       # one long-key launch on attn128q_kernel (one or several segments, one ragged tail each)
        assert c["attn_phased"] + c["attn_q64"] == 1 and c["attn_q64"] == 1, c
        assert torch.isfinite(out.float()).all()
        rows = torch.arange(0, Lq, 3).to(DEV)
        want = torch_groups(q, refs, [list(range(len(lens)))], B, Lq, n, d, rows)
        assert rel_err(out[:, rows].float(), want.float()) < 8e-3
