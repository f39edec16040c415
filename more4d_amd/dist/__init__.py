"""Sequence-parallel ("T-sharded") group for the denoise loop — built from scratch: the reference imports
`MoRe4D.dist` (get_sequence_parallel_rank/world_size, get_sp_group, usp_attn_forward,
xFuserLongContextAttention; wan_transformer4d.py:23-25) but ships no such package.

Design for MI355X: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI).  Tokens are
f-major, so a contiguous split of the token axis is a split along T (reference intent, :1187-1198).  Every
op but self-attention is token-local; per layer each rank all-gathers K [B, Ls, C] and V^T [C, B*Ls] and hands
the attention kernel one K/V *segment per rank* (no concat copy).  The final head output is all-gathered
along tokens (:1320-1321).
"""
import os

import torch
import torch.distributed as dist

from ..ops import KV

_SP_GROUP = None
_CFG = None     # (cfg_rank, world group size) when the two CFG branches run on different halves of the world


class SequenceParallelGroup:
    # self-attention attends the local K/V shard first (overlapping the all-gather) and merges the remote part afterwards
    # (m4d_attn_merge); M4D_SP_LOCAL_FIRST=0 restores the single attention call over all gathered segments
    local_first = os.environ.get("M4D_SP_LOCAL_FIRST", "1") != "0"
    # "allgather" (north_star: RCCL all-gather of K / V^T per layer) | "ulysses": head-split all-to-all — every rank attends ALL
    # tokens for heads/W of the heads; q, k, v^T go out and o comes back: 4 Ls C elements per rank and layer instead of the
    # 2 (W-1) Ls C an all-gather receives (3.5x fewer bytes at W = 8).  Selectable for A/B on a real node:
    # init_sequence_parallel(mode=...) or M4D_SP_MODE.
    mode = os.environ.get("M4D_SP_MODE", "allgather")

    def __init__(self, group=None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_gather(self, x, dim=1):
        """Concatenate equal-sized shards along `dim` (reference get_sp_group().all_gather, :1321)."""
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(self.world_size)]
        dist.all_gather(parts, x, group=self.group)
        return torch.cat(parts, dim=dim)

    def _gather_into(self, dst, src):
        """One flat all-gather (RCCL: a single collective into a contiguous buffer); list form for backends without it."""
        try:
            dist.all_gather_into_tensor(dst, src, group=self.group)
        except (RuntimeError, NotImplementedError, AttributeError):
            dist.all_gather(list(dst.unbind(0)), src, group=self.group)

    def all_to_all(self, x):
        """x [W, ...]: chunk j goes to rank j; returns [W, ...] whose chunk j came from rank j (one RCCL all-to-all)."""
        x = x.contiguous()
        out = torch.empty_like(x)
        try:
            dist.all_to_all_single(out, x, group=self.group)
        except (RuntimeError, NotImplementedError, AttributeError):
            parts = [torch.empty_like(x) for _ in range(self.world_size)]      # backends without all-to-all (CPU tests)
            dist.all_gather(parts, x, group=self.group)
            for j in range(self.world_size):
                out[j].copy_(parts[j][self.rank])
        return out

    def gather_start(self, x):
        """Begin an all-gather of this rank's shard; returns (gathered buffer [W, *x.shape], work handle).  The
        collective runs on RCCL's stream after the work already enqueued on the current stream, so the caller can keep
        launching kernels (the next projection GEMM) while the shards travel over xGMI."""
        x = x.contiguous()
        buf = torch.empty((self.world_size,) + tuple(x.shape), device=x.device, dtype=x.dtype)
        try:
            work = dist.all_gather_into_tensor(buf, x, group=self.group, async_op=True)
        except (RuntimeError, NotImplementedError, AttributeError):
            work = dist.all_gather(list(buf.unbind(0)), x, group=self.group, async_op=True)
        return buf, work, x

    def gather_finish(self, hk, hv, B, Ls, C, key_len):
        """Wait for both gathers (the current stream waits, not the host) and describe one K/V segment per rank."""
        (kg, wk, _), (vg, wv, _) = hk, hv
        for w in (wk, wv):
            if w is not None:
                w.wait()
        segs = []
        for r in range(self.world_size):
            n = max(0, min(Ls, key_len - r * Ls))
            segs.append(KV(kg[r].reshape(-1), vg[r], Ls * C, C, Ls, B * Ls, n))
        return segs

    def gather_kv(self, k, vt, B, Ls, C, key_len):
        """k: T [B*Ls, C] (this rank's keys after RMSNorm+RoPE), vt: T [C, B*Ls].  Returns one KV segment per
        rank; segment r covers global tokens [r*Ls, (r+1)*Ls) of each sample, of which
        clamp(key_len - r*Ls, 0, Ls) are valid keys."""
        W = self.world_size
        kg = torch.empty((W,) + tuple(k.shape), device=k.device, dtype=k.dtype)
        vg = torch.empty((W,) + tuple(vt.shape), device=vt.device, dtype=vt.dtype)
        self._gather_into(kg, k.contiguous())
        self._gather_into(vg, vt.contiguous())
        segs = []
        for r in range(W):
            n = max(0, min(Ls, key_len - r * Ls))
            segs.append(KV(kg[r].view(-1), vg[r], Ls * C, C, Ls, B * Ls, n))
        return segs


def init_sequence_parallel(group=None, cfg_parallel=False, mode=None):
    """Make `group` (default: WORLD) the sequence-parallel group.  torch.distributed must be initialised.
    mode: "allgather" (default) | "ulysses" (see SequenceParallelGroup.mode).

    cfg_parallel=True (even world size): the unconditional branch of classifier-free guidance runs on ranks
    [0, W/2), the conditional branch on [W/2, W); each half is its own sequence-parallel group (W/2 = 1: no token
    sharding at all).  The two branches never talk inside a DiT forward — only the [B,16,F,H,W] velocity is exchanged once
    per step (`cfg_exchange`) — so per-layer K/V all-gather traffic per rank drops by (W-1)*2 / (W/2-1) (2.3x at W = 8:
    half the batch, 3 peers instead of 7) and a 2-GPU run has no per-layer collective at all (SURVEY §8e "2-way CFG split
    x 4-way T split")."""
    global _SP_GROUP, _CFG
    _CFG = None
    if mode is not None:
        if mode not in ("allgather", "ulysses"):
            raise ValueError(f"unknown sequence-parallel mode {mode!r}")
        SequenceParallelGroup.mode = mode
    if cfg_parallel:
        if group is not None:
            raise ValueError("cfg_parallel builds its own groups from WORLD")
        W = dist.get_world_size()
        if W % 2:
            raise ValueError("cfg_parallel needs an even world size")
        half = W // 2
        if half == 1:           # one rank per branch: nothing to shard, no group to build
            _SP_GROUP = SequenceParallelGroup.__new__(SequenceParallelGroup)
            _SP_GROUP.group, _SP_GROUP.world_size, _SP_GROUP.rank = None, 1, 0
        else:
            mine = None
            for b in range(2):  # every rank must take part in the creation of every group
                g = dist.new_group(list(range(b * half, (b + 1) * half)))
                if dist.get_rank() // half == b:
                    mine = g
            _SP_GROUP = SequenceParallelGroup(mine)
        _CFG = (dist.get_rank() // half, W)
        return _SP_GROUP
    _SP_GROUP = SequenceParallelGroup(group)
    return _SP_GROUP


def get_cfg_parallel_rank():
    """0 = this rank computes the unconditional branch, 1 = the conditional one, None = CFG is batched on every rank."""
    return None if _CFG is None else _CFG[0]


def cfg_exchange(v):
    """v: this rank's branch velocity [B, ...] -> [2B, ...] = (unconditional, conditional), identical on every rank.
    One small all-gather over WORLD per denoise step (2.6 MB per rank at 49x480x832)."""
    if _CFG is None:
        raise RuntimeError("cfg_exchange without init_sequence_parallel(cfg_parallel=True)")
    W = _CFG[1]
    v = v.contiguous()
    buf = torch.empty((W,) + tuple(v.shape), device=v.device, dtype=v.dtype)
    try:
        dist.all_gather_into_tensor(buf, v)
    except (RuntimeError, NotImplementedError, AttributeError):
        dist.all_gather(list(buf.unbind(0)), v)
    return torch.cat([buf[0], buf[W // 2]])


def get_sp_group():
    if _SP_GROUP is None:
        raise RuntimeError("sequence parallelism is not initialised: call more4d_amd.dist.init_sequence_parallel()")
    return _SP_GROUP


def get_sequence_parallel_world_size():
    return 1 if _SP_GROUP is None else _SP_GROUP.world_size


def get_sequence_parallel_rank():
    return 0 if _SP_GROUP is None else _SP_GROUP.rank
