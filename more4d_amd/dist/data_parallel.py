"""Data-parallel training state for N MI355X (BASELINE configs[4]: `train_wan.sh` under `accelerate launch` = torch DDP,
train_wan.py:678-687, one sample per GPU) — built MI355X-first instead of wrapping DDP's ring all-reduce:

* parameters and gradients live in a few flat **buckets** (≈ one DiT block each, 0.8 GB bf16 at 14B); `p.data` / `p.grad` are
  views, so autograd accumulates straight into the bucket and nothing is copied before a collective;
* the moment the last gradient of a bucket has been accumulated during backward, ONE `reduce_scatter` (RCCL, the 7 xGMI links
  of a GPU in parallel) is issued asynchronously — it overlaps the rest of the backward, which is ≈ 40 block-times long;
* every rank then owns the summed gradient of 1/N of each bucket: the global gradient norm is one sum-of-squares pass over that
  slice + one scalar all-reduce, and the fused clip + AdamW kernel (`m4d_adamw`) updates **only that slice** — optimizer state
  and update traffic shrink N-fold (14B bf16: 56 GB of moments -> 7 GB per rank at N = 8);
* ONE `all_gather` per bucket returns the updated parameters (in place, into the flat buffer).

Bytes on the wire per rank and step: (N-1)/N x 33 GB reduce-scatter + the same all-gather — what a ring all-reduce moves, but as
two direct collectives RCCL can spread over all links, and with the optimizer in between sharded (SURVEY §8d config 5:
≈ 0.06 s vs ≈ 0.38 s for a single-link ring).  Averaging (DDP semantics) is folded into the clip coefficient handed to the AdamW
kernel: gradients stay sums on the wire.

Backends without reduce-scatter / flat all-gather (gloo, the CPU test-suite) fall back to all-reduce + slice / list gather; the
arithmetic is the same.  With world size 1 no collective is issued and this is a flat-buffer AdamW.
"""
import torch
import torch.distributed as dist

from .. import ops

_ALIGN = 128      # elements: every parameter starts on a 256-byte (bf16) / 512-byte (fp32) boundary of its bucket


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "flat_p", "flat_g", "pending", "work", "exp_avg", "exp_avg_sq", "ready")


class ShardedDataParallel:
    """dp = ShardedDataParallel(model, lr=..., ...);  loss.backward();  norm = dp.reduce_gradients();  dp.step(clip=...)"""

    def __init__(self, model, *, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, state_dtype=None, group=None,
                 bucket_bytes=1 << 30):
        self.model = model
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state_dtype = state_dtype
        self.step_count = 0
        self._handles = []
        self._defer = False         # no_sync(): micro-batches whose gradients only accumulate locally
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("ShardedDataParallel: the model has no trainable parameter")
        if len({p.dtype for p in params}) != 1 or len({p.device for p in params}) != 1:
            raise TypeError("ShardedDataParallel: trainable parameters must share one dtype and device")
        self.dtype, self.device = params[0].dtype, params[0].device
        # buckets in REVERSE registration order ~ the order in which backward finishes them
        self.buckets = []
        cur, cur_n = [], 0
        limit = max(1, bucket_bytes // params[0].element_size())
        for p in reversed(params):
            n = -(-p.numel() // _ALIGN) * _ALIGN
            if cur and cur_n + n > limit:
                self.buckets.append(self._make_bucket(cur))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += n
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self._owner = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._owner[id(p)] = (b, i)
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ construction
    def _make_bucket(self, params):
        b = _Bucket()
        b.params, b.offsets = list(params), []
        off = 0
        for p in params:
            b.offsets.append(off)
            off += -(-p.numel() // _ALIGN) * _ALIGN
        unit = _ALIGN * self.world
        b.numel = -(-off // unit) * unit                     # every rank's slice has the same aligned length
        b.flat_p = torch.zeros(b.numel, device=self.device, dtype=self.dtype)
        b.flat_g = torch.zeros(b.numel, device=self.device, dtype=self.dtype)
        with torch.no_grad():
            for p, o in zip(params, b.offsets):
                n = p.numel()
                b.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = b.flat_p[o:o + n].view(p.shape)     # the parameter now lives in the bucket
                p.grad = b.flat_g[o:o + n].view(p.shape)     # autograd accumulates in place into the bucket
        b.pending, b.work, b.ready = len(params), None, False
        sd = self.state_dtype or self.dtype
        s = b.numel // self.world
        b.exp_avg = torch.zeros(s, device=self.device, dtype=sd)
        b.exp_avg_sq = torch.zeros(s, device=self.device, dtype=sd)
        return b

    def _slice(self, b, flat):
        s = b.numel // self.world
        return flat[self.rank * s:(self.rank + 1) * s]

    # ------------------------------------------------------------------ backward side
    def _on_grad(self, p):
        b, i = self._owner[id(p)]
        o = b.offsets[i]
        if p.grad is None or p.grad.data_ptr() != b.flat_g.data_ptr() + o * b.flat_g.element_size():
            # someone replaced .grad (zero_grad(set_to_none=True) + autograd's "steal"): fold it back into the bucket view
            view = b.flat_g[o:o + p.numel()].view(p.shape)
            if p.grad is not None:
                view.add_(p.grad.to(view.dtype))
            p.grad = view
        if b.ready and self.world == 1:
            # nothing was reduced at world 1: a second backward simply accumulates, like a plain model (and torch DDP at world 1)
            b.ready, b.pending = False, len(b.params)
        if b.ready:
            # the bucket's reduce-scatter is already on the wire (or done): this rank's slice holds GLOBAL sums, a second
            # backward would add local gradients on top of them and never reduce those
            raise RuntimeError("ShardedDataParallel: backward() reached a bucket whose gradients were already reduced; run the "
                               "earlier micro-batches under `with dp.no_sync():` (gradient accumulation) or call zero_grad()")
        b.pending -= 1
        if b.pending == 0 and not self._defer:
            self._launch(b)
        if b.pending == 0 and self._defer:
            b.pending = len(b.params)          # re-arm for the next micro-batch; the LAST one launches

    def _launch(self, b):
        if b.ready:
            return
        b.ready = True
        if self.world == 1:
            return
        try:
            b.work = dist.reduce_scatter_tensor(self._slice(b, b.flat_g), b.flat_g, op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True)
        except (RuntimeError, NotImplementedError, AttributeError):
            b.work = dist.all_reduce(b.flat_g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def no_sync(self):
        """Context manager for gradient accumulation (`accelerator.accumulate` / --gradient_accumulation_steps,
        train_wan.py:1916): backward() inside it only accumulates into the local buckets; the first backward outside it
        issues the reduce-scatters.  Mirrors torch DDP's no_sync() — which, like this one, SUMS the micro-batch gradients:
        `accelerator.backward` divides the loss by gradient_accumulation_steps before it calls backward, so a port of the
        reference loop either does the same (`(loss / steps).backward()`) or passes `accumulation_steps=steps` to
        reduce_gradients() / step(), which folds 1 / steps into the norm and into the fused update scale."""
        dp = self

        class _NoSync:
            def __enter__(self):
                self.prev, dp._defer = dp._defer, True

            def __exit__(self, *exc):
                dp._defer = self.prev
                for b in dp.buckets:             # a partially finished bucket (unused parameters) restarts its count
                    if not b.ready:
                        b.pending = len(b.params)
                return False

        return _NoSync()

    def _wait_all(self):
        """Launch what backward left unlaunched and make the current stream wait for every outstanding reduce-scatter."""
        for b in self.buckets:
            self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None

    def reduce_gradients(self, accumulation_steps=1):
        """Issue the reduce-scatter of every bucket backward did not complete (unused parameters keep zero gradients), wait for
        all of them (the current stream waits, not the host) and return the global L2 norm of the AVERAGED gradients as a 0-d
        device tensor (`torch.norm(stack(norm(g)))` of train_wan.py:1991-1993 on what DDP would have left in .grad).
        accumulation_steps: micro-batches summed under no_sync() whose losses were NOT pre-divided (see no_sync)."""
        self._wait_all()
        acc = torch.zeros((), device=self.device, dtype=torch.float32)
        for b in self.buckets:
            ops.sumsq(self._slice(b, b.flat_g), acc)
        if self.world > 1:
            dist.all_reduce(acc, group=self.group)
        return acc.sqrt() / (self.world * accumulation_steps)

    # ------------------------------------------------------------------ optimizer side
    @torch.no_grad()
    def step(self, max_norm=None, total_norm=None, accumulation_steps=1):
        """Clip (coefficient fused into the update: no pass over the gradients) + AdamW on this rank's slice of every bucket +
        all-gather of the updated parameters.  `total_norm`: what reduce_gradients() returned (required with max_norm; pass the
        same accumulation_steps to both)."""
        # "ready" only means the reduce-scatter was LAUNCHED (asynchronously, from the backward hooks): the update below reads
        # the gradient slice on the compute stream, so every outstanding collective is waited for here, unconditionally
        self._wait_all()
        scale = torch.full((), 1.0 / (self.world * accumulation_steps), device=self.device, dtype=torch.float32)       # sum -> mean
        if max_norm is not None:
            if total_norm is None:
                raise ValueError("step(max_norm=...) needs the total_norm returned by reduce_gradients()")
            scale = scale * (max_norm / (total_norm + 1e-6)).clamp(max=1.0)
        scale = scale.to(torch.float32).contiguous()
        self.step_count += 1
        works = []
        for b in self.buckets:
            ops.adamw_(self._slice(b, b.flat_p), self._slice(b, b.flat_g), b.exp_avg, b.exp_avg_sq, lr=float(self.lr),
                       beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay,
                       step=self.step_count, grad_scale=scale)
            if self.world > 1:
                try:
                    works.append(dist.all_gather_into_tensor(b.flat_p, self._slice(b, b.flat_p), group=self.group, async_op=True))
                except (RuntimeError, NotImplementedError, AttributeError):
                    s = b.numel // self.world
                    works.append(dist.all_gather([b.flat_p[r * s:(r + 1) * s] for r in range(self.world)],
                                                 self._slice(b, b.flat_p).clone(), group=self.group, async_op=True))
        for w in works:
            w.wait()
        for b in self.buckets:
            for p in b.params:
                torch.autograd.graph.increment_version(p)       # raw-pointer update: refresh version-keyed caches
        return total_norm

    def zero_grad(self):
        """Clear the buckets (one memset each) and re-arm the per-bucket counters; .grad stays the bucket view."""
        for b in self.buckets:
            if b.work is not None:          # never clear a buffer a collective may still be writing
                b.work.wait()
            b.flat_g.zero_()
            b.pending, b.ready, b.work = len(b.params), False, None
            for p, o in zip(b.params, b.offsets):
                if p.grad is None or p.grad.data_ptr() != b.flat_g.data_ptr() + o * b.flat_g.element_size():
                    p.grad = b.flat_g[o:o + p.numel()].view(p.shape)

    # ------------------------------------------------------------------ checkpoint / resume (accelerator.save_state)
    def state_dict(self):
        """This rank's shard of the optimizer state (moments of its slice of every bucket) + the step counter and the
        hyper-parameters.  Parameters are saved through the model (they are replicated); the shard is only valid for the same
        world size, rank and bucket layout, which `load_state_dict` checks."""
        return {"step": self.step_count, "world": self.world, "rank": self.rank,
                "layout": [(b.numel, len(b.params)) for b in self.buckets],
                "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                "exp_avg": [b.exp_avg.detach().clone() for b in self.buckets],
                "exp_avg_sq": [b.exp_avg_sq.detach().clone() for b in self.buckets]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        if sd["world"] != self.world or sd["rank"] != self.rank:
            raise ValueError(f"optimizer shard of rank {sd['rank']}/{sd['world']} loaded on rank {self.rank}/{self.world}")
        if [tuple(x) for x in sd["layout"]] != [(b.numel, len(b.params)) for b in self.buckets]:
            raise ValueError("ShardedDataParallel.load_state_dict: bucket layout differs (model or bucket_bytes changed)")
        self.step_count = int(sd["step"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
        for b, m, v in zip(self.buckets, sd["exp_avg"], sd["exp_avg_sq"]):
            b.exp_avg.copy_(m.to(b.exp_avg.device, b.exp_avg.dtype))
            b.exp_avg_sq.copy_(v.to(b.exp_avg_sq.device, b.exp_avg_sq.dtype))

    def state_bytes(self):
        return sum(b.exp_avg.numel() * b.exp_avg.element_size() * 2 for b in self.buckets)

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []
