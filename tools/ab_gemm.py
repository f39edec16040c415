#!/usr/bin/env python
"""Same-box A/B of the production GEMM structures (M4D_GEMM_VARIANT is read once per process, so every variant runs in
its own child, alternating): sustained timing at the DiT's three shapes on N(0,1) operands + bit-comparison of the results
(all structures accumulate K in the same MFMA order).  Usage: python tools/ab_gemm.py 4 5 5:0 [--reps 2] [--n 40]
("5:1:1" = persistent kernel with the XCD-wide tile rounds, M4D_GEMM_SYNC=1; "5:0" = variant 5 with M4D_GEMM_PERSIST=0, i.e. the one-tile-per-workgroup form of the wide kernel)"""
import hashlib, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [("qkvo", 43680, 5120, 5120), ("ffn_up", 43680, 13824, 5120), ("ffn_down", 43680, 5120, 13824),
          ("shard8", 5460, 5120, 5120), ("ragged", 1000, 776, 1088)]


def child(n):
    import torch
    from more4d_amd import ops
    res = {}
    for name, M, N, K in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(7)
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16, generator=g) * K ** -0.5
        b = torch.randn(N, device="cuda", dtype=torch.bfloat16, generator=g)
        out = ops.gemm_bt(a, w, b)
        rows = min(M, 2048)
        ref = torch.nn.functional.linear(a[-rows:].float(), w.float(), b.float())
        err = float((out[-rows:].float() - ref).abs().max() / ref.abs().max())
        # f32 gated-residual epilogue (the other store path)
        resid = torch.zeros(M, N, device="cuda", dtype=torch.float32)
        gate = torch.randn(1, N, device="cuda", dtype=torch.float32, generator=g)
        ops.gemm_bt(a, w, b, out=resid, epilogue=ops.EPI_RESID_GATE, gate=gate, gate_stride=N, rows_per_sample=M)
        # no-bias and tanh-GELU stores (the two forms the persistent kernel also serves); repeated: a persistent workgroup's tile
        # hand-over races would not be deterministic
        nb_ = ops.gemm_bt(a, w, None)
        ge_ = ops.gemm_bt(a, w, b, epilogue=ops.EPI_GELU_TANH)
        for _ in range(3):
            if not (torch.equal(ops.gemm_bt(a, w, None), nb_) and torch.equal(ops.gemm_bt(a, w, b, epilogue=ops.EPI_GELU_TANH), ge_)
                    and torch.equal(ops.gemm_bt(a, w, b), out)):
                raise SystemExit(f"{name}: results differ between runs")
        digest = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes() + resid.cpu().numpy().tobytes()).hexdigest()[:16]
        digest += "/" + hashlib.sha1(nb_.view(torch.int16).cpu().numpy().tobytes() + ge_.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:8]
        for _ in range(5):
            ops.gemm_bt(a, w, b, out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(n):
            ops.gemm_bt(a, w, b, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        res[name] = dict(ms=round(ms, 4), tflops=round(2 * M * N * K / ms / 1e9, 1), relerr=err, digest=digest)
    print("RESULT " + json.dumps(res), flush=True)


def main():
    if "--child" in sys.argv:
        return child(int(sys.argv[sys.argv.index("--n") + 1]))
    variants = []
    for a in sys.argv[1:]:
        if not a[0].isdigit():
            break
        variants.append(a)
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
    n = sys.argv[sys.argv.index("--n") + 1] if "--n" in sys.argv else "40"
    table = {}
    for rep in range(reps):
        for v in variants:
            env = dict(os.environ, M4D_GEMM_VARIANT=v.split(":")[0])
            if ":" in v:
                env["M4D_GEMM_PERSIST"] = v.split(":")[1]
            if v.count(":") > 1:
                env["M4D_GEMM_SYNC"] = v.split(":")[2]
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--n", n], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(f"variant {v}: FAILED rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}", flush=True)
                continue
            d = json.loads(line[0][7:])
            table.setdefault(v, []).append(d)
            print(f"variant {v} rep {rep}: " + "  ".join(f"{k} {x['tflops']:.0f}TF err {x['relerr']:.1e} {x['digest']}" for k, x in d.items()), flush=True)
    print(json.dumps(table))


if __name__ == "__main__":
    main()
