#!/usr/bin/env python
"""conv_halo64_kernel (M4D_CONV_HALO64=2: planar-16 AND channels-last inputs; tiled weights) against conv_halo_kernel<3,3,12,32,3,3> (=0)
with plain and with tiled weights (ops.conv_pack_weights) on the same inputs — the kernels share the accumulation order and the epilogue
source, so every output must agree BIT FOR BIT — and against an fp32 reference; the pack kernel against a torch restatement of the tiled
order; then timing at the VAE's dominant shape (96 -> 96 channels, 480 x 832, 4 output frames).
    python tools/check_conv64.py [--time]"""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda"
BF = torch.bfloat16


def planar(x_cl, T, H, W, C):
    """[T*H*W, C] channels-last -> Planar16 view [C/16, T, H*W, 16]"""
    from more4d_amd import ops
    t = x_cl.view(T, H * W, C // 16, 16).permute(2, 0, 1, 3).contiguous()
    return ops.Planar16(t)


def tiled_ref(w, Cin):
    """torch restatement of m4d_conv_pack_weights: [Cout, taps * Cin] -> [ceil(Cout / 32)][Cin / 16][taps][32 rows][2 halves][8], rows past
    Cout - 1 repeat the last row, half h of row r holds channels 8 * (h ^ ((r >> 3) & 1)) + [0, 8) of the chunk"""
    Cout, K = w.shape
    taps, nrb = K // Cin, (Cout + 31) // 32
    rows = torch.arange(nrb * 32, device=w.device).clamp(max=Cout - 1)
    wv = w[rows].view(nrb, 32, taps, Cin // 16, 2, 8).clone()
    sw = ((torch.arange(32, device=w.device) >> 3) & 1).bool()
    wv[:, sw] = wv[:, sw].flip(-2)
    return wv.permute(0, 3, 2, 1, 4, 5).contiguous().view(-1)


TILED = False


def case(name, T, H, W, Cin, Cout, layout, bias=True, resid=False, norm=False, keep_raw=True, seed=0):
    from more4d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(T * H * W, Cin, device=DEV, generator=g).to(BF)
    w = (torch.randn(Cout, 27 * Cin, device=DEV, generator=g) * (27 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=DEV, generator=g).to(BF) if bias else None
    To = T - 2
    M = To * H * W
    r = torch.randn(M, Cout, device=DEV, generator=g).to(BF) if resid else None
    ops.launch_counts(reset=True)
    outs = []
    w_ref = w
    wt = None
    if TILED:
        wt = ops.conv_pack_weights(w, Cin)
        if not torch.equal(wt.view(torch.int16), tiled_ref(w, Cin).view(torch.int16)):
            print(f"{name}: conv_pack_weights differs from the torch restatement", flush=True)
            return True, []
    if layout == "cl":
        out = ops.conv_cl(x, w, b, Tin=T, Hin=H, Win=W, Cin=Cin, k=(3, 3, 3), pad=(0, 1, 1), out_thw=(To, H, W), resid=r, w_tiled=wt)
        outs.append(out)
    else:
        xp = planar(x, T, H, W, Cin)
        nrm = None
        if norm:
            gamma = torch.rand(Cout, device=DEV, generator=g) + 0.5
            dst = ops.Planar16(torch.full((Cout // 16, To, H * W, 16), float("nan"), device=DEV, dtype=BF))
            nrm = (gamma, dst, True)
        out = ops.conv_cl_planar(xp, w, b, Tin=T, Hin=H, Win=W, kt=3, resid=r, norm=nrm, keep_raw=keep_raw, w_tiled=wt)
        if out is not None:
            outs.append(out)
        if norm:
            outs.append(nrm[1].t)
    torch.cuda.synchronize()
    cnt = {k: v for k, v in ops.launch_counts().items() if v}
    # fp32 reference of the raw conv (+ bias + resid)
    # (27 shifted fp32 GEMMs instead of torch's conv3d: no MIOpen search on a fresh box)
    xf = torch.zeros(T, H + 2, W + 2, Cin, device=DEV)
    xf[:, 1:-1, 1:-1] = x.float().view(T, H, W, Cin)
    wf = w_ref.float().view(Cout, 3, 3, 3, Cin)
    ref = torch.zeros(To, H, W, Cout, device=DEV)
    for dt in range(3):
        for dh in range(3):
            for dw in range(3):
                ref += xf[dt:dt + To, dh:dh + H, dw:dw + W] @ wf[:, dt, dh, dw].t()
    if bias:
        ref += b.float()
    ref = ref.reshape(M, Cout)
    msg = ""
    bad = False
    if keep_raw or not norm:
        raw = outs[0].float()
        want = ref.to(BF).float()
        if resid:
            want = (want + r.float())
        e = float((raw - want).abs().max() / want.abs().max())
        bad |= not e < 1.2e-2
        msg = f"raw err {e:.3e}"
    if os.environ.get("CONV64_DEBUG") and (keep_raw or not norm):
        d = (outs[0].float() - want).abs().view(To, H, W, Cout)
        badm = d > 0.05 * float(want.abs().max())
        print("  bad fraction", float(badm.float().mean()))
        print("  bad by h%10:", [round(float(badm[:, i::10].float().mean()), 4) for i in range(10)])
        print("  bad by w%32:", [round(float(badm[:, :, i::32].float().mean()), 3) for i in range(32)])
        print("  bad by c//32:", [round(float(badm[..., i * 32:(i + 1) * 32].float().mean()), 4) for i in range(Cout // 32)])
        print("  bad by c%32:", [round(float(badm[..., i::32].float().mean()), 3) for i in range(32)])
        print("  bad by t:", [round(float(badm[i].float().mean()), 4) for i in range(To)])
        print("  bad by h (first 24):", [round(float(badm[:, i].float().mean()), 3) for i in range(min(24, H))])
    dig = " ".join(f"{float(o.float().nan_to_num().abs().double().sum()):.10e}" for o in outs)
    fin = all(bool(torch.isfinite(o.float()).all()) for o in outs)
    bad |= not fin
    print(f"{name}: {msg} finite {fin} digest {dig}", cnt, "FAIL" if bad else "ok", flush=True)
    return bad, [o.cpu() for o in outs]


# (shapes on which the dispatcher takes the three-tiles-per-wave 12 x 32 kernel when M4D_CONV_HALO64=0: rows divisible by 12 and by 10,
#  more than 256 workgroups unless the next layer's norm is fused)
CASES = (
    ("cl_96_96", 4, 120, 416, 96, 96, "cl"),
    ("cl_96_192_resid", 3, 120, 416, 96, 192, "cl", True, True),
    ("cl_ragged_edge", 4, 120, 400, 48, 96, "cl", False),               # 400 columns: a partial 32-column patch; Cin = 48 (3 chunks)
    ("planar_96_96", 4, 120, 416, 96, 96, "planar"),
    ("planar_norm_raw", 4, 60, 64, 96, 96, "planar", True, False, True, True),
    ("planar_norm_resid_noraw", 3, 60, 32, 192, 96, "planar", True, True, True, False),
    ("planar_384_384", 3, 120, 224, 384, 384, "planar", True, True),
    ("planar_384_384_208_columns", 3, 120, 208, 384, 384, "planar", True, True),   # 6.5 patches wide: the 24 x 16 kernel's map (M4D_CONV_HALO64_NARROW)
)


def main():
    global TILED
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        mode, tiled, what = sys.argv[2], sys.argv[3] == "1", sys.argv[4]
        TILED = tiled
        os.environ["M4D_CONV_HALO64"] = mode
        from more4d_amd import ops
        if what == "time":
            T, H, W, C = 6, 480, 832, 96
            g = torch.Generator(device=DEV).manual_seed(0)
            x = torch.randn(T * H * W, C, device=DEV, generator=g).to(BF)
            w = (torch.randn(C, 27 * C, device=DEV, generator=g) * (27 * C) ** -0.5).to(BF)
            b = torch.randn(C, device=DEV, generator=g).to(BF)
            wt = ops.conv_pack_weights(w, C) if tiled else None
            xp = planar(x, T, H, W, C)
            out = torch.empty((T - 2) * H * W, C, device=DEV, dtype=BF)
            for lay in ("planar", "cl"):
                def run():
                    if lay == "planar":
                        ops.conv_cl_planar(xp, w, b, Tin=T, Hin=H, Win=W, kt=3, out=out, w_tiled=wt)
                    else:
                        ops.conv_cl(x, w, b, Tin=T, Hin=H, Win=W, Cin=C, k=(3, 3, 3), pad=(0, 1, 1), out_thw=(T - 2, H, W), out=out, w_tiled=wt)
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
                ops.launch_counts(reset=True)
                from bench import ClockMonitor
                n = 40
                for _ in range(10 * n):          # (long enough for the clock samples: the board settles on its power limit)
                    run()
                torch.cuda.synchronize()
                mon = ClockMonitor(0).start()
                t0 = time.perf_counter()
                for _ in range(50 * n):
                    run()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / (50 * n) * 1e3
                mon.stop()
                ck = mon.region()
                fl = 2.0 * (T - 2) * H * W * C * 27 * C
                kern = "+".join(k for k, v in ops.launch_counts().items() if v and k.startswith("conv_halo"))
                print(f"halo64={mode} tiled={int(tiled)} {lay} [{kern}]: {ms:.4f} ms  {fl / ms / 1e9:.0f} TF  frac {fl / ms / 1e9 / 2500:.3f}  "
                      f"clock {ck.get('effective_clock_mhz') or 0:.0f} MHz power {(ck.get('socket_power_w') or {}).get('mean') or 0:.0f} W  "
                      f"digest {float(out.float().abs().double().sum()):.10e}", flush=True)
            return
        bad = False
        outs = []
        for c in CASES:
            b_, o = case(*c)
            bad |= b_
            outs.append(o)
        torch.save(outs, f"/tmp/conv64_mode{mode}{int(tiled)}.pt")
        print("RESULT halo64", mode, "tiled", int(tiled), "FAIL" if bad else "PASS", flush=True)
        sys.exit(1 if bad else 0)
    rc = 0
    runs = (("0", "0"), ("0", "1"), ("2", "1"))       # (M4D_CONV_HALO64, tiled weights)
    for mode, tiled in runs:
        try:
            rc |= subprocess.run([sys.executable, __file__, "--child", mode, tiled, "check"], timeout=180).returncode
        except subprocess.TimeoutExpired:
            print("halo64", mode, "tiled", tiled, "TIMED OUT", flush=True)
            rc |= 1
    try:
        a = torch.load("/tmp/conv64_mode00.pt")
        for mode, tiled in runs[1:]:
            b = torch.load(f"/tmp/conv64_mode{mode}{tiled}.pt")
            for (nm, *_), x, y in zip(CASES, a, b):
                same = len(x) == len(y) and len(x) > 0 and all(torch.equal(p.view(torch.int16), q.view(torch.int16)) for p, q in zip(x, y))
                print(f"{nm}: halo64={mode} tiled={tiled} vs the two-wave kernel on plain weights:", "bit-identical" if same else "DIFFERENT")
                rc |= int(not same)
    except Exception as ex:      # noqa: BLE001
        print("compare failed:", ex)
        rc |= 1
    if "--time" in sys.argv:
        for rnd in range(2):
            for mode, tiled in runs:
                try:
                    subprocess.run([sys.executable, __file__, "--child", mode, tiled, "time"], timeout=120)
                except subprocess.TimeoutExpired:
                    print("halo64", mode, "tiled", tiled, "time TIMED OUT", flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
