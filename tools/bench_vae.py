#!/usr/bin/env python
"""BASELINE.json configs[2]: Motion-Sensitive 3D-VAE encode+decode on a 49x480x832x3 trajectory tensor, 1 MI355X, bf16.
enc-adaptor -> *2-1 -> vae.encode (mode) -> vae.decode -> dec-adaptor.  Random-init weights (recipe), synthetic input.
    python tools/bench_vae.py [T H W]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(T=49, H=480, W=832, iters=2, dev="cuda", verbose=True):
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    dt = torch.bfloat16
    torch.manual_seed(0)
    vae = AutoencoderKLWan().eval()
    for n, p in vae.named_parameters():
        with torch.no_grad():
            if n.endswith("gamma"):
                p.fill_(1.0)
            elif p.dim() > 1:
                p.normal_(0, (p[0].numel()) ** -0.5)
            else:
                p.zero_()
    vae = vae.to(dev, dt)
    ea, da = VAEEncoderadaptor().eval(), VAEDecoderadaptor().eval()
    with torch.no_grad():
        ea.conv_out.weight.normal_(0, 0.02)
    ea, da = ea.to(dev, dt), da.to(dev, dt)
    traj = (torch.randn(1, T, H, W, 3, device=dev) * 0.1).permute(0, 4, 1, 2, 3).contiguous().to(dt)

    def step(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if verbose:
            print(f"{name}: {dt_*1e3:.1f} ms", flush=True)
        return out, dt_

    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        for it in range(iters):     # the last iteration is reported
            if verbose:
                print("iter", it)
            a, t1 = step("enc_adaptor", lambda: ea(traj))
            lat, t2 = step("vae_encode", lambda: vae.encode(a * 2 - 1)[0].mode())
            rec, t3 = step("vae_decode", lambda: vae.decode(lat).sample)
            out, t4 = step("dec_adaptor", lambda: da(rec))
    px = H * W
    flops = dict(enc=px * (6.657e6 + (T - 1) * 5.003e6), dec=px * (10.748e6 + (T - 1) * 8.445e6),
                 ea=11.8e12 * (T * px) / (49 * 480 * 832), da=23.4e12 * (T * px) / (49 * 480 * 832))
    return dict(shape=[T, H, W], ms=dict(enc_adaptor=t1 * 1e3, encode=t2 * 1e3, decode=t3 * 1e3, dec_adaptor=t4 * 1e3),
                roundtrip_ms=(t1 + t2 + t3 + t4) * 1e3,
                tflops=dict(encode=flops["enc"] / t2 / 1e12, decode=flops["dec"] / t3 / 1e12, enc_adaptor=flops["ea"] / t1 / 1e12,
                            dec_adaptor=flops["da"] / t4 / 1e12),
                finite=bool(torch.isfinite(out.float()).all()), peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)


if __name__ == "__main__":
    shape = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else [49, 480, 832]
    print(json.dumps(run(*shape)))
