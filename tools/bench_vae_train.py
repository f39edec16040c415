#!/usr/bin/env python
"""One train_vae.py step (scripts/4D_STraG_training/train_vae.py:434-495, `--finetune_vae_decoder`) at 49x480x832, bf16, 1 MI355X:
encoder adaptor -> *2-1 -> encode (no grad, :444-448) -> sample -> decode_memory_saver (grad) -> decoder adaptor -> L1 + 1e-6 KL ->
backward -> clip -> AdamW over decoder prompt + VAE decoder.  Random-init weights, synthetic trajectories.
    python tools/bench_vae_train.py [T H W] [--steps N] [--through-encoder]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(T=49, H=480, W=832, steps=1, warmup=1, dev="cuda", through_encoder=False):
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    from more4d_amd.optim import AdamW, clip_grad_norm_
    dt = torch.bfloat16
    torch.manual_seed(0)
    vae = AutoencoderKLWan()
    with torch.no_grad():
        for n, p in vae.named_parameters():
            if n.endswith("gamma"):
                p.fill_(1.0)
            elif p.dim() > 1:
                p.normal_(0, (p[0].numel()) ** -0.5)
            else:
                p.zero_()
    vae = vae.to(dev, dt)
    ea, da = VAEEncoderadaptor(), VAEDecoderadaptor()
    with torch.no_grad():
        ea.conv_out.weight.normal_(0, 0.02)
    ea, da = ea.to(dev, dt), da.to(dev, dt)
    ea.requires_grad_(True).train()
    da.requires_grad_(True).train()
    vae.model.encoder.requires_grad_(False).eval()
    vae.model.conv1.requires_grad_(False)
    vae.model.decoder.requires_grad_(True).train()
    params = list(ea.parameters()) + list(da.parameters()) + list(vae.model.decoder.parameters())
    opt = AdamW(params, lr=5e-6, weight_decay=1e-2)
    coords = (torch.randn(1, T, H, W, 3, device=dev) * 0.1).permute(0, 4, 1, 2, 3).contiguous()
    targets = (coords - coords[:, :, 0:1]).to(dt)
    times = {}

    def mark(name, t0):
        torch.cuda.synchronize()
        times[name] = times.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    def step():
        t0 = time.perf_counter()
        pseudo = ea(targets) * 2 - 1
        t0 = mark("enc_adaptor_fwd", t0)
        if through_encoder:
            posterior = vae.encode_memory_saver(pseudo).latent_dist
        else:
            with torch.no_grad():
                posterior = vae.encode_memory_saver(pseudo).latent_dist
        lat = posterior.sample()
        t0 = mark("encode_fwd", t0)
        recon = vae.decode_memory_saver(lat).sample
        t0 = mark("decode_fwd", t0)
        rec2 = da(recon)
        t0 = mark("dec_adaptor_fwd", t0)
        rec_loss = (rec2.float() - targets.float()).abs()
        loss = rec_loss.sum() / rec_loss.shape[0] + 1e-6 * posterior.kl().sum()
        t0 = mark("loss", t0)
        loss.backward()
        t0 = mark("backward", t0)
        clip_grad_norm_(params, 1.0, optimizer=opt)
        opt.step()
        opt.zero_grad()
        mark("optimizer", t0)
        return loss

    for _ in range(warmup):
        step()
    times.clear()
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt_ = (time.perf_counter() - t0) / steps
    px = H * W
    dec = px * (10.748e6 + (T - 1) * 8.445e6)
    da_f = 23.4e12 * (T * px) / (49 * 480 * 832)
    # decoder + decoder prompt: forward + recompute + data gradient + weight gradient = 4x forward FLOPs; encoder side forward only
    flops = 4 * (dec + da_f) + px * (6.657e6 + (T - 1) * 5.003e6) + 11.8e12 * (T * px) / (49 * 480 * 832)
    return dict(metric="train_vae.py step seconds (--finetune_vae_decoder), 49x480x832 bf16", value=dt_, unit="s/step", shape=[T, H, W],
                parts_ms={k: v / steps * 1e3 for k, v in times.items()}, loss=float(loss.detach()),
                model_tflop=flops / 1e12, mfma_frac=flops / dt_ / 1e12 / 2500.0, through_encoder=through_encoder,
                peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    shape = [int(a) for a in args[:3]] if len(args) >= 3 else [49, 480, 832]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
    print(json.dumps(run(*shape, steps=steps, through_encoder="--through-encoder" in sys.argv)))
