import sys, torch, time
sys.path.insert(0, "/root/repo")
from more4d_amd.models.wan_vae import AutoencoderKLWan
torch.manual_seed(0)
vae = AutoencoderKLWan().eval()
with torch.no_grad():
    for n, p_ in vae.named_parameters():
        if n.endswith("gamma"): p_.fill_(1.0)
        elif p_.dim() > 1: p_.normal_(0, (p_[0].numel()) ** -0.5)
        else: p_.zero_()
vae = vae.to("cuda", torch.bfloat16)
x = (torch.randn(1, 3, 13, 720, 1280, device="cuda") * 0.3).bfloat16()
with torch.no_grad():
    z = vae.encode(x)[0].mode(); torch.cuda.synchronize(); t0 = time.time()
    y = vae.decode(z).sample; torch.cuda.synchronize()
    part = vae.decode(z[:, :, :2].contiguous()).sample
print("720p", tuple(z.shape), tuple(y.shape), bool(torch.isfinite(y.float()).all()), "prefix equal:", bool(torch.equal(part, y[:, :, :5])), f"decode {time.time()-t0:.2f}s", "chunk", vae._chunk_latent(720, 1280))
