"""Round-5 host logic (no GPU)."""
import torch


def test_fused_qk_weights_are_views_not_copies():
    """the fused [Wq; Wk] operand of the self-attention must not be a second resident copy that can go stale.  The
    parameters become the two halves of ONE storage; in-place writes through either side are seen by the other; a dtype / device move
    re-fuses; parameters that live in someone else's flat buffer (dist/data_parallel buckets) are left alone."""
    from more4d_amd.models.wan_transformer4d import WanSelfAttention
    m = WanSelfAttention(64, 4)
    keys = sorted(m.state_dict())
    w, b = m._qk_weights()
    assert m.q.weight.data_ptr() == w.data_ptr() and m.k.weight.data_ptr() == w.data_ptr() + 64 * 64 * 4
    assert m.q.bias.data_ptr() == b.data_ptr() and m.k.bias.data_ptr() == b.data_ptr() + 64 * 4
    assert sorted(m.state_dict()) == keys and m.state_dict()["k.weight"].shape == (64, 64)
    m.q.weight.data.add_(1.0)                       # a raw write that does not bump the version counter
    m.k.bias.data.fill_(3.0)
    w2, b2 = m._qk_weights()
    assert w2 is w and torch.equal(w[:64], m.q.weight.detach()) and float(b2[64:].mean()) == 3.0
    m.load_state_dict({k: torch.full_like(v, 2.0) for k, v in m.state_dict().items()})
    assert m._qk_weights()[0] is w and float(w.mean()) == 2.0
    m = m.to(torch.bfloat16)                        # new storage: re-fused, re-pointed
    w3, _ = m._qk_weights()
    assert w3.dtype == torch.bfloat16 and m.k.weight.data_ptr() == w3.data_ptr() + 64 * 64 * 2
    flat = torch.zeros(64 * 64 * 3)
    m2 = WanSelfAttention(64, 4)
    m2.q.weight.data = flat[:4096].view(64, 64)
    assert m2._qk_weights() == (None, None) and m2.q.weight.data_ptr() == flat.data_ptr()
    m._qk_weights()
    m.q.weight.data = flat[:4096].view(64, 64).bfloat16()       # (own storage again)
    assert m._qk_weights()[0] is not w3


def test_pipeline_start_image_goes_into_frame_zero_of_the_second_control_group(monkeypatch):
    """start_image (reference pipeline_wan_fun_control.py:664-685, :773-777): resized / normalised like a control video, VAE-encoded, its
    latent written into frame 0 of the second 16-channel group of y; without it that group is zeros.  control_camera_video still
    raises: the adapter it feeds is undefined in the reference itself (wan_transformer4d.py:941)."""
    import pytest
    from types import SimpleNamespace
    import more4d_amd.pipeline.pipeline_wan_fun_control as pm
    seen = {}

    class Vae:
        spatial_compression_ratio, temporal_compression_ratio, latent_channels = 8, 4, 16
        dtype = torch.float32
        config = SimpleNamespace(latent_channels=16)

        def encode(self, v):
            seen.setdefault("encoded", []).append(v.clone())
            lat = v[:, :1].mean(dim=(3, 4), keepdim=True).expand(-1, 16, -1, -1, -1)      # [B, 16, f, 1, 1]: a recognisable "latent"
            f = (v.shape[2] - 1) // 4 + 1
            lat = lat[:, :, :f].expand(-1, -1, -1, v.shape[3] // 8, v.shape[4] // 8).contiguous()
            return [SimpleNamespace(mode=lambda: lat)]

    class Dit:
        dtype = torch.float32
        config = {"add_ref_conv": False}

        def prepare_context(self, ctx, clip):
            return "cc"

    def fake_loop(model, sch, lat, ts, gs, cc, y=None, full_ref=None, first_frame_features=None):
        seen["y"] = y
        return lat
    monkeypatch.setattr(pm, "denoise_latents", fake_loop)
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler
    pipe = pm.WanFunControlPipeline(vae=Vae(), transformer=Dit(), scheduler=FlowDPMSolverMultistepScheduler(solver_order=1, shift=1.0))
    monkeypatch.setattr(type(pipe), "_execution_device", property(lambda self: torch.device("cpu")), raising=False)
    pe = [torch.zeros(4, 8)]
    kw = dict(height=32, width=48, num_frames=9, num_inference_steps=2, guidance_scale=1.0, prompt_embeds=pe, negative_prompt_embeds=pe,
              output_type="latent")
    start = torch.rand(1, 3, 1, 40, 60)                       # [0, 1], another size: resized to 32 x 48 and mapped to [-1, 1]
    pipe(start_image=start, **kw)
    y = seen["y"]
    assert y.shape == (1, 32, 3, 4, 6)
    enc = seen["encoded"][-1]
    assert enc.shape == (1, 3, 1, 32, 48) and float(enc.min()) < 0
    assert torch.equal(y[:, :16], torch.zeros_like(y[:, :16]))                     # no control video
    assert torch.allclose(y[:, 16:, 0], enc[:, :1, 0].mean(dim=(2, 3), keepdim=True).expand(-1, 16, 4, 6))
    assert torch.equal(y[:, 16:, 1:], torch.zeros_like(y[:, 16:, 1:]))
    seen.clear()
    pipe(**kw)
    assert torch.equal(seen["y"], torch.zeros(1, 32, 3, 4, 6))
    with pytest.raises(NotImplementedError):
        pipe(control_camera_video=torch.zeros(1, 6, 9, 4, 6), **kw)
