"""Oracle: the geometry either side of the stage-1 sampler (scripts/inference/infer.py) — depth back-projection, the depth
control image, and the recovery of 3-D point trajectories from the decoded displacement video.  TEST INFRASTRUCTURE — see
oracle/__init__.py.  Pinned to tests/golden/pipeline_chain.npz (the reference's own functions run on seeded inputs).
"""
import torch
import torch.nn.functional as F

H_ORI, W_ORI = 540, 960          # infer.py:53


def intrinsics(H, W):
    """infer.py:161-176: normalised pinhole (principal point 0.5, 0.5); the longer relative side keeps focal 1."""
    if W_ORI / W > H_ORI / H:
        fx, fy = 1.0, W_ORI / H_ORI / (W / H)
    else:
        fy, fx = 1.0, H_ORI / W_ORI / (H / W)
    return fx, fy


def back_project_coords(depth, H, W):
    """infer.py:179-195: depth [h, w] -> bilinear (align_corners=False) to [H, W]; pixel grid u, v on linspace(0, 1);
    rays = K^-1 (u, v, 1) = ((u - .5)/fx, (v - .5)/fy, 1); points = rays * depth.  Returns [H, W, 3]."""
    d = F.interpolate(depth[None, None].float(), size=(H, W), mode="bilinear", align_corners=False)[0, 0]
    fx, fy = intrinsics(H, W)
    u = torch.linspace(0, 1, W)
    v = torch.linspace(0, 1, H)
    uu, vv = torch.meshgrid(u, v, indexing="xy")
    x = uu * (1.0 / fx) + (-0.5 / fx)
    y = vv * (1.0 / fy) + (-0.5 / fy)
    return torch.stack([x * d, y * d, d], dim=-1)


def depth_control_image(first_frame_coords):
    """infer.py:820-828: z of the back-projected first frame [1,3,1,H,W] -> 3 identical channels, clamped to [0, 1e4],
    inf / nan / < 1e-5 replaced by 1 (after the clamp: +inf is 1e4 by then), min-max normalised to [-1, 1]."""
    z = first_frame_coords[:, 2, :, :].unsqueeze(1).repeat(1, 3, 1, 1, 1)
    z = torch.clamp(z, min=0.0, max=10000.0)
    z = torch.where(torch.isinf(z) | torch.isnan(z) | (z < 1e-5), torch.ones_like(z), z)
    lo, hi = z.min(), z.max()
    return 2 * (z - lo) / (hi - lo + 1e-8) - 1


def recover_flow(rel_flow, first_frame_coords):
    """infer.py:198-219 (`inverse_flow_norm_transform_no_diff`): rel_flow [B,3,F,H,W] are displacements in units of the first
    frame's extent; diff = max over x,y,z of (max - min over pixels) of the first frame (0 -> 1);
    out = (rel + frame0 / diff) * diff.  Returns ([B,3,F,H,W], diff [B,3])."""
    B = rel_flow.shape[0]
    f0 = first_frame_coords[:, :, 0].float().expand(B, -1, -1, -1)          # [B,3,H,W]
    flat = f0.reshape(B, 3, -1)
    diff = (flat.max(dim=2).values - flat.min(dim=2).values).max(dim=1)[0]  # [B]
    diff = torch.where(diff == 0, torch.ones_like(diff), diff)
    d = diff.view(B, 1, 1, 1, 1)
    out = (rel_flow.float() + (f0 / diff.view(B, 1, 1, 1)).unsqueeze(2)) * d
    return out, diff.view(B, 1).repeat(1, 3)


def stage1_coords(rel_flow, first_frame_coords):
    """infer.py:870: the stored cloud = the first frame's coordinates followed by frames 1.. of the recovered flow."""
    flow, _ = recover_flow(rel_flow, first_frame_coords)
    return torch.cat([first_frame_coords.float().expand(flow.shape[0], -1, -1, -1, -1), flow[:, :, 1:]], dim=2)


def preprocess_image(x):
    """diffusers VaeImageProcessor.preprocess on a float tensor that already has the target size (third-party, restated:
    parity unpinned): [0, 1] -> [-1, 1]; a tensor with negative values is taken to be in [-1, 1] already and passed through."""
    return x if float(x.min()) < 0 else x * 2.0 - 1.0
