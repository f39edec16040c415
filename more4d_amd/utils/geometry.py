"""Stage-1 geometry on the device (scripts/inference/infer.py): what sits directly in front of the pipeline's VAE encodes and
directly behind the decoder prompt.  Same function names and argument meaning as the inference script; every tensor-sized
step is a HIP kernel (csrc/geometry.hip) — no `.cpu().numpy()` round trip between the depth model, the sampler and the point
cloud.  Pinned to the reference's own functions through tests/golden/pipeline_chain.npz."""
import torch

from .. import ops

DEFAULT_H_ORI, DEFAULT_W_ORI = 540, 960         # the capture resolution the normalised intrinsics refer to (infer.py:53)


def get_intrinsics(H, W):
    """(fx, fy) of the normalised pinhole K = [[fx,0,.5],[0,fy,.5],[0,0,1]] (infer.py:161-176)."""
    if DEFAULT_W_ORI / W > DEFAULT_H_ORI / H:
        return 1.0, DEFAULT_W_ORI / DEFAULT_H_ORI / (W / H)
    return DEFAULT_H_ORI / DEFAULT_W_ORI / (H / W), 1.0


def get_intrinsic_matrix(H, W, device):
    fx, fy = get_intrinsics(H, W)
    return torch.tensor([[fx, 0, 0.5], [0, fy, 0.5], [0, 0, 1]], dtype=torch.float32, device=device)


def _resized_depth(depth_map, H, W):
    d = depth_map.float().contiguous()
    if tuple(d.shape) != (H, W):
        d = ops.bilinear_cl(d.view(1, d.shape[0], d.shape[1], 1), (H, W)).view(H, W)      # align_corners=False (:181-182)
    return d


def back_project_coords(depth_map, H, W, device=None):
    """depth [h, w] -> 3-D points [H, W, 3] (infer.py:179-195)."""
    dev = device if device is not None else depth_map.device
    fx, fy = get_intrinsics(H, W)
    coords, _ = ops.backproject(_resized_depth(depth_map.to(dev), H, W), 1.0 / fx, 1.0 / fy)
    return coords.permute(1, 2, 0)


def depth_conditioning(depth_map, H, W, dtype=torch.float32):
    """infer.py:820-828 in one go: depth [h, w] -> (first_frame_coords float32 [1,3,1,H,W], depth_pixel_values `dtype`
    [1,3,1,H,W] in [-1,1]) — the two tensors `process_stage1_sample` derives from the depth model's output."""
    fx, fy = get_intrinsics(H, W)
    coords, zc = ops.backproject(_resized_depth(depth_map, H, W), 1.0 / fx, 1.0 / fy)
    dpv = ops.depth_control(zc, ops.minmax(zc, 1), dtype)
    return coords.view(1, 3, 1, H, W), dpv.view(1, 3, 1, H, W)


def inverse_flow_norm_transform_no_diff(rel_flow, first_frame_coords):
    """Decoded displacement video [B,3,F,H,W] + first-frame coordinates [1 or B,3,1,H,W] -> (recovered points [B,3,F,H,W],
    diff [B,3]) (infer.py:198-219), frame 0 included: (rel[:, :, 0] + f0 / diff) * diff like the reference (its caller drops that frame
    for the first-frame coordinates, :870 — `recover_stage1_coords` below).  The per-channel min / max come from `m4d_minmax` (fminf /
    fmaxf: a NaN coordinate is skipped, where torch.min / max would propagate it)."""
    B, _, F, H, W = rel_flow.shape
    f0 = first_frame_coords[:, :, 0].to(rel_flow.device, torch.float32).expand(B, 3, H, W).contiguous()
    mm = ops.minmax(f0, B * 3)
    out = ops.flow_recover(rel_flow, f0, mm, first_frame="recovered")
    ext = (mm[:, 1] - mm[:, 0]).view(B, 3).max(dim=1).values
    diff = torch.where(ext == 0, torch.ones_like(ext), ext)
    return out, diff.view(B, 1).repeat(1, 3)


def recover_stage1_coords(recon_video, first_frame_coords, normalize_track_z=False):
    """The `coords_data` of infer.py:857-870: first-frame coordinates followed by the recovered trajectory frames 1.. ."""
    B, _, F, H, W = recon_video.shape
    f0 = first_frame_coords[:, :, 0].to(recon_video.device, torch.float32).expand(B, 3, H, W).contiguous()
    if normalize_track_z:
        return ops.flow_recover(recon_video, f0, None, track_z=True)
    return ops.flow_recover(recon_video, f0, ops.minmax(f0, B * 3))
