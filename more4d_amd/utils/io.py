"""Host-side wire formats either side of the hot path: the resumable-sampler pickle of the training
save / load hooks (scripts/4D_STraG_training/train_wan.py:978-979, 983-988) and the per-frame point-cloud text dump +
coordinate recovery of the inference script (scripts/inference/infer.py:447-460, 857-871).  Plain Python / numpy: nothing
here touches the GPU path."""
import os
import pickle

import numpy as np
import torch

SAMPLER_FILE = "sampler_pos_start.pkl"


def save_sampler_state(output_dir, pos_start, first_epoch):
    """`[batch_sampler.sampler._pos_start, first_epoch]` pickled next to the weights (train_wan.py:978-979)."""
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, SAMPLER_FILE), "wb") as fh:
        pickle.dump([int(pos_start), int(first_epoch)], fh)


def load_sampler_state(input_dir, dataloader_num_workers=0, num_processes=1):
    """Returns (resume position, saved epoch) or None.  The position is moved back by the samples the workers may have
    prefetched: max(saved - workers * processes * 2, 0) (train_wan.py:983-988)."""
    path = os.path.join(input_dir, SAMPLER_FILE)
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fh:
        loaded_number, epoch = pickle.load(fh)
    return max(int(loaded_number) - dataloader_num_workers * num_processes * 2, 0), epoch


def recover_coords(recon_video, first_frame_coords):
    """`--normalize_track_z` branch of infer.py:857-861: the decoded trajectory video [1, 3, F, H, W] holds per-frame
    displacements; adding the first frame's 3-D coordinates [1, 3, F0, H, W] (frame 0) gives [1, 3, F, H, W] points.  The
    stored cloud is frame 0's coordinates followed by frames 1.. of the recovered flow (:870)."""
    flow = recon_video.float().cpu() + first_frame_coords[0, :, 0].unsqueeze(0).unsqueeze(2).float().cpu()
    return torch.cat([first_frame_coords.float().cpu(), flow[:, :, 1:]], dim=2)


def image_colors(image):
    """[-1, 1] image [B, 3, H, W] -> uint8 colours [B, H*W, 3] (infer.py:865-868)."""
    color = (image + 1) / 2
    color = color.reshape(color.shape[0], 3, -1).permute(0, 2, 1)
    return (color * 255).clamp(0, 255).to(torch.uint8)


def save_pointcloud_data(recon_flow, colors, video_name, output_dir, seed):
    """One `<output_dir>/pts/seed_<seed>/<video>_frame_%04d.txt` per frame, rows `x y z r g b` in numpy.savetxt's default
    format (infer.py:447-460).  recon_flow [B, 3, F, H, W] (sample 0 is written), colors [B, H*W, 3]."""
    pts_dir = os.path.join(output_dir, "pts", f"seed_{seed}")
    os.makedirs(pts_dir, exist_ok=True)
    B, C, F, H, W = recon_flow.shape
    files = []
    for frame_idx in range(F):
        coords = recon_flow[0, :, frame_idx].permute(1, 2, 0).reshape(-1, 3)
        data = torch.cat([coords.cpu().float(), colors[0].reshape(-1, 3).cpu().float()], dim=1)
        path = os.path.join(pts_dir, f"{video_name}_frame_{frame_idx:04d}.txt")
        np.savetxt(path, data.numpy())
        files.append(path)
    return files
