"""Round-6 fixtures, produced by RUNNING THE REFERENCE in the build container (needs /root/reference):

    python tests/golden/make_golden_r6.py [subject_ref|guid_pertoken|all]

* dit_tiny_subject_ref.npz — the `subject_ref` branch of WanTransformer4DModel.forward (wan_transformer4d.py:1092-1097, 1328-1331): extra
  frames that go through `patch_embedding`, are appended BEHIND the video tokens (RoPE frame index continues) and are cut off after the
  head.  The dit_tiny.npz model and inputs plus subject_ref [B, 64, 2, 16, 16] (two frames = 128 tokens), with and without the reference row.
  Reference quirk recorded by this fixture: `subject_ref_length = subject_ref[0].size(1)` (:1329) is the model WIDTH (subject_ref is
  [B, Ls, dim] there), so the reference cuts `dim` tokens; unpatchify then takes the first prod(grid) tokens, which makes the result right
  whenever at least `dim` tokens follow the video tokens — true here (128 subject tokens, dim = 128).
* dit_tiny_guid_pertoken_grads.npz — spatial guidance (:757-783) TOGETHER with per-token timesteps (:655-657, t [B, seq_len]) in training:
  make_golden.py:make_dit_guid_grads (the reference's forward + backward with the stand-in feature extractor) re-run with the t_tok of
  dit_tiny_pertoken.npz; prediction, loss, norms + sampled values of every gradient.
Data only; no reference source is stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
from make_golden import TINY_DIT, load_recipe, npz_save  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def make_subject_ref(ref):
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, "dit_tiny_keys.json", seed=1234)
    z = dict(np.load(os.path.join(HERE, "dit_tiny.npz")))
    x, y, full_ref, clip, t = (torch.from_numpy(z[k]) for k in ("x", "y", "full_ref", "clip", "t"))
    ctx = [torch.from_numpy(z["ctx0"]), torch.from_numpy(z["ctx1"])]
    g = torch.Generator().manual_seed(61)
    subject_ref = torch.randn(x.shape[0], 64, 2, 16, 16, generator=g)
    L = 2 * 8 * 8
    out_ref = m(x=x, t=t, context=ctx, seq_len=L + 5, clip_fea=clip, y=y, full_ref=full_ref, subject_ref=subject_ref)
    out_noref = m(x=x, t=t, context=ctx, seq_len=L, clip_fea=clip, y=y, full_ref=None, subject_ref=subject_ref)
    base = torch.from_numpy(z["out_ref"])
    print("subject_ref moves the output by", float((out_ref - base).abs().max() / base.abs().max()))
    npz_save("dit_tiny_subject_ref.npz", subject_ref=subject_ref, seq_len_pad=np.int64(L + 5), seq_len=np.int64(L), out_ref=out_ref,
             out_noref=out_noref)


def make_guid_pertoken(ref):
    from make_golden import make_dit_guid_grads
    t_tok = torch.from_numpy(np.load(os.path.join(HERE, "dit_tiny_pertoken.npz"))["t_tok"])
    torch.set_grad_enabled(True)
    try:
        make_dit_guid_grads(ref, t_override=t_tok, out_name="dit_tiny_guid_pertoken_grads.npz")
    finally:
        torch.set_grad_enabled(False)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref = _ref_import.load_reference()
    if what in ("subject_ref", "all"):
        make_subject_ref(ref)
    if what in ("guid_pertoken", "all"):
        make_guid_pertoken(ref)
