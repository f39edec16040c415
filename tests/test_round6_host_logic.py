"""Round-6 host-side checks (no GPU)."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gen,inc", [("gen_attn_q64.py", "attention_q64_gen.inc"), ("gen_attn_bwd64.py", "attention_bwd64_dq_gen.inc"),
                                     ("gen_attn_bwd64_kv.py", "attention_bwd64_kv_gen.inc"), ("gen_conv_halo64.py", "conv_halo64_gen.inc"),
                                     ("gen_conv_halo64.py --shape 1x4x4", "conv_halo64k1_gen.inc")])
def test_generated_streams_match_their_generators(gen, inc):
    """the committed .inc files (hashed into the library) are what the generators emit (nothing checked that), and the
    generators' --help works"""
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "x.inc")
        gen, *flags = gen.split()
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "-o", outp] + flags, check=True, capture_output=True)
        a = [l for l in open(outp) if not l.startswith("//")]
        b = [l for l in open(os.path.join(ROOT, "more4d_amd", "csrc", inc)) if not l.startswith("//")]
        assert a == b, inc
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage" in r.stdout.lower(), r.stderr


@pytest.mark.parametrize("flags,reads,loop_mfmas", [([], 8 + 8 + 2 * 27 * 8, 810), (["--ahead", "2"], 8 + 16 + 27 * 8, 405), (["--window"], 8 + 10 + 2 * 3 * 48, 810),
                                                    (["--gather"], 8 + 8 + 2 * 27 * 8, 810)])
def test_conv_halo64_generator_variants(flags, reads, loop_mfmas):
    """the measured-and-not-kept forms of the conv stream stay generable: fragments two taps ahead, the sliding window of row
    fragments (21 instead of 45 pixel-fragment reads per frame: 48 reads per nine taps), plain-order weights; every form issues the
    same 15 MFMAs per tap"""
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "x.inc")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_conv_halo64.py"), "-o", outp] + flags, check=True, capture_output=True)
        txt = open(outp).read()
    assert txt.count("ds_read_b128") == reads, txt.count("ds_read_b128")
    assert txt.count("v_mfma_f32_32x32x16_bf16") == loop_mfmas
