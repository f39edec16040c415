"""CPU oracle for the 4D-STraG denoising hot path (TEST INFRASTRUCTURE, not product).

A plain fp32 torch/numpy restatement of the reference's algorithm for the path named in
BASELINE.json (Wan2.1-DiT forward, Motion-Sensitive 3D-VAE encode/decode, flow-matching
Euler denoise loop with CFG).  Every function cites the reference file:line it follows.

Rules:
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
    package; the product package `more4d_amd` never does;
  * parity is PINNED: tests/golden/*.npz were produced by importing the reference itself
    in the build container (tests/golden/make_golden.py) and `tests/test_oracle_golden.py`
    checks this restatement against them.  The diffusers boundary
    (FlowMatchEulerDiscreteScheduler, DiagonalGaussianDistribution) is third-party code
    absent from /root/reference ("diffusers>=0.30.1", unpinned): those two pieces are
    restated from the published algorithm and are "parity unpinned" — the Euler/sigma
    schedule is instead pinned to the in-tree order-1 FlowDPMSolverMultistepScheduler.
"""
