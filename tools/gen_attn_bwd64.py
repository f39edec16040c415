#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of the one-wave-per-SIMD flash-attention backward passes
(more4d_amd/csrc/attention_bwd64.h: attn_bwd_dq64_kernel).

    python tools/gen_attn_bwd64.py [--plain] [--cap N] [-o more4d_amd/csrc/attention_bwd64_dq_gen.inc]

Same construction as tools/gen_attn_q64.py (the forward): the kernel body is ONE inline-asm block with hand-assigned registers, a
workgroup = 4 waves = one wave per SIMD, each wave owns 64 X rows (two halves of 32) and the SIMD's whole 512-register file; the
elementwise arithmetic is a list of fillers placed into the issue shadows of one continuous MFMA stream.

dQ pass (X = queries: Q, dO fragments resident in AGPRs; Y = keys: K, V tiles streamed through LDS), per 32-key UNIT u of a 64-key tile:
    SG(u):  S'^T = K(u) Q~^T - lse      G'^T = V(u) dO^T - delta        32 MFMAs (2 products x 2 query halves x 8 k-steps, 4 chains)
    E(u):   P = exp2(S')                                                32 v_exp_f32 per lane
    M(u):   dS = P G' -> bf16 pairs, IN PLACE in the first 8 registers of each G' block        32 v_mul_f32 + 16 v_cvt_pk_bf16_f32
    dQ(u):  dQ^T += K(u)^T dS^T   (K^T fragments by ds_read_b64_tr_b16 out of the row-major K tile)   16 MFMAs (8 chains)
-lse and -delta are lane constants (lane = query) and enter as the C operand of the first k-step (16-register tuples), so a score
costs exp + multiply + 1/2 pack = 2.5 VALU (the two-waves-per-SIMD kernel: 4.5 and two v_fma).  The softmax scale rides in Q~ (the
caller folds it into q's RMSNorm weight: sc = 1, Q~ = Q's bits) and the gradient's factor `scale` is applied to the accumulators in
the epilogue (dS is linear in it).  S' / G' are double-buffered per unit (X = u & 1), which makes the stream
    iteration u:   [SG(u+1) || M(u)]  [dQ(u) || E(u+1)]                 48 MFMAs, 80 VALU
K / V tiles (row-major, 16 KiB each, XOR-swizzled on the DMA source address): four 32 KiB LDS stages, tile t in stage t & 3, tile t + 3
requested while tile t is computed, one s_barrier per tile behind s_waitcnt vmcnt(8).  Fragments arrive through an 8-slot ring of
VGPRs, read 8 fragments ahead, every fragment feeds the two query halves; the lgkmcnt in front of every fragment's first MFMA is
computed from the reads issued behind it.
Lane tables (fragment addresses, DMA lane offsets, row offsets) are computed by the C++ wrapper and handed over through LDS.
"""
import argparse
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--plain", action="store_true", help="debug: no interleaving (MFMA stream of a phase, then its fillers)")
ap.add_argument("--cap", type=int, default=5, help="instructions per MFMA gap besides the MFMA")
ap.add_argument("--first-gap", type=int, default=1, help="dQ phase: no fillers behind its first N MFMAs (the first exps read S' accumulators)")
ap.add_argument("--greedy", action="store_true", help="A/B: pour the fillers into the first gaps up to the cap instead of spreading them over the phase")
ap.add_argument("--ahead", type=int, default=8, help="fragments in flight (ring slots: 8)")
ap.add_argument("-o", default="more4d_amd/csrc/attention_bwd64_dq_gen.inc")
args = ap.parse_args()

STAGE, VOFF, RAGOFF, TABOFF = 32768, 16384, 4 * 32768, 3 * 32768

# ---------------- register map ----------------
# VGPRs (v0..v3 stay with the compiler: the work-item id)
RA = [4 + i for i in range(8)]                  # row-fragment addresses, k-step kk (K at +0, V at +VOFF, 32-row half at +8192)
TA = [[12 + j * 4 + d for d in range(4)] for j in range(2)]      # transposing-read addresses [jj][d-block] (+ half * 8192 + chunk * 4096)
DK = [20 + p for p in range(4)]                 # DMA lane offsets, K pieces
DV = [24 + p for p in range(4)]
TMP = [28, 29, 30, 31]


def SB(X, h, r=0):
    return 32 + X * 32 + h * 16 + r


def GB(X, h, r=0):
    return 96 + X * 32 + h * 16 + r


def NL(h, r=0):
    return 160 + h * 16 + r


def ND(h, r=0):
    return 192 + h * 16 + r


def RING(slot):
    return 224 + slot * 4


# AGPRs
def O(h, d, r=0):
    return (h * 4 + d) * 16 + r


def QF(h, kk, r=0):
    return 128 + (h * 8 + kk) * 4 + r


def DF(h, kk, r=0):
    return 192 + (h * 8 + kk) * 4 + r


# SGPRs (s48..s101 are clobbered; inputs live below)
KP, VP, KSTEP, VSTEP, KCNT = 48, 50, 52, 53, 54
WB, REM, SC, SCALE = 55, 56, 57, 58
QP, DOP, OP, LP, DP = 60, 62, 64, 66, 68
OLS, NROWS, RAG, LDS0, WAVE = 70, 71, 72, 73, 74
ST = list(range(76, 88))
RET, TGT, EXS = 88, 90, 92


def vr(a, n):
    return f"v[{a}:{a + n - 1}]"


def ar(a, n):
    return f"a[{a}:{a + n - 1}]"


def sr(a, n=2):
    return f"s[{a}:{a + n - 1}]"


out = []


def emit(s):
    if callable(s):
        s = s()
    if isinstance(s, (list, tuple)):
        for x in s:
            emit(x)
    else:
        out.append(s)


def label(name):
    return f".Lb64_{name}_%="


def put_label(name):
    emit(label(name) + ":")


# ---------------- instruction builders ----------------
def mfma_s(X, h, kk, slot):
    c = vr(NL(h), 16) if kk == 0 else vr(SB(X, h), 16)
    return f"v_mfma_f32_32x32x16_bf16 {vr(SB(X, h), 16)}, {vr(RING(slot), 4)}, {ar(QF(h, kk), 4)}, {c}"


def mfma_g(X, h, kk, slot):
    c = vr(ND(h), 16) if kk == 0 else vr(GB(X, h), 16)
    return f"v_mfma_f32_32x32x16_bf16 {vr(GB(X, h), 16)}, {vr(RING(slot), 4)}, {ar(DF(h, kk), 4)}, {c}"


def mfma_dq(X, h, d, c, slot):
    return f"v_mfma_f32_32x32x16_bf16 {ar(O(h, d), 16)}, {vr(RING(slot), 4)}, {vr(GB(X, h, 4 * c), 4)}, {ar(O(h, d), 16)}"


def read_row(kind, sub, kk, slot):
    off = sub * 8192 + (VOFF if kind == "v" else 0)
    return [f"ds_read_b128 {vr(RING(slot), 4)}, v{RA[kk]} offset:{off}"]


def read_tr(half, c, d, slot):
    off = half * 8192 + c * 4096
    return [f"ds_read_b64_tr_b16 {vr(RING(slot), 2)}, v{TA[0][d]} offset:{off}",
            f"ds_read_b64_tr_b16 {vr(RING(slot) + 2, 2)}, v{TA[1][d]} offset:{off}"]


def step_to(stage):       # address delta that moves a pointer INTO `stage` from the previous one
    return 0xFFFE8000 if stage == 0 else 0x8000


def vadd_imm(reg, imm):
    return f"v_add_u32 v{reg}, 0x{imm & 0xFFFFFFFF:x}, v{reg}"


def mul_units(X):
    """M(u): dS = P G' and the bf16 pairs, in place: pair p of a block lands in register p of the G' block (pair p reads G' registers
    2p, 2p + 1 >= p, whose products are done by then)"""
    ins = []
    for h in range(2):
        for r in range(16):
            ins.append(f"v_mul_f32 v{SB(X, h, r)}, v{SB(X, h, r)}, v{GB(X, h, r)}")
            if r & 1:
                p = r >> 1
                ins.append(f"v_cvt_pk_bf16_f32 v{GB(X, h, p)}, v{SB(X, h, r - 1)}, v{SB(X, h, r)}")
    return ins


def exp_units(X):
    return [f"v_exp_f32 v{SB(X, h, r)}, v{SB(X, h, r)}" for h in range(2) for r in range(16)]


# ---------------- the fragment stream ----------------
class Frag:
    def __init__(self, reads, mfmas):
        self.reads, self.mfmas = reads, mfmas


def sg_frags(X, sub):
    """SG of the unit in buffer X whose rows are half `sub` of the tile RA[] points at: 16 fragments (k-step kk: K then V)"""
    fr = []
    for kk in range(8):
        for kind in ("k", "v"):
            n = len(fr)
            slot = n % 8
            mk = mfma_s if kind == "k" else mfma_g
            fr.append(Frag(read_row(kind, sub, kk, slot), [mk(X, 0, kk, slot), mk(X, 1, kk, slot)]))
    return fr


def dq_frags(X, half):
    """dQ of the unit in buffer X = half `half` of the tile TA[] points at: 8 fragments (chunk c, d-block d), two reads each"""
    fr = []
    for c in range(2):
        for d in range(4):
            slot = len(fr) % 8
            fr.append(Frag(read_tr(half, c, d, slot), [mfma_dq(X, 0, d, c, slot), mfma_dq(X, 1, d, c, slot)]))
    return fr


AHEAD = args.ahead


def emit_phase(frags, nxt, fillers, extra_post=None, first_gap=0, tail=None):
    """frags: this phase's fragments (their reads are in flight or issued by the previous phases); nxt: the fragments that follow in
    program order (at least AHEAD of them; reads of fragment n + AHEAD are issued behind the MFMAs of fragment n).
    extra_post: {mfma index: [pinned instructions]}; tail: instructions behind the last MFMA (after the fillers)."""
    allf = frags + nxt
    n_m = 2 * len(frags)
    pre = [[] for _ in range(n_m)]
    post = [[] for _ in range(n_m)]
    for n, f in enumerate(frags):
        behind = sum(len(allf[j].reads) for j in range(n + 1, n + AHEAD))
        pre[2 * n].append(f"s_waitcnt lgkmcnt({min(behind, 15)})")
        post[2 * n + 1] += allf[n + AHEAD].reads
    for k, ins in (extra_post or {}).items():
        post[k] += ins
    mf = [m for f in frags for m in f.mfmas]
    fl = list(fillers)
    if args.plain:
        for k in range(n_m):
            emit(pre[k])
            emit(mf[k])
            emit(post[k])
        emit("s_nop 15")
        emit("s_nop 15")
        emit(fl)
        emit("s_nop 7")
        emit(tail or [])
        return
    for k in range(n_m):
        emit(pre[k])
        emit(mf[k])
        used = len(pre[k + 1]) if k + 1 < n_m else 0
        for s_ in post[k]:
            emit(s_)
            used += 1
        room = args.cap - used if k >= first_gap else 0
        if not args.greedy and room > 0:      # spread over the phase (two gaps short of its end)
            room = min(room, -(-len(fl) // max(1, n_m - 2 - k)))
        if k == n_m - 1:
            room = len(fl)
        while room > 0 and fl:
            emit(fl.pop(0))
            room -= 1
    assert not fl
    emit(tail or [])


def dma_piece(kind, lds_off, p):
    reg, ptr = (DK[p], KP) if kind == "k" else (DV[p], VP)
    return ([f"s_add_u32 m0, s{WB}, 0x{lds_off:x}", "s_nop 0"] if p == 0 else []) + [f"global_load_lds_dwordx4 v{reg}, {sr(ptr)} offset:{p * 1024}"]


def adv(kind):
    """tile pointer to the next request; behind the last tile the step becomes 0 (later requests re-fetch the last tile into a free stage)"""
    ptr, step = (KP, KSTEP) if kind == "k" else (VP, VSTEP)
    ins = []
    if kind == "v":          # (one count for both pointers: V moves second)
        ins += [f"s_cmp_eq_u32 s{KCNT}, 0", f"s_cselect_b32 s{KSTEP}, 0, s{KSTEP}", f"s_cselect_b32 s{VSTEP}, 0, s{VSTEP}",
                f"s_sub_u32 s{KCNT}, s{KCNT}, 1"]
    ins += [f"s_add_u32 s{ptr}, s{ptr}, s{step}", f"s_addc_u32 s{ptr + 1}, s{ptr + 1}, 0"]
    return ins


def request_tile(stage):
    """all 8 pieces of one tile back to back (prologue)"""
    for p in range(4):
        emit(dma_piece("k", stage * STAGE, p))
    for p in range(4):
        emit(dma_piece("v", stage * STAGE + VOFF, p))
    # K pointer moves with the step of THIS tile (the count is tested when V moves: K first would step once too often at the end)
    emit([f"s_cmp_eq_u32 s{KCNT}, 0", f"s_cselect_b32 s{KSTEP}, 0, s{KSTEP}", f"s_cselect_b32 s{VSTEP}, 0, s{VSTEP}",
          f"s_sub_u32 s{KCNT}, s{KCNT}, 1",
          f"s_add_u32 s{KP}, s{KP}, s{KSTEP}", f"s_addc_u32 s{KP + 1}, s{KP + 1}, 0",
          f"s_add_u32 s{VP}, s{VP}, s{VSTEP}", f"s_addc_u32 s{VP + 1}, s{VP + 1}, 0"])


def plain_frags(frags):
    """lock-step: 8 fragments' reads, wait, their MFMAs"""
    for i in range(0, len(frags), 8):
        grp = frags[i:i + 8]
        for f in grp:
            emit(f.reads)
        emit("s_waitcnt lgkmcnt(0)")
        for f in grp:
            emit(f.mfmas)


_calls = [0]


def call(name, ret=RET):
    _calls[0] += 1
    n = _calls[0]
    here = label(f"pc{n}")
    emit(f"s_getpc_b64 {sr(TGT)}")
    put_label(f"pc{n}")
    emit(f"s_add_u32 s{TGT}, s{TGT}, {label(name)}-{here}")
    emit(f"s_addc_u32 s{TGT + 1}, s{TGT + 1}, 0")
    emit(f"s_swappc_b64 {sr(ret)}, {sr(TGT)}")


# =====================================================================================================================
# kernel body
# =====================================================================================================================
t = TMP
for dst, src in ((KP, "kp_lo"), (KP + 1, "kp_hi"), (VP, "vp_lo"), (VP + 1, "vp_hi"), (KSTEP, "kstep"), (VSTEP, "vstep"),
                 (QP, "qp_lo"), (QP + 1, "qp_hi"), (DOP, "dop_lo"), (DOP + 1, "dop_hi"), (OP, "op_lo"), (OP + 1, "op_hi"),
                 (LP, "lp_lo"), (LP + 1, "lp_hi"), (DP, "dp_lo"), (DP + 1, "dp_hi"), (OLS, "olsb"), (NROWS, "nrows"), (SC, "sc"),
                 (SCALE, "scale"), (RAG, "rag"), (LDS0, "lds0")):
    emit(f"s_mov_b32 s{dst}, %[{src}]")
emit(f"s_sub_u32 s{KCNT}, %[nt], 1")                         # tiles behind the one the pointers stand on
emit(f"v_bfe_u32 v{t[0]}, %[tid], 6, 2")                     # (the upper bits of the work-item id register are not zero)
emit("s_nop 3")
emit(f"v_readfirstlane_b32 s{WAVE}, v{t[0]}")
emit("s_nop 3")
emit(f"s_lshl_b32 s{WB}, s{WAVE}, 12")
emit(f"s_add_u32 s{WB}, s{WB}, s{LDS0}")
# ---- lane table (written by the wrapper into stage 3, 128 bytes per work item): RA[8] TA[8] DK[4] DV[4] QO[2] DO[2] ST[2] - - ----
emit(f"v_and_b32 v{t[0]}, 0xff, %[tid]")
emit(f"v_lshlrev_b32 v{t[0]}, 7, v{t[0]}")
emit(f"v_add_u32 v{t[0]}, s{LDS0}, v{t[0]}")
emit(f"v_add_u32 v{t[0]}, 0x{TABOFF:x}, v{t[0]}")
emit(f"ds_read_b128 {vr(RA[0], 4)}, v{t[0]} offset:0")
emit(f"ds_read_b128 {vr(RA[4], 4)}, v{t[0]} offset:16")
emit(f"ds_read_b128 {vr(TA[0][0], 4)}, v{t[0]} offset:32")
emit(f"ds_read_b128 {vr(TA[1][0], 4)}, v{t[0]} offset:48")
emit(f"ds_read_b128 {vr(DK[0], 4)}, v{t[0]} offset:64")
emit(f"ds_read_b128 {vr(DV[0], 4)}, v{t[0]} offset:80")
# (the six row offsets are parked in the tuple registers until the loads through them have been issued)
emit(f"ds_read_b128 {vr(NL(0), 4)}, v{t[0]} offset:96")
emit(f"ds_read_b64 {vr(NL(0) + 4, 2)}, v{t[0]} offset:112")
QO, DOO, STO = NL(0), NL(0) + 2, NL(0) + 4
emit("s_waitcnt lgkmcnt(0)")
emit("s_barrier")                                             # every wave has its table: stage 3 may be overwritten by tile 3
# ---- X fragments: Q rows -> v32..v95, dO rows -> v96..v159 (16 bytes per lane and k-step), statistics -> v28..v31 ----
for h in range(2):
    for kk in range(8):
        emit(f"global_load_dwordx4 {vr(32 + (h * 8 + kk) * 4, 4)}, v{QO + h}, {sr(QP)} offset:{kk * 32}")
for h in range(2):
    for kk in range(8):
        emit(f"global_load_dwordx4 {vr(96 + (h * 8 + kk) * 4, 4)}, v{DOO + h}, {sr(DOP)} offset:{kk * 32}")
for h in range(2):
    emit(f"global_load_dword v{t[h]}, v{STO + h}, {sr(LP)}")
    emit(f"global_load_dword v{t[2 + h]}, v{STO + h}, {sr(DP)}")
# ---- first tile requests: tiles 0, 1, 2 -> stages 0, 1, 2 (24 pieces per wave) ----
for tile in range(3):
    request_tile(tile)
# ---- accumulators ----
Z0, Z1, Z2 = RING(0), RING(0) + 1, RING(0) + 2                 # (the ring is idle in the prologue)
emit(f"v_mov_b32 v{Z2}, 0")
for r in range(128):
    emit(f"v_accvgpr_write_b32 a{r}, v{Z2}")
emit("s_waitcnt vmcnt(24)")                                   # the X loads have landed (in-order retirement); the DMA pieces may stay in flight
# ---- Q~ = bf16(Q * sc) (sc == 1: Q's bits), dO -> AGPRs; -lse / -delta tuples ----
emit(f"s_cmp_eq_u32 s{SC}, 0x3f800000")
emit(f"s_cbranch_scc1 {label('qcopy')}")
for d in range(64):
    src = 32 + d
    emit(f"v_lshlrev_b32 v{Z0}, 16, v{src}")
    emit(f"v_and_b32 v{Z1}, 0xffff0000, v{src}")
    emit(f"v_mul_f32 v{Z0}, s{SC}, v{Z0}")
    emit(f"v_mul_f32 v{Z1}, s{SC}, v{Z1}")
    emit(f"v_cvt_pk_bf16_f32 v{src}, v{Z0}, v{Z1}")
put_label("qcopy")
for d in range(64):
    emit(f"v_accvgpr_write_b32 a{128 + d}, v{32 + d}")
for d in range(64):
    emit(f"v_accvgpr_write_b32 a{192 + d}, v{96 + d}")
for h in range(2):
    for r in range(16):
        emit(f"v_xor_b32 v{NL(h, r)}, 0x80000000, v{t[h]}")
        emit(f"v_xor_b32 v{ND(h, r)}, 0x80000000, v{t[2 + h]}")
emit("s_nop 7")
emit("s_nop 7")


def valu_plain(X, mask=False):
    """E and M of the unit in buffer X, nothing interleaved (first unit, ragged tile)"""
    emit("s_nop 15")
    emit("s_nop 15")
    if mask:
        # key of register r (hi = 0): 32 sub + 16 (r >> 3) + 4 (r & 3) + ((r >> 2) & 1), + 2 hi; invalid keys: S' = -inf => P = 0, dS = 0
        emit(f"v_bfe_u32 v{t[0]}, %[tid], 5, 1")
        emit(f"v_lshlrev_b32 v{t[0]}, 1, v{t[0]}")            # 2 hi
        emit(f"v_mov_b32 v{t[1]}, 0xff800000")
        for r in range(16):
            kb_ = 32 * X + 16 * (r >> 3) + 4 * (r & 3) + ((r >> 2) & 1)
            emit(f"s_sub_i32 s{ST[0]}, s{RAG}, {kb_}")
            emit(f"v_cmp_gt_i32 vcc, s{ST[0]}, v{t[0]}")      # key < rag
            for h in range(2):
                emit(f"v_cndmask_b32 v{SB(X, h, r)}, v{t[1]}, v{SB(X, h, r)}, vcc")
    emit(exp_units(X))
    emit("s_nop 1")
    emit(mul_units(X))
    emit("s_nop 7")


# ---- ragged key tail (staged by the wrapper behind the four stages, zero rows beyond RAG keys): one whole tile in plain order ----
emit(f"s_cmp_eq_u32 s{RAG}, 0")
emit(f"s_cbranch_scc1 {label('norag')}")
for a_ in RA + TA[0] + TA[1]:
    emit(vadd_imm(a_, RAGOFF))
for sub in range(2):
    plain_frags(sg_frags(sub, sub))
for sub in range(2):
    valu_plain(sub, mask=True)
for sub in range(2):
    plain_frags(dq_frags(sub, sub))
for a_ in RA + TA[0] + TA[1]:
    emit(vadd_imm(a_, -RAGOFF))
put_label("norag")
# ---- tiles 0..2 landed; S'(0), G'(0) in lock step, E(0); the ring is primed with the first fragments of the loop ----
emit("s_waitcnt vmcnt(0)")
emit("s_barrier")
plain_frags(sg_frags(0, 0))
emit("s_nop 15")
emit("s_nop 15")
emit(exp_units(0))
emit(f"s_mov_b32 s{REM}, %[nt]")


def body_frags(c):
    """fragment list of one tile body (copy c): it A = SG(2t+1) [X = 1, sub 1, tile t], dQ(2t) [X = 0, half 0]; it B = SG(2t+2) [X = 0,
    sub 0, tile t + 1], dQ(2t+1) [X = 1, half 1]"""
    return [sg_frags(1, 1), dq_frags(0, 0), sg_frags(0, 0), dq_frags(1, 1)]


first = body_frags(0)
for f in (first[0] + first[1])[:AHEAD]:
    emit(f.reads)
for c in range(4):
    put_label(f"copy{c}")
    emit(f"s_cmp_eq_u32 s{REM}, 0")
    emit(f"s_cbranch_scc1 {label('done')}")
    emit(f"s_sub_u32 s{REM}, s{REM}, 1")
    ph = body_frags(c)
    nxt_body = body_frags((c + 1) & 3)
    seq = ph[0] + ph[1] + ph[2] + ph[3] + nxt_body[0]
    # --- it A, SG(2t+1): M(2t); the K pieces of tile t + 3 -> stage (c + 3) & 3; RA -> tile t + 1 behind its last read of tile t
    ks = ((c + 3) & 3) * STAGE
    xp = {}
    for p in range(4):
        xp[2 + 4 * p] = dma_piece("k", ks, p)
    for j in range(8):          # RA[4..7] are read (fragments 8..15 of this phase) behind MFMA pairs 0..7; RA[0..3] were read in the previous phase
        xp.setdefault(17 + j, []).append(vadd_imm(RA[j], step_to((c + 1) & 3)))
    emit_phase(ph[0], seq[16:], mul_units(0), extra_post=xp)
    # --- it A, dQ(2t): E(2t+1)
    emit_phase(ph[1], seq[24:], exp_units(1), first_gap=args.first_gap)
    # --- it B, SG(2t+2): M(2t+1); the V pieces of tile t + 3, then both pointers move on
    xp = {}
    for p in range(4):
        xp[2 + 4 * p] = dma_piece("v", ks + VOFF, p)
    adv_ins = adv("v")[:4] + adv("k") + adv("v")[4:]
    for i, ins in enumerate(adv_ins):
        xp.setdefault(18 + i, []).append(ins)
    emit_phase(ph[2], seq[40:], mul_units(1), extra_post=xp)
    # --- it B, dQ(2t+1): E(2t+2); TA -> tile t + 1 (its last reads of tile t were issued in the second half of the SG phase above)
    xp = {}
    for j in range(8):
        xp.setdefault(1 + j, []).append(vadd_imm((TA[0] + TA[1])[j], step_to((c + 1) & 3)))
    emit_phase(ph[3], seq[48:], exp_units(0), extra_post=xp, first_gap=args.first_gap, tail=["s_waitcnt vmcnt(8)", "s_barrier"])
    if c == 3:
        emit(f"s_branch {label('copy0')}")
put_label("done")
emit("s_waitcnt lgkmcnt(0)")
emit("s_waitcnt vmcnt(0)")
emit("s_nop 15")
emit("s_nop 15")
# ---- epilogue: dQ rows: out[row][dblk*32 + 8 rq + 4 hi + e] = dQ^T[h][dblk][4 rq + e] * scale, 8-byte stores ----
emit(f"v_and_b32 v{t[0]}, 31, %[tid]")
emit(f"v_bfe_u32 v{t[1]}, %[tid], 5, 1")
emit(f"s_lshl_b32 s{ST[0]}, s{WAVE}, 6")
emit(f"v_add_u32 v{t[2]}, s{ST[0]}, v{t[0]}")                 # row of half 0
emit(f"v_lshlrev_b32 v{t[1]}, 3, v{t[1]}")                   # hi * 4 elements * 2 bytes
E_ = [SB(0, 0, e) for e in range(4)]
for h in range(2):
    if h:
        emit(f"v_add_u32 v{t[2]}, 32, v{t[2]}")
    emit(f"v_mul_lo_u32 v{t[3]}, v{t[2]}, s{OLS}")
    emit(f"v_add_u32 v{t[3]}, v{t[3]}, v{t[1]}")
    emit(f"v_cmp_gt_u32 vcc, s{NROWS}, v{t[2]}")
    emit(f"s_and_saveexec_b64 {sr(EXS)}, vcc")
    for d in range(4):
        for rq in range(4):
            for e in range(4):
                emit(f"v_accvgpr_read_b32 v{E_[e]}, a{O(h, d, rq * 4 + e)}")
            for e in range(4):
                emit(f"v_mul_f32 v{E_[e]}, s{SCALE}, v{E_[e]}")
            emit(f"v_cvt_pk_bf16_f32 v{E_[0]}, v{E_[0]}, v{E_[1]}")
            emit(f"v_cvt_pk_bf16_f32 v{E_[1]}, v{E_[2]}, v{E_[3]}")
            emit(f"global_store_dwordx2 v{t[3]}, {vr(E_[0], 2)}, {sr(OP)} offset:{(d * 32 + rq * 8) * 2}")
            emit("s_nop 1")
    emit(f"s_mov_b64 exec, {sr(EXS)}")
emit("s_waitcnt vmcnt(0)")

# ---------------- write ----------------
n_mfma = sum(1 for s_ in out if s_.startswith("v_mfma"))
with open(args.o, "w") as fh:
    fh.write("// GENERATED by tools/gen_attn_bwd64.py %s — do not edit; %d instructions, %d MFMAs\n" %
             (" ".join(a for a in sys.argv[1:] if not a.startswith("more4d") and a != "-o"),
              sum(1 for s_ in out if not s_.endswith(":") and not s_.startswith(";")), n_mfma))
    for s_ in out:
        if s_.startswith(";"):
            continue
        fh.write('"%s\\n\\t"\n' % s_)
print(f"{args.o}: {len(out)} lines, {n_mfma} MFMAs", file=sys.stderr)
