#!/usr/bin/env python3
"""Generator of the hand-placed instruction streams of attn_bwd_kv64_kernel (more4d_amd/csrc/attention_bwd64.h): the fused dK / dV pass
of the flash-attention backward as ONE WAVE PER SIMD, role-split.

    python tools/gen_attn_bwd64_kv.py [--plain] [--cap N] [-o more4d_amd/csrc/attention_bwd64_kv_gen.inc]

A workgroup owns 128 keys = two PAIRS of waves (waves 0, 1: keys 0..63; waves 2, 3: keys 64..127); in a pair
    P-wave  (even):  S''  = Q (-K)^T + lse,  P = exp2(-S'')  -> bf16 pairs -> registers + the pair's LDS mailbox,   dV^T += dO^T P
    dS-wave (odd):   G''  = dO (-V)^T + delta,  dS = -P G'' (P from the mailbox) -> bf16 pairs,                   dK^T += Q^T dS
(the role split and the mailbox of attn_bwd_kvp_kernel, attention_bwd_kvp.h, whose two-waves-per-SIMD schedule this replaces).  Every
wave owns BOTH 32-key halves of its pair's 64 keys (fragments of K resp. V resident in AGPRs, negated once so that the streamed
statistics enter the MFMAs un-negated as C operands and exp / multiply take the sign as a source modifier), so every streamed
fragment feeds two MFMAs, and the whole 512-register file: 128 accumulators + 64 fragment registers in AGPRs.
Streamed side: 64-query tiles of Q and dO (row-major, 16 KiB each, XOR-swizzled on the DMA source address) + the tile's lse / delta in
accumulator-register order (a DMA gather), four stages of 32 KiB + 512 B; unit u = 32 queries.  Iteration i (one s_barrier each):
    P-wave:   [S(i+1): 16 MFMAs || pack P(i), mailbox]  [dV(i): 16 MFMAs || exp2 of S(i+1)]
    dS-wave:  [G(i):   16 MFMAs || mailbox P(i-1), dS(i-1) ...]  [dK(i-2): 16 MFMAs || ... dS(i-1)]
P(u) is written in iteration u and read in iteration u + 1 (two mailbox slots per pair).  Tile t is live from iteration 2t - 2 to
2t + 3; tile t + 2 is requested in the first iteration of body t (= iterations 2t, 2t + 1) and waited for at the body's end.
The ragged query tail is staged by the wrapper in stage 3 (rows beyond it zero, lse = +inf) and processed first, in plain order.

The emission engine places pinned instructions and fillers around the MFMAs and then computes every s_waitcnt lgkmcnt from the
POSITIONS of the LDS operations in the final order (reads, mailbox writes), per loop copy against the copy in front of it.
"""
import argparse
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--plain", action="store_true", help="debug: no interleaving (the MFMAs of a phase, then its fillers)")
ap.add_argument("--cap", type=int, default=5, help="instructions per MFMA gap besides the MFMA")
ap.add_argument("--first-gap", type=int, default=1, help="no fillers behind the first N MFMAs of a phase whose fillers read fresh accumulators")
ap.add_argument("--ahead", type=int, default=6, help="fragments in flight (the ring has 8 slots; lgkmcnt counts to 15)")
ap.add_argument("-o", default="more4d_amd/csrc/attention_bwd64_kv_gen.inc")
args = ap.parse_args()

AHEAD = args.ahead
STAGE, VOFF, STOFF = 32768 + 512, 16384, 32768          # stage: Q tile, dO tile, lse[64], delta[64] (accumulator-register order)
NSTG = 4
MAILOFF = NSTG * STAGE                                  # 2 pairs x 2 slots x 4 KiB
TABSTAGE = 2                                            # the lane table travels in stage 2 (tile 2 is requested in the loop)

# ---------------- register map ----------------
RA = [4 + i for i in range(8)]                  # row-fragment addresses (stage base + row * 256 + swizzled k-step chunk)
TA = [[12 + j * 4 + d for d in range(4)] for j in range(2)]
DQ = [20 + p for p in range(4)]                 # DMA lane offsets, Q pieces
DD = [24 + p for p in range(4)]                 # dO pieces
TMP = [28, 29, 30, 31]
STG, MB, OSC = 32, 33, 34                       # statistics gather lane offset; mailbox lane address; per-lane output factor
U0, U1 = 36, 37                                 # unpack temporaries (dS-wave)


def SG(X, h, r=0):          # S'' (P-wave) / G'' (dS-wave) of the unit in buffer X, key half h
    return 64 + X * 32 + h * 16 + r


def PK(Y, h, j=0):          # bf16 pairs: P (P-wave, Y = 0 only) / dS (dS-wave, Y = unit parity)
    return 128 + Y * 16 + h * 8 + j


def TU(r=0):                # statistics tuple of the unit whose S / G is computed next
    return 160 + r


def MP(h, j=0):             # mailbox P pairs (dS-wave)
    return 176 + h * 8 + j


def RING(slot):
    return 224 + slot * 4


def O(h, d, r=0):
    return (h * 4 + d) * 16 + r


def XF(h, kk, r=0):
    return 128 + (h * 8 + kk) * 4 + r


# SGPRs (s48..s101 clobbered)
YQ, YD, QSTEP, DSTEP, KCNT = 48, 50, 52, 53, 54
WB, REM, SCALE, ROLE = 55, 56, 57, 58
XP, OP, SP = 60, 62, 64                                  # X rows (K or V), output (dV or dK), statistics (lse or delta; P-waves)
OLS, NVAL, NSTO, RAG, LDS0, WAVE, SDST, SSTEP = 66, 67, 68, 69, 70, 71, 72, 73
ST = list(range(76, 88))
EXS = 92


def vr(a, n):
    return f"v[{a}:{a + n - 1}]"


def ar(a, n):
    return f"a[{a}:{a + n - 1}]"


def sr(a, n=2):
    return f"s[{a}:{a + n - 1}]"


def label(name):
    return f".Lkv64_{name}_%="


class Op:
    """an LDS operation (counts in lgkmcnt, retires in order) carrying a tag"""
    def __init__(self, tag, text):
        self.tag, self.text = tag, text


class Wait:
    """s_waitcnt lgkmcnt(N): everything up to the LAST operation tagged `tag` has retired"""
    def __init__(self, tag):
        self.tag = tag


out = []


def emit(s, dst=None):
    dst = out if dst is None else dst
    if isinstance(s, (list, tuple)):
        for x in s:
            emit(x, dst)
    else:
        dst.append(s)


def resolve(items, context=()):
    """Wait -> s_waitcnt lgkmcnt(N) with N = LDS operations between the last `tag` operation and the wait; `context` = the items in
    front (the previous loop copy).  A wait whose operation is not found drains the queue."""
    lin = list(context) + list(items)
    base = len(context)
    res = []
    for i in range(base, len(lin)):
        it = lin[i]
        if isinstance(it, Wait):
            n, found = 0, False
            for j in range(i - 1, -1, -1):
                if isinstance(lin[j], Op):
                    if lin[j].tag == it.tag:
                        found = True
                        break
                    n += 1
            assert found or not context, f"wait for {it.tag}: no such operation in front"
            if not found:
                n = 0
            n = min(n, 15)          # (a smaller count only waits for more than needed; the counter has 4 bits)
            res.append(f"s_waitcnt lgkmcnt({n})")
        elif isinstance(it, Op):
            res.append(it.text)
        else:
            res.append(it)
    return res


# ---------------- instruction builders ----------------
def mfma_sg(X, h, kk, slot):
    c = vr(TU(), 16) if kk == 0 else vr(SG(X, h), 16)
    return f"v_mfma_f32_32x32x16_bf16 {vr(SG(X, h), 16)}, {vr(RING(slot), 4)}, {ar(XF(h, kk), 4)}, {c}"


def mfma_acc(Y, h, d, c, slot):
    return f"v_mfma_f32_32x32x16_bf16 {ar(O(h, d), 16)}, {vr(RING(slot), 4)}, {vr(PK(Y, h, 4 * c), 4)}, {ar(O(h, d), 16)}"


class Frag:
    def __init__(self, tag, reads, mfmas):
        self.tag, self.reads, self.mfmas = tag, reads, mfmas


def row_frags(tagp, role, X, sub):
    """S'' (role P: Q tile) or G'' (role D: dO tile) of the 32-query half `sub` of the tile RA[] points at, into buffer X: 8 fragments;
    the first one also loads the unit's statistics tuple (4 x 16 bytes, the same address for all lanes of a half: a broadcast)"""
    toff = 0 if role == "P" else VOFF
    soff = STOFF + (0 if role == "P" else 256) + sub * 128
    fr = []
    for kk in range(8):
        slot = kk
        tag = (tagp, kk)
        rd = [Op(tag, f"ds_read_b128 {vr(RING(slot), 4)}, v{RA[kk]} offset:{sub * 8192 + toff}")]
        if kk == 0:
            rd += [Op(tag, f"ds_read_b128 {vr(TU(4 * i), 4)}, v{TMP[3]} offset:{soff + 16 * i}") for i in range(4)]
        fr.append(Frag(tag, rd, [mfma_sg(X, 0, kk, slot), mfma_sg(X, 1, kk, slot)]))
    return fr


def tr_frags(tagp, role, Y, half):
    """dV^T += dO^T P (role P: dO tile) or dK^T += Q^T dS (role D: Q tile) of half `half` of the tile TA[] points at"""
    toff = VOFF if role == "P" else 0
    fr = []
    for c in range(2):
        for d in range(4):
            slot = len(fr)
            tag = (tagp, 8 + slot)
            off = toff + half * 8192 + c * 4096
            rd = [Op(tag, f"ds_read_b64_tr_b16 {vr(RING(slot), 2)}, v{TA[0][d]} offset:{off}"),
                  Op(tag, f"ds_read_b64_tr_b16 {vr(RING(slot) + 2, 2)}, v{TA[1][d]} offset:{off}")]
            fr.append(Frag(tag, rd, [mfma_acc(Y, 0, d, c, slot), mfma_acc(Y, 1, d, c, slot)]))
    return fr


def step_to(stage):       # address delta that moves a pointer INTO `stage` from the previous one
    return (-(NSTG - 1) * STAGE) if stage == 0 else STAGE


def vadd_imm(reg, imm):
    return f"v_add_u32 v{reg}, 0x{imm & 0xFFFFFFFF:x}, v{reg}"


def exp_units(X):           # P = exp2(-S'')
    return [f"v_exp_f32_e64 v{SG(X, h, r)}, -v{SG(X, h, r)}" for h in range(2) for r in range(16)]


def pack_p(X, slot_off, tag):
    """P-wave: bf16 pairs of P (buffer X) -> PK(0), and the four mailbox writes (slot_off = 0 / 4096)"""
    ins = []
    for h in range(2):
        for p in range(8):
            ins.append(f"v_cvt_pk_bf16_f32 v{PK(0, h, p)}, v{SG(X, h, 2 * p)}, v{SG(X, h, 2 * p + 1)}")
            if p & 3 == 3:
                i = h * 2 + (p >> 2)
                ins.append(Op(tag, f"ds_write_b128 v{MB}, {vr(PK(0, h, p - 3), 4)} offset:{slot_off + i * 1024}"))
    return ins


def mail_reads(slot_off, tag):
    return [Op(tag, f"ds_read_b128 {vr(MP(i >> 1, (i & 1) * 4), 4)}, v{MB} offset:{slot_off + i * 1024}") for i in range(4)]


def ds_units(X, Y, tag):
    """dS-wave: dS = -P G'' of the unit in G buffer X, P pairs in MP -> bf16 pairs PK(Y)"""
    ins = [Wait(tag)]
    for h in range(2):
        for p in range(8):
            a, b_ = SG(X, h, 2 * p), SG(X, h, 2 * p + 1)
            ins += [f"v_lshlrev_b32 v{U0}, 16, v{MP(h, p)}", f"v_and_b32 v{U1}, 0xffff0000, v{MP(h, p)}",
                    f"v_mul_f32_e64 v{a}, -v{U0}, v{a}", f"v_mul_f32_e64 v{b_}, -v{U1}, v{b_}",
                    f"v_cvt_pk_bf16_f32 v{PK(Y, h, p)}, v{a}, v{b_}"]
    return ins


def phase(frags, nxt, fillers, extra_post=None, first_gap=0, tail=None):
    """the items of one phase: fragment waits, MFMAs, the reads of fragment n + AHEAD behind the MFMAs of fragment n, pinned
    instructions (extra_post: {mfma index: [...]}) and fillers poured evenly into what the cap leaves of every gap"""
    allf = frags + nxt
    n_m = 2 * len(frags)
    pre = [[] for _ in range(n_m)]
    post = [[] for _ in range(n_m)]
    for n, f in enumerate(frags):
        pre[2 * n].append(Wait(f.tag))
        post[2 * n + 1] += allf[n + AHEAD].reads
    for k, ins in (extra_post or {}).items():
        post[k] += ins
    mf = [m for f in frags for m in f.mfmas]
    fl = list(fillers)
    res = []
    if args.plain:
        for k in range(n_m):
            emit(pre[k], res)
            emit(mf[k], res)
            emit(post[k], res)
        emit(["s_nop 15", "s_nop 15"], res)
        emit(fl, res)
        emit("s_nop 7", res)
        emit(tail or [], res)
        return res
    for k in range(n_m):
        emit(pre[k], res)
        emit(mf[k], res)
        used = len(pre[k + 1]) if k + 1 < n_m else 0
        emit(post[k], res)
        used += len(post[k])
        room = args.cap - used if k >= first_gap else 0
        if room > 0:
            room = min(room, -(-len(fl) // max(1, n_m - 1 - k)))
        if k == n_m - 1:
            room = len(fl)
        while room > 0 and fl:
            res.append(fl.pop(0))
            room -= 1
    emit(tail or [], res)
    return res


def dma_piece(kind, lds_off, p):
    reg, ptr = (DQ[p], YQ) if kind == "q" else (DD[p], YD)
    return ([f"s_add_u32 m0, s{WB}, 0x{lds_off:x}", "s_nop 0"] if p == 0 else []) + [f"global_load_lds_dwordx4 v{reg}, {sr(ptr)} offset:{p * 1024}"]


def dma_stat(stage):
    """P-waves: the tile's lse (wave 0) / delta (wave 2) in accumulator-register order, one dword per lane"""
    return [f"s_add_u32 m0, s{SDST}, 0x{stage * STAGE:x}", "s_nop 0", f"global_load_lds_dword v{STG}, {sr(SP)}"]


def adv_all(role):
    """tile pointers to the next request; behind the last tile the steps become 0 (later requests re-fetch the last tile)"""
    ins = [f"s_cmp_eq_u32 s{KCNT}, 0", f"s_cselect_b32 s{QSTEP}, 0, s{QSTEP}", f"s_cselect_b32 s{DSTEP}, 0, s{DSTEP}"]
    if role == "P":
        ins.append(f"s_cselect_b32 s{SSTEP}, 0, s{SSTEP}")
    ins += [f"s_sub_u32 s{KCNT}, s{KCNT}, 1",
            f"s_add_u32 s{YQ}, s{YQ}, s{QSTEP}", f"s_addc_u32 s{YQ + 1}, s{YQ + 1}, 0",
            f"s_add_u32 s{YD}, s{YD}, s{DSTEP}", f"s_addc_u32 s{YD + 1}, s{YD + 1}, 0"]
    if role == "P":
        ins += [f"s_add_u32 s{SP}, s{SP}, s{SSTEP}", f"s_addc_u32 s{SP + 1}, s{SP + 1}, 0"]
    return ins


def request_tile(role, stage):
    ins = []
    for p in range(4):
        ins += dma_piece("q", stage * STAGE, p)
    for p in range(4):
        ins += dma_piece("d", stage * STAGE + VOFF, p)
    if role == "P":
        ins += dma_stat(stage)
    return ins + adv_all(role)


def plain_frags(frags):
    """lock-step: 8 fragments' reads, wait, their MFMAs"""
    res = []
    for i in range(0, len(frags), 8):
        grp = frags[i:i + 8]
        for f in grp:
            emit(f.reads, res)
        res.append("s_waitcnt lgkmcnt(0)")
        for f in grp:
            emit(f.mfmas, res)
    return res


def strip(items):
    return [it.text if isinstance(it, Op) else it for it in items]


# =====================================================================================================================
# common prologue
# =====================================================================================================================
t = TMP
for dst, src in ((YQ, "yq_lo"), (YQ + 1, "yq_hi"), (YD, "yd_lo"), (YD + 1, "yd_hi"), (QSTEP, "qstep"), (DSTEP, "dstep"),
                 (SCALE, "scale"), (NVAL, "nval"), (NSTO, "nsto"), (RAG, "rag"), (LDS0, "lds0")):
    emit(f"s_mov_b32 s{dst}, %[{src}]")
emit(f"s_sub_u32 s{KCNT}, %[nt], 1")
emit(f"s_mov_b32 s{SSTEP}, 256")
emit(f"v_bfe_u32 v{t[0]}, %[tid], 6, 2")
emit("s_nop 3")
emit(f"v_readfirstlane_b32 s{WAVE}, v{t[0]}")
emit("s_nop 3")
emit(f"s_and_b32 s{ROLE}, s{WAVE}, 1")                       # 0: P-wave, 1: dS-wave
emit(f"s_lshl_b32 s{WB}, s{WAVE}, 12")
emit(f"s_add_u32 s{WB}, s{WB}, s{LDS0}")
# role-dependent pointers: X rows (K | V), output (dV | dK), output row stride, output factor; statistics (wave 0: lse, wave 2: delta)
emit(f"s_cmp_eq_u32 s{ROLE}, 0")
for dst, a_, b_ in ((XP, "xk_lo", "xv_lo"), (XP + 1, "xk_hi", "xv_hi"), (OP, "ov_lo", "ok_lo"), (OP + 1, "ov_hi", "ok_hi"),
                    (OLS, "ovls", "okls")):
    emit(f"s_cselect_b32 s{dst}, %[{a_}], %[{b_}]")
emit(f"s_cselect_b32 s{SCALE}, 0x3f800000, s{SCALE}")        # dV carries no factor, dK = scale * sum
emit(f"s_cmp_eq_u32 s{WAVE}, 0")
emit(f"s_cselect_b32 s{SP}, %[lse_lo], %[del_lo]")
emit(f"s_cselect_b32 s{SP + 1}, %[lse_hi], %[del_hi]")
emit(f"s_cselect_b32 s{SDST}, 0, 256")
emit(f"s_add_u32 s{SDST}, s{SDST}, 0x{STOFF:x}")
emit(f"s_add_u32 s{SDST}, s{SDST}, s{LDS0}")
# ---- lane table (stage 2, 128 bytes per work item): RA[8] TA[8] DQ[4] DD[4] XK[2] XV[2] STG - - - ----
emit(f"v_and_b32 v{t[0]}, 0xff, %[tid]")
emit(f"v_lshlrev_b32 v{t[0]}, 7, v{t[0]}")
emit(f"v_add_u32 v{t[0]}, s{LDS0}, v{t[0]}")
emit(f"v_add_u32 v{t[0]}, 0x{TABSTAGE * STAGE:x}, v{t[0]}")
emit(f"ds_read_b128 {vr(RA[0], 4)}, v{t[0]} offset:0")
emit(f"ds_read_b128 {vr(RA[4], 4)}, v{t[0]} offset:16")
emit(f"ds_read_b128 {vr(TA[0][0], 4)}, v{t[0]} offset:32")
emit(f"ds_read_b128 {vr(TA[1][0], 4)}, v{t[0]} offset:48")
emit(f"ds_read_b128 {vr(DQ[0], 4)}, v{t[0]} offset:64")
emit(f"ds_read_b128 {vr(DD[0], 4)}, v{t[0]} offset:80")
emit(f"ds_read_b128 {vr(TU(0), 4)}, v{t[0]} offset:96")      # XK[2] XV[2] parked in the tuple registers
emit(f"ds_read_b32 v{STG}, v{t[0]} offset:112")
emit("s_waitcnt lgkmcnt(0)")
emit("s_barrier")                                             # every wave has its table
# X rows of this wave's role (K for the P-wave, V for the dS-wave), 2 key halves x 8 k-steps -> v64..v127 -> negated -> a128..a191
emit(f"s_cmp_eq_u32 s{ROLE}, 0")
emit(f"s_cbranch_scc1 {label('xk')}")
emit(f"v_mov_b32 v{TU(0)}, v{TU(2)}")
emit(f"v_mov_b32 v{TU(1)}, v{TU(3)}")
out.append(label("xk") + ":")
for h in range(2):
    for kk in range(8):
        emit(f"global_load_dwordx4 {vr(64 + (h * 8 + kk) * 4, 4)}, v{TU(h)}, {sr(XP)} offset:{kk * 32}")
# mailbox lane address: MAILOFF + pair * 8192 + lane * 16
emit(f"v_and_b32 v{MB}, 63, %[tid]")
emit(f"v_lshlrev_b32 v{MB}, 4, v{MB}")
emit(f"s_lshr_b32 s{ST[0]}, s{WAVE}, 1")
emit(f"s_lshl_b32 s{ST[0]}, s{ST[0]}, 13")
emit(f"s_add_u32 s{ST[0]}, s{ST[0]}, s{LDS0}")
emit(f"s_add_u32 s{ST[0]}, s{ST[0]}, 0x{MAILOFF:x}")
emit(f"v_add_u32 v{MB}, s{ST[0]}, v{MB}")
# statistics tuple address: stage 0 + hi * 64 (+ STOFF (+256) + sub * 128 in the instruction offset)
emit(f"v_bfe_u32 v{t[3]}, %[tid], 5, 1")
emit(f"v_lshlrev_b32 v{t[3]}, 6, v{t[3]}")
emit(f"v_add_u32 v{t[3]}, s{LDS0}, v{t[3]}")
# per-lane output factor: key = 64 * pair + 32 h + li valid (< NVAL) ? scale : 0 — per half; OSC = half 0, OSC + 1 = half 1
emit(f"v_and_b32 v{t[0]}, 31, %[tid]")
emit(f"s_lshr_b32 s{ST[0]}, s{WAVE}, 1")
emit(f"s_lshl_b32 s{ST[0]}, s{ST[0]}, 6")
emit(f"v_add_u32 v{t[0]}, s{ST[0]}, v{t[0]}")
emit(f"v_mov_b32 v{t[1]}, s{SCALE}")
for h in range(2):
    emit(f"v_cmp_gt_u32 vcc, s{NVAL}, v{t[0]}")
    emit("s_nop 1")
    emit(f"v_cndmask_b32 v{OSC + h}, 0, v{t[1]}, vcc")
    emit(f"v_add_u32 v{t[0]}, 32, v{t[0]}")
# ---- first tile requests: tiles 0, 1 -> stages 0, 1; both roles branch into their own stream from here ----
emit(f"s_cmp_eq_u32 s{ROLE}, 1")
emit(f"s_cbranch_scc1 {label('role_d')}")


def role_stream(role):
    R = role
    nreq = 9 if R == "P" else 8
    for tile in range(2):
        emit(request_tile(R, tile))
    emit(f"v_mov_b32 v{U0}, 0")
    for r in range(128):
        emit(f"v_accvgpr_write_b32 a{r}, v{U0}")
    for r in range(32):
        emit(f"v_mov_b32 v{PK(0, 0, r)}, 0")                  # dS pairs of units -2, -1 (dS-wave) / unused (P-wave)
    emit(f"s_waitcnt vmcnt({2 * nreq})")                      # the X rows have landed
    for d in range(64):
        emit(f"v_xor_b32 v{64 + d}, 0x80008000, v{64 + d}")   # -K / -V
    for d in range(64):
        emit(f"v_accvgpr_write_b32 a{128 + d}, v{64 + d}")
    emit("s_nop 7")
    if R == "D":
        for r in range(32):
            emit(f"v_mov_b32 v{SG(1, 0, r)}, 0")              # G''(-1) = 0 -> dS(-1) = 0 (the mailbox slot of P(-1) is zero as well)
    # ---- ragged query tail in stage 3 (RAG valid rows; lse = +inf beyond them), plain order, two extra barriers ----
    emit(f"s_cmp_eq_u32 s{RAG}, 0")
    emit(f"s_cbranch_scc1 {label('norag' + R)}")
    mv = RA + [t[3]] + ((TA[0] + TA[1]) if R == "P" else [])    # (the dS-wave's TA[] already stands on stage 3 = "tile -1")
    for a_ in mv:
        emit(vadd_imm(a_, 3 * STAGE))
    if R == "P":
        for sub in range(2):
            emit(strip(plain_frags(row_frags(("rag", sub), R, sub, sub))))
        emit(["s_nop 15", "s_nop 15"])
        for sub in range(2):
            emit(exp_units(sub))
        emit("s_nop 1")
        for sub in range(2):
            emit(strip(pack_p(sub, sub * 4096, "ragw")))
            emit("s_nop 7")
            emit(strip(plain_frags(tr_frags(("ragt", sub), R, 0, sub))))
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
        emit("s_barrier")
    else:
        for sub in range(2):
            emit(strip(plain_frags(row_frags(("rag", sub), R, sub, sub))))
        emit("s_barrier")
        for sub in range(2):
            emit(strip(mail_reads(sub * 4096, "ragm")))
            emit("s_waitcnt lgkmcnt(0)")
            emit("s_nop 15")
            emit(strip(ds_units(sub, sub, "ragm")[1:]))
        emit("s_nop 7")
        for sub in range(2):
            emit(strip(plain_frags(tr_frags(("ragt", sub), R, sub, sub))))
        # back to the initial state: dS pairs, G''(-1) and the mailbox slot of P(-1) zero
        for r in range(32):
            emit(f"v_mov_b32 v{PK(0, 0, r)}, 0")
        for r in range(32):
            emit(f"v_mov_b32 v{SG(1, 0, r)}, 0")
        for i in range(4):
            emit(f"ds_write_b128 v{MB}, {vr(SG(1, 0, 0), 4)} offset:{4096 + i * 1024}")
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
    for a_ in mv:
        emit(vadd_imm(a_, -3 * STAGE))
    out.append(label("norag" + R) + ":")
    # ---- tiles 0, 1 landed ----
    emit("s_waitcnt vmcnt(0)")
    emit("s_barrier")
    if R == "P":
        emit(strip(plain_frags(row_frags(("pro", 0), R, 0, 0))))
        emit(["s_nop 15", "s_nop 15"])
        emit(exp_units(0))
    emit(f"s_mov_b32 s{REM}, %[nt]")

    # ---- loop bodies ----
    def body(c):
        """[phase frags] of body t (copy c = t & 3): iterations 2t, 2t + 1.
        P:  S(2t+1) [X 1, sub 1, tile t]   dV(2t)   [half 0, tile t]     S(2t+2) [X 0, sub 0, tile t+1]   dV(2t+1) [half 1, tile t]
        D:  G(2t)   [X 0, sub 0, tile t]   dK(2t-2) [Y 0, half 0, t-1]   G(2t+1) [X 1, sub 1, tile t]     dK(2t-1) [Y 1, half 1, t-1]"""
        if R == "P":
            return [row_frags((c, 0), R, 1, 1), tr_frags((c, 1), R, 0, 0), row_frags((c, 2), R, 0, 0), tr_frags((c, 3), R, 0, 1)]
        return [row_frags((c, 0), R, 0, 0), tr_frags((c, 1), R, 0, 0), row_frags((c, 2), R, 1, 1), tr_frags((c, 3), R, 1, 1)]

    def body_items(c):
        ph = body(c)
        nb = body((c + 1) & 3)
        seq = ph[0] + ph[1] + ph[2] + ph[3] + nb[0]
        rs = (c + 2) & 3                                       # stage of the tile requested in this body (tile t + 2)
        items = []
        items += [f"{label(f'copy{R}{c}')}:", f"s_cmp_eq_u32 s{REM}, 0", f"s_cbranch_scc1 {label('done' + R)}", f"s_sub_u32 s{REM}, s{REM}, 1"]
        nxt_stage = (c + 1) & 3
        if R == "P":
            # it 2t, phase 1: S(2t+1) || pack P(2t) [buffer 0] -> mailbox slot 0; Q pieces of tile t + 2; RA / statistics address -> tile
            # t + 1 (their reads of tile t were issued in the previous iteration's second phase; the row reads of S(2t+2) follow behind
            # the MFMAs of the NEXT phase)
            xp = {}
            for p in range(4):
                xp[2 + 4 * p] = dma_piece("q", rs * STAGE, p)
            for j, a_ in enumerate(RA + [t[3]]):
                xp.setdefault((3, 4, 5, 7, 8, 9, 11, 12, 13)[j], []).append(vadd_imm(a_, step_to(nxt_stage)))
            items += phase(ph[0], seq[8:], pack_p(0, 0, ("mw", c, 0)), extra_post=xp)
            # phase 2: dV(2t) || exp2 of S''(2t+1) [buffer 1]; dO pieces + statistics of tile t + 2
            xp = {}
            for p in range(4):
                xp[2 + 3 * p] = dma_piece("d", rs * STAGE + VOFF, p)
            xp[14] = dma_stat(rs)
            items += phase(ph[1], seq[16:], exp_units(1), extra_post=xp, first_gap=args.first_gap, tail=[Wait(("mw", c, 0)), "s_barrier"])
            # it 2t+1, phase 1: S(2t+2) || pack P(2t+1) [buffer 1] -> slot 1; tile pointers move on
            xp = {}
            for i_, ins in enumerate(adv_all(R)):
                xp.setdefault(1 + i_, []).append(ins)
            items += phase(ph[2], seq[24:], pack_p(1, 4096, ("mw", c, 1)), extra_post=xp)
            # phase 2: dV(2t+1) || exp2 of S''(2t+2) [buffer 0]; TA -> tile t + 1 (its reads of tile t were issued in phase 1)
            xp = {}
            for j, a_ in enumerate(TA[0] + TA[1]):
                xp.setdefault(2 + j, []).append(vadd_imm(a_, step_to(nxt_stage)))
            items += phase(ph[3], seq[32:], exp_units(0), extra_post=xp, first_gap=args.first_gap,
                           tail=[Wait(("mw", c, 1)), "s_waitcnt vmcnt(0)", "s_barrier"])
        else:
            # it 2t, phase 1: G(2t) [buffer 0] || P(2t-1) from slot 1, dS(2t-1) = -P G''(2t-1) [buffer 1] -> pairs PK(1)
            fl = mail_reads(4096, ("mr", c, 0)) + ds_units(1, 1, ("mr", c, 0))
            xp = {}
            for p in range(4):
                xp[2 + 4 * p] = dma_piece("q", rs * STAGE, p)
            n1 = len(fl) * 16 // 32
            items += phase(ph[0], seq[8:], fl[:n1], extra_post=xp)
            # phase 2: dK(2t-2) [pairs PK(0), half 0 of tile t - 1] || the rest of dS(2t-1)
            xp = {}
            for p in range(4):
                xp[2 + 3 * p] = dma_piece("d", rs * STAGE + VOFF, p)
            # RA (dO tile t, both halves: G(2t) read in the previous body's last phase, G(2t+1) behind THIS phase's MFMAs) stays;
            items += phase(ph[1], seq[16:], fl[n1:], extra_post=xp, tail=["s_barrier"])
            # it 2t+1, phase 1: G(2t+1) [buffer 1] || P(2t) from slot 0, dS(2t) [buffer 0] -> PK(0)
            fl = mail_reads(0, ("mr", c, 1)) + ds_units(0, 0, ("mr", c, 1))
            xp = {}
            for i_, ins in enumerate(adv_all(R)):
                xp.setdefault(1 + i_, []).append(ins)
            items += phase(ph[2], seq[24:], fl[:n1], extra_post=xp)
            # phase 2: dK(2t-1) [PK(1), half 1 of tile t - 1] || rest of dS(2t); RA / statistics address -> tile t + 1 (the reads of
            # G(2t+1) were issued in the phase before THIS one... no: in phase 2 of iteration 2t; G(2t+2)'s follow behind these MFMAs, so
            # the move is pinned in front of the first lookahead read: behind MFMA 0); TA -> tile t behind its last reads (phase 1)
            xp = {}
            for j, a_ in enumerate([RA[0], t[3]] + RA[1:]):      # RA[k] is read again behind MFMA 5 + 2k (the next phase's fragment k)
                xp.setdefault((0, 0, 2, 2, 4, 6, 8, 10, 12)[j], []).append(vadd_imm(a_, step_to(nxt_stage)))
            for j, a_ in enumerate(TA[0] + TA[1]):
                xp.setdefault((5, 7, 9, 11, 13, 15, 14, 12)[j], []).append(vadd_imm(a_, step_to(c)))
            items += phase(ph[3], seq[32:], fl[n1:], extra_post=xp, tail=["s_waitcnt vmcnt(0)", "s_barrier"])
        if c == 3:
            items.append(f"s_branch {label(f'copy{R}0')}")
        return items

    bodies = [body_items(c) for c in range(4)]
    # the ring is primed with the first fragments of body 0 and DRAINED (the steady-state counts of copy 0 assume the previous copy)
    for f in (body(0)[0] + body(0)[1])[:AHEAD]:
        emit(strip(f.reads))
    emit("s_waitcnt lgkmcnt(0)")
    for c in range(4):
        emit(resolve(bodies[c], context=bodies[(c - 1) & 3]))
    out.append(label("done" + R) + ":")
    emit("s_waitcnt lgkmcnt(0)")
    if R == "D":
        # the last unit's dS and the last tile's dK: P(2NT-1) from slot 1 (written in the last iteration), G''(2NT-1) in buffer 1;
        # pairs PK(0) hold dS(2NT-2); TA stands on tile NT - 1
        emit(strip(mail_reads(4096, "fin")))
        emit("s_waitcnt lgkmcnt(0)")
        emit(["s_nop 15", "s_nop 15"])
        emit(strip(ds_units(1, 1, "fin")[1:]))
        emit("s_nop 7")
        emit(strip(plain_frags(tr_frags(("fin", 0), R, 0, 0))))
        emit(strip(plain_frags(tr_frags(("fin", 1), R, 1, 1))))
    emit(f"s_branch {label('epi')}")


role_stream("P")
out.append(label("role_d") + ":")
# the dS-wave's transposing reads start on "tile -1" = stage 3 (zero or the ragged tail: finite) with dS = 0
for a_ in TA[0] + TA[1]:
    emit(vadd_imm(a_, 3 * STAGE))
role_stream("D")
# ---- epilogue (both roles): out[row][dblk*32 + 8 rq + 4 hi + e] = acc^T[h][dblk][4 rq + e] * factor, rows < NSTO ----
out.append(label("epi") + ":")
emit("s_waitcnt vmcnt(0)")
emit(["s_nop 15", "s_nop 15"])
emit(f"v_and_b32 v{t[0]}, 31, %[tid]")
emit(f"v_bfe_u32 v{t[1]}, %[tid], 5, 1")
emit(f"s_lshr_b32 s{ST[0]}, s{WAVE}, 1")
emit(f"s_lshl_b32 s{ST[0]}, s{ST[0]}, 6")
emit(f"v_add_u32 v{t[2]}, s{ST[0]}, v{t[0]}")
emit(f"v_lshlrev_b32 v{t[1]}, 3, v{t[1]}")
E_ = [SG(0, 0, e) for e in range(4)]
for h in range(2):
    if h:
        emit(f"v_add_u32 v{t[2]}, 32, v{t[2]}")
    emit(f"v_mul_lo_u32 v{t[3]}, v{t[2]}, s{OLS}")
    emit(f"v_add_u32 v{t[3]}, v{t[3]}, v{t[1]}")
    emit(f"v_cmp_gt_u32 vcc, s{NSTO}, v{t[2]}")
    emit(f"s_and_saveexec_b64 {sr(EXS)}, vcc")
    for d in range(4):
        for rq in range(4):
            for e in range(4):
                emit(f"v_accvgpr_read_b32 v{E_[e]}, a{O(h, d, rq * 4 + e)}")
            for e in range(4):
                emit(f"v_mul_f32 v{E_[e]}, v{OSC + h}, v{E_[e]}")
            emit(f"v_cvt_pk_bf16_f32 v{E_[0]}, v{E_[0]}, v{E_[1]}")
            emit(f"v_cvt_pk_bf16_f32 v{E_[1]}, v{E_[2]}, v{E_[3]}")
            emit(f"global_store_dwordx2 v{t[3]}, {vr(E_[0], 2)}, {sr(OP)} offset:{(d * 32 + rq * 8) * 2}")
            emit("s_nop 1")
    emit(f"s_mov_b64 exec, {sr(EXS)}")
emit("s_waitcnt vmcnt(0)")

# ---------------- write ----------------
assert all(isinstance(s_, str) for s_ in out), [s_ for s_ in out if not isinstance(s_, str)][:3]
n_mfma = sum(1 for s_ in out if s_.startswith("v_mfma"))
with open(args.o, "w") as fh:
    fh.write("// GENERATED by tools/gen_attn_bwd64_kv.py %s — do not edit; %d instructions, %d MFMAs\n" %
             (" ".join(a for a in sys.argv[1:] if not a.startswith("more4d") and a != "-o"),
              sum(1 for s_ in out if not s_.endswith(":") and not s_.startswith(";")), n_mfma))
    for s_ in out:
        if s_.startswith(";"):
            continue
        fh.write('"%s\\n\\t"\n' % s_)
print(f"{args.o}: {len(out)} lines, {n_mfma} MFMAs", file=sys.stderr)
