#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attn128q_kernel (more4d_amd/csrc/attention_q64.h).

    python tools/gen_attn_q64.py [--plain] [--cap N] [--eb N] [--stamps] [-o more4d_amd/csrc/attention_q64_gen.inc]

The kernel is ONE inline-asm block with hand-assigned registers: one wave per SIMD (4 waves x 64 query rows, the whole 512-register
file per lane), softmax arithmetic placed as fillers into the issue shadows of a continuous MFMA stream.  This script owns the
register map and the schedule; the emitted .inc is committed next to it.  The C++ wrapper (attention_q64.h) maps the workgroup to
(batch, head, 256-query tile), stages the ragged key tail, and passes pointers / strides as SGPR inputs.

Per 64-key tile and wave: QK^T 32 MFMAs (v_mfma_f32_32x32x16_bf16: 2 key blocks x 2 query halves x 8 k-steps) + PV 32 MFMAs
(4 head-dim blocks x 2 query halves x 4 k-steps) = 64 x 32 cycles of the SIMD's matrix pipe.  Everything else has to fit into the
~7 issue slots between two MFMAs.  Steady state, iteration i (S buffers X = i & 1, Y = X ^ 1):
  phase A(i): QK(i+1) -> S'(Y)      || finish softmax(i) on S'(X): remaining exp2 / row sums, bf16 packing -> P(i)
  phase B(i): O += V^T(i) P(i)      || start softmax(i+1) on S'(Y): row max + check against the lazy reference, first exp2s
S' = K Q~^T - R: Q~ = Q * scale * log2(e) (bf16, prescaled once per workgroup) and the lazy reference maximum R of the row enters as
the C operand of the first k-step (a 16-register tuple of -R per query half), so a score costs exp2 + row-sum add + 1/2 pack + 1/2
max3 and no multiply-subtract.  R moves only when a tile maximum exceeds it by 2^8 (first tile: always): the rare path finishes the
phase without fillers, rescales O / l, shifts S' and rewrites the -R tuples (subroutine fix_Y).

Register map (arch VGPRs v16.., AGPRs, SGPRs s40..): see the constants below.  K / V^T fragments arrive through an 8-slot ring of
AGPRs by ds_read_b128, one read per fragment, each fragment feeds two MFMAs (the two query halves), read 8 fragments ahead.
K / V^T tiles: four 32 KiB LDS stages (K at +0, V^T at +16 KiB), tile t in stage t & 3, global->LDS DMA: K(i+4) and V^T(i+3) are
requested in phase B(i); one s_barrier per tile (end of phase B) behind s_waitcnt vmcnt(12) (K(i+3) landed).
"""
import argparse
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--plain", action="store_true", help="debug: no interleaving (MFMA stream, then the fillers)")
ap.add_argument("--cap", type=int, default=5, help="instructions per MFMA gap besides the MFMA")
ap.add_argument("--eb", type=int, default=10, help="softmax units (2 scores per query half) exponentiated in phase B")
ap.add_argument("--stamps", action="store_true", help="s_memtime stamps of wave 0 of workgroup 1000 at the phase boundaries of tiles 100..107")
ap.add_argument("--wait1", action="store_true", help="one s_waitcnt lgkmcnt per fragment instead of one per TWO fragments (A/B: +0.8 %% time)")
ap.add_argument("--first-gap", type=int, default=1, help="phase B: no fillers behind its first N MFMAs (0: -0.8 %% time, but the first row-max "
                "instructions would read accumulators one MFMA behind their last write)")
ap.add_argument("--dma-m0", action="store_true", help="A/B: one m0 write per DMA piece instead of one per K / V^T request of four pieces with the pieces' LDS steps in the "
                "instruction offset (which moves the global address too: cancelled in the lane offsets); -0.6 %% time for the default")
ap.add_argument("--exp-alt", action="store_true", help="A/B: exp / add alternating instead of exp exp add add")
ap.add_argument("--abl", default="", help="timing ablations for the stamps build (results WRONG): letters e = no exp / add / pack fillers, m = no row-max list, "
                "d = no DMA requests, r = no fragment reads, w = no lgkm waits, a = no address updates, b = no barrier / vmcnt")
ap.add_argument("-o", default="more4d_amd/csrc/attention_q64_gen.inc")
args = ap.parse_args()

STAGE, VOFF, RAGOFF = 32768, 16384, 4 * 32768

# ---------------- register map ----------------
R = [16, 17]                      # lazy reference maximum per query half (log2 domain)
TMP = list(range(18, 32))         # scratch (prologue / epilogue / fix-up)


def S(buf, h, sub, r=0):
    return 32 + buf * 64 + h * 32 + sub * 16 + r


def P(h, c, j=0):
    return 160 + h * 16 + c * 4 + j


def NM(h, r=0):
    return 192 + h * 16 + r


KA = [224 + i for i in range(8)]
VA = [232 + i for i in range(4)]
DK = [236 + i for i in range(4)]
DV = [240 + i for i in range(4)]


def L(h, j):
    return 244 + h * 2 + j


def TM(h, sub):
    return 248 + h * 2 + sub


MX = [252, 253]
U = [254, 255]


def O(h, d, r=0):
    return (h * 4 + d) * 16 + r


def Q(h, kk, r=0):
    return 128 + (h * 8 + kk) * 4 + r


def RING(slot):
    return 192 + slot * 4


# SGPRs
KP, VP, KTS, KSTEP, VSTEP, KCNT = 40, 42, 44, 45, 46, 47
WB, REM, THR, SC = 48, 49, 50, 51
QP, OP, LP = 52, 54, 56
QLS, OLS, NROWS, RAG, LDS0, WAVE = 58, 59, 60, 61, 62, 63
ST = list(range(64, 80))          # scalar temps
RET, TGT, EXS = 80, 82, 84
KLS, VLS = 86, 87
DBG, STMP = 88, 90
VCNT, RET2, KSEG, VSEG = 91, 92, 94, 95
KARG, BIDX, HIDX, NSEG, NRAG = 96, 98, 99, 100, 101
RAGLO, RAGHI = 78, 79      # (= ST[14], ST[15]; prologue only) k_lim of ragged slots 0..3 / 4, one byte each; RAG = k_lim of the tile in flight
# byte offsets inside the kernel-argument segment (AttnArgs: q, out, kv{k[8], vt[8], k_bs[8], k_ls[8], vt_bs[8], vt_ls[8], len[8], ...};
# static_asserts in attention_q64.h)
KA_K, KA_VT, KA_KBS, KA_VTBS, KA_LEN = 16, 80, 144, 272, 400


def vr(a, n):
    return f"v[{a}:{a + n - 1}]"


def ar(a, n):
    return f"a[{a}:{a + n - 1}]"


def sr(a, n=2):
    return f"s[{a}:{a + n - 1}]"


out = []
_uid = [0]


def emit(s):
    """a line, a list of lines, or a callable that returns either (items that carry labels are generated afresh per emission: the
    skeleton of phase B is emitted twice, in the loop and in the rare path)"""
    if callable(s):
        s = s()
    if isinstance(s, (list, tuple)):
        for x in s:
            emit(x)
    else:
        out.append(s)


def label(name):
    return f".Lq64_{name}_%="


def put_label(name):
    emit(label(name) + ":")


# ---------------- instruction builders ----------------
def mfma_qk(buf, h, sub, kk, slot):
    c = vr(NM(h), 16) if kk == 0 else vr(S(buf, h, sub), 16)
    return f"v_mfma_f32_32x32x16_bf16 {vr(S(buf, h, sub), 16)}, {ar(RING(slot), 4)}, {ar(Q(h, kk), 4)}, {c}"


def mfma_pv(h, d, c, slot):
    return f"v_mfma_f32_32x32x16_bf16 {ar(O(h, d), 16)}, {ar(RING(slot), 4)}, {vr(P(h, c), 4)}, {ar(O(h, d), 16)}"


def read_k(f, slot):      # K fragment f = kk * 2 + sub of the tile KA[] points at
    return f"ds_read_b128 {ar(RING(slot), 4)}, v{KA[f >> 1]} offset:{(f & 1) * 8192}"


def read_v(g, slot):      # V^T fragment g = c * 4 + dblk of the tile VA[] points at
    return f"ds_read_b128 {ar(RING(slot), 4)}, v{VA[g >> 2]} offset:{(g & 3) * 4096}"


def step_to(stage):       # address delta that moves a pointer INTO `stage` from the previous one
    return 0xFFFE8000 if stage == 0 else 0x8000


def vadd_imm(reg, imm):
    return f"v_add_u32 v{reg}, 0x{imm & 0xFFFFFFFF:x}, v{reg}"


def max_list(buf):
    """row maximum of S'(buf) over the 64 keys of the tile, both query halves, compared with THR -> vcc"""
    ins = []
    for k in range(8):
        for sub in range(2):          # the order the QK chains finished in: (h0, sub0), (h1, sub0), (h0, sub1), (h1, sub1)
            for h in range(2):
                t, b = TM(h, sub), S(buf, h, sub)
                if k == 0:
                    ins.append(f"v_max3_f32 v{t}, v{b}, v{b + 1}, v{b + 2}")
                elif k < 7:
                    ins.append(f"v_max3_f32 v{t}, v{t}, v{b + 2 * k + 1}, v{b + 2 * k + 2}")
                else:
                    ins.append(f"v_max3_f32 v{t}, v{t}, v{b + 15}, v{b + 15}")
    for h in range(2):
        ins.append(f"v_max_f32 v{MX[h]}, v{TM(h, 0)}, v{TM(h, 1)}")
    ins.append("s_nop 0")
    for h in range(2):
        ins.append(f"v_mov_b32 v{U[h]}, v{MX[h]}")
    ins.append("s_nop 1")
    for h in range(2):
        ins.append(f"v_permlane32_swap_b32 v{U[h]}, v{MX[h]}")
    ins.append("s_nop 1")
    for h in range(2):
        ins.append(f"v_max_f32 v{MX[h]}, v{U[h]}, v{MX[h]}")
    ins.append(f"v_max_f32 v{U[0]}, v{MX[0]}, v{MX[1]}")
    ins.append(f"v_cmp_lt_f32 vcc, s{THR}, v{U[0]}")
    ins.append("s_nop 1")
    return ins


def unit_regs(buf, u):
    c, h, j = u >> 3, (u >> 2) & 1, u & 3
    b = S(buf, h, c >> 1) + (c & 1) * 8 + 2 * j
    return h, c, j, b


def exp_units(buf, u0, u1, cvt=True, lead_cvt=()):
    """softmax units [u0, u1) of S'(buf): exp2 in place, row sums, bf16 packing (cvt) one unit behind the exps.
    lead_cvt: units whose exps / sums are already done (phase B) and only need packing"""
    ins = []
    for u in lead_cvt:
        h, c, j, b = unit_regs(buf, u)
        ins.append(f"v_cvt_pk_bf16_f32 v{P(h, c, j)}, v{b}, v{b + 1}")
    prev = None
    for u in range(u0, u1):
        h, c, j, b = unit_regs(buf, u)
        if args.exp_alt and prev is not None:
            tl = tail_of_unit(buf, prev, cvt)
            ins += [f"v_exp_f32 v{b}, v{b}", tl[0], f"v_exp_f32 v{b + 1}, v{b + 1}"] + tl[1:]
            prev = u
            continue
        ins.append(f"v_exp_f32 v{b}, v{b}")
        ins.append(f"v_exp_f32 v{b + 1}, v{b + 1}")
        if prev is not None:
            ins += tail_of_unit(buf, prev, cvt)
        prev = u
    if prev is not None:
        ins.append("s_nop 0")
        ins += tail_of_unit(buf, prev, cvt)
    return ins


def tail_of_unit(buf, u, cvt):
    h, c, j, b = unit_regs(buf, u)
    ins = [f"v_add_f32 v{L(h, 0)}, v{L(h, 0)}, v{b}", f"v_add_f32 v{L(h, 1)}, v{L(h, 1)}, v{b + 1}"]
    if cvt:
        ins.append(f"v_cvt_pk_bf16_f32 v{P(h, c, j)}, v{b}, v{b + 1}")
    return ins


# ---------------- stream emitter ----------------
class Cont:
    """what is left of a phase after the branch point (slow path = the same skeleton without fillers)"""
    def __init__(self):
        self.items = None


def _abl_keep(item):
    if callable(item):
        return "d" not in args.abl
    t_ = item.split()[0]
    if t_ == "ds_read_b128":
        return "r" not in args.abl
    if t_ == "s_waitcnt":
        return ("b" not in args.abl) if "vmcnt" in item else ("w" not in args.abl)
    if t_ == "s_barrier":
        return "b" not in args.abl
    if t_ in ("global_load_lds_dwordx4",) or (t_ == "s_add_u32" and " m0," in item):
        return "d" not in args.abl
    if t_ == "v_add_u32":
        return "a" not in args.abl
    if t_ in ("s_add_u32", "s_addc_u32") and "d" in args.abl:
        return False
    return True


def emit_stream(mfmas, pre, post, fillers, cont=None, branch_label=None, first_gap=0):
    if args.abl:
        pre = [[x for x in l if _abl_keep(x)] for l in pre]
        post = [[x for x in l if _abl_keep(x)] for l in post]
    """mfmas[k]; pre[k] = instructions right in front of MFMA k; post[k] = pinned instructions behind it; fillers = ordered list
    placed into what the cap leaves of every gap.  The item "BRANCH" in the fillers becomes s_cbranch_vccnz branch_label and records
    the remainder of the skeleton in cont.items."""
    n = len(mfmas)
    fl = list(fillers)
    if args.plain:
        for k in range(n):
            for s in pre[k]:
                emit(s)
            emit(mfmas[k])
            for s in post[k]:
                emit(s)
        emit("s_nop 15")
        emit("s_nop 15")
        for f in fl:
            if f == "BRANCH":
                emit(f"s_cbranch_vccnz {branch_label}")
                cont.items = []
            else:
                emit(f)
        return
    for k in range(n):
        for s in pre[k]:
            emit(s)
        emit(mfmas[k])
        used = len(pre[k + 1]) if k + 1 < n else 0
        for s in post[k]:
            emit(s)
            used += 4 if callable(s) else 1
        room = args.cap - used if k >= first_gap else 0
        if k == n - 1:
            room = len(fl)          # whatever is left goes behind the last MFMA
        gap = []
        while room > 0 and fl:
            gap.append(fl.pop(0))
            room -= 1
        for f in gap:
            if f == "BRANCH":
                emit(f"s_cbranch_vccnz {branch_label}")
                cont.items = []
                for kk in range(k + 1, n):
                    cont.items += list(pre[kk]) + [mfmas[kk]] + list(post[kk])
            else:
                emit(f)
    assert not fl


def wait_lgkm(n):
    return f"s_waitcnt lgkmcnt({n})"


def frag_wait(n):
    """wait in front of fragment n of the continuous stream (8 reads in flight, in-order returns)"""
    if args.wait1:
        return [wait_lgkm(7)]
    return [wait_lgkm(6)] if n % 2 == 0 else []


# ---------------- phases ----------------
def phase_a(c, fillers):
    """QK(i+1) into S'(Y), i = c mod 4; reads K(i+1) fragments 8..15 (KA[4..7]) and V^T(i) fragments 0..7 (VA[0..1])"""
    Y = (c + 1) & 1
    mf, pre, post = [], [], []
    for f in range(16):
        for h in range(2):
            mf.append(mfma_qk(Y, h, f & 1, f >> 1, f % 8))
            pre.append(frag_wait(f) if h == 0 else [])
            post.append([])
        k = 2 * f + 1
        post[k].append(read_k(f + 8, f % 8) if f < 8 else read_v(f - 8, f % 8))
    # pointer advances, at least one gap behind the last read through them
    for j in range(4):          # KA[4 + j]: fragments 8 + 2j, 9 + 2j, read behind MFMA 2 (2j + 1) + 1
        post[2 * (2 * j + 1) + 3].append(vadd_imm(KA[4 + j], step_to((c + 2) & 3)))
    post[2 * 11 + 3].append(vadd_imm(VA[0], step_to((c + 1) & 3)))
    post[31].append(vadd_imm(VA[1], step_to((c + 1) & 3)))
    return mf, pre, post, fillers


def dma_piece(kind, lds_off, p):
    """piece p (1 KiB per wave) of a K / V^T tile request into LDS offset lds_off (+ the wave's 4 KiB)"""
    reg, ptr = (DK[p], KP) if kind == "k" else (DV[p], VP)
    if not args.dma_m0:      # (the lane offsets were reduced by p KiB in the prologue)
        return ([f"s_add_u32 m0, s{WB}, 0x{lds_off:x}"] if p == 0 else []) + [f"global_load_lds_dwordx4 v{reg}, {sr(ptr)} offset:{p * 1024}"]
    return [f"s_add_u32 m0, s{WB}, 0x{lds_off + p * 1024:x}", f"global_load_lds_dwordx4 v{reg}, {sr(ptr)}"]


def phase_b(c, fillers, last_barrier=True):
    """O += V^T(i) P(i); reads V^T(i) fragments 8..15 (VA[2..3]) and K(i+2) fragments 0..7 (KA[0..3]); requests K(i+4), V^T(i+3)"""
    mf, pre, post = [], [], []
    for g in range(16):
        for h in range(2):
            mf.append(mfma_pv(h, g & 3, g >> 2, g % 8))
            pre.append(frag_wait(g) if h == 0 else [])
            post.append([])
        k = 2 * g + 1
        post[k].append(read_v(g + 8, g % 8) if g < 8 else read_k(g - 8, g % 8))
    post[2 * 3 + 3].append(vadd_imm(VA[2], step_to((c + 1) & 3)))
    post[2 * 7 + 3].append(vadd_imm(VA[3], step_to((c + 1) & 3)))
    for j in range(3):
        post[2 * (8 + 2 * j + 1) + 3].append(vadd_imm(KA[j], step_to((c + 3) & 3)))
    post[31].append(vadd_imm(KA[3], step_to((c + 3) & 3)))
    # DMA: K(i+4) -> K half of stage c, V^T(i+3) -> V half of stage (c+3)&3; one piece behind every second MFMA, then the pointers move on
    ks, vs = c * STAGE, ((c + 3) & 3) * STAGE + VOFF
    for p in range(4):
        post[2 + 2 * p] += dma_piece("k", ks, p)
    for p in range(4):
        post[10 + 2 * p] += dma_piece("v", vs, p)
    post[18].append(adv_test("k"))
    post[20] += adv_step("k")
    post[22].append(adv_test("v"))
    post[24] += adv_step("v")
    if last_barrier:
        post[31] += ["s_waitcnt vmcnt(12)", "s_barrier"]
    return mf, pre, post, fillers


def plain_qk(buf):
    """S'(buf) = K Q~^T + (-R) of the tile KA[] points at, no fillers; the ring is idle on entry and on exit"""
    for f in range(8):
        emit(read_k(f, f))
    for f in range(16):
        emit(wait_lgkm(min(7, 15 - f)))
        for h in range(2):
            emit(mfma_qk(buf, h, f & 1, f >> 1, f % 8))
        if f + 8 < 16:
            emit(read_k(f + 8, f % 8))


def plain_pv():
    for g in range(8):
        emit(read_v(g, g))
    for g in range(16):
        emit(wait_lgkm(min(7, 15 - g)))
        for h in range(2):
            emit(mfma_pv(h, g & 3, g >> 2, g % 8))
        if g + 8 < 16:
            emit(read_v(g + 8, g % 8))


_calls = [0]


def call(name, ret=None):
    """s_swappc to a subroutine placed BEHIND every call site (positive offset); ret = SGPR pair that takes the return address"""
    _calls[0] += 1
    n = _calls[0]
    here = label(f"pc{n}")
    emit(f"s_getpc_b64 {sr(TGT)}")
    put_label(f"pc{n}")
    emit(f"s_add_u32 s{TGT}, s{TGT}, {label(name)}-{here}")
    emit(f"s_addc_u32 s{TGT + 1}, s{TGT + 1}, 0")
    emit(f"s_swappc_b64 {sr(RET if ret is None else ret)}, {sr(TGT)}")


def call_fix(buf):
    call(f"fix{buf}")


# K / V^T tile iterators over the segment list (SGPRs: pointer of the NEXT request, tiles left in its segment, step).  The fast path of an
# advance is five scalar instructions; the end of a segment branches to an out-of-line block that calls the segment routine.
_sites = []


def adv_test(kind):
    """first half of an advance (a callable: fresh labels per emission): count down, leave for the segment switch at zero"""
    def gen():
        _calls[0] += 1
        n = _calls[0]
        cnt = KCNT if kind == "k" else VCNT
        site, back = label(f"sw{kind}{n}"), label(f"bk{kind}{n}")
        _sites.append((site, back, kind))
        return [f"s_sub_u32 s{cnt}, s{cnt}, 1", f"s_cmp_eq_u32 s{cnt}, 0", f"s_cbranch_scc1 {site}", back + ":"]
    return gen


def adv_step(kind):
    ptr, step = (KP, KSTEP) if kind == "k" else (VP, VSTEP)
    return [f"s_add_u32 s{ptr}, s{ptr}, s{step}", f"s_addc_u32 s{ptr + 1}, s{ptr + 1}, 0"]


def stamp(slot):
    if not args.stamps:
        return
    # wave 0 of workgroup 1000, tiles 100..107 (STMP = tile counter, -1 = off): dbg[(tile - 100) * 4 + slot] = s_memtime
    skip = label(f"st{len(out)}")
    emit(f"s_cmp_lt_u32 s{STMP}, 8")
    emit(f"s_cbranch_scc0 {skip}")
    emit(f"s_memtime {sr(ST[12])}")
    emit(f"s_lshl_b32 s{ST[14]}, s{STMP}, 5")
    emit(f"s_add_u32 s{ST[14]}, s{ST[14]}, {slot * 8}")
    emit(f"v_mov_b32 v{TMP[12]}, s{ST[14]}")
    emit("s_waitcnt lgkmcnt(0)")
    emit(f"v_mov_b32 v{TMP[10]}, s{ST[12]}")
    emit(f"v_mov_b32 v{TMP[11]}, s{ST[13]}")
    emit(f"global_store_dwordx2 v{TMP[12]}, {vr(TMP[10], 2)}, {sr(DBG)}")
    emit(skip + ":")


# =====================================================================================================================
# kernel body
# =====================================================================================================================
t = TMP
emit("; ---- inputs -> fixed registers ----")
for dst, src in ((KARG, "ka_lo"), (KARG + 1, "ka_hi"), (BIDX, "b"), (HIDX, "h"), (NSEG, "nseg"), (NRAG, "nrag"), (RAGLO, "rag_lo"),
                 (RAGHI, "rag_hi"), (QP, "qp_lo"), (QP + 1, "qp_hi"), (OP, "op_lo"), (OP + 1, "op_hi"), (LP, "lp_lo"), (LP + 1, "lp_hi"),
                 (KLS, "klsb"), (VLS, "vlsb"), (QLS, "qlsb"), (OLS, "olsb"), (NROWS, "nrows"), (SC, "sc"), (LDS0, "lds0"),
                 (DBG, "dbg_lo"), (DBG + 1, "dbg_hi"), (STMP, "stamp")):
    emit(f"s_mov_b32 s{dst}, %[{src}]")
emit(f"s_lshl_b32 s{KTS}, s{KLS}, 6")                      # bytes per 64-key K tile
# lane constants: t0 = lane, t1 = li, t2 = hi, t3 = wave
emit(f"v_and_b32 v{t[0]}, 63, %[tid]")
emit(f"v_and_b32 v{t[1]}, 31, %[tid]")
emit(f"v_bfe_u32 v{t[2]}, %[tid], 5, 1")
emit(f"v_bfe_u32 v{t[3]}, %[tid], 6, 2")                    # (the upper bits of the work-item id register are not zero)
# measured on gfx950 (tools/q64_dump.py, round 5): a v_readfirstlane issued right behind the VALU that writes its source reads the OLD
# register (hipcc pads this hazard itself; it does not look inside inline asm)
emit("s_nop 3")
emit(f"v_readfirstlane_b32 s{WAVE}, v{t[3]}")
emit("s_nop 3")
if args.stamps:      # only wave 0 stamps
    emit(f"s_cmp_lg_u32 s{WAVE}, 0")
    emit(f"s_cselect_b32 s{STMP}, 0x80000000, s{STMP}")
emit(f"s_lshl_b32 s{WB}, s{WAVE}, 12")
emit(f"s_add_u32 s{WB}, s{WB}, s{LDS0}")
# kr = perm23(li) = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1)
emit(f"v_and_b32 v{t[4]}, 0x13, v{t[1]}")
emit(f"v_and_b32 v{t[5]}, 4, v{t[1]}")
emit(f"v_lshlrev_b32 v{t[5]}, 1, v{t[5]}")
emit(f"v_or_b32 v{t[4]}, v{t[4]}, v{t[5]}")
emit(f"v_and_b32 v{t[5]}, 8, v{t[1]}")
emit(f"v_lshrrev_b32 v{t[5]}, 1, v{t[5]}")
emit(f"v_or_b32 v{t[4]}, v{t[4]}, v{t[5]}")                 # t4 = kr
emit(f"v_and_b32 v{t[5]}, 15, v{t[4]}")                    # t5 = kr & 15
emit(f"v_lshlrev_b32 v{t[6]}, 8, v{t[4]}")                 # t6 = kr * 256
emit(f"v_add_u32 v{t[6]}, s{LDS0}, v{t[6]}")
for kk in range(8):                                         # KA[kk] = lds0 + kr*256 + (((kk*2 + hi) ^ (kr & 15)) << 4)
    emit(f"v_add_u32 v{t[7]}, {kk * 2}, v{t[2]}")
    emit(f"v_xor_b32 v{t[7]}, v{t[7]}, v{t[5]}")
    emit(f"v_lshlrev_b32 v{t[7]}, 4, v{t[7]}")
    emit(f"v_add_u32 v{KA[kk]}, v{t[6]}, v{t[7]}")
emit(f"v_lshlrev_b32 v{t[6]}, 7, v{t[1]}")                 # li * 128
emit(f"v_add_u32 v{t[6]}, s{LDS0}, v{t[6]}")
emit(f"v_add_u32 v{t[6]}, 0x{VOFF:x}, v{t[6]}")
emit(f"v_bfe_u32 v{t[5]}, v{t[1]}, 1, 3")                  # (li >> 1) & 7
for c in range(4):                                          # VA[c] = lds0 + VOFF + li*128 + (((c*2 + hi) ^ ((li>>1)&7)) << 4)
    emit(f"v_add_u32 v{t[7]}, {c * 2}, v{t[2]}")
    emit(f"v_xor_b32 v{t[7]}, v{t[7]}, v{t[5]}")
    emit(f"v_lshlrev_b32 v{t[7]}, 4, v{t[7]}")
    emit(f"v_add_u32 v{VA[c]}, v{t[6]}, v{t[7]}")
# DMA source offsets.  K piece p of wave w: row = 16w + 4p + (lane >> 4), 16-byte chunk (lane & 15) ^ (row & 15)
emit(f"v_lshrrev_b32 v{t[4]}, 4, v{t[0]}")                 # lane >> 4
emit(f"v_and_b32 v{t[5]}, 15, v{t[0]}")                    # lane & 15
for p in range(4):
    emit(f"v_add_u32 v{t[6]}, {4 * p}, v{t[4]}")           # row & 15
    emit(f"v_xor_b32 v{t[7]}, v{t[5]}, v{t[6]}")
    emit(f"v_lshlrev_b32 v{t[7]}, 4, v{t[7]}")
    emit(f"s_lshl_b32 s{ST[0]}, s{WAVE}, 4")
    emit(f"v_add_u32 v{t[6]}, s{ST[0]}, v{t[6]}")          # row
    emit(f"v_mul_lo_u32 v{t[6]}, v{t[6]}, s{KLS}")
    emit(f"v_add_u32 v{DK[p]}, v{t[6]}, v{t[7]}")
    if not args.dma_m0 and p:
        emit(f"v_subrev_u32 v{DK[p]}, {p * 1024}, v{DK[p]}")
# V^T piece p of wave w: row = 32w + 8p + (lane >> 3), chunk (lane & 7) ^ ((row >> 1) & 7)
emit(f"v_lshrrev_b32 v{t[4]}, 3, v{t[0]}")
emit(f"v_and_b32 v{t[5]}, 7, v{t[0]}")
for p in range(4):
    emit(f"s_lshl_b32 s{ST[0]}, s{WAVE}, 5")
    emit(f"v_add_u32 v{t[6]}, {8 * p}, v{t[4]}")
    emit(f"v_add_u32 v{t[6]}, s{ST[0]}, v{t[6]}")          # row
    emit(f"v_bfe_u32 v{t[7]}, v{t[6]}, 1, 3")
    emit(f"v_xor_b32 v{t[7]}, v{t[5]}, v{t[7]}")
    emit(f"v_lshlrev_b32 v{t[7]}, 4, v{t[7]}")
    emit(f"v_mul_lo_u32 v{t[6]}, v{t[6]}, s{VLS}")
    emit(f"v_add_u32 v{DV[p]}, v{t[6]}, v{t[7]}")
    if not args.dma_m0 and p:
        emit(f"v_subrev_u32 v{DV[p]}, {p * 1024}, v{DV[p]}")
# ---- Q rows: row = 64 w + 32 h + li (clamped to nrows - 1), 16 loads of 16 bytes per lane into v32..v95 ----
emit(f"s_sub_u32 s{ST[1]}, s{NROWS}, 1")
for h in range(2):
    emit(f"s_lshl_b32 s{ST[0]}, s{WAVE}, 6")
    emit(f"v_add_u32 v{t[6]}, s{ST[0]}, v{t[1]}")
    if h:
        emit(f"v_add_u32 v{t[6]}, 32, v{t[6]}")
    emit(f"v_min_u32 v{t[6]}, s{ST[1]}, v{t[6]}")
    emit(f"v_mul_lo_u32 v{t[6]}, v{t[6]}, s{QLS}")
    emit(f"v_lshlrev_b32 v{t[7]}, 4, v{t[2]}")             # hi * 16 bytes
    emit(f"v_add_u32 v{t[8 + h]}, v{t[6]}, v{t[7]}")
    for kk in range(8):
        emit(f"global_load_dwordx4 {vr(32 + (h * 8 + kk) * 4, 4)}, v{t[8 + h]}, {sr(QP)} offset:{kk * 32}")


def dma_prologue():
    """iterators to the first tile, then the requests K0 V0 K1 K2 V1 K3 V2 (>= 4 full tiles: host-checked); what the loop expects: the next
    K request is tile 4, the next V^T request tile 3"""
    emit(f"s_mov_b32 s{KSEG}, -1")
    emit(f"s_mov_b32 s{VSEG}, -1")
    emit(f"s_mov_b32 s{KSTEP}, s{KTS}")
    emit(f"s_mov_b32 s{VSTEP}, 128")
    call("kseg_next")
    call("vseg_next")
    for kind, tile in (("k", 0), ("v", 0), ("k", 1), ("k", 2), ("v", 1), ("k", 3), ("v", 2)):
        for p in range(4):
            emit(dma_piece(kind, (tile & 3) * STAGE + (0 if kind == "k" else VOFF), p))
        emit(adv_test(kind))
        emit(adv_step(kind))


# ---- DMA prologue right away unless ragged tails sit in the pipeline stages (more than one ragged tail: slots 1..4 = stages 0..3) ----
emit(f"s_cmp_gt_u32 s{NRAG}, 1")
emit(f"s_cbranch_scc1 {label('nodma0')}")
dma_prologue()
put_label("nodma0")
# ---- state: O = 0, l = 0, -R tuples = 0, R = 0, THR = -inf ----
emit(f"v_mov_b32 v{t[0]}, 0")
for r in range(128):
    emit(f"v_accvgpr_write_b32 a{r}, v{t[0]}")
for h in range(2):
    emit(f"v_mov_b32 v{R[h]}, 0")
    for j in range(2):
        emit(f"v_mov_b32 v{L(h, j)}, 0")
    for r in range(16):
        emit(f"v_mov_b32 v{NM(h, r)}, 0")
emit(f"s_mov_b32 s{THR}, 0xff800000")
# ---- Q~ = bf16(Q * sc) -> a128..a191 ----
emit(f"s_cmp_gt_u32 s{NRAG}, 1")
emit(f"s_cbranch_scc1 {label('qwait0')}")
emit("s_waitcnt vmcnt(28)")                                  # the 28 DMA pieces behind the Q loads may stay in flight
put_label("qwait0")
emit(f"s_cmp_le_u32 s{NRAG}, 1")
emit(f"s_cbranch_scc1 {label('qwait1')}")
emit("s_waitcnt vmcnt(0)")
put_label("qwait1")
for d in range(64):
    src = 32 + d
    emit(f"v_lshlrev_b32 v{t[0]}, 16, v{src}")
    emit(f"v_and_b32 v{t[1]}, 0xffff0000, v{src}")
    emit(f"v_mul_f32 v{t[0]}, s{SC}, v{t[0]}")
    emit(f"v_mul_f32 v{t[1]}, s{SC}, v{t[1]}")
    emit(f"v_cvt_pk_bf16_f32 v{t[2]}, v{t[0]}, v{t[1]}")
    emit(f"v_accvgpr_write_b32 a{128 + d}, v{t[2]}")
emit("s_nop 7")
# ---- ragged tails (one per K / V^T segment whose length is not a multiple of 64; staged by the wrapper: slot 0 behind the four stages,
# slots 1..4 in stages 0..3): complete tiles in plain order, subroutine rag_tile ----
emit(f"s_mov_b32 s{REM}, 0")
put_label("ragloop")
emit(f"s_cmp_ge_u32 s{REM}, s{NRAG}")
emit(f"s_cbranch_scc1 {label('ragdone')}")
emit(f"s_lshl_b32 s{ST[1]}, s{REM}, 3")
emit(f"s_lshr_b32 s{RAG}, s{RAGLO}, s{ST[1]}")
emit(f"s_cmp_eq_u32 s{REM}, 4")
emit(f"s_cselect_b32 s{RAG}, s{RAGHI}, s{RAG}")
emit(f"s_and_b32 s{RAG}, s{RAG}, 0xff")                      # k_lim of this slot
emit(f"s_sub_u32 s{ST[1]}, s{REM}, 1")
emit(f"s_lshl_b32 s{ST[1]}, s{ST[1]}, 15")
emit(f"s_cmp_eq_u32 s{REM}, 0")
emit(f"s_cselect_b32 s{ST[1]}, 0x{RAGOFF:x}, s{ST[1]}")     # LDS offset of the slot
for a_ in KA + VA:
    emit(f"v_add_u32 v{a_}, s{ST[1]}, v{a_}")
call("rag_tile", RET2)
for a_ in KA + VA:
    emit(f"v_subrev_u32 v{a_}, s{ST[1]}, v{a_}")
emit(f"s_add_u32 s{REM}, s{REM}, 1")
emit(f"s_branch {label('ragloop')}")
put_label("ragdone")
emit(f"s_cmp_le_u32 s{NRAG}, 1")
emit(f"s_cbranch_scc1 {label('nodma1')}")
emit("s_barrier")                                            # every wave has read the ragged slots in stages 0..3
dma_prologue()
put_label("nodma1")
# ---- tile 0: S'(0) in lock step, start of its softmax, K(1) fragments 0..7 on their way ----
emit("s_waitcnt vmcnt(12)")
emit("s_barrier")
plain_qk(0)
emit("s_barrier")                                           # nobody overwrites K(0) (request of K(4) in phase B(0)) before everyone has read it
emit("s_nop 15")
emit("s_nop 15")
for s_ in max_list(0):
    emit(s_)
emit(f"s_cbranch_vccz {label('t0nofix')}")
call_fix(0)
put_label("t0nofix")
for s_ in exp_units(0, 0, args.eb, cvt=False):
    emit(s_)
for a_ in KA:
    emit(vadd_imm(a_, STAGE))                               # -> tile 1
for f in range(8):
    emit(read_k(f, f))
for j in range(4):
    emit(vadd_imm(KA[j], STAGE))                            # KA[0..3] -> tile 2
emit(f"s_sub_u32 s{REM}, %[nt], 1")                         # NT - 1 full iterations, then the tail
# ---- the loop: four copies (stage constants), S buffers alternate ----
slow = []
for c in range(4):
    X, Y = c & 1, (c + 1) & 1
    put_label(f"copy{c}")
    emit(f"s_cmp_eq_u32 s{REM}, 0")
    emit(f"s_cbranch_scc1 {label(f'tail{X}')}")
    emit(f"s_sub_u32 s{REM}, s{REM}, 1")
    if args.stamps:
        emit(f"s_add_u32 s{STMP}, s{STMP}, 1")
    stamp(0)
    fa = [] if "e" in args.abl else exp_units(X, args.eb, 32, lead_cvt=range(args.eb))
    emit_stream(*phase_a(c, fa))
    stamp(1)
    cont = Cont()
    fb = ([] if "m" in args.abl else max_list(Y) + ["BRANCH"]) + ([] if "e" in args.abl else exp_units(Y, 0, args.eb, cvt=False))
    # the first fillers read S'(Y), whose last MFMA closed phase A: with first_gap = 1 and the row-max chains in the order the QK chains
    # finished, every accumulator is read at least two MFMAs (>= 64 cycles) behind its last write
    emit_stream(*phase_b(c, fb), cont=cont, branch_label=label(f"slow{c}"), first_gap=args.first_gap)
    stamp(2)
    if c == 3:
        emit(f"s_branch {label('copy0')}")
    if cont.items is not None:      # (None: a timing-ablation build without the row-max check has no rare path)
        slow.append((c, Y, cont.items))
# ---- tails: the last tile (finish its softmax, PV), X = its S buffer ----
for X in range(2):
    put_label(f"tail{X}")
    emit("s_waitcnt lgkmcnt(0)")
    for s_ in exp_units(X, args.eb, 32, lead_cvt=range(args.eb)):
        emit(s_)
    emit("s_nop 1")
    plain_pv()
    emit(f"s_branch {label('epilogue')}")
# ---- slow paths: the rest of phase B without fillers, the fix-up, the exps phase B would have done ----
for c, Y, items in slow:
    put_label(f"slow{c}")
    for s_ in items:
        emit(s_)
    call_fix(Y)
    for s_ in exp_units(Y, 0, args.eb, cvt=False):
        emit(s_)
    emit(f"s_branch {label(f'copy{(c + 1) & 3}')}")
# ---- epilogue ----
put_label("epilogue")
emit("s_nop 15")
emit("s_nop 15")
# t0 = li, t1 = hi, t2/t3 = row of half 0 / 1 (unclamped)
emit(f"v_and_b32 v{t[0]}, 31, %[tid]")
emit(f"v_bfe_u32 v{t[1]}, %[tid], 5, 1")
emit(f"s_lshl_b32 s{ST[0]}, s{WAVE}, 6")
emit(f"v_add_u32 v{t[2]}, s{ST[0]}, v{t[0]}")
emit(f"v_add_u32 v{t[3]}, 32, v{t[2]}")
for h in range(2):
    lt, inv, tmp = t[4 + h], t[6 + h], t[8]
    emit(f"v_add_f32 v{lt}, v{L(h, 0)}, v{L(h, 1)}")
    emit(f"v_mov_b32 v{tmp}, v{lt}")
    emit("s_nop 1")
    emit(f"v_permlane32_swap_b32 v{tmp}, v{lt}")
    emit("s_nop 1")
    emit(f"v_add_f32 v{lt}, v{tmp}, v{lt}")                 # l of the row (both lane halves)
    emit(f"v_rcp_f32 v{inv}, v{lt}")
    emit(f"v_cmp_lt_f32 vcc, 0, v{lt}")
    emit("s_nop 0")
    emit(f"v_cndmask_b32 v{inv}, 0, v{inv}, vcc")
# lse (optional): R + log2(l) for the lanes hi == 0 of valid rows
emit(f"s_or_b32 s{ST[0]}, s{LP}, s{LP + 1}")
emit(f"s_cmp_eq_u32 s{ST[0]}, 0")
emit(f"s_cbranch_scc1 {label('nolse')}")
for h in range(2):
    emit(f"v_log_f32 v{t[8]}, v{t[4 + h]}")
    emit(f"v_lshlrev_b32 v{t[9]}, 2, v{t[2 + h]}")
    emit(f"v_cmp_gt_u32 vcc, s{NROWS}, v{t[2 + h]}")
    emit(f"v_cmp_eq_u32 s[{ST[2]}:{ST[3]}], 0, v{t[1]}")
    emit(f"s_and_b64 vcc, vcc, s[{ST[2]}:{ST[3]}]")
    emit(f"v_add_f32 v{t[8]}, v{t[8]}, v{R[h]}")
    emit(f"s_and_saveexec_b64 {sr(EXS)}, vcc")
    emit(f"global_store_dword v{t[9]}, v{t[8]}, {sr(LP)}")
    emit(f"s_mov_b64 exec, {sr(EXS)}")
put_label("nolse")
# O rows: out[row][dblk*32 + 8 rq + 4 hi + e] = O[h][dblk][4 rq + e] * inv, 8-byte stores
for h in range(2):
    emit(f"v_mul_lo_u32 v{t[8]}, v{t[2 + h]}, s{OLS}")
    emit(f"v_lshlrev_b32 v{t[9]}, 3, v{t[1]}")             # hi * 4 elements * 2 bytes
    emit(f"v_add_u32 v{t[8]}, v{t[8]}, v{t[9]}")
    emit(f"v_cmp_gt_u32 vcc, s{NROWS}, v{t[2 + h]}")
    emit(f"s_and_saveexec_b64 {sr(EXS)}, vcc")
    for d in range(4):
        for rq in range(4):
            for e in range(4):
                emit(f"v_accvgpr_read_b32 v{t[10 + e]}, a{O(h, d, rq * 4 + e)}")
            for e in range(4):
                emit(f"v_mul_f32 v{t[10 + e]}, v{t[10 + e]}, v{t[6 + h]}")
            emit(f"v_cvt_pk_bf16_f32 v{t[10]}, v{t[10]}, v{t[11]}")
            emit(f"v_cvt_pk_bf16_f32 v{t[11]}, v{t[12]}, v{t[13]}")
            emit(f"global_store_dwordx2 v{t[8]}, {vr(t[10], 2)}, {sr(OP)} offset:{(d * 32 + rq * 8) * 2}")
            emit("s_nop 0")
    emit(f"s_mov_b64 exec, {sr(EXS)}")
emit(f"s_branch {label('end')}")
# ---- out-of-line blocks: a tile iterator reached the end of its segment ----
for site, back, kind in list(_sites):
    emit(site + ":")
    call("kseg_next" if kind == "k" else "vseg_next")
    ptr, step = (KP, KSTEP) if kind == "k" else (VP, VSTEP)
    emit(f"s_sub_u32 s{ptr}, s{ptr}, s{step}")               # (the unconditional step behind `back` puts it back)
    emit(f"s_subb_u32 s{ptr + 1}, s{ptr + 1}, 0")
    emit(f"s_branch {back}")
# ---- ragged tile (subroutine, return address in RET2): a whole tile in plain order from the slot KA[] / VA[] point at, RAG valid keys ----
put_label("rag_tile")
plain_qk(1)
emit("s_nop 15")
emit("s_nop 15")
emit(f"v_bfe_u32 v{t[0]}, %[tid], 5, 1")
emit(f"v_lshlrev_b32 v{t[0]}, 3, v{t[0]}")                 # 8 * hi
emit(f"v_mov_b32 v{t[1]}, 0xff800000")
for sub in range(2):
    for r in range(16):
        kb_ = sub * 32 + 16 * (r >> 3) + (r & 7)            # key of this register for hi = 0
        emit(f"s_sub_i32 s{ST[0]}, s{RAG}, {kb_}")
        emit(f"v_cmp_gt_i32 vcc, s{ST[0]}, v{t[0]}")        # key < rag
        for h in range(2):
            emit(f"v_cndmask_b32 v{S(1, h, sub, r)}, v{t[1]}, v{S(1, h, sub, r)}, vcc")
for s_ in max_list(1):
    emit(s_)
emit(f"s_cbranch_vccz {label('ragnofix')}")
call_fix(1)
put_label("ragnofix")
for s_ in exp_units(1, 0, 32):
    emit(s_)
emit("s_nop 1")
plain_pv()
emit(f"s_setpc_b64 {sr(RET2)}")
# ---- segment routines (return address in RET): move a tile iterator to the next K / V^T segment that has a full tile, or park it ----
for kind in ("k", "v"):
    seg, cnt, ptr, step = (KSEG, KCNT, KP, KSTEP) if kind == "k" else (VSEG, VCNT, VP, VSTEP)
    o_ptr, o_bs = (KA_K, KA_KBS) if kind == "k" else (KA_VT, KA_VTBS)
    T_ = ST[2:14]
    put_label(f"{kind}seg_next")
    emit(f"s_add_u32 s{seg}, s{seg}, 1")
    emit(f"s_cmp_ge_u32 s{seg}, s{NSEG}")
    emit(f"s_cbranch_scc1 {label(kind + 'park')}")
    emit(f"s_lshl_b32 s{T_[0]}, s{seg}, 3")
    emit(f"s_add_u32 s{T_[0]}, s{KARG}, s{T_[0]}")
    emit(f"s_addc_u32 s{T_[1]}, s{KARG + 1}, 0")
    emit(f"s_load_dwordx2 {sr(T_[2])}, {sr(T_[0])}, 0x{o_ptr:x}")
    emit(f"s_load_dwordx2 {sr(T_[4])}, {sr(T_[0])}, 0x{o_bs:x}")
    emit(f"s_load_dwordx2 {sr(T_[6])}, {sr(T_[0])}, 0x{KA_LEN:x}")
    emit("s_waitcnt lgkmcnt(0)")                             # (drains the fragment reads in flight as well: harmless, the counted waits stay valid)
    emit(f"s_ashr_i32 s{cnt}, s{T_[6]}, 6")                  # full tiles of the segment (lengths below 2^31)
    emit(f"s_cmp_le_i32 s{cnt}, 0")
    emit(f"s_cbranch_scc1 {label(kind + 'seg_next')}")
    # byte offset of (batch b, head h) inside the segment: K: (b * k_bs + h * 128) * 2;  V^T: (b * vt_bs) * 2 + h * 128 * vls_bytes
    emit(f"s_mul_hi_u32 s{T_[9]}, s{BIDX}, s{T_[4]}")
    emit(f"s_mul_i32 s{T_[8]}, s{BIDX}, s{T_[4]}")
    emit(f"s_lshl_b64 {sr(T_[8])}, {sr(T_[8])}, 1")
    if kind == "k":
        emit(f"s_lshl_b32 s{T_[10]}, s{HIDX}, 8")
        emit(f"s_mov_b32 s{T_[11]}, 0")
    else:
        emit(f"s_lshl_b32 s{T_[10]}, s{HIDX}, 7")
        emit(f"s_mul_hi_u32 s{T_[11]}, s{T_[10]}, s{VLS}")
        emit(f"s_mul_i32 s{T_[10]}, s{T_[10]}, s{VLS}")
    emit(f"s_add_u32 s{T_[8]}, s{T_[8]}, s{T_[10]}")
    emit(f"s_addc_u32 s{T_[9]}, s{T_[9]}, s{T_[11]}")
    emit(f"s_add_u32 s{ptr}, s{T_[2]}, s{T_[8]}")
    emit(f"s_addc_u32 s{ptr + 1}, s{T_[3]}, s{T_[9]}")
    emit(f"s_setpc_b64 {sr(RET)}")
    put_label(kind + "park")                                 # no tile left: the requests that still follow re-fetch the last one
    emit(f"s_mov_b32 s{cnt}, 0x7fffffff")
    emit(f"s_mov_b32 s{step}, 0")
    emit(f"s_setpc_b64 {sr(RET)}")
# ---- fix-up subroutines (behind every call site): the reference maximum of some rows moves ----
for Y in range(2):
    put_label(f"fix{Y}")
    D, F = [t[0], t[1]], [t[2], t[3]]
    for h in range(2):
        emit(f"v_cmp_lt_f32 vcc, s{THR}, v{MX[h]}")
        emit("s_nop 1")
        emit(f"v_cndmask_b32 v{D[h]}, 0, v{MX[h]}, vcc")    # delta = this row's maximum moved ? its excess : 0
        emit(f"v_add_f32 v{R[h]}, v{R[h]}, v{D[h]}")
        emit(f"v_exp_f32_e64 v{F[h]}, -v{D[h]}")            # 2^-delta
    for h in range(2):
        for sub in range(2):
            for r in range(16):
                emit(f"v_sub_f32 v{S(Y, h, sub, r)}, v{S(Y, h, sub, r)}, v{D[h]}")
        for r in range(16):
            emit(f"v_xor_b32 v{NM(h, r)}, 0x80000000, v{R[h]}")
        for j in range(2):
            emit(f"v_mul_f32 v{L(h, j)}, v{L(h, j)}, v{F[h]}")
    emit("s_nop 15")
    emit("s_nop 15")                                         # the last PV MFMAs have written O
    for h in range(2):
        for r0 in range(0, 64, 8):
            for e in range(8):
                emit(f"v_accvgpr_read_b32 v{t[4 + e]}, a{h * 64 + r0 + e}")
            for e in range(8):
                emit(f"v_mul_f32 v{t[4 + e]}, v{t[4 + e]}, v{F[h]}")
            for e in range(8):
                emit(f"v_accvgpr_write_b32 a{h * 64 + r0 + e}, v{t[4 + e]}")
    emit(f"s_mov_b32 s{THR}, 0x41000000")                    # 8.0
    emit("s_nop 7")
    emit("s_nop 7")
    emit(f"s_setpc_b64 {sr(RET)}")
put_label("end")

# ---------------- write ----------------
n_mfma = sum(1 for s_ in out if s_.startswith("v_mfma"))
with open(args.o, "w") as fh:
    fh.write("// GENERATED by tools/gen_attn_q64.py %s — do not edit; %d instructions, %d MFMAs\n" %
             (" ".join(a for a in sys.argv[1:] if not a.startswith("more4d") and a != "-o"), sum(1 for s_ in out if not s_.endswith(":") and not s_.startswith(";")), n_mfma))
    for s_ in out:
        if s_.startswith(";"):
            continue
        fh.write('"%s\\n\\t"\n' % s_)
print(f"{args.o}: {len(out)} lines, {n_mfma} MFMAs", file=sys.stderr)
