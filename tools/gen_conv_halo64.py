#!/usr/bin/env python3
"""Generator of the main loop of conv_halo64_kernel (more4d_amd/csrc/conv_halo64.h): the 3 x 3 x 3 convolution of the VAE's 96-channel
tiles as ONE WAVE PER SIMD.

    python tools/gen_conv_halo64.py [-o more4d_amd/csrc/conv_halo64_gen.inc]

Why: conv_halo_kernel<3, 3, 12, 32, 3, 3> (conv_halo.h; half of the VAE's kernel time) feeds MT x NT = 3 x 3 MFMAs from 3 + 3 fragment
reads — 0.67 ds_read_b128 per MFMA, i.e. the CU's LDS pipe is 2/3 busy when the matrix pipe is full, and with two workgroups of
256-register waves per CU there is no room for a bigger register tile (PMC: MFMA busy 0.51).  Here a wave owns the SIMD's whole
512-register file: MT = 5 pixel tiles (32 pixels = one image row each) x NT = 3 channel tiles = 15 accumulators in 240 AGPRs, 8 reads
per 15 MFMAs (0.53); a workgroup is TWO waves (a 10 x 32 patch), two workgroups per CU = one wave on every SIMD, and each workgroup's
prologue / epilogue is covered by the other one (the C++ epilogue of conv_halo.h is shared: same arithmetic, same bits).
The halo is not loaded per 16-channel chunk (3 frames at once, a barrier-fenced reload per chunk) but as SLABS = one frame of one
chunk (12 x 40 halo pixels x 32 B = 15 KiB), streamed through a ring of three, requested six steps ahead; weight groups (one (dt, dh)
row of three taps, 9 KiB) through a ring of three, landed one step early, so that the first fragments of step s + 1 are read in the
shadow of step s's last MFMAs and the MFMA stream never waits for LDS:
    step s = (chunk c, dt, dh):   s_waitcnt vmcnt / s_barrier / request weights(s + 2) [, slab(j + 2) when dh == 0, j = 3c + dt]
                                  3 taps x 15 MFMAs; the 5 + 3 fragments of the next tap are read behind the first MFMAs of a tap
One loop iteration = one chunk = 9 steps (all LDS offsets are immediates).  Accumulation order = conv_halo_kernel's (chunk, dt, dh, dw):
bit-identical results.  Requests beyond the last chunk read past the weight / input extents (buffer range check: zeros) and are never
consumed.  Lane tables (fragment addresses, DMA lane offsets) come from the C++ wrapper through LDS.
"""
import argparse
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--abl", default="", help="timing ablations (results wrong): w = no weight requests, s = no slab requests, r = no fragment reads, "
                "p = no pixel-fragment reads, q = no weight-fragment reads")
ap.add_argument("--gather", action="store_true", help="weights in the plain [Cout][27][Cin] order (the first version: every 16-byte piece of a request from another row); "
                "default: the tiled order of m4d_conv_pack_weights, every request one contiguous KiB")
ap.add_argument("--window", action="store_true", help="pixel fragments as a sliding window of image rows per (dt, dw): the five rows of tap (dh, dw) are rows dh .. dh + 4 of "
                "seven, every row fragment is read ONCE per frame (21 reads per 9 taps instead of 45); same MFMA order, same bits")
ap.add_argument("--ahead", type=int, default=1, help="fragments are read this many taps ahead of their MFMAs (1: two register sets; 2: three sets — built and measured: bit-identical, same time, the reads cost LDS throughput, not latency)")
ap.add_argument("--shape", default="3x5x3", help="KT x MT x NT: 3x5x3 = the 3x3x3 conv, 10 x 32 patches, 96-channel tiles (conv_halo64_kernel); "
                "1x4x4 = the 3x3 conv, 8 x 32 patches, 128-channel tiles (conv_halo64k1_kernel: the adaptors' convs)")
ap.add_argument("-o", default=None)
args = ap.parse_args()

KT, MT, NT = (int(v) for v in args.shape.split("x"))
assert (KT, MT, NT) in ((3, 5, 3), (1, 4, 4))
if args.o is None:
    args.o = "more4d_amd/csrc/conv_halo64_gen.inc" if KT == 3 else "more4d_amd/csrc/conv_halo64k1_gen.inc"
NG = 3 * KT                                      # (dt, dh) groups of three taps per chunk
WTAP = NT * 1024
WG_BYTES = 3 * WTAP
SLAB = 16384 if MT == 5 else 14336               # one frame of one chunk: (2 MT + 2) halo rows x 40 pixels x 32 B, in whole KiB pieces per wave
NSP, NWP = SLAB // 2048, (WG_BYTES // 1024 + 1) // 2      # DMA pieces per wave: slab / weight group
CKSTEP = 32 if args.gather else 9 * KT * 1024    # weight soffset step per 16-channel chunk
assert not (args.gather or args.window or args.ahead != 1) or KT == 3
W_OFF, S_OFF = 0, 3 * WG_BYTES                 # weight ring, slab ring (bytes from the workgroup's LDS base)
PITCHB = 40 * 32                               # bytes of one halo row

# VGPRs
AB = [[4 + mi * 3 + dw for dw in range(3)] for mi in range(MT)]       # pixel-fragment addresses (slab ring base included)
WF = 19                                                               # weight-fragment address (row li of a tap's tile)
HO = [20 + i for i in range(8)]                                       # slab DMA lane offsets (piece i of this wave)
WO = [28 + i for i in range(6)]                                       # weight DMA lane offsets
TMP = [34, 35]


NBUF = args.ahead + 1              # fragment register sets


def FA(buf, mi):
    return 64 + (buf * MT + mi) * 4


def FW(buf, ni):
    return 64 + NBUF * MT * 4 + (buf * NT + ni) * 4


def ACC(mi, ni):
    return (mi * NT + ni) * 16


# SGPRs (s48..s101 clobbered)
RX, RW = 48, 52                  # buffer resources of x / w
WBW, WBS = 56, 57                # m0 bases of this wave's DMA pieces: weights / slabs
CHO, CKB = 58, 59                # soffset of the current chunk's frame-0 slab; 2 * ck0 of the current chunk (weights)
FRB, CHB, GS = 60, 61, 62        # bytes per input frame; slab soffset step per chunk; weight soffset step per group (3 taps)
NCH, LDS0, WAVE = 63, 64, 65
ST = [66, 67, 68, 69]


def vr(a, n):
    return f"v[{a}:{a + n - 1}]"


def ar(a, n):
    return f"a[{a}:{a + n - 1}]"


def sr(a, n=4):
    return f"s[{a}:{a + n - 1}]"


def label(name):
    return f".Lcv64_{name}_%="


out = []


def emit(s):
    if isinstance(s, (list, tuple)):
        for x in s:
            emit(x)
    else:
        out.append(s)


def frag_reads(buf, slab_slot, dh, w_slot, dw):
    """the 5 pixel + 3 weight fragments of tap (dh, dw) of the step that reads slab slot / weight slot"""
    ins = []
    for mi in range(MT):
        if "p" not in args.abl:
            ins.append(f"ds_read_b128 {vr(FA(buf, mi), 4)}, v{AB[mi][dw]} offset:{slab_slot * SLAB + dh * PITCHB}")
    for ni in range(NT):
        if "q" not in args.abl:
            ins.append(f"ds_read_b128 {vr(FW(buf, ni), 4)}, v{WF} offset:{w_slot * WG_BYTES + dw * WTAP + ni * 1024}")
    return ins


def tap_mfmas(buf):
    return [f"v_mfma_f32_32x32x16_bf16 {ar(ACC(mi, ni), 16)}, {vr(FW(buf, ni), 4)}, {vr(FA(buf, mi), 4)}, {ar(ACC(mi, ni), 16)}"
            for mi in range(MT) for ni in range(NT)]


def dma_w(slot, soff_reg):
    """this wave's pieces of a weight group (pieces 2 i + wave; nine pieces: wave 1's fifth repeats piece 8: same bytes, same place) as a
    list of instruction GROUPS (one per MFMA gap)"""
    # (piece 8 of nine is the group's last KiB: both waves write it to the same place — wave-independent base)
    odd = (WG_BYTES // 1024) % 2 == 1
    return [[(f"s_add_u32 m0, s{LDS0}, 0x{slot * WG_BYTES + WG_BYTES - 1024:x}" if odd and i == NWP - 1 else f"s_add_u32 m0, s{WBW}, 0x{slot * WG_BYTES + i * 2048:x}"), "s_nop 0",
             f"buffer_load_dwordx4 v{WO[i]}, {sr(RW)}, s{soff_reg} offen lds"] for i in range(NWP)]


def dma_slab(slot, soff_reg):
    return [[f"s_add_u32 m0, s{WBS}, 0x{slot * SLAB + i * 2048:x}", "s_nop 0",
             f"buffer_load_dwordx4 v{HO[i]}, {sr(RX)}, s{soff_reg} offen lds"] for i in range(NSP)]


def w_soff(g, chunk_ahead, dst):
    """soffset of weight group g of the chunk `chunk_ahead` chunks after the current one: g * GS + CKB + 32 * chunk_ahead"""
    if not args.gather:      # tiled: unit (row block, chunk, tap) = 1 KiB, taps of a chunk consecutive
        return [f"s_add_u32 s{dst}, s{CKB}, 0x{g * 3 * 1024 + chunk_ahead * CKSTEP:x}"]
    ins = [f"s_mul_i32 s{dst}, s{GS}, {g}", f"s_add_u32 s{dst}, s{dst}, s{CKB}"]
    if chunk_ahead:
        ins.append(f"s_add_u32 s{dst}, s{dst}, {CKSTEP * chunk_ahead}")
    return ins


def s_soff(dt, chunk_ahead, dst):
    ins = [f"s_mul_i32 s{dst}, s{FRB}, {dt}", f"s_add_u32 s{dst}, s{dst}, s{CHO}"]
    for _ in range(chunk_ahead):
        ins.append(f"s_add_u32 s{dst}, s{dst}, s{CHB}")
    return ins


# =====================================================================================================================
for dst, src in ((RX, "rx0"), (RX + 1, "rx1"), (RX + 2, "rx2"), (RX + 3, "rx3"), (RW, "rw0"), (RW + 1, "rw1"), (RW + 2, "rw2"), (RW + 3, "rw3"),
                 (CHO, "cho"), (FRB, "frb"), (CHB, "chb"), (GS, "gs"), (NCH, "nch"), (LDS0, "lds0")):
    emit(f"s_mov_b32 s{dst}, %[{src}]")
emit(f"s_mov_b32 s{CKB}, 0")
emit(f"v_bfe_u32 v{TMP[0]}, %[tid], 6, 1")
emit("s_nop 3")
emit(f"v_readfirstlane_b32 s{WAVE}, v{TMP[0]}")
emit("s_nop 3")
emit(f"s_lshl_b32 s{WBW}, s{WAVE}, 10")
emit(f"s_add_u32 s{WBW}, s{WBW}, s{LDS0}")
emit(f"s_add_u32 s{WBS}, s{WBW}, 0x{S_OFF:x}")
# lane table (LDS offset 0, 128 bytes per work item): AB[15] WF HO[8] WO[5] - - -
emit(f"v_and_b32 v{TMP[0]}, 0x7f, %[tid]")
emit(f"v_lshlrev_b32 v{TMP[0]}, 7, v{TMP[0]}")
emit(f"v_add_u32 v{TMP[0]}, s{LDS0}, v{TMP[0]}")
for i in range(8):
    emit(f"ds_read_b128 {vr(4 + 4 * i, 4)}, v{TMP[0]} offset:{16 * i}")       # v4..v35
emit("s_waitcnt lgkmcnt(0)")
emit("s_barrier")                                              # both waves have their tables: the weight ring may be overwritten
# first requests: slabs 0, 1 (frames 0, 1 of chunk 0), weight groups 0, 1
emit(s_soff(0, 0, ST[0]))
emit(dma_slab(0, ST[0]))
emit(w_soff(0, 0, ST[1]))
emit(dma_w(0, ST[1]))
emit(s_soff(1, 0, ST[0]) if KT == 3 else s_soff(0, 1, ST[0]))       # (KT = 1: a chunk has one slab — the second request is chunk 1's)
emit(dma_slab(1, ST[0]))
emit(w_soff(1, 0, ST[1]))
emit(dma_w(1, ST[1]))
emit(f"v_mov_b32 v{TMP[1]}, 0")
for r in range(MT * NT * 16):
    emit(f"v_accvgpr_write_b32 a{r}, v{TMP[1]}")
emit("s_waitcnt vmcnt(0)")
emit("s_barrier")
def tap_coords(T):
    """(slab slot, dh, weight slot, dw) of tap T of the loop body.  Slabs and weight groups go through rings of three: a slab serves nine taps
    (KT = 3: frame dt of the chunk; KT = 1: the chunk), a weight group three — the slots repeat every 27 taps"""
    T %= 27
    return (T // 9) % 3, (T // 3) % 3, (T // 3) % 3, T % 3


if args.window:
    # ---- sliding window: register slot (R % 5, dw) holds row R of the wave's seven halo rows (R = mi + dh) at column shift dw.  Row 5
    # takes over row 0's registers once tap (dh 0, dw) is done, row 6 row 1's; the next frame's rows 0..4 follow two taps after the
    # last use of what they replace.  Reads issued in tap t (relative to the frame's nine taps), each one tap or more before its
    # first use and one tap or more after the last use of the registers' previous content:
    WSCHED = {0: [(0, 2, 1), (0, 3, 1), (0, 4, 1), (0, 0, 2)], 1: [(0, 1, 2), (0, 2, 2), (0, 3, 2), (0, 4, 2)], 2: [(0, 5, 0), (0, 5, 1)],
              3: [(0, 5, 2)], 4: [(0, 6, 0)], 5: [(0, 6, 1)], 6: [(0, 6, 2)], 7: [(1, 0, 0), (1, 1, 0), (1, 2, 0)],
              8: [(1, 3, 0), (1, 4, 0), (1, 0, 1), (1, 1, 1)]}          # (frame ahead, row, dw)
    assert args.ahead == 1 and sum(len(v) for v in WSCHED.values()) == 21

    def PS(slot, dw):
        return 64 + (slot * 3 + dw) * 4

    def FWW(buf, ni):
        return 64 + 60 + (buf * NT + ni) * 4

    def row_read(dt, R, dw):
        return f"ds_read_b128 {vr(PS(R % 5, dw), 4)}, v{AB[0][dw]} offset:{(dt % 3) * SLAB + R * PITCHB}"

    def w_reads(buf, T):
        _, _, w_slot, dw = tap_coords(T)
        return [f"ds_read_b128 {vr(FWW(buf, ni), 4)}, v{WF} offset:{w_slot * WG_BYTES + dw * WTAP + ni * 1024}" for ni in range(NT)]

    for t in (7, 8):               # what the two taps in front of frame 0 would have read
        emit([row_read(0, R, dw) for (_, R, dw) in WSCHED[t]])
    emit(w_reads(0, 0))
    emit(label("chunk") + ":")
    tail_check = [f"s_add_u32 s{CKB}, s{CKB}, 0x{CKSTEP:x}", f"s_add_u32 s{CHO}, s{CHO}, s{CHB}", f"s_sub_u32 s{NCH}, s{NCH}, 1", f"s_cmp_eq_u32 s{NCH}, 0"]
    for u in range(2):
        for dt in range(3):
            for dh in range(3):
                g = dt * 3 + dh
                emit([f"s_waitcnt vmcnt({8 if dh == 1 and 's' not in args.abl else 0})", "s_barrier"])
                g2, c2 = (g + 2) % 9, (g + 2) // 9
                groups = [w_soff(g2, c2, ST[1]) + dma_w((g + 2) % 3, ST[1])[0]] + dma_w((g + 2) % 3, ST[1])[1:]
                if dh == 0:
                    j2 = dt + 2
                    sl = dma_slab(j2 % 3, ST[0])
                    groups += [s_soff(j2 % 3, j2 // 3, ST[0]) + sl[0]] + sl[1:]
                for dw in range(3):
                    T = u * 27 + g * 3 + dw
                    buf = T % 2
                    mf = [f"v_mfma_f32_32x32x16_bf16 {ar(ACC(mi, ni), 16)}, {vr(FWW(buf, ni), 4)}, {vr(PS((mi + dh) % 5, dw), 4)}, {ar(ACC(mi, ni), 16)}"
                          for mi in range(MT) for ni in range(NT)]
                    nxt = w_reads(buf ^ 1, g * 3 + dw + 1) + [row_read(dt + ahead_f, R, dw_) for (ahead_f, R, dw_) in WSCHED[dh * 3 + dw]]
                    assert len(nxt) <= 8
                    emit("s_waitcnt lgkmcnt(0)")
                    for i, m in enumerate(mf):
                        emit(m)
                        if 1 <= i <= 8 and i - 1 < len(nxt):
                            emit(nxt[i - 1])
                        elif i >= 9 and groups:
                            emit(groups.pop(0))
                assert not groups
        emit(tail_check)
        emit(f"s_cbranch_scc1 {label('done')}" if u == 0 else f"s_cbranch_scc0 {label('chunk')}")
else:
    TPC = 9 * KT                       # taps per chunk
    # chunks per loop body: the taps must line up with the register sets, and (KT = 1: one slab per chunk, ring of three) with the slab slots
    UNROLL = {(3, 2): 2, (3, 3): 1, (1, 2): 6}[(KT, NBUF)]
    RD = MT + NT                       # fragment reads per tap: behind MFMAs 1 .. RD, DMA pieces behind the MFMAs after them
    for T in range(args.ahead):        # taps 0 .. ahead - 1 of chunk 0
        emit(frag_reads(T % NBUF, *tap_coords(T)))
    emit(label("chunk") + ":")
    tail_check = [f"s_add_u32 s{CKB}, s{CKB}, 0x{CKSTEP:x}", f"s_add_u32 s{CHO}, s{CHO}, s{CHB}", f"s_sub_u32 s{NCH}, s{NCH}, 1", f"s_cmp_eq_u32 s{NCH}, 0"]
    for u in range(UNROLL):
        for dt in range(KT):
            for dh in range(3):
                g = dt * 3 + dh
                # ---- step boundary: weight group g + 1 (and the slab of step g + 1) have landed; weight slot (g + 2) % 3 and, when dh == 0,
                # the slab slot two slabs on are free.  In flight and allowed to stay: the slab requested in the previous step (NSP pieces,
                # issued behind that step's weight pieces).  (The very first boundary has nothing to wait for: same code, the counters are zero.)
                emit([f"s_waitcnt vmcnt({NSP if dh == 1 and 's' not in args.abl else 0})", "s_barrier"])
                g2, c2 = (g + 2) % NG, (g + 2) // NG
                groups = [w_soff(g2, c2, ST[1]) + dma_w((g + 2) % 3, ST[1])[0]] + dma_w((g + 2) % 3, ST[1])[1:]
                if "w" in args.abl:
                    groups = []
                if dh == 0 and "s" not in args.abl:
                    if KT == 3:      # slab j = 3 c + dt: frame dt + 2 (of this or the next chunk) into slot (dt + 2) % 3
                        j2 = dt + 2
                        sl = dma_slab(j2 % 3, ST[0])
                        groups += [s_soff(j2 % 3, j2 // 3, ST[0]) + sl[0]] + sl[1:]
                    else:            # slab j = c: the slab of chunk c + 2 into slot (c + 2) % 3 (the loop body starts at a chunk c = 0 mod 3)
                        sl = dma_slab((u + 2) % 3, ST[0])
                        groups += [s_soff(0, 2, ST[0]) + sl[0]] + sl[1:]
                for dw in range(3):
                    T = u * TPC + g * 3 + dw                 # tap counter over the loop body
                    buf = T % NBUF
                    mf = tap_mfmas(buf)
                    # the fragments of tap T + ahead go into the register set tap T - 1 has just released
                    nxt = frag_reads((T + args.ahead) % NBUF, *tap_coords(T + args.ahead))
                    # the reads of the taps in between (issued behind the previous taps' MFMAs, LDS returns in order) may stay in flight
                    emit(f"s_waitcnt lgkmcnt({0 if 'r' in args.abl else min(15, (args.ahead - 1) * RD)})")
                    for i, m in enumerate(mf):
                        emit(m)
                        if 1 <= i <= RD:
                            if "r" not in args.abl and i - 1 < len(nxt):
                                emit(nxt[i - 1])                 # one fragment read behind each of MFMAs 1..RD
                        elif i > RD and groups:
                            emit(groups.pop(0))              # one DMA piece (m0, nop, request) behind each of the last MFMAs: weights first
                assert not groups
        emit(tail_check)
        emit(f"s_cbranch_scc1 {label('done')}" if u < UNROLL - 1 else f"s_cbranch_scc0 {label('chunk')}")
emit(label("done") + ":")
# drain: the last prefetched fragments and the over-requested DMA pieces must not land in the epilogue's staging blocks
emit("s_waitcnt lgkmcnt(0)")
emit("s_waitcnt vmcnt(0)")
emit("s_nop 15")
emit("s_nop 15")
emit("s_barrier")

n_mfma = sum(1 for s_ in out if s_.startswith("v_mfma"))
with open(args.o, "w") as fh:
    fh.write("// GENERATED by tools/gen_conv_halo64.py %s — do not edit; %d instructions, %d MFMAs\n" %
             (" ".join(a for a in sys.argv[1:] if not a.startswith("more4d") and a != "-o"),
              sum(1 for s_ in out if not s_.endswith(":")), n_mfma))
    for s_ in out:
        fh.write('"%s\\n\\t"\n' % s_)
print(f"{args.o}: {len(out)} lines, {n_mfma} MFMAs", file=sys.stderr)
