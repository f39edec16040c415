// attn128p_kernel: the production forward kernel (attention.hip: attn128_kernel<8>) re-scheduled so that the softmax
// arithmetic of one wave group hides under the MFMAs of the other.
//
// Measured on attn128_kernel<8> (profiles/r01e): MFMA busy 46 % of SIMD cycles although the kernel does little else —
// per 64-key tile a wave issues 32 MFMAs (1024 matrix-pipe cycles) and ~190 VALU instructions of which 32 are
// quarter-rate v_exp_f32 (~1100 VALU cycles), and because all 8 waves pass the per-tile barrier together, the two
// waves of a SIMD are in their MFMA phase at the same time and in their softmax phase at the same time: the pipes take
// turns instead of overlapping.  Here every wave alternates
//     V(i):  softmax of tile i (VALU only)                      | barrier
//     M(i):  O^T += V^T(i) P^T(i)  and  S^T(i+1) = K(i+1) Q^T    | barrier      (32 MFMAs back to back, raised priority)
// and the wave group {4..7} runs exactly one barrier behind {0..3}; each SIMD hosts one wave of each group, so its
// matrix pipe always has an M phase to run while the other wave is in its V phase.
//
// K / V^T tiles go through FOUR 32 KiB LDS stages by global->LDS DMA: tile i+3 is requested at the start of M(i) into
// the stage tile i-1 left (its last reader, PV(i-1) of the late group, finished before the barrier that opened this
// phase) and is waited for at the end of M(i+1), one barrier before QK(i+3) may start in either group.
// The ragged last tile of the key range is peeled off and processed FIRST in lock-step (softmax is order-free), so
// the pipelined loop only sees full DMA tiles.  Single K/V segment only (the multi-segment T-sharded call keeps
// attn128_kernel<8>).
#pragma once
#include <type_traits>
#ifndef M4D_ATTN_MSTREAM
#define M4D_ATTN_MSTREAM 0      // side builds (tools/side_lib.sh): 1 = M stream without fragment reads, 2 = without MFMAs
#endif
#ifndef M4D_ATTN_STAMPS
#define M4D_ATTN_STAMPS 0      // the stamp code costs ~8 % even when it is switched off at run time: its own side build
#endif
#ifndef M4D_ATTN_ONE_BARRIER
#define M4D_ATTN_ONE_BARRIER 1
#endif

// ---- M-phase instruction stream: steps 0..15 = PV (c = J/4 key group, d = J%4 head-dim block), steps 16..31 = QK of the
// next tile (kk = (J-16)/2, sub = (J-16)%2).  A ring of 8 fragment registers, step J consumes ring[J % 8] and re-fills
// it with the fragment of step J + 8: eight 1 KiB LDS reads are always in flight per wave.
template <int OFF> M4D_DEV void dsr128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> M4D_DEV void lgkm_le() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// (An inline-asm form of these three with O accumulating in AGPRs and the Q fragments in AGPRs was measured in round 3: 12 % slower —
// hipcc splits a 256-register budget 128 / 128 as soon as a kernel uses AGPRs and the softmax runs short of arch registers — and the
// MFMA issue rate does not depend on the accumulator file, tools/probes/mfma_rate.hip.  Removed.)
M4D_DEV void mma_o(const bf16x8& a, const bf16x8& b, f32x16& c) { mma32(a, b, c); }
M4D_DEV void mma_s(const bf16x8& a, const bf16x8& q, f32x16& c) { mma32(a, q, c); }
M4D_DEV void mma_s0(const bf16x8& a, const bf16x8& q, f32x16& c) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q, zero, 0, 0, 0);      // C = 0 as an inline constant, no 32 v_mov to clear the accumulators
}
// RD = ring depth (fragment reads in flight per wave) in the steady part of the stream.
// Where the time of an M phase went (tools/attn_clock.py phase stamps, round 3): the 32 MFMAs themselves issue every 33-34 cycles
// (tools/probes/mfma_rate.hip: 32.5-33.1 in every combination of accumulator file, chain length, neighbour wave), but the phase
// took ~1 500 cycles because everything else sat in FRONT of the first MFMA or BEHIND the last one: the K / V^T tile request
// (scalar address arithmetic, 4 DMA instructions), four more fragment prefetches, 12 address updates.  The stream now starts with
// the first MFMA right after the barrier; steps 0..3 issue the four missing prefetches next to their regular read, and a `hook`
// called once per step places the tile request (steps 3..7) and the address updates (steps 8..11, 10..24) in MFMA shadows.
constexpr int RD = 8;      // (12 and 14 were measured with the round-2 structure: slower)
template <int J> struct MStep {
    static constexpr bool pv = J < 16;
    static constexpr int idx = J < 16 ? J : J - 16;
};
template <int J> M4D_DEV void m_read(bf16x8 (&ring)[RD], const unsigned (&va)[4], const unsigned (&ka)[8]) {
    constexpr int I = MStep<J>::idx;
    if constexpr (MStep<J>::pv) dsr128<(I & 3) * 4096>(ring[J % RD], va[I >> 2]);
    else dsr128<(I & 1) * 8192>(ring[J % RD], ka[I >> 1]);
}
// issue order of the fragment reads: r0..r3 in the softmax phase, then step j issues r(j+4) (j < 4) and r(j+RD)
constexpr int m_issued_by_step(int j, int LAST) { return (j < 4 && j + 4 < LAST ? 1 : 0) + (j + RD < LAST ? 1 : 0); }
constexpr int m_issued_before(int J, int LAST) { int n = 4; for (int j = 0; j < J; ++j) n += m_issued_by_step(j, LAST); return n; }
constexpr int m_pos(int r, int LAST) {      // position of read r in the issue order
    if (r < 4) return r;
    if (r < 8) return m_issued_before(r - 4, LAST);                                          // first read of step r-4
    return m_issued_before(r - RD, LAST) + ((r - RD) < 4 && (r - RD) + 4 < LAST ? 1 : 0);     // after that step's late prefetch
}
// MODE (tool builds): 0 = the stream, 1 = MFMAs on whatever the ring holds (no fragment reads), 2 = fragment reads only
template <int J, int LAST, int MODE = 0, typename Hook>
M4D_DEV void m_steps(bf16x8 (&ring)[RD], const unsigned (&va)[4], const unsigned (&ka)[8], const bf16x8 (&pf)[4], const bf16x8 (&qf)[8],
                     f32x16 (&o)[4], f32x16 (&s)[2], Hook&& hook) {
    static_assert(RD == 8, "the late prefetch of steps 0..3 fills ring slots 4..7");
    if constexpr (J < LAST) {
        constexpr int I = MStep<J>::idx;
        if constexpr (MODE != 1) lgkm_le<m_issued_before(J, LAST) - m_pos(J, LAST) - 1>();      // fragment J has arrived (LDS returns in order)
        if constexpr (MODE == 2) asm volatile("" :: "v"(ring[J % RD]));
        else if constexpr (MStep<J>::pv) mma_o(ring[J % RD], pf[I >> 2], o[I & 3]);
        else if constexpr (I < 2) mma_s0(ring[J % RD], qf[0], s[I]);       // first k-step of S: C = 0
        else mma_s(ring[J % RD], qf[I >> 1], s[I & 1]);
        if constexpr (MODE != 1) {
            if constexpr (J < 4 && J + 4 < LAST) m_read<J + 4>(ring, va, ka);
            if constexpr (J + RD < LAST) m_read<J + RD>(ring, va, ka);
        }
        hook(std::integral_constant<int, J>{});
        __builtin_amdgcn_sched_barrier(0);
        m_steps<J + 1, LAST, MODE>(ring, va, ka, pf, qf, o, s, hook);
    }
}
template <int J, int END> M4D_DEV void m_prefetch(bf16x8 (&ring)[RD], const unsigned (&va)[4], const unsigned (&ka)[8]) {
    if constexpr (J < END) { m_read<J>(ring, va, ka); m_prefetch<J + 1, END>(ring, va, ka); }
}

// SMX: softmax arithmetic, 0 = packed fp32 (v_pk_fma_f32 / v_pk_add_f32), 1 = scalar v_fma_f32 / v_add_f32 (default)
// PRIO: 0 = no s_setprio, 1 = raised priority around every MFMA phase (default), 2 = static: the younger wave group at priority 1
template <int SMX, int PRIO>
__global__ __launch_bounds__(512, 2) void attn128p_kernel(AttnArgs p) {
    typedef bf16_t T;
    constexpr int D = 128, KVB = 64, STAGE = 32768, VOFF = 16384, QB = 256, NST = 4;
    extern __shared__ __attribute__((aligned(16))) char psmem[];   // NST * STAGE

    unsigned long long ts0 = 0, rt0 = 0;
    if (M4D_ABL(p) & 64) { ts0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const int HB = p.heads * p.B;
    int qt, hb;
    if ((HB & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        hb = xcd * (HB >> 3) + idx / p.nq_tiles;
        qt = idx % p.nq_tiles;
    } else {
        hb = blockIdx.x / p.nq_tiles;
        qt = blockIdx.x % p.nq_tiles;
    }
    const int b = hb / p.heads, h = hb % p.heads;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hi = lane >> 5;
    const int grp = wave >> 2;
    const int64_t qrow = (int64_t)qt * QB + wave * 32 + li;
    const bool qvalid = qrow < p.Lq;

    bf16x8 qf[8];
    {
        const T* qp = (const T*)p.q + b * p.q_bs + qrow * p.q_ls + (int64_t)h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (qvalid) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 16);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[kk][j] = (T)0.f;
            }
        }
    }
    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // absolute per-lane fragment addresses: ka = K side of the tile whose S is computed next, va = V^T side of the tile
    // whose P is applied next; both start in stage 0 and step one stage (mod 4) per tile
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)psmem;
    unsigned ka[8], va[4];
    {
        const int kr = perm23(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] = lds0 + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) va[c] = lds0 + VOFF + li * 128 + (((c * 2 + hi) ^ ((li >> 1) & 7)) << 4);
    }
    const int k_r = lane >> 4, k_lc0 = lane & 15;
    const int v_r = lane >> 3, v_pc = lane & 7;

    // K/V may arrive as several segments (one per rank of the T-sharded loop) that share their strides (host-checked).
    // Full 64-key tiles of all segments form one pipelined tile list; each segment's ragged tail is peeled first.
    const int64_t kls = p.kv.k_ls[0], vls = p.kv.vt_ls[0];
    int NT = 0;                                 // full tiles of all segments, pipelined
    for (int sg = 0; sg < p.kv.nseg; ++sg) NT += p.kv.len[sg] > 0 ? (int)(p.kv.len[sg] / KVB) : 0;
    auto seg_k = [&](int sg) { return (const T*)p.kv.k[sg] + b * p.kv.k_bs[sg] + (int64_t)h * D; };
    auto seg_v = [&](int sg) { return (const T*)p.kv.vt[sg] + b * p.kv.vt_bs[sg] + (int64_t)h * D * p.kv.vt_ls[sg]; };
    // DMA iterator over the tile list: (segment, first key); wave-uniform
    int dseg = 0;
    int64_t dk0 = 0;
    while (dseg < p.kv.nseg && p.kv.len[dseg] < KVB) ++dseg;
    // base pointers / length of the iterator's CURRENT segment live in scalars and are re-read only when it moves on
    // (indexing the kernel-argument arrays per tile costs scalar memory loads on the critical path of every iteration)
    const T* dkb = dseg < p.kv.nseg ? seg_k(dseg) : nullptr;
    const T* dvb = dseg < p.kv.nseg ? seg_v(dseg) : nullptr;
    int64_t dlen = dseg < p.kv.nseg ? p.kv.len[dseg] : 0;

    // DMA sources as 32-bit per-lane byte offsets from a wave-uniform base (sgpr_base + vgpr_offset addressing)
    unsigned offk[2], offv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int blk = wave * 2 + i;
        const int krow = blk * 4 + k_r, vrow = blk * 8 + v_r;
        offk[i] = (unsigned)((krow * kls + (k_lc0 ^ (krow & 15)) * 8) * 2);
        offv[i] = (unsigned)((vrow * vls + (v_pc ^ ((vrow >> 1) & 7)) * 8) * 2);
    }
    auto uniform_ptr = [](const char* q) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi2 << 32) | lo);
    };
    // request of the NEXT tile of the list, in two parts so that the pipelined loop can place them in MFMA shadows: the scalar part
    // (source pointers, iterator advance), then the 4 DMA instructions one at a time
    const char* dma_kp = nullptr;
    const char* dma_vp = nullptr;
    unsigned dma_dst = 0;
    auto dma_prepare = [&](int stage) {
        dma_kp = uniform_ptr((const char*)(dkb + dk0 * kls));
        dma_vp = uniform_ptr((const char*)(dvb + dk0));
        dk0 += KVB;
        if (dk0 + KVB > dlen) {
            dk0 = 0;
            ++dseg;
            while (dseg < p.kv.nseg && p.kv.len[dseg] < KVB) ++dseg;
            if (dseg < p.kv.nseg) { dkb = seg_k(dseg); dvb = seg_v(dseg); dlen = p.kv.len[dseg]; }
        }
        dma_dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + wave * 2048);
    };
    auto dma_issue = [&](int n) {          // n = 0..3 (a literal at every call site): K rows, V^T rows, K rows, V^T rows
        const unsigned dst = dma_dst;
        const char* const kp = dma_kp;
        const char* const vp = dma_vp;
        const unsigned ok0 = offk[0], ok1 = offk[1], ov0 = offv[0], ov1 = offv[1];
        if (n == 0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(ok0), "s"(kp) : "memory", "m0");
        else if (n == 1) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + VOFF), "v"(ov0), "s"(vp) : "memory", "m0");
        else if (n == 2) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + 1024), "v"(ok1), "s"(kp) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + VOFF + 1024), "v"(ov1), "s"(vp) : "memory", "m0");
    };
    auto dma_tile = [&](int stage, int64_t /*unused*/) {
        dma_prepare(stage);
        dma_issue(0); dma_issue(1); dma_issue(2); dma_issue(3);
    };

#define M4D_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define M4D_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    bf16x8 ring[RD];
    f32x16 s[2];
    bf16x8 pf[4];

    // S^T = K Q^T of the tile in stage `st` (16 MFMAs, K fragments two kk ahead)
#define M4D_QK(B, KK, SUB, OFF, W) do { M4D_LGKM(W); mma_s(B, qf[KK], s[SUB]); if ((KK) + 2 < 8) M4D_DSR(B, ka[((KK) + 2) & 7], OFF); } while (0)
#define M4D_QK0(B, SUB, OFF, W) do { M4D_LGKM(W); mma_s0(B, qf[0], s[SUB]); M4D_DSR(B, ka[2], OFF); } while (0)
#define M4D_QK_TILE(FIRSTWAIT)                                                                                        \
    do {                                                                                                             \
        M4D_QK0(ring[0], 0, 0, 3); M4D_QK0(ring[1], 1, 8192, 3); M4D_QK(ring[2], 1, 0, 0, 3); M4D_QK(ring[3], 1, 1, 8192, 3);    \
        M4D_QK(ring[0], 2, 0, 0, 3); M4D_QK(ring[1], 2, 1, 8192, 3); M4D_QK(ring[2], 3, 0, 0, 3); M4D_QK(ring[3], 3, 1, 8192, 3);    \
        M4D_QK(ring[0], 4, 0, 0, 3); M4D_QK(ring[1], 4, 1, 8192, 3); M4D_QK(ring[2], 5, 0, 0, 3); M4D_QK(ring[3], 5, 1, 8192, 3);    \
        M4D_QK(ring[0], 6, 0, 0, 3); M4D_QK(ring[1], 6, 1, 8192, 2); M4D_QK(ring[2], 7, 0, 0, 1); M4D_QK(ring[3], 7, 1, 8192, 0);    \
    } while (0)
#define M4D_QK_PREFETCH() do { M4D_DSR(ring[0], ka[0], 0); M4D_DSR(ring[1], ka[0], 8192); M4D_DSR(ring[2], ka[1], 0); M4D_DSR(ring[3], ka[1], 8192); } while (0)
    // O^T += V^T P^T (16 MFMAs, V^T fragments four steps ahead); the last four steps start the K prefetch of the next QK
#define M4D_PV(B, C, DD, OFF, W) do { M4D_LGKM(W); mma_o(B, pf[C], o[DD]); if ((C) + 1 < 4) M4D_DSR(B, va[((C) + 1) & 3], OFF); } while (0)
#define M4D_PV_PREFETCH() do { M4D_DSR(ring[0], va[0], 0); M4D_DSR(ring[1], va[0], 4096); M4D_DSR(ring[2], va[0], 8192); M4D_DSR(ring[3], va[0], 12288); } while (0)
#define M4D_PV_TILE()                                                                                                 \
    do {                                                                                                             \
        M4D_PV(ring[0], 0, 0, 0, 3); M4D_PV(ring[1], 0, 1, 4096, 3); M4D_PV(ring[2], 0, 2, 8192, 3); M4D_PV(ring[3], 0, 3, 12288, 3);  \
        M4D_PV(ring[0], 1, 0, 0, 3); M4D_PV(ring[1], 1, 1, 4096, 3); M4D_PV(ring[2], 1, 2, 8192, 3); M4D_PV(ring[3], 1, 3, 12288, 3);  \
        M4D_PV(ring[0], 2, 0, 0, 3); M4D_PV(ring[1], 2, 1, 4096, 3); M4D_PV(ring[2], 2, 2, 8192, 3); M4D_PV(ring[3], 2, 3, 12288, 3);  \
        M4D_PV(ring[0], 3, 0, 0, 3); M4D_PV(ring[1], 3, 1, 4096, 2); M4D_PV(ring[2], 3, 2, 8192, 1); M4D_PV(ring[3], 3, 3, 12288, 0);  \
    } while (0)
    // online softmax of s (exp2 domain) -> pf; `k_lim` masks keys >= k_lim of the tile (only the peeled ragged tile)
    auto softmax = [&](int k_lim) {
        if (k_lim < KVB) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7) >= k_lim) s[sub][r] = -INFINITY;
        }
        // row max: 16 three-operand max instead of 32 canonicalising v_max_f32 pairs
        float mx;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]));
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[0][r]), "v"(s[0][r + 1]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[0][15]), "v"(s[1][0]));
#pragma unroll
        for (int r = 1; r + 1 < 16; r += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[1][r]), "v"(s[1][r + 1]));
        mx = fmaxf(mx, s[1][15]);
        {
            const unsigned u = __float_as_uint(mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // lazy reference maximum: it only moves when the tile maximum exceeds it by more than 2^8 (probabilities stay <= 256,
        // harmless in fp32 / bf16; O / l and the LSE do not depend on the reference) -> the O rescale almost never runs
        const float cand = mx * p.sc;
        const float m_new = cand > m_run + 8.f ? cand : m_run;
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        // p = exp2(s * sc - m): the multiply-subtract and the row sum run two elements per instruction (v_pk_fma_f32 /
        // v_pk_add_f32); only the 32 v_exp_f32 are scalar
        if constexpr (SMX == 0) {
            const f32x2 sc2 = {p.sc, p.sc}, nm2 = {-m_run, -m_run};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 x = {s[sub][r], s[sub][r + 1]};
                    x = __builtin_elementwise_fma(x, sc2, nm2);
                    x[0] = __builtin_amdgcn_exp2f(x[0]);
                    x[1] = __builtin_amdgcn_exp2f(x[1]);
                    s[sub][r] = x[0];
                    s[sub][r + 1] = x[1];
                    ps2 += x;
                }
            l_run += ps2[0] + ps2[1];
        } else {
            // single-issue forms pinned by inline asm (hipcc SLP-packs adjacent scalar f32 ops into v_pk_* otherwise).  The
            // whole stream is volatile asm in a fixed order: hipcc's hazard recogniser does not see through inline asm, and
            // a VALU that consumes a transcendental's result needs one other instruction in between (gfx940+ trans-use
            // hazard) — here every v_add_f32 trails its v_exp_f32 by four instructions.
            const float nm = -m_run;
            float pa = 0.f, pb = 0.f;
#define M4D_SM_A(SUB, R) do { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s[SUB][R]) : "v"(s[SUB][R]), "s"(p.sc), "v"(nm)); \
                              asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s[SUB][(R) + 1]) : "v"(s[SUB][(R) + 1]), "s"(p.sc), "v"(nm)); } while (0)
#define M4D_SM_B(SUB, R) do { asm volatile("v_exp_f32 %0, %1" : "=v"(s[SUB][R]) : "v"(s[SUB][R])); \
                              asm volatile("v_exp_f32 %0, %1" : "=v"(s[SUB][(R) + 1]) : "v"(s[SUB][(R) + 1])); } while (0)
#define M4D_SM_C(SUB, R) do { asm volatile("v_add_f32 %0, %1, %2" : "=v"(pa) : "v"(pa), "v"(s[SUB][R])); \
                              asm volatile("v_add_f32 %0, %1, %2" : "=v"(pb) : "v"(pb), "v"(s[SUB][(R) + 1])); } while (0)
            M4D_SM_A(0, 0); M4D_SM_B(0, 0);
#pragma unroll
            for (int i = 1; i < 16; ++i) {          // pair i = (sub i >> 3, r = 2 * (i & 7))
                M4D_SM_A(i >> 3, 2 * (i & 7));
                M4D_SM_B(i >> 3, 2 * (i & 7));
                M4D_SM_C((i - 1) >> 3, 2 * ((i - 1) & 7));
            }
            asm volatile("s_nop 1");
            M4D_SM_C(1, 14);
#undef M4D_SM_A
#undef M4D_SM_B
#undef M4D_SM_C
            l_run += pa + pb;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) pf[c] = pack8<T>(s[c >> 1], (c & 1) * 8);
    };

    // ---- peeled ragged tail: lock-step, stage 0, zero-filled through registers ----
    for (int sg = 0; sg < p.kv.nseg; ++sg) {
        const int64_t len = p.kv.len[sg];
        if (len <= 0) continue;
        const int64_t tail0 = (len / KVB) * KVB;
        if (tail0 >= len) continue;
        const T* kbase = seg_k(sg);
        const T* vbase = seg_v(sg);
        char* base = psmem;
        // all four 16-byte pieces of a thread are requested unconditionally from clamped, in-bounds addresses and masked afterwards
        // (predicated — `key < len ? load : 0` — each load sits in a basic block of its own and is waited for alone: four serial
        // round trips in front of every workgroup's first MFMA).  K rows beyond the segment re-read its last row; V^T chunks beyond it
        // re-read the first chunk of the tail; a partial chunk's padding columns exist (strides are multiples of 8) and are zeroed here.
        const int rem = (int)(len - tail0);                      // 1 .. 63 valid keys
        uint4 kraw[2], vraw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + 512 * i;
            const int row = c >> 4, ch = c & 15;
            kraw[i] = *reinterpret_cast<const uint4*>(kbase + (tail0 + (row < rem ? row : rem - 1)) * kls + ch * 8);
            const int vrow = c >> 3, vch = c & 7;
            vraw[i] = *reinterpret_cast<const uint4*>(vbase + vrow * vls + tail0 + (vch * 8 < rem ? vch * 8 : 0));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + 512 * i;
            {
                const int row = c >> 4, ch = c & 15;
                *reinterpret_cast<uint4*>(base + swz_off<256>(row, ch)) = row < rem ? kraw[i] : make_uint4(0, 0, 0, 0);
            }
            {
                const int row = c >> 3, ch = c & 7;
                union { uint4 u; unsigned short e[8]; } tmp;
                tmp.u = vraw[i];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (ch * 8 + j >= rem) tmp.e[j] = 0;
                *reinterpret_cast<uint4*>(base + VOFF + swz_off<128>(row, ch)) = tmp.u;
            }
        }
        __syncthreads();
        M4D_QK_PREFETCH();
        M4D_QK_TILE(0);
        M4D_PV_PREFETCH();
        softmax((int)(len - tail0));
        __builtin_amdgcn_sched_barrier(0);
        M4D_PV_TILE();
        __syncthreads();
    }

    if (NT > 0) {
        // ---- pipeline prologue: tiles 0..2 requested, S(0) computed in lock-step, then the groups split ----
        dma_tile(0, 0);
        if (NT > 1) dma_tile(1, KVB);
        if (NT > 2) dma_tile(2, 2 * KVB);
        if (NT > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // tiles 0 and 1 landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        M4D_QK_PREFETCH();
        M4D_QK_TILE(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] += STAGE;                    // K side now points at tile 1 (stage 1)
        if constexpr (PRIO == 2) { if (grp == 1) __builtin_amdgcn_s_setprio(1); }
        // side build -DM4D_ATTN_STAMPS=1 (tools/side_lib.sh) with abl & 128: phase stamps of waves 0 and 4 of workgroup 1000, tiles 100..107, kept in LDS behind the stages
        int i = 0;
#if M4D_ATTN_STAMPS
        const bool stamp_on = (M4D_ABL(p) & 128) && blockIdx.x == 1000 && __builtin_amdgcn_readfirstlane(wave & 3) == 0;
#else
        constexpr bool stamp_on = false;
#endif
        char* const stamp_base = psmem + NST * STAGE + __builtin_amdgcn_readfirstlane(wave >> 2) * 512;
#define stamp(SLOT)                                                                                                   \
    do {                                                                                                             \
        if (stamp_on && i >= 100 && i < 108) {                                                                       \
            unsigned long long tc_;                                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc_) :: "memory");                            \
            *reinterpret_cast<volatile unsigned long long*>(stamp_base + ((i - 100) * 8 + (SLOT)) * 8) = tc_;         \
        }                                                                                                            \
    } while (0)
        // V(t): softmax of S(t) -> P(t); the first four fragments of the following PV are requested in front of it
#define M4D_V_BODY()                                                                                                  \
    do {                                                                                                             \
        stamp(0);                                                                                                    \
        m_prefetch<0, 4>(ring, va, ka);      /* V^T landed long ago; (only four: the softmax needs the registers) */   \
        if (!(M4D_ABL(p) & 1)) softmax(KVB);                                                                          \
        /* pin the whole softmax (exp2, row sums, bf16 packing) here: without these uses the compiler sinks the      \
           32 v_exp_f32 into the MFMA stream this schedule exists to keep clean */                                   \
        asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]));                                          \
        asm volatile("" : "+v"(l_run), "+v"(m_run));                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        stamp(1);                                                                                                    \
    } while (0)
        // M(i): one stream of 32 MFMAs (PV(i) then QK(i+1)); in its shadows: the request of tile i+3 (its stage held tile i-1, dead
        // since the barrier that opened this interval) and the advance of the fragment addresses one stage (mod 4) as soon as the last
        // read through them has been issued (va -> tile i+1 after step 7, ka[kk] -> tile i+2 after step 9 + 2 kk)
#define M4D_M_BODY()                                                                                                  \
    do {                                                                                                             \
        stamp(2);                                                                                                    \
        const bool do_dma = i + 3 < NT;                                                                              \
        const unsigned dv = ((i + 1) & 3) ? (unsigned)STAGE : (unsigned)(-3 * STAGE);                                \
        const unsigned dk = ((i + 2) & 3) ? (unsigned)STAGE : (unsigned)(-3 * STAGE);                                \
        auto hook = [&](auto JJ) {                                                                                   \
            constexpr int J = decltype(JJ)::value;                                                                   \
            if constexpr (J == 3) { if (do_dma) dma_prepare((i + 3) & 3); }                                          \
            if constexpr (J >= 4 && J < 8) { if (do_dma) dma_issue(J - 4); }                                         \
            if constexpr (J >= 8 && J < 12) va[J - 8] += dv;                                                         \
            if constexpr (J >= 10 && J <= 24 && (J & 1) == 0) ka[(J - 10) >> 1] += dk;                               \
        };                                                                                                           \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                      \
        if (!(M4D_ABL(p) & 2)) m_steps<0, 32, M4D_ATTN_MSTREAM>(ring, va, ka, pf, qf, o, s, hook);                    \
        else {                                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
            if (do_dma) dma_tile((i + 3) & 3, 0);                                                                    \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) va[c] += dv;                                               \
            _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) ka[kk] += dk;                                           \
        }                                                                                                            \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                      \
        stamp(3);                                                                                                    \
    } while (0)
        // end of interval i: tile i+2 must have landed before anyone reads it in interval i+1; tile i+3 (just requested) may stay in flight
#define M4D_END_INTERVAL()                                                                                            \
    do {                                                                                                             \
        if (i + 3 < NT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                             \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        stamp(4);                                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                \
    } while (0)
#define M4D_LAST_PV()                                                                                                 \
    do {                                                                                                             \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                      \
        m_steps<0, 16>(ring, va, ka, pf, qf, o, s, [](auto) {});                                                     \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#if M4D_ATTN_ONE_BARRIER
        // ONE barrier per tile.  In interval i both groups read V^T(i) and K(i+1) and nothing else; the early group runs
        // V(i), M(i) and the late group M(i), V(i+1) (its V(0) is done here, in front of the loop), so on every SIMD one wave is in its
        // softmax while the other streams MFMAs without a barrier between the halves: an interval costs V + M instead of
        // 2 max(V, M) + two hand-overs (phase stamps, round 3: V 1 120, M 1 300, waits 700 cycles per tile and group).
        if (grp == 0) {
            for (; i + 1 < NT; ++i) { M4D_V_BODY(); M4D_M_BODY(); M4D_END_INTERVAL(); }
            M4D_V_BODY();
            M4D_LAST_PV();
        } else {
            M4D_V_BODY();
            for (; i + 1 < NT; ++i) { M4D_M_BODY(); M4D_V_BODY(); M4D_END_INTERVAL(); }
            M4D_LAST_PV();
        }
#else
        // TWO barriers per tile: the late group runs exactly one barrier behind the early one
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (; i + 1 < NT; ++i) {
            M4D_V_BODY();
            __builtin_amdgcn_s_barrier();
            M4D_M_BODY();
            M4D_END_INTERVAL();
        }
        M4D_V_BODY();
        __builtin_amdgcn_s_barrier();
        M4D_LAST_PV();
        __builtin_amdgcn_s_barrier();
        if (grp == 0) __builtin_amdgcn_s_barrier();                        // balance the barrier count
#endif
#undef M4D_V_BODY
#undef M4D_M_BODY
#undef M4D_END_INTERVAL
#undef M4D_LAST_PV
#undef stamp
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    }
#undef M4D_PV_TILE
#undef M4D_PV_PREFETCH
#undef M4D_PV
#undef M4D_QK_PREFETCH
#undef M4D_QK_TILE
#undef M4D_QK
#undef M4D_QK0
#undef M4D_LGKM
#undef M4D_DSR

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (p.lse && qvalid && hi == 0) p.lse[((int64_t)b * p.heads + h) * p.Lq + qrow] = m_run + log2f(l_tot);
    if (qvalid) {
        T* op = (T*)p.out + b * p.o_bs + qrow * p.o_ls + (int64_t)h * D + hi * 4;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][rq * 4 + e] * inv;
                T* dst = op + d * 32 + rq * 8;
                if (p.accumulate) {
                    f32x4 prev = load4(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]) + prev[e];
                }
                store4(dst, v);
            }
    }
    if (M4D_ATTN_STAMPS && (M4D_ABL(p) & 128) && p.dbg && blockIdx.x == 1000) {      // phase stamps: [group][tile][slot] -> dbg + 8 Mi words
        __syncthreads();
        if (t < 128) p.dbg[(1 << 19) + t] = reinterpret_cast<unsigned long long*>(psmem + NST * STAGE)[t];
    }
    if ((M4D_ABL(p) & 64) && p.dbg && t == 0) {     // tool build: shader cycles and 100 MHz wall clock of this workgroup's lifetime
        unsigned long long* d = p.dbg + (size_t)blockIdx.x * 4;
        d[0] = ts0; d[1] = __builtin_readcyclecounter(); d[2] = rt0; d[3] = __builtin_amdgcn_s_memrealtime();
    }
}
