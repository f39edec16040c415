// attn128x_kernel: attention over SHORT key lists (WanI2VCrossAttention: 512 text + 257 image keys against 21 840 queries,
// reference wan_transformer4d.py:533-552) as ONE persistent pipeline per CU.
//
// Why a kernel of its own: attn128p_kernel (attention_phased.h) at 8 key tiles spends 11 of its 24 us per workgroup outside the
// tile loop (workgroup launch, Q fetch from HBM, three-tile pipeline fill, lock-step S(0), drain, epilogue) and with 128 KiB
// of LDS per workgroup nothing else is resident on the CU to cover it; attn128_kernel<4> (two 4-wave workgroups per CU) covers it
// but has no V / M alternation (0.215 of the MFMA peak).  Here a workgroup (8 waves x 32 queries, the phased kernel's schedule:
// V(g) = softmax of tile g on the VALU, M(g) = PV(g) + QK(g+1) as one stream of 32 MFMAs, wave groups half an interval apart, one
// barrier per 64-key tile, four 32 KiB stages, tile g+3 requested in M(g)) walks a LIST OF (query tile, key tile) pairs:
//   * the K / V^T tile stream runs on from one query tile ("item") into the next: no fill, no drain, no lock-step S(0);
//   * the next item's Q fragments are fetched into the registers of the old ones as soon as the item's last QK has been issued
//     (they land under the softmax and the PV half of the stream);
//   * ragged key tiles (257 = 4 x 64 + 1) stay IN the pipeline: the DMA cannot mask, so rows / chunks beyond the segment are redirected
//     to in-bounds addresses per lane, the partial and the redirected V^T chunks are zeroed in LDS by the lanes that requested them
//     (after their vmcnt wait, before the barrier that publishes the tile), and the scores of the missing keys are masked in V(g);
//   * a softmax group that ends (kv.new_softmax: text | image; or the item) is flushed = normalised and stored, or added to what the
//     previous group stored (round_T(round_T(o) + out), the reference's x + img_x in bf16): transposed through 4 KiB of wave-private LDS
//     and written as full 128-byte lines.  The early wave group does it at the head of the next interval (its stores are older than that
//     interval's tile request); the late group's closing s_waitcnt leaves its stores (issued behind the tile request) in flight.
// Work split: items in the XCD-aware order of the other kernels ((b, h) groups pinned to an XCD, consecutive query tiles of one
// (b, h) on one workgroup so its K / V^T stay in that L2), a contiguous run of items per workgroup, gridDim = number of CUs.
#pragma once
#include "attention_phased.h"
#ifndef XP_VAR
#define XP_VAR 0      // side builds (tools/side_lib.sh): timing variants of the loop, results wrong — 1 no ragged mask in V, 2 no repair check,
#endif                // 4 no flush / fetch tests in the loop, 8 interval-closing wait without the switch, 16 tile iterator without bookkeeping

template <int PRIO>
__global__ __launch_bounds__(512, 2) void attn128x_kernel(AttnArgs p) {
    typedef bf16_t T;
    constexpr int D = 128, KVB = 64, STAGE = 32768, VOFF = 16384, QB = 256;
    constexpr int F_GROUP_END = 1, F_NEXT_ITEM = 2, F_ADD = 4;
    extern __shared__ __attribute__((aligned(16))) char xsmem[];   // 4 * STAGE + 8 waves x 4 KiB of epilogue staging

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hi = lane >> 5;
    const int grp = wave >> 2;

    // The loop is short of SGPRs (the phased kernel's 75 + an iterator over items / segments / tiles): whatever only the rare paths need
    // (segment change, Q fetch, flush) is NOT kept in registers.  Kernel arguments of those paths are read through an opaque copy of the
    // kernarg pointer (scalar loads where they are used, not ~25 SGPRs that live through the loop), per-lane constants are re-derived from
    // an opaque copy of the lane id (hoisted by the compiler they cost ~30 VGPRs).  With everything live hipcc parked loop state in VGPR
    // lanes and put v_readlane / v_writelane pairs into the MFMA stream: 20 % per interval (tools/xp_vs_phased.py, side builds -DXP_VAR).
    typedef const AttnArgs __attribute__((address_space(4))) * kernarg_t;
    auto args = [&]() { kernarg_t a = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(a)); return a; };
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };

    // ---- this workgroup's items ----
    // Items = (b * heads + h, query tile), dealt round-robin: workgroup slot s of an XCD takes items s, s + nslots, ... of that XCD's list, so
    // the workgroups of an XCD are always within one or two (b, h) of each other and their K / V^T (394 KB per (b, h)) stay in the 4 MiB L2.
    auto item_step = [&](kernarg_t a) { return ((a->heads * a->B) & 7) == 0 ? (int)(gridDim.x >> 3) : (int)gridDim.x; };
    int c_hb, c_qt, G;
    {
        const int nq = p.nq_tiles, HB = p.heads * p.B;
        int e0, E, hb0, istep;
        if ((HB & 7) == 0) {
            e0 = blockIdx.x >> 3; istep = gridDim.x >> 3;
            E = (HB >> 3) * nq;
            hb0 = (blockIdx.x & 7) * (HB >> 3);
        } else {
            e0 = (int)blockIdx.x; istep = (int)gridDim.x;
            E = HB * nq;
            hb0 = 0;
        }
        if (e0 >= E) return;
        c_hb = hb0 + e0 / nq; c_qt = e0 % nq;
        int NTI = 0;                                 // key tiles per item
        for (int sg = 0; sg < p.kv.nseg; ++sg) NTI += p.kv.len[sg] > 0 ? (int)((p.kv.len[sg] + KVB - 1) / KVB) : 0;
        G = ((E - e0 + istep - 1) / istep) * NTI;    // intervals of this workgroup
    }
    auto advance = [&](kernarg_t a, int& hb, int& qt) {
        const int nq = a->nq_tiles;
        qt += item_step(a);
        while (qt >= nq) { qt -= nq; ++hb; }
    };

    // ---- consumer side: Q fragments of the item whose tiles are being computed ----
    bf16x8 qf[8];
    // eight 16-byte pieces of the lane's query row, requested WITHOUT a wait (inline asm: a compiler-visible load into registers that live
    // across the loop makes hipcc guard their first use, the first QK MFMA of every M stream, with s_waitcnt vmcnt(0) — behind the tile
    // request issued twelve MFMAs earlier).  The wait is placed by hand: M stream, step 15, of the interval whose QK half is the first to use
    // them.  Rows beyond Lq re-read the last row (their outputs are never stored).
    auto load_q = [&](int hb, int qt) {
        kernarg_t a = args();
        const int heads = a->heads, b = hb / heads, h = hb - b * heads;
        const int l = opaque_lane();
        const int64_t Lq = a->Lq, row = (int64_t)qt * QB + wave * 32 + (l & 31);
        const int64_t r = row < Lq ? row : Lq - 1;
        const T* qp = (const T*)a->q + b * a->q_bs + r * a->q_ls + (int64_t)h * D + (l >> 5) * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[0]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(qf[1]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(qf[2]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:96" : "=v"(qf[3]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(qf[4]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:160" : "=v"(qf[5]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:192" : "=v"(qf[6]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:224" : "=v"(qf[7]) : "v"(qp) : "memory");
    };
    load_q(c_hb, c_qt);

    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)xsmem;
    unsigned ka[8], va[4];
    {
        const int kr = perm23(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] = lds0 + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) va[c] = lds0 + VOFF + li * 128 + (((c * 2 + hi) ^ ((li >> 1) & 7)) << 4);
    }

    // ---- DMA side: the NEXT tile to request (three tiles ahead of the consumer) as running source pointers + keys left in its segment ----
    const char* kp = nullptr;                     // K rows of the next tile: (b, first key, h, 0)
    const char* vp = nullptr;                     // V^T columns of the next tile: (b, h, d = 0, first key)
    int drem = 0;                                 // keys of the current segment from the next tile's first key on
    unsigned dkls2 = 0, dvls2 = 0;                // row strides of the current segment in bytes (segments need not share them)
    int dsegx = 0;                                // current segment | 256 if the softmax group in progress is ADDED to the output
    int d_hb = c_hb, d_qt = c_qt, d_left = G;     // (d_left: tiles still to request)
    auto enter_segment = [&](kernarg_t a, int sg) {
        const int heads = a->heads, b = d_hb / heads, h = d_hb - b * heads;
        const int64_t vls = a->kv.vt_ls[sg];
        kp = (const char*)((const T*)a->kv.k[sg] + b * a->kv.k_bs[sg] + (int64_t)h * D);
        vp = (const char*)((const T*)a->kv.vt[sg] + b * a->kv.vt_bs[sg] + (int64_t)h * D * vls);
        drem = (int)a->kv.len[sg];
        dkls2 = (unsigned)(a->kv.k_ls[sg] * 2); dvls2 = (unsigned)(vls * 2);
        dsegx = (dsegx & 256) | sg;
    };
    auto first_seg = [&](kernarg_t a) { int sg = 0; while (sg < a->kv.nseg && a->kv.len[sg] <= 0) ++sg; return sg; };
    {
        kernarg_t a = args();
        dsegx = a->accumulate != 0 ? 256 : 0;
        enter_segment(a, first_seg(a));
    }
    const unsigned dst_w = __builtin_amdgcn_readfirstlane(lds0 + wave * 2048);      // this wave's 2 KiB of every K / V^T image
    // the tile just requested is behind us: returns its code = valid keys | flags << 8 and moves on to the next tile of the list
    auto next_tile = [&]() -> int {
        const int lim = drem < KVB ? drem : KVB;
        int flags = 0;
        kp += (size_t)KVB * dkls2; vp += KVB * 2; drem -= KVB; --d_left;
        if (drem <= 0 && !(XP_VAR & 16)) {
            kernarg_t a = args();
            int sg = (dsegx & 255) + 1;
            while (sg < a->kv.nseg && a->kv.len[sg] <= 0) ++sg;
            if (sg < a->kv.nseg) {
                if ((a->kv.new_softmax >> sg) & 1) { flags = F_GROUP_END | (dsegx & 256 ? F_ADD : 0); dsegx |= 256; }
            } else {
                flags = F_GROUP_END | (dsegx & 256 ? F_ADD : 0);
                dsegx = a->accumulate != 0 ? 256 : 0;
                sg = first_seg(a);
                if (d_left > 0) { flags |= F_NEXT_ITEM; advance(a, d_hb, d_qt); }
            }
            if (d_left > 0) enter_segment(a, sg);
        } else if (drem <= 0) { drem += 8192; }
        return lim | (flags << 8);
    };
    // per-lane source offsets (K piece = 4 rows x 256 B, V^T piece = 8 rows x 128 B, XOR swizzle on the source chunk).  Ragged tile: K rows
    // beyond the segment re-read its last row (their scores are masked), V^T chunks entirely beyond it re-read chunk 0 of their row (zeroed
    // in LDS afterwards, like the tail of the partial chunk).  Four per-lane constants: K row / swizzled chunk of this lane's first K piece,
    // V^T row / chunk of its first V^T piece; the second pieces are 4 / 8 rows further, which flips bit 2 of the swizzle term.
    const int dk_row = wave * 8 + (lane >> 4), dv_row = wave * 16 + (lane >> 3);
    const unsigned dk_c = (unsigned)(((lane & 15) ^ (dk_row & 15)) << 4);
    const int dv_lc = (lane & 7) ^ ((dv_row >> 1) & 7);
    auto dma_issue = [&](int n, int stage) {          // n = 0..3 (a literal at every call site): K rows, V^T rows, K rows, V^T rows
        const unsigned dst = dst_w + stage * STAGE;
        if ((n & 1) == 0) {
            int krow = dk_row + (n >> 1) * 4;
            const unsigned kc = dk_c ^ ((n >> 1) * 64);
            if (drem < KVB) krow = krow < drem ? krow : drem - 1;
            const unsigned ok = (unsigned)krow * dkls2 + kc;
            if (n == 0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(ok), "s"(kp) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + 1024), "v"(ok), "s"(kp) : "memory", "m0");
        } else {
            const int vrow = dv_row + (n >> 1) * 8;
            int lc = dv_lc ^ ((n >> 1) * 4);
            if (drem < KVB) lc = lc * 8 < drem ? lc : 0;
            const unsigned ov = (unsigned)vrow * dvls2 + (unsigned)(lc << 4);
            if (n == 1) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + VOFF), "v"(ov), "s"(vp) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + VOFF + 1024), "v"(ov), "s"(vp) : "memory", "m0");
        }
    };
    // zero what the V^T image of a ragged tile holds beyond its `lim` keys: every lane repairs the two 16-byte chunks it requested
    auto sanitize = [&](int stage, int lim) {
        const int l = opaque_lane();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = wave * 2 + i;
            const int vrow = blk * 8 + (l >> 3);
            const int lc = (l & 7) ^ ((vrow >> 1) & 7);
            const int nv = lim - lc * 8;                 // valid keys of this chunk
            if (nv < 8) {
                uint4* cp = reinterpret_cast<uint4*>(xsmem + stage * STAGE + VOFF + blk * 1024 + l * 16);
                union { uint4 u; unsigned short e[8]; } w;
                w.u = make_uint4(0, 0, 0, 0);
                if (nv > 0) {
                    w.u = *cp;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j >= nv) w.e[j] = 0;
                }
                *cp = w.u;
            }
        }
    };

#define M4D_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define M4D_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    bf16x8 ring[RD];
    f32x16 s[2];
    bf16x8 pf[4];
#define M4D_QK(B, KK, SUB, OFF, W) do { M4D_LGKM(W); mma_s(B, qf[KK], s[SUB]); if ((KK) + 2 < 8) M4D_DSR(B, ka[((KK) + 2) & 7], OFF); } while (0)
#define M4D_QK0(B, SUB, OFF, W) do { M4D_LGKM(W); mma_s0(B, qf[0], s[SUB]); M4D_DSR(B, ka[2], OFF); } while (0)
#define M4D_QK_TILE()                                                                                                 \
    do {                                                                                                             \
        M4D_QK0(ring[0], 0, 0, 3); M4D_QK0(ring[1], 1, 8192, 3); M4D_QK(ring[2], 1, 0, 0, 3); M4D_QK(ring[3], 1, 1, 8192, 3);    \
        M4D_QK(ring[0], 2, 0, 0, 3); M4D_QK(ring[1], 2, 1, 8192, 3); M4D_QK(ring[2], 3, 0, 0, 3); M4D_QK(ring[3], 3, 1, 8192, 3);    \
        M4D_QK(ring[0], 4, 0, 0, 3); M4D_QK(ring[1], 4, 1, 8192, 3); M4D_QK(ring[2], 5, 0, 0, 3); M4D_QK(ring[3], 5, 1, 8192, 3);    \
        M4D_QK(ring[0], 6, 0, 0, 3); M4D_QK(ring[1], 6, 1, 8192, 2); M4D_QK(ring[2], 7, 0, 0, 1); M4D_QK(ring[3], 7, 1, 8192, 0);    \
    } while (0)
#define M4D_QK_PREFETCH() do { M4D_DSR(ring[0], ka[0], 0); M4D_DSR(ring[1], ka[0], 8192); M4D_DSR(ring[2], ka[1], 0); M4D_DSR(ring[3], ka[1], 8192); } while (0)

    // online softmax of s (exp2 domain) -> pf; keys >= k_lim of the tile are masked (ragged tiles only).  Same arithmetic, in the same
    // order, as attn128p_kernel's scalar form.
    auto softmax = [&](int k_lim) {
        if (k_lim < KVB) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7) >= k_lim) s[sub][r] = -INFINITY;
        }
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
        for (int i = 1; i < 16; ++i) {
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
#pragma unroll
        for (int c = 0; c < 4; ++c) pf[c] = pack8<T>(s[c >> 1], (c & 1) * 8);
    };

    // close a softmax group: normalise o, store it (or add it to what the previous group stored), start the next group from zero; returns the
    // number of store instructions certainly issued.  F_NEXT_ITEM: the group was the item's last, the consumer moves on to the item whose Q
    // fragments are already in qf.
    auto flush = [&](int flags) -> int {
        int issued = 0;
        float l_tot;
        {
            const unsigned u = __float_as_uint(l_run);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = l_tot > 0.f ? __builtin_amdgcn_rcpf(l_tot) : 0.f;
        kernarg_t a = args();
        const int heads = a->heads, c_b = c_hb / heads, c_h = c_hb - c_b * heads;
        const int64_t Lq = a->Lq;
        const int l = opaque_lane();
        const int64_t row0 = (int64_t)c_qt * QB + __builtin_amdgcn_readfirstlane(wave) * 32;
        if (a->lse && row0 + (l & 31) < Lq && l < 32) a->lse[((int64_t)c_b * heads + c_h) * Lq + row0 + l] = m_run + log2f(l_tot);
        if (!(M4D_ABL(p) & 4)) {
            // The accumulators hold 4 consecutive d per (lane, register quad): stored from there, every instruction touches 32 rows with
            // 16 bytes each.  Each wave transposes its 32 x 128 tile through 4 KiB of LDS of its own, half of D at a time (row-major, 16-byte
            // chunk ^ ((row >> 1) & 7)), and writes rows: eight lanes cover the 128 contiguous bytes of a row, an instruction 8 full lines.
            char* const ep = xsmem + 4 * STAGE + wave * 4096;
            const int er = l >> 3, ej = l & 7, fli = l & 31, fhi = l >> 5;
            const int64_t o_ls = a->o_ls;
            T* const obase = (T*)a->out + (c_b * a->o_bs + (int64_t)c_h * D + row0 * o_ls);      // wave-uniform base + 32-bit lane offsets
            const unsigned ols = (unsigned)o_ls;
            const int nrows = Lq - row0 < 32 ? (int)(Lq - row0) : 32;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        bf16x4 w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = (T)(o[half * 2 + dd][rq * 4 + e] * inv);
                        *reinterpret_cast<bf16x4*>(ep + fli * 128 + (((dd * 4 + rq) ^ ((fli >> 1) & 7)) << 4) + fhi * 8) = w;
                    }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 v[4];
                unsigned go[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 8 + er;
                    v[it] = *reinterpret_cast<const bf16x8*>(ep + r * 128 + ((ej ^ ((r >> 1) & 7)) << 4));
                    go[it] = (unsigned)(r < nrows ? r : nrows - 1) * ols + (unsigned)(half * 64 + ej * 8);
                }
                if ((flags & F_ADD) && nrows > 0 && !(M4D_ABL(p) & 32)) {      // (a wave entirely beyond Lq has no rows to read)
                    bf16x8 prev[4];       // all four previous pieces requested before the first is used
#pragma unroll
                    for (int it = 0; it < 4; ++it) prev[it] = *reinterpret_cast<const bf16x8*>(obase + go[it]);
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[it][e] = (T)((float)v[it][e] + (float)prev[it][e]);
                }
                if (nrows >= 32 && !(M4D_ABL(p) & 32)) {          // (wave-uniform) all 32 rows exist: four unpredicated stores
#pragma unroll
                    for (int it = 0; it < 4; ++it) *reinterpret_cast<bf16x8*>(obase + go[it]) = v[it];
                    issued += 4;
                } else if (!(M4D_ABL(p) & 32)) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        if (it * 8 + er < nrows) *reinterpret_cast<bf16x8*>(obase + go[it]) = v[it];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        m_run = -INFINITY; l_run = 0.f;
        if (flags & F_NEXT_ITEM) advance(a, c_hb, c_qt);
        return issued;
    };
    // the item's last QK has been issued: fetch the next item's Q fragments into qf
    auto fetch_next_q = [&]() -> int {
        if (M4D_ABL(p) & 8) return 0;
        int n_hb = c_hb, n_qt = c_qt;
        advance(args(), n_hb, n_qt);
        load_q(n_hb, n_qt);
        return 8;
    };

    // codes of tiles g-1 (early group: its flush is still owed), g, g+1, g+2, g+3: 12 bits each in one scalar pair
    unsigned long long codes = (unsigned long long)KVB * 0x0001001001001000ull;
#define cprev ((int)(codes & 0xfff))
#define c0 ((int)((codes >> 12) & 0xfff))
#define c1 ((int)((codes >> 24) & 0xfff))
#define c2 ((int)((codes >> 36) & 0xfff))
#define SET_CODE(I, V) (codes = (codes & ~(0xfffull << (12 * ((I) + 1)))) | ((unsigned long long)(V) << (12 * ((I) + 1))))
    // ---- pipeline prologue: tiles 0..2 requested, S(0) computed in lock-step, then the groups split ----
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (i < G) { dma_issue(0, i); dma_issue(1, i); dma_issue(2, i); dma_issue(3, i); const int c = next_tile(); SET_CODE(i, c); }
    if (G > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // tiles 0 and 1 (and the Q fragments) landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((c0 & 255) < KVB) sanitize(0, c0 & 255);
    if (G > 1 && (c1 & 255) < KVB) sanitize(1, c1 & 255);
    __builtin_amdgcn_s_barrier();
    M4D_QK_PREFETCH();
    M4D_QK_TILE();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) ka[kk] += STAGE;                    // K side now points at tile 1 (stage 1)

    if constexpr (PRIO == 2) { if (grp == 1) __builtin_amdgcn_s_setprio(1); }      // static: the younger wave group at priority 1, no per-phase flips
    int g = 0;
    // V(t): softmax of S(t) -> P(t); the first four fragments of the following PV are requested in front of it
#define M4D_V_BODY(CODE)                                                                                              \
    do {                                                                                                             \
        m_prefetch<0, 4>(ring, va, ka);                                                                              \
        if (!(M4D_ABL(p) & 1)) softmax((XP_VAR & 1) ? KVB : ((CODE) & 255));                                         \
        asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]));                                          \
        asm volatile("" : "+v"(l_run), "+v"(m_run));                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
    // M(g): PV(g) then QK(g+1) as one stream of 32 MFMAs; in its shadows the request of tile g+3 (steps 4..7), the iterator's step to the
    // tile after it (step 8) and the fragment-address advances
#define M4D_M_BODY()                                                                                                  \
    do {                                                                                                             \
        const bool do_dma = g + 3 < G;                                                                               \
        const unsigned dv = ((g + 1) & 3) ? (unsigned)STAGE : (unsigned)(-3 * STAGE);                                \
        const unsigned dk = ((g + 2) & 3) ? (unsigned)STAGE : (unsigned)(-3 * STAGE);                                \
        auto hook = [&](auto JJ) {                                                                                   \
            constexpr int J = decltype(JJ)::value;                                                                   \
            if constexpr (J >= 4 && J < 8) { if (do_dma && !(M4D_ABL(p) & 16)) dma_issue(J - 4, (g + 3) & 3); }      \
            if constexpr (J == 8) { if (do_dma) { const int c = next_tile(); SET_CODE(3, c); } }                     \
            if constexpr (J >= 8 && J < 12) va[J - 8] += dv;                                                         \
            if constexpr (J >= 10 && J <= 24 && (J & 1) == 0) ka[(J - 10) >> 1] += dk;                               \
            if constexpr (J == 15) {      /* QK(g+1) opens the next item: its Q fragments (older than this interval's tile request) */ \
                if ((c0 >> 8) & F_NEXT_ITEM) {                                                                       \
                    if (do_dma) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                     \
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
                }                                                                                                    \
            }                                                                                                        \
        };                                                                                                           \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                      \
        if (!(M4D_ABL(p) & 2)) m_steps<0, 32, 0>(ring, va, ka, pf, qf, o, s, hook);                                  \
        else { if (do_dma) { dma_issue(0, (g + 3) & 3); dma_issue(1, (g + 3) & 3); dma_issue(2, (g + 3) & 3); dma_issue(3, (g + 3) & 3);   \
                             const int c = next_tile(); SET_CODE(3, c); }                                            \
               _Pragma("unroll") for (int c = 0; c < 4; ++c) va[c] += dv;                                            \
               _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) ka[kk] += dk; }                                      \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                      \
    } while (0)
    // end of interval g: tile g+2 must have landed (and, if ragged, been repaired) before anyone reads it in interval g+1.  vmcnt retires in
    // issue order (loads and stores alike on gfx9), so the wait may leave outstanding whatever was issued AFTER this interval's tile request:
    // its 4 DMA pieces + AFTER = the stores of a flush / the Q fragments of the next item that are known to have been issued behind it.
#define M4D_END_INTERVAL(AFTER)                                                                                       \
    do {                                                                                                             \
        switch ((XP_VAR & 8) ? (g + 3 < G ? 4 : 0) : (g + 3 < G ? 4 : 0) + (AFTER)) {                                \
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;                                          \
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;                                          \
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;                                          \
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;                                        \
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;                                        \
            case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;                                        \
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        }                                                                                                            \
        if (!(XP_VAR & 2) && g + 2 < G && (c2 & 255) < KVB) sanitize((g + 2) & 3, c2 & 255);                         \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
        codes >>= 12;                                                                                                \
    } while (0)
#define M4D_LAST_PV()                                                                                                 \
    do {                                                                                                             \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                      \
        m_steps<0, 16>(ring, va, ka, pf, qf, o, s, [](auto) {});                                                     \
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
    // The early group flushes tile g at the head of interval g+1 (its stores are then older than that interval's tile request), the late
    // group between M(g) and V(g+1).  (Both groups flushing in interval g — the early one behind its M(g) — was measured 2 % slower: the
    // flush's VALU work then runs beside the late group's softmax instead of beside its MFMA stream.)
    if (grp == 0) {
        // early group, interval g: [flush of tile g-1] [next item's Q if tile g is its item's last] V(g) M(g)
        for (; g + 1 < G; ++g) {
            if (!(XP_VAR & 4)) {
                if ((cprev >> 8) & F_GROUP_END) flush(cprev >> 8);
                if ((c0 >> 8) & F_NEXT_ITEM) fetch_next_q();
            }
            M4D_V_BODY(c0);
            M4D_M_BODY();
            M4D_END_INTERVAL(0);
        }
        if ((cprev >> 8) & F_GROUP_END) flush(cprev >> 8);
        M4D_V_BODY(c0);
        M4D_LAST_PV();
    } else {
        // late group, interval g: M(g) [flush of tile g] [next item's Q if tile g+1 is its item's last] V(g+1)
        if ((c0 >> 8) & F_NEXT_ITEM) fetch_next_q();
        M4D_V_BODY(c0);
        for (; g + 1 < G; ++g) {
            M4D_M_BODY();
            int after = 0;
            if (!(XP_VAR & 4)) {
                if ((c0 >> 8) & F_GROUP_END) after = flush(c0 >> 8);
                if ((c1 >> 8) & F_NEXT_ITEM) after += fetch_next_q();
            }
            M4D_V_BODY(c1);
            M4D_END_INTERVAL(after);
        }
        M4D_LAST_PV();
    }
    flush((c0 >> 8) & ~F_NEXT_ITEM);
#undef cprev
#undef c0
#undef c1
#undef c2
#undef SET_CODE
#undef M4D_V_BODY
#undef M4D_M_BODY
#undef M4D_END_INTERVAL
#undef M4D_LAST_PV
#undef M4D_QK_PREFETCH
#undef M4D_QK_TILE
#undef M4D_QK
#undef M4D_QK0
#undef M4D_LGKM
#undef M4D_DSR
}
