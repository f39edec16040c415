// attn_bwd_dqp_kernel: the dQ pass of the flash-attention backward (bf16, head_dim 128) with the forward kernel's schedule, like the fused
// dK / dV pass next to it (attention_bwd_kvp.h).  Same math and operand layout as attn_bwd128_kernel<BWD_DQ> (attention_bwd128.h):
//   X = 32 query rows per wave (Q and dO fragments in registers, lse / delta lane-local), Y = 64-key tiles of (K, V), row-major,
//   S^T = K Q^T,  G^T = V dO^T,  dS = P (G - delta) scale,  dQ^T += K^T dS^T   (K^T fragments by ds_read_b64_tr_b16 out of the K tile)
// but the unit of work is a 32-key HALF tile u (S and G of a whole tile would need 64 registers next to the 128 of the X fragments):
//   V(u): elementwise step on S(u), G(u) -> dS(u) (two 16-key B fragments)                                    72 VALU, 16 of them v_exp_f32
//   M(u): dQ^T += K(u)^T dS(u)  (8 MFMAs, transposing reads)  then  S(u+1), G(u+1)  (16 MFMAs, ds_read_b128)   one stream of 24 MFMAs
// Every wave runs V(u) M(u) V(u+1) M(u+1) ...; the wave group {4..7} runs one phase behind {0..3}, so on every SIMD one wave streams MFMAs
// while its partner is in the elementwise step (the lock-step kernel took 59 cycles per MFMA and SIMD, the forward kernel takes 47).
// One workgroup barrier per TILE (two half steps); K / V tiles through four 32 KiB stages by DMA: in interval i the early group reads tiles
// i and i + 1, the late group i - 1 and i; tile i + 2 is requested at the start of interval i into the stage tile i - 2 left and has the
// whole interval (~2 us) to land.
#pragma once
#include <type_traits>

#ifndef DQP_RD
#define DQP_RD 6
#endif
namespace dqp {
constexpr int RD = DQP_RD;            // fragments in flight per wave (accumulate steps: two reads each)
constexpr int NST = 4, STAGE = 32768, VOFF = 16384;
constexpr int LDS_BYTES = NST * STAGE;
// fragment J of an M stream: J < 8 accumulate step (chunk J >> 2, d-block J & 3), two transposing reads; J >= 8: k-step (J - 8) >> 1 of
// S (even) or G (odd), one 16-byte read
constexpr int nreads(int J) { return J < 8 ? 2 : 1; }
constexpr int behind(int J) {
    int n = 0;
    for (int k = J + 1; k < J + RD && k < 24; ++k) n += nreads(k);
    return n;
}
struct Ring { bf16x4 tl[RD], th[RD]; bf16x8 rb[RD]; };

// HA: half of the tile behind `ta` the accumulate steps read; HS: half of the tile behind `ra` whose S / G are computed
template <int J, int HA, int HS> M4D_DEV void read(Ring& r, const unsigned (&ta)[2][4], const unsigned (&ra)[8]) {
    if constexpr (J < 8) {
        bwd_tr_read<HA * 8192 + (J >> 2) * 4096>(r.tl[J % RD], ta[0][J & 3]);
        bwd_tr_read<HA * 8192 + (J >> 2) * 4096>(r.th[J % RD], ta[1][J & 3]);
    } else {
        constexpr int I = J - 8;
        bkv_dsr<HS * 8192 + (I & 1) * VOFF>(r.rb[J % RD], ra[I >> 1]);
    }
}
template <int J, int END, int HA, int HS> M4D_DEV void prefetch(Ring& r, const unsigned (&ta)[2][4], const unsigned (&ra)[8]) {
    if constexpr (J < END) { read<J, HA, HS>(r, ta, ra); prefetch<J + 1, END, HA, HS>(r, ta, ra); }
}
template <int J, int HA, int HS, typename Hook>
M4D_DEV void steps(Ring& r, const unsigned (&ta)[2][4], const unsigned (&ra)[8], const bf16x8 (&pf)[2], const bf16x8 (&xq)[8],
                   const bf16x8 (&xd)[8], f32x16 (&acc)[4], f32x16& s, f32x16& g, Hook&& hook) {
    if constexpr (J < 24) {
        bkv_lgkm<behind(J)>();
        if constexpr (J < 8) mma32(bwd_tr_join(r.tl[J % RD], r.th[J % RD]), pf[J >> 2], acc[J & 3]);
        else {
            constexpr int I = J - 8, KK = I >> 1;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr ((I & 1) == 0) {
                if constexpr (KK == 0) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.rb[J % RD], xq[0], zero, 0, 0, 0);      // first k-step: C = 0
                else mma32(r.rb[J % RD], xq[KK], s);
            } else {
                if constexpr (KK == 0) g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.rb[J % RD], xd[0], zero, 0, 0, 0);
                else mma32(r.rb[J % RD], xd[KK], g);
            }
        }
        if constexpr (J + RD < 24) read<J + RD, HA, HS>(r, ta, ra);
        hook(std::integral_constant<int, J>{});
        __builtin_amdgcn_sched_barrier(0);
        steps<J + 1, HA, HS>(r, ta, ra, pf, xq, xd, acc, s, g, hook);
    }
}
}  // namespace dqp

__global__ __launch_bounds__(512, 2) void attn_bwd_dqp_kernel(BwdArgs p) {
    typedef bf16_t T;
    using namespace dqp;
    constexpr int D = 128, YB = 64, XB = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int HB = p.heads * p.B;
    int xt, hb;
    if ((HB & 7) == 0) {     // (b, h) groups pinned per XCD: the Y operands of a head stay in that XCD's L2
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        hb = xcd * (HB >> 3) + idx / p.nx_tiles;
        xt = idx % p.nx_tiles;
    } else {
        hb = blockIdx.x / p.nx_tiles;
        xt = blockIdx.x % p.nx_tiles;
    }
    const int b = hb / p.heads, h = hb % p.heads;
    const int t = threadIdx.x, lane = t & 63, li = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool late = wave >= 4;                    // wave-uniform: the group that runs one phase behind
    const int64_t xrow = (int64_t)xt * XB + wave * 32 + li;
    const bool xvalid = xrow < p.LX;

    bf16x8 xq[8], xd[8];                            // Q rows and dO rows of this lane's query (rows beyond LX: zero; their column of every
    {                                               // product is garbage nobody stores — a column depends on its own X row only)
        const T* pq = (const T*)p.xa + b * p.xa_bs + xrow * p.xa_ls + (int64_t)h * D + hi * 8;
        const T* pd = (const T*)p.xb + b * p.xb_bs + xrow * p.xb_ls + (int64_t)h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (xvalid) { xq[kk] = *reinterpret_cast<const bf16x8*>(pq + kk * 16); xd[kk] = *reinterpret_cast<const bf16x8*>(pd + kk * 16); }
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { xq[kk][j] = (T)0.f; xd[kk][j] = (T)0.f; }
            }
        }
    }
    float lse_x = 0.f, nd_x = 0.f;                  // nd_x = delta * scale
    if (xvalid) {
        const int64_t si = ((int64_t)b * p.heads + h) * p.Lq + xrow;
        lse_x = p.lse[si];
        nd_x = p.delta[si] * p.scale;
    }
    f32x16 acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    // fragment addresses, both sets start on stage 0: ra = row fragments of the tile whose S / G are computed next (K; V at + VOFF),
    // ta = transposing reads of the K tile whose dS is accumulated next
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    unsigned ra[8], ta[2][4];
    {
        const int kr = bwd_tr_row(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ra[kk] = lds0 + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) ta[jj][dd] = lds0 + bwd_tr_addr(li, hi, jj, dd);
    }
    int slot_r = 0, slot_t = 0;

    // ---- tile requests: scalar bases + per-lane 32-bit offsets, four instructions per wave and tile (two K pieces, two V pieces) ----
    const int NT = (int)((p.LY + YB - 1) / YB);
    const bool ragged = (p.LY % YB) != 0;
    const int k_r = lane >> 4, k_lc0 = lane & 15;
    const T* gk = (const T*)p.ya + b * p.ya_bs + (int64_t)h * D;
    const T* gv = (const T*)p.yb + b * p.yb_bs + (int64_t)h * D;
    unsigned ok_[2], ov_[2], ok_l[2], ov_l[2];      // *_l: the ragged last tile, rows beyond LY clamped to row LY - 1 (their dS is masked to 0)
    {
        const int rem = (int)(p.LY - (int64_t)(NT - 1) * YB);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + k_r;
            const int rowc = row < rem ? row : rem - 1;
            const unsigned sw = (unsigned)((k_lc0 ^ (row & 15)) * 16);
            ok_[i] = (unsigned)(row * p.ya_ls * 2) + sw;
            ov_[i] = (unsigned)(row * p.yb_ls * 2) + sw;
            ok_l[i] = (unsigned)(rowc * p.ya_ls * 2) + sw;
            ov_l[i] = (unsigned)(rowc * p.yb_ls * 2) + sw;
        }
    }
    auto uniform_ptr = [](const char* q) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi2 << 32) | lo);
    };
#define DQP_GLDS(DST, VOFF_, SRC) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(DST), "v"(VOFF_), "s"(SRC) : "memory", "m0")
    const char *rk_b = nullptr, *rv_b = nullptr;
    unsigned r_dst = 0;
    bool r_last = false;
    int rq_slot = 0;                                // stage of the NEXT request
    auto req_prepare = [&](int tile) {              // tiles beyond the last one re-request the last tile into the (free) stage
        tile = tile < NT ? tile : NT - 1;
        r_last = ragged && tile == NT - 1;
        rk_b = uniform_ptr((const char*)(gk + (int64_t)tile * YB * p.ya_ls));
        rv_b = uniform_ptr((const char*)(gv + (int64_t)tile * YB * p.yb_ls));
        r_dst = __builtin_amdgcn_readfirstlane(lds0 + rq_slot * STAGE + wave * 2048);
        rq_slot = rq_slot == NST - 1 ? 0 : rq_slot + 1;
    };
    auto req_k = [&](int i) { const unsigned off = r_last ? ok_l[i] : ok_[i]; DQP_GLDS(r_dst + i * 1024, off, rk_b); };
    auto req_v = [&](int i) { const unsigned off = r_last ? ov_l[i] : ov_[i]; DQP_GLDS(r_dst + VOFF + i * 1024, off, rv_b); };

    Ring ring;
    f32x16 s, g;
    bf16x8 pf[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[c][e] = (T)0.f;

    // hooks of the two stream shapes.  M_even (half 0 accumulated, S / G of half 1 of the same tile): carries the interval's tile request
    // and moves `ra` to the next tile once its last read has been issued; M_odd (half 1 accumulated, S / G of half 0 of the next tile):
    // moves `ta`.
    bool do_req = false;
    int req_tile = 0;
    unsigned dl_r = 0, dl_t = 0;
    auto hook_even = [&](auto JJ) {
        constexpr int J = decltype(JJ)::value;
        if constexpr (J == 1) { if (do_req) req_prepare(req_tile); }
        if constexpr (J == 2) { if (do_req) req_k(0); }
        if constexpr (J == 5) { if (do_req) req_v(0); }
        if constexpr (J == 8) { if (do_req) req_k(1); }
        if constexpr (J == 11) { if (do_req) req_v(1); }
        if constexpr (J >= 16 && J < 24) ra[J - 16] += dl_r;
    };
    auto hook_odd = [&](auto JJ) {
        constexpr int J = decltype(JJ)::value;
        if constexpr (J >= 4 && J < 8) { ta[0][J - 4] += dl_t; ta[1][J - 4] += dl_t; }
    };
    auto next_delta = [&](int& slot, bool adv) -> unsigned {
        if (!adv) return 0u;
        const unsigned d = slot == NST - 1 ? (unsigned)(-(NST - 1) * STAGE) : (unsigned)STAGE;
        slot = slot == NST - 1 ? 0 : slot + 1;
        return d;
    };

    // ---- elementwise step of half u: dS = P (G scale - delta scale), P = exp2(S sc - lse) -> pf[0..1].  MASK: the ragged last tile ----
    auto v_step = [&](int u, auto MASKED) {
        constexpr bool MASK = decltype(MASKED)::value;
        // (hipcc's hazard recogniser does not look inside inline asm: a 16-pass MFMA result needs 18 wait states before a VALU reads it)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 3");
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int rb = c * 8;
            float x[8], tt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(x[e]) : "v"(s[rb + e]), "s"(p.sc), "v"(lse_x));
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_exp_f32 %0, %1" : "=v"(x[e]) : "v"(x[e]));
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(tt[e]) : "v"(g[rb + e]), "s"(p.scale), "v"(nd_x));
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[e]) : "v"(x[e]), "v"(tt[e]));
            if constexpr (MASK) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((int64_t)(u >> 1) * YB + bwd_tr_stat((u & 1) * 32 + c * 16 + 8 * hi + e) >= p.LY) x[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[c][e] = (T)x[e];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#define DQP_V(U)                                                                                                       \
    do {                                                                                                               \
        if (ragged && ((U) >> 1) == NT - 1) v_step((U), std::true_type{}); else v_step((U), std::false_type{});        \
    } while (0)
#define DQP_M_EVEN()                                                                                                   \
    do {                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        steps<0, 0, 1>(ring, ta, ra, pf, xq, xd, acc, s, g, hook_even);                                                \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)
#define DQP_M_ODD()                                                                                                    \
    do {                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        steps<0, 1, 0>(ring, ta, ra, pf, xq, xd, acc, s, g, hook_odd);                                                 \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)
#define DQP_END()                                                                                                      \
    do {                                                                                                               \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* this interval's request (tile i + 2) has landed */ \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                  \
    } while (0)

    // ---- prologue: tiles 0 and 1 requested and landed ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // X fragments
    req_prepare(0); req_k(0); req_v(0); req_k(1); req_v(1);
    req_prepare(1); req_k(0); req_v(0); req_k(1); req_v(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // One stream shape per parity everywhere (accumulators that flow through differently shaped branches cost hipcc copies and spills):
    // the dry run in front of the first elementwise step is an M_odd on dS = 0 against tile 0 (accumulators unchanged) that leaves
    // S(0), G(0); the last M_odd computes S / G of a stale stage that nobody reads.
    if (!late) {
        dl_t = 0u;
        prefetch<0, RD, 1, 0>(ring, ta, ra);
        DQP_M_ODD();                                              // S(0), G(0)
        for (int i = 0; i < NT; ++i) {
            do_req = i + 2 < NT;
            req_tile = i + 2;
            prefetch<0, RD, 0, 1>(ring, ta, ra);
            DQP_V(2 * i);
            dl_r = next_delta(slot_r, true);
            DQP_M_EVEN();                                         // dQ += K(2i)^T dS(2i); S, G (2i + 1); ra -> tile i + 1
            prefetch<0, RD, 1, 0>(ring, ta, ra);
            DQP_V(2 * i + 1);
            dl_t = next_delta(slot_t, true);
            DQP_M_ODD();                                          // dQ += K(2i+1)^T dS(2i+1); S, G (2i + 2) of tile i + 1; ta -> tile i + 1
            DQP_END();
        }
    } else {
        prefetch<0, RD, 1, 0>(ring, ta, ra);                     // interval 0 starts with the dry run (dS = 0 against tile 0)
        for (int i = 0; i < NT; ++i) {
            do_req = i + 2 < NT;
            req_tile = i + 2;
            dl_t = next_delta(slot_t, i > 0);                    // ta: tile i - 1 -> tile i (stays on tile 0 in interval 0)
            DQP_M_ODD();                                          // dQ += K(2i-1)^T dS(2i-1); S, G (2i)
            prefetch<0, RD, 0, 1>(ring, ta, ra);
            DQP_V(2 * i);
            dl_r = next_delta(slot_r, true);
            DQP_M_EVEN();                                         // carries this group's share of the request
            prefetch<0, RD, 1, 0>(ring, ta, ra);
            DQP_V(2 * i + 1);
            DQP_END();
        }
        dl_t = 0u;
        DQP_M_ODD();                                              // dQ += K(2NT-1)^T dS(2NT-1)
    }
#undef DQP_END
#undef DQP_M_ODD
#undef DQP_M_EVEN
#undef DQP_V
#undef DQP_GLDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // (epilogue-only kernel arguments through an opaque copy of the kernarg pointer: attention_bwd_kvp.h)
    typedef const BwdArgs __attribute__((address_space(4))) * kernarg_t;
    kernarg_t pa = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pa));
    const int accumulate = pa->accumulate;
    if (xrow < pa->LXs) {
        T* oa = (T*)pa->out_a + b * pa->oa_bs + xrow * pa->oa_ls + (int64_t)h * D + hi * 4;
        // accumulate mode: ALL sixteen previous quads are requested before the first store (interleaved load / add / store, every
        // load is waited for alone behind the store in front of it: sixteen serial round trips per workgroup)
        f32x4 prev[4][4];
        if (accumulate) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) prev[d][rq] = load4(oa + d * 32 + rq * 8);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[d][rq * 4 + e];
                T* dst = oa + d * 32 + rq * 8;
                if (accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += prev[d][rq][e];
                }
                store4(dst, v);
            }
    }
}

inline int launch_bwd_dqp(const BwdArgs& p, hipStream_t st) {
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)attn_bwd_dqp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dqp::LDS_BYTES) != hipSuccess) return -3;
        configured.mark();
    }
    dim3 grid((unsigned)((int64_t)p.nx_tiles * p.heads * p.B));
    hipLaunchKernelGGL(attn_bwd_dqp_kernel, grid, dim3(512), dqp::LDS_BYTES, st, p);
    return 0;
}
