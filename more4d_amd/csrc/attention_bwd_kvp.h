// attn_bwd_kvp_kernel: the fused dK / dV pass of the flash-attention backward (bf16, head_dim 128) with the FORWARD kernel's schedule
// (attention_phased.h): every wave alternates a VALU phase and ONE stream of 32 MFMAs behind a ring of hand-issued fragment reads, and the
// two waves that share a SIMD run half an interval apart, so the SIMD's matrix pipe always has a stream to run while its partner is in the
// elementwise step.
//
// Roles (attention_bwd128.h: attn_bwd_kv128_kernel, same math, same role split, same mailbox): 128 key rows per workgroup, the work of 32
// key rows is split over the wave pair (w, w + 4):
//   P-wave  (w < 4,  X = K rows):  V: P(j) = exp2(S(j) sc - lse)  -> registers + mailbox      M: dV^T += dO(j)^T P(j) ;  S(j+1) = Q(j+1) K^T
//   dS-wave (w >= 4, X = V rows):  M: dK^T += Q(j-1)^T dS(j-1) ;  G(j) = dO(j) V^T             V: dS(j) = P(j) (G(j) - delta)
// Interval j (one workgroup barrier each) = {P-wave: V, M | dS-wave: M, V}.  P(j) crosses inside the interval through the pair's LDS
// mailbox and a flag (the P-wave posts it ~600 cycles into the interval, the dS-wave needs it ~1 100 cycles in).  Both M streams are the
// same code: 16 accumulate steps (A = transposed fragment of a row-major tile by two ds_read_b64_tr_b16, B = P / dS fragment) then 16
// S / G steps (A = row fragment by ds_read_b128, B = X fragment), ring of 6 fragments, the tile request and the address updates in the
// shadows of the MFMAs.
//
// Y tiles (64 queries) are ROW-MAJOR ONLY (bwd_tr_* in attention_bwd128.h): Q in a ring of 5 slots (live: Q(j-1) for dK, Q(j+1) for S;
// Q(j+3) is requested in interval j), dO and the statistics in a ring of 3 (live: dO(j); dO(j+2) requested in interval j): every request
// has two whole intervals to land.  16 B per MFMA cycle and CU of DMA traffic — the forward kernel's figure — against 32 for the first
// fused kernel with its four tiles per step.
#pragma once
#include <type_traits>

#ifndef KVP_RD
#define KVP_RD 6        // fragments in flight per wave (4, 5, 7 measured: within 0.5 %)
#endif
#ifndef KVP_PRIO
#define KVP_PRIO 1      // 0 no s_setprio, 1 raised priority around every M stream, 2 static: the dS-waves (younger half) at priority 1 (all within 1 %)
#endif
#ifndef KVP_ABL
#define KVP_ABL 0       // side builds (tools/side_lib.sh, tools/abl_kvp.sh), timing only, results wrong: 1 no elementwise arithmetic, 2 no MFMAs,
#endif                  // 4 no fragment reads, 8 no mailbox / flag, 16 no tile requests
#ifndef KVP_STAMPS
#define KVP_STAMPS 0    // side build: s_memtime stamps of waves 0 / 4 of workgroup 1000 (intervals 100..115), printed by the kernel (tools/kvp_stamps.py)
#endif
namespace kvp {
constexpr int RD = KVP_RD;            // fragments in flight per wave (accumulate steps: two reads each -> lgkmcnt <= 12)
constexpr int NQ = 5, ND = 3;         // ring slots
constexpr int TILE = 16384;
constexpr int Q_OFF = 0, DO_OFF = NQ * TILE, ST_OFF = DO_OFF + ND * TILE, MAIL_OFF = ST_OFF + ND * 512, FLAG_OFF = MAIL_OFF + 16384;
constexpr int STAMP_OFF = FLAG_OFF + 64;
constexpr int LDS_BYTES = STAMP_OFF + (KVP_STAMPS ? 2048 : 0);
// fragment J of an M stream: J < 16 accumulate step (chunk J >> 2, d-block J & 3), two transposing reads; J >= 16 S / G step
// (32-row half (J - 16) >> 3, k-step (J - 16) & 7), one 16-byte read
constexpr int nreads(int J) { return J < 16 ? 2 : 1; }
// reads in flight behind fragment J when step J waits for it (stream = fragments [.., J1); fragments J + 1 .. J + RD - 1 were issued after it)
constexpr int behind(int J, int J1) {
    int n = 0;
    for (int k = J + 1; k < J + RD && k < J1; ++k) n += nreads(k);
    return n;
}
struct Ring { bf16x4 tl[RD], th[RD]; bf16x8 rb[RD]; };

template <int J> M4D_DEV void read(Ring& r, const unsigned (&ta)[2][4], const unsigned (&ra)[8]) {
    if constexpr (KVP_ABL & 4) return;
    if constexpr (J < 16) {
        bwd_tr_read<(J >> 2) * 4096>(r.tl[J % RD], ta[0][J & 3]);
        bwd_tr_read<(J >> 2) * 4096>(r.th[J % RD], ta[1][J & 3]);
    } else {
        constexpr int I = J - 16;
        bkv_dsr<(I >> 3) * 8192>(r.rb[J % RD], ra[I & 7]);
    }
}
template <int J, int END> M4D_DEV void prefetch(Ring& r, const unsigned (&ta)[2][4], const unsigned (&ra)[8]) {
    if constexpr (J < END) { read<J>(r, ta, ra); prefetch<J + 1, END>(r, ta, ra); }
}
// steps J .. J1 - 1 of the stream that started at J0 (its first RD fragments already requested by prefetch<J0, J0 + RD>)
template <int J, int J0, int J1, typename Hook>
M4D_DEV void steps(Ring& r, const unsigned (&ta)[2][4], const unsigned (&ra)[8], const bf16x8 (&pf)[4], const bf16x8 (&xf)[8],
                   f32x16 (&acc)[4], f32x16 (&sg)[2], Hook&& hook) {
    if constexpr (J < J1) {
        bkv_lgkm<behind(J, J1)>();
        if constexpr (KVP_ABL & 2) {}
        else if constexpr (J < 16) mma32(bwd_tr_join(r.tl[J % RD], r.th[J % RD]), pf[J >> 2], acc[J & 3]);
        else {
            constexpr int I = J - 16;
            if constexpr ((I & 7) == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                sg[I >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.rb[J % RD], xf[0], zero, 0, 0, 0);      // first k-step: C = 0
            } else mma32(r.rb[J % RD], xf[I & 7], sg[I >> 3]);
        }
        if constexpr (J + RD < J1) read<J + RD>(r, ta, ra);
        hook(std::integral_constant<int, J - J0>{}, std::integral_constant<int, J>{});
        __builtin_amdgcn_sched_barrier(0);
        steps<J + 1, J0, J1>(r, ta, ra, pf, xf, acc, sg, hook);
    }
}
}  // namespace kvp

__global__ __launch_bounds__(512, 2) void attn_bwd_kvp_kernel(BwdArgs p) {
    typedef bf16_t T;
    using namespace kvp;
    constexpr int D = 128, YB = 64, XB = 128;
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
    const int pair = wave & 3;
    const bool ds_role = wave >= 4;                 // wave-uniform
    const int64_t xrow = (int64_t)xt * XB + pair * 32 + li;
    const bool xvalid = xrow < p.LX;

    bf16x8 xf[8];                                   // K rows (P-wave) or V rows (dS-wave); rows beyond LX are zero: their columns of
    {                                               // S / G / the accumulators are garbage nobody stores (a column depends on its own X row only)
        const T* px = ds_role ? (const T*)p.xb + b * p.xb_bs + xrow * p.xb_ls : (const T*)p.xa + b * p.xa_bs + xrow * p.xa_ls;
        px += (int64_t)h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (xvalid) xf[kk] = *reinterpret_cast<const bf16x8*>(px + kk * 16);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[kk][j] = (T)0.f;
            }
        }
    }
    f32x16 acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    // fragment addresses.  ra: row fragments of the S / G operand (P-wave: Q ring, dS-wave: dO ring); ta: transposing reads of the
    // accumulate operand (P-wave: dO ring, dS-wave: Q ring).  Both start in slot 0 of their ring and move one slot per tile.
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    unsigned ra[8], ta[2][4];
    {
        const unsigned rbase = lds0 + (ds_role ? DO_OFF : Q_OFF), tbase = lds0 + (ds_role ? Q_OFF : DO_OFF);
        const int kr = bwd_tr_row(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ra[kk] = rbase + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) ta[jj][dd] = tbase + bwd_tr_addr(li, hi, jj, dd);
    }
    const int n_r = ds_role ? ND : NQ, n_t = ds_role ? NQ : ND;       // ring sizes behind ra / ta (scalars)
    int slot_r = 0, slot_t = 0;
    const unsigned mail = lds0 + MAIL_OFF + pair * 4096 + lane * 16;
    const unsigned flag = lds0 + FLAG_OFF + pair * 4;
    if (!ds_role && lane == 0) *reinterpret_cast<volatile unsigned*>(smem + FLAG_OFF + pair * 4) = 0u;

    // ---- tile requests: scalar bases + per-lane 32-bit offsets; FIVE instructions per wave and request (two Q pieces, two dO pieces,
    // one statistics row: even waves lse, odd waves delta, in accumulator-register order bwd_tr_stat) ----
    const int NT = (int)((p.LY + YB - 1) / YB);
    const bool ragged = (p.LY % YB) != 0;
    const int k_r = lane >> 4, k_lc0 = lane & 15;
    const T* gq = (const T*)p.ya + b * p.ya_bs + (int64_t)h * D;
    const T* gdo = (const T*)p.yb + b * p.yb_bs + (int64_t)h * D;
    const float* gst = (wave & 1 ? p.delta : p.lse) + ((int64_t)b * p.heads + h) * p.Lq;
    unsigned oq[2], odo[2], oq_l[2], odo_l[2], ostat, ostat_l;       // *_l: the ragged last tile, rows beyond LY clamped to row LY - 1
    {                                                              // (finite values of a real row; their P is masked to zero)
        const int rem = (int)(p.LY - (int64_t)(NT - 1) * YB);     // rows of the last tile (1..64)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + k_r;
            const int rowc = row < rem ? row : rem - 1;
            const unsigned sw = (unsigned)((k_lc0 ^ (row & 15)) * 16);
            oq[i] = (unsigned)(row * p.ya_ls * 2) + sw;
            odo[i] = (unsigned)(row * p.yb_ls * 2) + sw;
            oq_l[i] = (unsigned)(rowc * p.ya_ls * 2) + sw;
            odo_l[i] = (unsigned)(rowc * p.yb_ls * 2) + sw;
        }
        const int sy = bwd_tr_stat(lane);
        ostat = (unsigned)sy * 4u;
        ostat_l = (unsigned)(sy < rem ? sy : rem - 1) * 4u;
    }
    auto uniform_ptr = [](const char* q) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi2 << 32) | lo);
    };
#define KVP_GLDS(DST, VOFF, SRC) if constexpr (!(KVP_ABL & 16)) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(DST), "v"(VOFF), "s"(SRC) : "memory", "m0")
    const char *rq_b = nullptr, *rdo_b = nullptr, *rst_b = nullptr;
    unsigned rq_dst = 0, rdo_dst = 0, rst_dst = 0;
    bool rq_last = false, rdo_last = false;
    int qd_slot = 0, dd_slot = 0;         // ring slots of the NEXT Q / dO request
    // request of Q(tq) and dO(td) + statistics(td) in two parts so that the streams can place the instructions in MFMA shadows;
    // indices beyond the last tile re-request the last tile into the (free) slot: every wave issues the same five instructions
    auto req_prepare = [&](int tq, int td) {
        tq = tq < NT ? tq : NT - 1;
        td = td < NT ? td : NT - 1;
        rq_last = ragged && tq == NT - 1;
        rdo_last = ragged && td == NT - 1;
        rq_b = uniform_ptr((const char*)(gq + (int64_t)tq * YB * p.ya_ls));
        rdo_b = uniform_ptr((const char*)(gdo + (int64_t)td * YB * p.yb_ls));
        rst_b = uniform_ptr((const char*)(gst + (int64_t)td * YB));
        rq_dst = __builtin_amdgcn_readfirstlane(lds0 + Q_OFF + qd_slot * TILE + wave * 2048);
        rdo_dst = __builtin_amdgcn_readfirstlane(lds0 + DO_OFF + dd_slot * TILE + wave * 2048);
        rst_dst = __builtin_amdgcn_readfirstlane(lds0 + ST_OFF + dd_slot * 512 + (wave & 1) * 256);
        qd_slot = qd_slot == NQ - 1 ? 0 : qd_slot + 1;
        dd_slot = dd_slot == ND - 1 ? 0 : dd_slot + 1;
    };
    auto req_q = [&](int i) {            // i = 0, 1: a literal at every call site
        const unsigned off = rq_last ? oq_l[i] : oq[i];
        KVP_GLDS(rq_dst + i * 1024, off, rq_b);
    };
    auto req_do = [&](int i) {
        const unsigned off = rdo_last ? odo_l[i] : odo[i];
        KVP_GLDS(rdo_dst + i * 1024, off, rdo_b);
    };
    auto req_st = [&]() {
        const unsigned off = rdo_last ? ostat_l : ostat;
        if constexpr (!(KVP_ABL & 16)) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" :: "s"(rst_dst), "v"(off), "s"(rst_b) : "memory", "m0");
    };

    Ring ring;
    f32x16 sg[2];
    bf16x8 pf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[c][e] = (T)0.f;

    // hook of an M stream: K = step counted from the start of the stream (the request sits in the first steps), J = fragment index
    // (address sets move one ring slot as soon as their last read of the tile has been issued: ta[..][dd] by step 6 + dd, ra[kk] by
    // step 18 + kk)
    bool do_req = false;
    int req_tq = 0, req_td = 0;
    unsigned dl_t = 0, dl_r = 0;
    auto hook = [&](auto KK, auto JJ) {
        constexpr int K = decltype(KK)::value, J = decltype(JJ)::value;
        if constexpr (K == 1) { if (do_req) req_prepare(req_tq, req_td); }
        if constexpr (K == 2) { if (do_req) req_q(0); }
        if constexpr (K == 3) { if (do_req) req_do(0); }
        if constexpr (K == 4) { if (do_req) req_q(1); }
        if constexpr (K == 5) { if (do_req) req_do(1); }
        if constexpr (K == 6) { if (do_req) req_st(); }
        if constexpr (J >= 12 && J < 16) { ta[0][J - 12] += dl_t; ta[1][J - 12] += dl_t; }
        if constexpr (J >= 22 && J < 30) ra[J - 22] += dl_r;
    };
    // deltas for the stream about to run + slot bookkeeping (a set that must stay where it is gets delta 0)
    auto advance_deltas = [&](bool adv_t, bool adv_r) {
        dl_t = !adv_t ? 0u : slot_t == n_t - 1 ? (unsigned)(-(n_t - 1) * TILE) : (unsigned)TILE;
        dl_r = !adv_r ? 0u : slot_r == n_r - 1 ? (unsigned)(-(n_r - 1) * TILE) : (unsigned)TILE;
        if (adv_t) slot_t = slot_t == n_t - 1 ? 0 : slot_t + 1;
        if (adv_r) slot_r = slot_r == n_r - 1 ? 0 : slot_r + 1;
    };

    // ---- P-wave elementwise step: P(j) = exp2(S sc - lse) -> pf + mailbox + flag.  MASK: the ragged last tile (queries >= LY -> 0) ----
    auto v_p = [&](int j, auto MASKED) {
        constexpr bool MASK = decltype(MASKED)::value;
        const float* st = reinterpret_cast<const float*>(smem + ST_OFF + (j % ND) * 512);
        // (hipcc's hazard recogniser does not look inside inline asm: a 16-pass MFMA result needs 18 wait states before a VALU reads it)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 3");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int half = c >> 1, rb = (c & 1) * 8, yb = c * 16 + 8 * hi;
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(st + yb), l1 = *reinterpret_cast<const f32x4*>(st + yb + 4);
            float lv[8], x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { lv[e] = l0[e]; lv[4 + e] = l1[e]; }
            if constexpr (MASK) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((int64_t)j * YB + bwd_tr_stat(yb + e) >= p.LY) lv[e] = INFINITY;      // lse = +inf => probability exactly 0
            }
            // single-issue v_fma_f32 / v_exp_f32 as one volatile stream (packed fp32 VALU costs more than its two halves beside the partner
            // wave's MFMAs); a VALU consuming a v_exp_f32 result sits eight instructions behind it
            if constexpr (KVP_ABL & 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = sg[half][rb + e] + lv[e & 1];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(x[e]) : "v"(sg[half][rb + e]), "s"(p.sc), "v"(lv[e]));
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_exp_f32 %0, %1" : "=v"(x[e]) : "v"(x[e]));
                asm volatile("s_nop 1");
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[c][e] = (T)x[e];
        }
        if constexpr (!(KVP_ABL & 8)) {
            asm volatile("ds_write_b128 %0, %1" :: "v"(mail), "v"(pf[0]) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(mail), "v"(pf[1]) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:2048" :: "v"(mail), "v"(pf[2]) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:3072" :: "v"(mail), "v"(pf[3]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned seq = (unsigned)j + 1u;
            asm volatile("ds_write_b32 %0, %1" :: "v"(flag), "v"(seq) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- dS-wave elementwise step: dS(j) = P(j) (G - delta) (the softmax scale is applied to dK once, at the end) ----
    auto v_ds = [&](int j) {
        const float* st = reinterpret_cast<const float*>(smem + ST_OFF + (j % ND) * 512) + 64;
        const unsigned seq = (unsigned)j + 1u;
        bf16x8 pm[4];
        if constexpr (KVP_ABL & 8) { pm[0] = xf[0]; pm[1] = xf[1]; pm[2] = xf[2]; pm[3] = xf[3]; }
        else {
            for (int spin = 0; spin < (1 << 22); ++spin) {      // (bounded: a lost flag must end in wrong numbers the tests catch, never in a hung GPU)
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag) : "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= seq) break;
                __builtin_amdgcn_s_sleep(1);
            }
            bkv_dsr<0>(pm[0], mail); bkv_dsr<1024>(pm[1], mail); bkv_dsr<2048>(pm[2], mail); bkv_dsr<3072>(pm[3], mail);
            bkv_lgkm<0>();
        }
        asm volatile("s_nop 15\n\ts_nop 3");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int half = c >> 1, rb = (c & 1) * 8, yb = c * 16 + 8 * hi;
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(st + yb), d1 = *reinterpret_cast<const f32x4*>(st + yb + 4);
            float dv[8], tt[8], x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { dv[e] = d0[e]; dv[4 + e] = d1[e]; }
            if constexpr (KVP_ABL & 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { tt[e] = dv[e & 1]; x[e] = sg[half][rb + e] + tt[e] + (float)pm[c][e & 1]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(tt[e]) : "v"(sg[half][rb + e]), "v"(dv[e]));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pe = (float)pm[c][e];
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[e]) : "v"(pe), "v"(tt[e]));
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[c][e] = (T)x[e];
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: [Q(0) dO(0) st(0)] [Q(1) dO(1) st(1)] [Q(2)]; the P-waves compute S(0) in front of interval 0 ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // X fragments
    req_prepare(0, 0); req_q(0); req_do(0); req_q(1); req_do(1); req_st();
    req_prepare(1, 1); req_q(0); req_do(0); req_q(1); req_do(1); req_st();
    {   // Q(2) alone: the dO ring's third slot is filled by interval 0's request
        const int tq = 2 < NT ? 2 : NT - 1;
        rq_last = ragged && tq == NT - 1;
        rq_b = uniform_ptr((const char*)(gq + (int64_t)tq * YB * p.ya_ls));
        rq_dst = __builtin_amdgcn_readfirstlane(lds0 + Q_OFF + qd_slot * TILE + wave * 2048);
        qd_slot += 1;
        req_q(0); req_q(1);
    }
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");          // Q(0), dO(0), st(0) of this wave landed
    __builtin_amdgcn_s_barrier();
    // ONE stream shape everywhere (accumulators that flow through differently shaped branches cost hipcc copies of all 64 registers and
    // spills): where half of a stream has nothing to do it runs on harmless operands — P = dS = 0 against a landed tile (accumulators
    // unchanged), or S / G of a stale slot that nobody reads afterwards.
    int iv = -1;                                               // interval counter of the stamps
    const bool stamp_on = KVP_STAMPS && blockIdx.x == (gridDim.x > 1000 ? 1000 : 0) && pair == 0;
    // stamps (side build): slot 0 interval start, 1 between the wave's two phases, 2 before / 3 after the wait for the tile requests
#define KVP_STAMP(SLOT)                                                                                                \
    do {                                                                                                               \
        if constexpr (KVP_STAMPS) {                                                                                    \
            if (stamp_on && iv >= 100 && iv < 116) {                                                                   \
                unsigned long long tc_;                                                                                \
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc_) :: "memory");                          \
                if (lane == 0) *(volatile LDS_AS unsigned long long*)((LDS_AS char*)smem + STAMP_OFF + (ds_role ? 1024 : 0) + ((iv - 100) * 4 + (SLOT)) * 8) = tc_; \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)
#define KVP_END_INTERVAL()                                                                                             \
    do {                                                                                                               \
        KVP_STAMP(2);                                                                                                  \
        if (do_req) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      /* everything but this interval's own request has landed */ \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        KVP_STAMP(3);                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                  \
        ++iv;                                                                                                          \
        KVP_STAMP(0);                                                                                                  \
    } while (0)
#define KVP_M_STREAM()                                                                                                 \
    do {                                                                                                               \
        if constexpr (KVP_PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                    \
        steps<0, 0, 32>(ring, ta, ra, pf, xf, acc, sg, hook);                                                          \
        if constexpr (KVP_PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                    \
    } while (0)
    if constexpr (KVP_PRIO == 2) { if (ds_role) __builtin_amdgcn_s_setprio(1); }
    if (!ds_role) {
        // in front of interval 0: S(0) = Q(0) K^T (the accumulate half runs on P = 0 against dO(0))
        do_req = false;
        prefetch<0, RD>(ring, ta, ra);
        advance_deltas(false, true);
        KVP_M_STREAM();
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // Q(1), dO(1), st(1) landed
        __builtin_amdgcn_s_barrier();
        for (int j = 0; j < NT; ++j) {
            do_req = j + 2 < NT;                               // Q(j + 3) (clamped), dO(j + 2), st(j + 2)
            req_tq = j + 3; req_td = j + 2;
            prefetch<0, RD>(ring, ta, ra);                     // dO(j)^T fragments land under the elementwise step
            if (ragged && j == NT - 1) v_p(j, std::true_type{}); else v_p(j, std::false_type{});
            KVP_STAMP(1);
            advance_deltas(true, true);
            KVP_M_STREAM();                                    // dV += dO(j)^T P(j); S(j + 1) (of a stale slot after the last tile)
            KVP_END_INTERVAL();
        }
        do_req = false;
        KVP_END_INTERVAL();                                    // interval NT: the dS-waves' last dK steps
    } else {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prefetch<0, RD>(ring, ta, ra);                         // interval 0: dS(-1) = 0 against Q(0)
        for (int j = 0; j < NT; ++j) {
            do_req = j + 2 < NT;
            req_tq = j + 3; req_td = j + 2;
            advance_deltas(j > 0, true);                       // ta: Q(j - 1) -> Q(j) (stays on Q(0) in interval 0); ra: dO(j) -> dO(j + 1)
            KVP_M_STREAM();                                    // dK += Q(j - 1)^T dS(j - 1); G(j) = dO(j) V^T
            KVP_STAMP(1);
            v_ds(j);
            prefetch<0, RD>(ring, ta, ra);                     // Q(j)^T fragments of the next interval's dK steps (the tile landed long ago)
            KVP_END_INTERVAL();
        }
        do_req = false;
        advance_deltas(false, false);
        KVP_M_STREAM();                                        // interval NT: dK += Q(NT - 1)^T dS(NT - 1) (G of a stale slot, unused)
        KVP_END_INTERVAL();
    }
#undef KVP_M_STREAM
#undef KVP_END_INTERVAL
    if constexpr (KVP_STAMPS) {
        __syncthreads();
        if (stamp_on && t == 0) {
            const LDS_AS unsigned long long* sp = (const LDS_AS unsigned long long*)((LDS_AS char*)smem + STAMP_OFF);
            for (int k = 0; k < 16; ++k)
                printf("KVPSTAMP %d  P: %llu %llu %llu %llu   dS: %llu %llu %llu %llu\n", k, sp[k * 4], sp[k * 4 + 1], sp[k * 4 + 2], sp[k * 4 + 3],
                       sp[128 + k * 4], sp[128 + k * 4 + 1], sp[128 + k * 4 + 2], sp[128 + k * 4 + 3]);
        }
    }
#undef KVP_GLDS

    // (epilogue-only kernel arguments through an opaque copy of the kernarg pointer: scalar loads here instead of ~16 SGPRs that live
    // through the loop of a kernel that is out of them — attention_xp.h)
    typedef const BwdArgs __attribute__((address_space(4))) * kernarg_t;
    kernarg_t pa = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pa));
    const int accumulate = pa->accumulate;
    if (xrow < pa->LXs) {
        T* oa = ds_role ? (T*)pa->out_a + b * pa->oa_bs + xrow * pa->oa_ls : (T*)pa->out_b + b * pa->ob_bs + xrow * pa->ob_ls;
        oa += (int64_t)h * D + hi * 4;
        const float osc = ds_role ? pa->scale : 1.f;       // dK = scale * sum (P (G - delta))^T Q
        const bool keep = xvalid;                         // rows in [LX, LXs) are stored as zeros
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
                for (int e = 0; e < 4; ++e) v[e] = keep ? acc[d][rq * 4 + e] * osc : 0.f;
                T* dst = oa + d * 32 + rq * 8;
                if (accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += prev[d][rq][e];
                }
                store4(dst, v);
            }
    }
}

inline int launch_bwd_kvp(const BwdArgs& p, hipStream_t st) {
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)attn_bwd_kvp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kvp::LDS_BYTES) != hipSuccess) return -3;
        configured.mark();
    }
    dim3 grid((unsigned)((int64_t)p.nx_tiles * p.heads * p.B));
    hipLaunchKernelGGL(attn_bwd_kvp_kernel, grid, dim3(512), kvp::LDS_BYTES, st, p);
    return 0;
}
