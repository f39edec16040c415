// Production flash-attention backward for bf16, head_dim 128 (the Wan2.1 DiT's only training shape).
// Same math as attn_bwd_kernel (attention_bwd.hip) split into THREE single-accumulator passes so that every pass has
// the forward kernel's shape (attention.hip: attn128_kernel) and inherits its measured structure:
//   * 8 waves x 32 X-rows per workgroup share each Y tile (two waves per SIMD, <= 256 VGPRs each);
//   * Y tiles arrive by global->LDS DMA into FOUR stages (tile i + 3 requested while tile i is computed, counted vmcnt), ONE barrier per
//     tile, swizzle applied on the source address; ROW-MAJOR tiles only: the transposed operand of the accumulate product is read out of
//     the same tile with ds_read_b64_tr_b16 (bwd_tr_* below), so no transposed copies of Q / K / dO travel (round 4);
//   * LDS fragment reads are hand-pipelined (inline asm ds_read_b128 + counted lgkmcnt), exp2 is the raw v_exp_f32.
//   DQ:  X = (Q, dO)  Y = (K, V)      dS = P (G - delta_x) scale        dQ^T += K^T  dS^T      3 matmuls
//   DK:  X = (K, V)   Y = (Q, dO)     dS = P (G - delta_y) scale        dK^T += Q^T  dS^T      3 matmuls
//   DV:  X = (K)      Y = (Q, dO)     P  = exp2(S sc - lse_y)           dV^T += dO^T P^T       2 matmuls
// (8 matmul units instead of the fused two-pass kernel's 7, for twice the occupancy and no 512-VGPR waves.)
#pragma once

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

enum { BWD_DQ = 0, BWD_DK = 1, BWD_DV = 2 };

// ---- transposed operands without transposed copies: ds_read_b64_tr_b16 (round 4) ----
// The accumulate products need the Y tile transposed (dV^T += dO^T P^T, dK^T += Q^T dS^T, dQ^T += K^T dS^T): A fragment = 32 d rows x 16 y,
// lane (i = d, hi) holding 8 y values of column d of the row-major tile.  gfx950's transposing LDS read does exactly that gather.
// Semantics (probed, tools/probes/ds_read_tr.hip): within each group of 16 lanes, lane s supplies the address of one 8-byte piece =
// row (s >> 2), column quad (s & 3) of a [4 rows][16 columns] block of 16-bit elements, and lane t receives column t of the block
// (4 elements, rows 0..3).  The four rows of a block may be any four rows (each lane has its own address); here they are rows
// y0, y0 + 4, y0 + 8, y0 + 12 of a 16-row chunk, because with the tile's XOR swizzle (16-byte chunk ^ (row & 15), applied on the DMA source
// address) rows that differ in bits 2..3 and agree in bits 0..1 put their four chunks on 16 different bank slots: the 32 lanes serviced
// per LDS cycle cover all 64 banks exactly once.  Two reads (jj = 0, 1) fill one fragment, so fragment slot (hi, e) of chunk c is
//     y = 16 c + 2 hi + (e >> 2) + 4 (e & 3)
// — and the S / G accumulators must deliver P / dS in that order: the row-major fragment reads take tile row bwd_tr_row(i) for MFMA row i
// (C layout: register r of lane-half hi is row (r & 3) + 8 (r >> 2) + 4 hi), which makes registers 8 c' + e of a 32-row half exactly
// slot e of chunk c'.  Conflict-free for ds_read_b128 as well: the 16 lanes of a service group read rows that are distinct mod 16.
M4D_DEV int bwd_tr_row(int i) { return (i & 16) | ((i & 3) << 2) | ((i >> 1) & 2) | ((i >> 3) & 1); }
// byte offset (inside a row-major [64][128] bf16 tile, chunk c = 0) of the piece lane (li, hi) supplies for read jj of d-block dd
M4D_DEV unsigned bwd_tr_addr(int li, int hi, int jj, int dd) {
    const int g = li >> 4, s = li & 15;
    const int row = 2 * hi + jj + 4 * (s >> 2);
    const int chunk = (4 * dd + 2 * g + ((s >> 1) & 1)) ^ row;
    return (unsigned)(row * 256 + chunk * 16 + (s & 1) * 8);
}
// statistics (lse / delta of a tile's 64 y) sit in LDS in accumulator-register order: position 32 half + 16 c + 8 hi + e holds y =
M4D_DEV int bwd_tr_stat(int pos) { return (pos & 48) | (((pos >> 3) & 1) << 1) | ((pos >> 2) & 1) | ((pos & 3) << 2); }
template <int OFF> M4D_DEV void bwd_tr_read(bf16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
M4D_DEV bf16x8 bwd_tr_join(const bf16x4& lo, const bf16x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

template <int MODE, int SMX>   // SMX: 1 = single-issue fp32 VALU forms in the elementwise step (default), 0 = packed v_pk_*_f32 (A/B)
__global__ __launch_bounds__(512, 2) void attn_bwd128_kernel(BwdArgs p) {
    typedef bf16_t T;
    constexpr int D = 128, YB = 64, XB = 256;
    constexpr bool HAS_G = MODE != BWD_DV;
    constexpr bool STAT_Y = MODE != BWD_DQ;
    constexpr int T1 = 16384;                          // second row-major tile (G operand; dV: the accumulate operand dO)
    constexpr int TT = MODE == BWD_DV ? T1 : 0;        // tile the transposing reads of the accumulate product go to (K / Q / dO)
    constexpr int STAT_OFF = 32768;
    constexpr int STAGE = STAT_OFF + (STAT_Y ? 512 : 0);
    constexpr int NST = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NST * STAGE

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
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hi = lane >> 5;
    const int64_t xrow = (int64_t)xt * XB + wave * 32 + li;
    const bool xvalid = xrow < p.LX;

    bf16x8 xaf[8], xbf[HAS_G ? 8 : 1];
    {
        const T* pa = (const T*)p.xa + b * p.xa_bs + xrow * p.xa_ls + (int64_t)h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (xvalid) xaf[kk] = *reinterpret_cast<const bf16x8*>(pa + kk * 16);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xaf[kk][j] = (T)0.f;
            }
        }
        if constexpr (HAS_G) {
            const T* pb = (const T*)p.xb + b * p.xb_bs + xrow * p.xb_ls + (int64_t)h * D + hi * 8;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (xvalid) xbf[kk] = *reinterpret_cast<const bf16x8*>(pb + kk * 16);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xbf[kk][j] = (T)0.f;
                }
            }
        }
    }
    float lse_x = 0.f, delta_x = 0.f;
    if (!STAT_Y && xvalid) {
        const int64_t si = ((int64_t)b * p.heads + h) * p.Lq + xrow;
        lse_x = p.lse[si];
        delta_x = p.delta[si];
    }
    f32x16 acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    // absolute LDS addresses of this lane's fragments in the CURRENT stage; toggled by +-STAGE after every tile
    unsigned ka[8], ta[2][4];
    {
        const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
        const int kr = bwd_tr_row(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] = lds0 + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) ta[jj][dd] = lds0 + TT + bwd_tr_addr(li, hi, jj, dd);
    }
    const int k_r = lane >> 4, k_lc0 = lane & 15;     // row-major tile DMA: 4 rows x 256 B per instruction

    const T* gya = (const T*)p.ya + b * p.ya_bs + (int64_t)h * D;
    const T* gyb = (const T*)p.yb + b * p.yb_bs + (int64_t)h * D;
    const float* glse = p.lse + ((int64_t)b * p.heads + h) * p.Lq;
    const float* gdel = p.delta + ((int64_t)b * p.heads + h) * p.Lq;

    // tile request: scalar bases + per-lane 32-bit offsets, in two parts so that the loop can spread the instructions over the first
    // MFMAs of a tile (round 4: issued back to back at the top of the tile, the 8 waves' requests queue on the CU's one texture-address
    // unit with every wave stuck behind its own — measured on the fused dK / dV kernel below: 6 ms of a 25 ms pass)
    unsigned oya[2], oyb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int blk = (t >> 6) * 2 + i;
        const int row = blk * 4 + k_r;
        oya[i] = (unsigned)((row * p.ya_ls + (k_lc0 ^ (row & 15)) * 8) * 2);
        oyb[i] = (unsigned)((row * p.yb_ls + (k_lc0 ^ (row & 15)) * 8) * 2);
    }
    auto uniform_ptr = [](const char* q) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi2 << 32) | lo);
    };
    const char *dya_b = nullptr, *dyb_b = nullptr;
    unsigned d_dst = 0;
    const unsigned lds0_ = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    auto dma_prepare = [&](int stage, int64_t y0) {
        dya_b = uniform_ptr((const char*)(gya + y0 * p.ya_ls));
        dyb_b = uniform_ptr((const char*)(gyb + y0 * p.yb_ls));
        d_dst = __builtin_amdgcn_readfirstlane(lds0_ + stage * STAGE + wave * 2048);
    };
#define M4D_BGLDS(DST, VOFF, SRC) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(DST), "v"(VOFF), "s"(SRC) : "memory", "m0")
    auto dma_one = [&](int n) {          // n = 0..3, a literal at every call site: FOUR instructions per wave and tile (the counted waits rely on it)
        if (n == 0) M4D_BGLDS(d_dst, oya[0], dya_b);
        else if (n == 1) M4D_BGLDS(d_dst + T1, oyb[0], dyb_b);
        else if (n == 2) M4D_BGLDS(d_dst + 1024, oya[1], dya_b);
        else M4D_BGLDS(d_dst + T1 + 1024, oyb[1], dyb_b);
    };
    auto dma_tile = [&](int stage, int64_t y0) {
        dma_prepare(stage, y0);
        dma_one(0); dma_one(1); dma_one(2); dma_one(3);
    };
    auto reg_tile = [&](int stage, int64_t y0) {      // ragged last tile: zero filled, synchronous
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + 512 * i;
            const int row = c >> 4, ch = c & 15;
            const int64_t y = y0 + row;
            const bool ok = y < p.LY;
            *reinterpret_cast<uint4*>(base + swz_off<256>(row, ch)) =
                ok ? *reinterpret_cast<const uint4*>(gya + y * p.ya_ls + ch * 8) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(base + T1 + swz_off<256>(row, ch)) =
                ok ? *reinterpret_cast<const uint4*>(gyb + y * p.yb_ls + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto load_stat = [&](int64_t y0) -> float {       // threads 0..63: lse, 64..127: delta; LDS position t & 63 holds row bwd_tr_stat(t & 63)
        if (!STAT_Y || t >= 128) return 0.f;
        const int64_t y = y0 + bwd_tr_stat(t & 63);
        if (y >= p.LY) return t < 64 ? INFINITY : 0.f;   // lse = +inf => probability exactly 0
        return t < 64 ? glse[y] : gdel[y];
    };

    const int64_t y_begin = p.ws ? (int64_t)blockIdx.y * p.y_chunk : 0;
    const int64_t y_end = p.ws ? (y_begin + p.y_chunk < p.LY ? y_begin + p.y_chunk : p.LY) : p.LY;
    // NST stages: tile it + NST - 1 is requested while tile it is computed (first version: two stages, vmcnt(0) in front of every tile).
    // Full tiles travel by DMA, four instructions per wave and tile; a ragged last tile of the key / query range is staged synchronously.
    const int NT = y_begin < y_end ? (int)((y_end - y_begin + YB - 1) / YB) : 0;
    const int NF = p.LY - y_begin >= 0 ? (int)((p.LY - y_begin) / YB < NT ? (p.LY - y_begin) / YB : NT) : 0;      // full (DMA) tiles of this chunk
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
        if (j < NF) dma_tile(j, y_begin + (int64_t)j * YB);
    float rstat = load_stat(y_begin);
    int stage = 0;
    for (int it = 0; it < NT; ++it) {
        const int64_t y0 = y_begin + (int64_t)it * YB;
        if (STAT_Y && t < 128) reinterpret_cast<float*>(smem + stage * STAGE + STAT_OFF)[t] = rstat;
        if (it < NF) {
            const int younger = NF - 1 - it;          // requested tiles behind this one: min(younger, NST - 2) x 4 instructions may stay in flight
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else reg_tile(stage, y0);
        __syncthreads();
        const bool next_dma = it + NST - 1 < NF;
        if (next_dma) dma_prepare(stage == 0 ? NST - 1 : stage - 1, y0 + (int64_t)(NST - 1) * YB);
        if (it + 1 < NT) rstat = load_stat(y0 + YB);
#define M4D_BDMA(N) do { if (next_dma) dma_one(N); __builtin_amdgcn_sched_barrier(0); } while (0)

        bf16x8 fb0, fb1, fb2, fb3;
        bf16x4 t0l, t0h, t1l, t1h, t2l, t2h, t3l, t3h;      // transposed fragments: two ds_read_b64_tr_b16 each (chunk C = immediate offset)
#define M4D_TR(F, C, DD) do { bwd_tr_read<(C) * 4096>(F##l, ta[0][DD]); bwd_tr_read<(C) * 4096>(F##h, ta[1][DD]); } while (0)
        bf16x8 pf[4];
        const float* st = reinterpret_cast<const float*>(smem + stage * STAGE + STAT_OFF);
#define M4D_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define M4D_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        // P / dS of 8 consecutive y (accumulator registers rb..rb+7 of sub-tile `sub`) -> one bf16 fragment
        auto elementwise = [&](const f32x16& sv, const f32x16& gv, int sub, int half) -> bf16x8 {
            const int rb = half * 8, yb = sub * 32 + half * 16 + 8 * hi;
            float lv[8], dv[8];
            if constexpr (STAT_Y) {
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(st + yb), l1 = *reinterpret_cast<const f32x4*>(st + yb + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { lv[j] = l0[j]; lv[4 + j] = l1[j]; }
                if constexpr (HAS_G) {
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(st + 64 + yb), d1 = *reinterpret_cast<const f32x4*>(st + 64 + yb + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { dv[j] = d0[j]; dv[4 + j] = d1[j]; }
                }
            }
            bf16x8 out;
            if constexpr (SMX == 1) {
                // single-issue v_fma_f32 / v_exp_f32 / v_mul_f32 as one volatile asm stream in a fixed order: beside the partner
                // wave's MFMAs a packed fp32 VALU costs more than its two scalar halves (MI355X_MICROARCH.md price list) and hipcc
                // SLP-packs plain C; its hazard recogniser does not see through inline asm, so no VALU consumes a v_exp_f32 result
                // closer than eight instructions behind it (gfx940+ trans-use hazard needs one)
                float x[8];
                // ... nor through the MFMA -> VALU read-after-write hazard: the S / G accumulators were written by MFMAs that
                // may have issued a few cycles ago, and a 16-pass MFMA result needs 18 wait states before a VALU may read it
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 15\n\ts_nop 3");
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (STAT_Y) asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(x[j]) : "v"(sv[rb + j]), "s"(p.sc), "v"(lv[j]));
                    else asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(x[j]) : "v"(sv[rb + j]), "s"(p.sc), "v"(lse_x));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %1" : "=v"(x[j]) : "v"(x[j]));
                if constexpr (HAS_G) {
                    float tt[8], nd = 0.f;
                    if constexpr (!STAT_Y) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(nd) : "s"(p.scale), "v"(delta_x));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if constexpr (STAT_Y) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(nd) : "s"(p.scale), "v"(dv[j]));
                        asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(tt[j]) : "v"(gv[rb + j]), "s"(p.scale), "v"(nd));
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(tt[j]));
                } else {
                    asm volatile("s_nop 1");
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float pv = x[j];
                    if constexpr (STAT_Y) {
                        if (!xvalid) pv = 0.f;
                    } else {
                        if (y0 + bwd_tr_stat(yb + j) >= p.LY) pv = 0.f;
                    }
                    out[j] = (T)pv;
                }
                return out;
            }
            // two elements per instruction wherever the ISA has a packed fp32 form (v_pk_fma_f32 / v_pk_mul_f32); only
            // the exp2 is scalar.  dS = P * (G * scale - delta * scale)
            const f32x2 sc2 = {p.sc, p.sc}, scale2 = {p.scale, p.scale};
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                f32x2 x = {sv[rb + j], sv[rb + j + 1]};
                f32x2 nl;
                if constexpr (STAT_Y) nl = f32x2{-lv[j], -lv[j + 1]};
                else nl = f32x2{-lse_x, -lse_x};
                x = __builtin_elementwise_fma(x, sc2, nl);
                f32x2 pv = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                if constexpr (STAT_Y) {
                    if (!xvalid) pv = f32x2{0.f, 0.f};
                } else {
                    if (y0 + bwd_tr_stat(yb + j) >= p.LY) pv[0] = 0.f;
                    if (y0 + bwd_tr_stat(yb + j + 1) >= p.LY) pv[1] = 0.f;
                }
                if constexpr (HAS_G) {
                    f32x2 nd;
                    if constexpr (STAT_Y) nd = f32x2{-dv[j], -dv[j + 1]} * scale2;
                    else nd = f32x2{-delta_x, -delta_x} * scale2;
                    const f32x2 gg = {gv[rb + j], gv[rb + j + 1]};
                    pv = pv * __builtin_elementwise_fma(gg, scale2, nd);
                }
                out[j] = (T)pv[0];
                out[j + 1] = (T)pv[1];
            }
            return out;
        };
        if constexpr (HAS_G) {
            // one 32-row sub-tile at a time (S and G accumulators of only one sub-tile are live: 32 VGPRs instead of 64);
            // 16 steps per sub-tile, the (S, G) fragment pair is read two kk ahead
#define M4D_SGS(B, KK, OFF, W) do { M4D_LGKM(W); mma32(B, xaf[KK], s1); if ((KK) + 2 < 8) M4D_DSR(B, ka[((KK) + 2) & 7], OFF); } while (0)
#define M4D_SGG(B, KK, OFF, W) do { M4D_LGKM(W); mma32(B, xbf[KK], g1); if ((KK) + 2 < 8) M4D_DSR(B, ka[((KK) + 2) & 7], OFF); } while (0)
#define M4D_H(N, ON) do { if (ON) M4D_BDMA(N); } while (0)
#define M4D_SUB(SO, GO, ON)                                                                                          \
    do {                                                                                                             \
        M4D_DSR(fb0, ka[0], SO); M4D_DSR(fb1, ka[0], GO); M4D_DSR(fb2, ka[1], SO); M4D_DSR(fb3, ka[1], GO);          \
        M4D_SGS(fb0, 0, SO, 3); M4D_H(0, ON); M4D_SGG(fb1, 0, GO, 3); M4D_SGS(fb2, 1, SO, 3); M4D_H(1, ON); M4D_SGG(fb3, 1, GO, 3);  \
        M4D_SGS(fb0, 2, SO, 3); M4D_H(2, ON); M4D_SGG(fb1, 2, GO, 3); M4D_SGS(fb2, 3, SO, 3); M4D_H(3, ON); M4D_SGG(fb3, 3, GO, 3);  \
        M4D_SGS(fb0, 4, SO, 3); M4D_SGG(fb1, 4, GO, 3); M4D_SGS(fb2, 5, SO, 3); M4D_SGG(fb3, 5, GO, 3);  \
        M4D_SGS(fb0, 6, SO, 3); M4D_SGG(fb1, 6, GO, 2); M4D_SGS(fb2, 7, SO, 1); M4D_SGG(fb3, 7, GO, 0);              \
    } while (0)
            {
                f32x16 s1, g1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s1[r] = 0.f; g1[r] = 0.f; }
                M4D_SUB(0, 16384, true);
                pf[0] = elementwise(s1, g1, 0, 0);
                pf[1] = elementwise(s1, g1, 0, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                f32x16 s1, g1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s1[r] = 0.f; g1[r] = 0.f; }
                M4D_SUB(8192, 24576, false);
                // the first four transposed fragments land under the elementwise arithmetic (dK pass: behind it — with the tile statistics
                // in registers as well the early request spills)
                if constexpr (!STAT_Y) { M4D_TR(t0, 0, 0); M4D_TR(t1, 0, 1); M4D_TR(t2, 0, 2); M4D_TR(t3, 0, 3); }
                pf[2] = elementwise(s1, g1, 1, 0);
                pf[3] = elementwise(s1, g1, 1, 1);
                if constexpr (STAT_Y) { M4D_TR(t0, 0, 0); M4D_TR(t1, 0, 1); M4D_TR(t2, 0, 2); M4D_TR(t3, 0, 3); }
            }
#undef M4D_SUB
#undef M4D_H
#undef M4D_SGG
#undef M4D_SGS
        } else {
            f32x16 s[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
#define M4D_QK(B, KK, SUB, OFF, W) do { M4D_LGKM(W); mma32(B, xaf[KK], s[SUB]); if ((KK) + 2 < 8) M4D_DSR(B, ka[((KK) + 2) & 7], OFF); } while (0)
            M4D_DSR(fb0, ka[0], 0); M4D_DSR(fb1, ka[0], 8192); M4D_DSR(fb2, ka[1], 0); M4D_DSR(fb3, ka[1], 8192);
            M4D_QK(fb0, 0, 0, 0, 3); M4D_BDMA(0); M4D_QK(fb1, 0, 1, 8192, 3); M4D_BDMA(1); M4D_QK(fb2, 1, 0, 0, 3); M4D_BDMA(2); M4D_QK(fb3, 1, 1, 8192, 3); M4D_BDMA(3);
            M4D_QK(fb0, 2, 0, 0, 3); M4D_QK(fb1, 2, 1, 8192, 3); M4D_QK(fb2, 3, 0, 0, 3); M4D_QK(fb3, 3, 1, 8192, 3);
            M4D_QK(fb0, 4, 0, 0, 3); M4D_QK(fb1, 4, 1, 8192, 3); M4D_QK(fb2, 5, 0, 0, 3); M4D_QK(fb3, 5, 1, 8192, 3);
            M4D_QK(fb0, 6, 0, 0, 3); M4D_QK(fb1, 6, 1, 8192, 2); M4D_QK(fb2, 7, 0, 0, 1); M4D_QK(fb3, 7, 1, 8192, 0);
#undef M4D_QK
            M4D_TR(t0, 0, 0); M4D_TR(t1, 0, 1); M4D_TR(t2, 0, 2); M4D_TR(t3, 0, 3);
#pragma unroll
            for (int c = 0; c < 4; ++c) pf[c] = elementwise(s[c >> 1], s[c >> 1], c >> 1, c & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- acc^T += (Y tile)^T . pf ----  16 (c, d) steps, four fragments = eight transposing reads ahead
#define M4D_PV(F, C, DD, W) do { M4D_LGKM(W); mma32(bwd_tr_join(F##l, F##h), pf[C], acc[DD]); } while (0)
        M4D_PV(t0, 0, 0, 6); M4D_TR(t0, 1, 0); M4D_PV(t1, 0, 1, 6); M4D_TR(t1, 1, 1); M4D_PV(t2, 0, 2, 6); M4D_TR(t2, 1, 2); M4D_PV(t3, 0, 3, 6); M4D_TR(t3, 1, 3);
        M4D_PV(t0, 1, 0, 6); M4D_TR(t0, 2, 0); M4D_PV(t1, 1, 1, 6); M4D_TR(t1, 2, 1); M4D_PV(t2, 1, 2, 6); M4D_TR(t2, 2, 2); M4D_PV(t3, 1, 3, 6); M4D_TR(t3, 2, 3);
        M4D_PV(t0, 2, 0, 6); M4D_TR(t0, 3, 0); M4D_PV(t1, 2, 1, 6); M4D_TR(t1, 3, 1); M4D_PV(t2, 2, 2, 6); M4D_TR(t2, 3, 2); M4D_PV(t3, 2, 3, 6); M4D_TR(t3, 3, 3);
        M4D_PV(t0, 3, 0, 6); M4D_PV(t1, 3, 1, 4); M4D_PV(t2, 3, 2, 2); M4D_PV(t3, 3, 3, 0);
#undef M4D_PV
#undef M4D_TR
#undef M4D_LGKM
#undef M4D_DSR
#undef M4D_BDMA
        const unsigned dl = stage == NST - 1 ? (unsigned)(-(NST - 1) * STAGE) : (unsigned)STAGE;
        stage = stage == NST - 1 ? 0 : stage + 1;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] += dl;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) ta[jj][dd] += dl;
    }
#undef M4D_BGLDS

    if (p.ws) {
        if (xrow < p.LXs && xvalid) {
            float* w = p.ws + ((int64_t)b * p.LXs + xrow) * ((int64_t)p.heads * D) + (int64_t)h * D + hi * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(w + d * 32 + rq * 8 + e, acc[d][rq * 4 + e]);
        }
    } else if (xrow < p.LXs) {
        T* oa = (T*)p.out_a + b * p.oa_bs + xrow * p.oa_ls + (int64_t)h * D + hi * 4;
        // accumulate mode: ALL sixteen previous quads are requested before the first store (interleaved load / add / store, every
        // load is waited for alone behind the store in front of it: sixteen serial round trips per workgroup)
        f32x4 prev[4][4];
        if (p.accumulate) {
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
                if (p.accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += prev[d][rq][e];
                }
                store4(dst, v);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fused dK / dV pass (round 4): S and P are computed ONCE per (key, query) tile — 4 matmul units (S, G, dV, dK) instead of the 3 + 2 of
// the separate dK and dV passes, 4 Y tiles through LDS per step instead of 3 + 2, one exp per score instead of two.
//
// The obstacle was registers (two 32 x 128 fp32 accumulators + K and V fragments + S and G do not fit a 256-register wave).  Here the
// work of 32 key rows is ROLE-SPLIT over a PAIR of waves that share a SIMD (waves w and w + 4):
//   P-wave  (w < 4):  X = K rows   S^T = Q K^T  -> P = exp2(S sc - lse_y)  -> mailbox     dV^T += dO^T P^T
//   dS-wave (w >= 4): X = V rows   G^T = dO V^T     dS = P (G - delta_y) scale            dK^T += Q^T dS^T
// Each wave runs 2 matmul units and ONE 64-register accumulator — the shape of the dV pass — so both fit 2 waves per SIMD without
// AGPRs.  P crosses from the P-wave to the dS-wave through a lane-private LDS mailbox (64 bytes per lane: the S and G accumulators of
// the two waves have the same lane / register layout, so lane l of the dS-wave needs exactly what lane l of the P-wave holds), rounded
// to bf16 — the value the dV product uses anyway.
//
// What bounds these passes is not the MFMA pipe (timing ablations of the first version, tools/abl_bkv.sh: without ANY MFMA the pass
// was 3 % faster) but the serial chain of a tile: barrier, tile request, fragment latency, elementwise stream, barrier.  So:
//   * the pair synchronises through an LDS FLAG, not a workgroup barrier: the tile is cut in two 32-query halves, the P-wave posts
//     P(half) + flag as soon as it has it, the dS-wave (which computed G meanwhile) turns it into dS(half) and accumulates while the
//     P-wave is already in the next half's S MFMAs / exp stream — on the shared SIMD one wave of the pair streams MFMAs while the other
//     runs VALU, without either of them waiting for the six other waves;
//   * ONE workgroup barrier per tile (stage hand-over); the tile request is 8 x (s_mov m0 + global_load_lds) with scalar bases and
//     per-lane 32-bit offsets computed once (no 64-bit VALU address arithmetic in the loop).
// 128 key rows per workgroup (4 pairs x 32), 64-query Y tiles of (Q, dO, Q^T, dO^T) in two 64.5 KiB stages.
#ifndef BKV_NST
#define BKV_NST 4      // Y-tile stages of the fused dK / dV pass (tile i + NST - 1 is requested while tile i is computed)
#endif
#ifndef BKV_ABL
#define BKV_ABL 0      // side builds (tools/side_lib.sh): 1 no elementwise arithmetic, 2 no S / G MFMAs, 4 no accumulate MFMAs, 8 no mailbox /
#endif                 // flags, 16 no tile DMA, 32 no fragment reads — timing only, results wrong
template <int OFF> M4D_DEV void bkv_dsr(bf16x8& dst, unsigned addr) {
    if constexpr (BKV_ABL & 32) return;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> M4D_DEV void bkv_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512, 2) void attn_bwd_kv128_kernel(BwdArgs p) {
    typedef bf16_t T;
    constexpr int D = 128, YB = 64, XB = 128;
    constexpr int TQ = 0, TDO = 16384, STAT_OFF = 32768, STAGE = STAT_OFF + 512, NST = BKV_NST, MAIL = NST * STAGE, FLAGS = MAIL + 16384;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NST * STAGE + 4 pairs x 4 KiB mailbox + flags

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

    bf16x8 xf[8];                                   // K rows (P-wave) or V rows (dS-wave)
    {
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

    // per-lane fragment addresses in the CURRENT stage, the role's tiles folded in: row-major tile = Q (P-wave) / dO (dS-wave),
    // transposed tile = dO^T (P-wave) / Q^T (dS-wave)
    // The transposed operand of the accumulate product (dO^T for dV, Q^T for dK) is read out of the SAME row-major tile with
    // ds_read_b64_tr_b16 (bwd_tr_row / bwd_tr_addr above): no transposed copies in HBM, half the tile bytes through the DMA path.
    unsigned ka[8], ta[2][4];
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    {
        const unsigned rbase = lds0 + (ds_role ? TDO : TQ), tbase = lds0 + (ds_role ? TQ : TDO);
        const int kr = bwd_tr_row(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] = rbase + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) ta[jj][dd] = tbase + bwd_tr_addr(li, hi, jj, dd);
    }
    const unsigned mail = lds0 + MAIL + pair * 4096 + lane * 16;
    const unsigned flag = lds0 + FLAGS + pair * 8;           // two counters per pair: P of half 0 / half 1 posted up to tile n
    if (!ds_role && lane == 0) {
        volatile unsigned* fz = reinterpret_cast<volatile unsigned*>(smem + FLAGS + pair * 8);
        fz[0] = 0u; fz[1] = 0u;
    }

    // ---- tile request: scalar bases + per-lane 32-bit offsets (row-major tiles: 4 rows x 256 B per instruction; transposed: 8 rows x 128 B)
    const int k_r = lane >> 4, k_lc0 = lane & 15;
    const T* gq = (const T*)p.ya + b * p.ya_bs + (int64_t)h * D;
    const T* gdo = (const T*)p.yb + b * p.yb_bs + (int64_t)h * D;
    const float* glse = p.lse + ((int64_t)b * p.heads + h) * p.Lq;
    const float* gdel = p.delta + ((int64_t)b * p.heads + h) * p.Lq;
    unsigned oq[2], odo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int blk = wave * 2 + i;
        const int row = blk * 4 + k_r;
        oq[i] = (unsigned)((row * p.ya_ls + (k_lc0 ^ (row & 15)) * 8) * 2);
        odo[i] = (unsigned)((row * p.yb_ls + (k_lc0 ^ (row & 15)) * 8) * 2);
    }
    const unsigned ostat = (unsigned)bwd_tr_stat(lane) * 4u;      // LDS position `lane` of a tile's statistics holds query bwd_tr_stat(lane)
    auto uniform_ptr = [](const char* q) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi2 << 32) | lo);
    };
#define BKV_GLDS(DST, VOFF, SRC) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(DST), "v"(VOFF), "s"(SRC) : "memory", "m0")
    // the tile request in two parts so that the loop can spread its instructions over the first MFMAs of a tile (issued back to back
    // at the top of the tile, the 8 waves' 68 requests queue on the CU's one texture-address unit with every wave stuck behind its own)
    const char *dq_b = nullptr, *ddo_b = nullptr, *dst_b = nullptr;
    unsigned d_dst = 0, d_sdst = 0;
    auto dma_prepare = [&](int stage, int64_t y0) {
        dq_b = uniform_ptr((const char*)(gq + y0 * p.ya_ls));
        ddo_b = uniform_ptr((const char*)(gdo + y0 * p.yb_ls));
        dst_b = uniform_ptr((const char*)((wave & 1 ? gdel : glse) + y0));
        d_dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + wave * 2048);
        d_sdst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + STAT_OFF + (wave & 1) * 256);
    };
    auto dma_one = [&](int n) {          // n = 0..4, a literal at every call site: FIVE instructions per wave and tile (the counted waits rely on it)
        if constexpr (BKV_ABL & 16) return;
        if (n == 0) BKV_GLDS(d_dst + TQ, oq[0], dq_b);
        else if (n == 1) BKV_GLDS(d_dst + TDO, odo[0], ddo_b);
        else if (n == 2) BKV_GLDS(d_dst + TQ + 1024, oq[1], dq_b);
        else if (n == 3) BKV_GLDS(d_dst + TDO + 1024, odo[1], ddo_b);
        else      // the tile's statistics travel the same way (even waves: lse, odd waves: delta of the 64 queries, 4 bytes per lane, in the
                  // register order of the S / G accumulators; waves 2..7 repeat the copy of waves 0 / 1 so that every wave counts the same)
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" :: "s"(d_sdst), "v"(ostat), "s"(dst_b) : "memory", "m0");
    };
    auto dma_tile = [&](int stage, int64_t y0) {
        dma_prepare(stage, y0);
        dma_one(0); dma_one(1); dma_one(2); dma_one(3); dma_one(4);
    };
    auto reg_tile = [&](int stage, int64_t y0) {      // ragged last tile: zero filled, synchronous
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + 512 * i;
            {
                const int row = c >> 4, ch = c & 15;
                const int64_t y = y0 + row;
                const bool ok = y < p.LY;
                *reinterpret_cast<uint4*>(base + TQ + swz_off<256>(row, ch)) =
                    ok ? *reinterpret_cast<const uint4*>(gq + y * p.ya_ls + ch * 8) : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(base + TDO + swz_off<256>(row, ch)) =
                    ok ? *reinterpret_cast<const uint4*>(gdo + y * p.yb_ls + ch * 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto reg_stat = [&](int stage, int64_t y0) {      // ragged last tile: threads 0..63 lse, 64..127 delta of row y0 + (t & 63)
        if (t >= 128) return;
        const int64_t y = y0 + bwd_tr_stat(t & 63);
        float v = t < 64 ? INFINITY : 0.f;               // lse = +inf => probability exactly 0
        if (y < p.LY) v = t < 64 ? glse[y] : gdel[y];
        reinterpret_cast<float*>(smem + stage * STAGE + STAT_OFF)[t] = v;
    };
    // (An L2 "touch" of the tile three or six steps ahead — one dword load per thread and line, issued behind the tile request and left in
    // flight by a counted wait — was measured here and changes nothing: the request one tile ahead is not what the pass waits for.)
    // dS-wave: spin until the partner P-wave has posted P of (tile, half): the flag word holds the number of tiles posted so far
    auto wait_flag = [&](int half, unsigned seq) {
        if constexpr (BKV_ABL & 8) return;
        for (int spin = 0; spin < (1 << 22); ++spin) {      // (bounded: a lost flag must end in wrong numbers the tests catch, never in a hung GPU)
            unsigned v;
            if (half) asm volatile("ds_read_b32 %0, %1 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag) : "memory");
            else asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag) : "memory");
            if (__builtin_amdgcn_readfirstlane(v) >= seq) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // NST stages: tile it + NST - 1 is requested while tile it is computed, so a request has NST - 2 whole tiles to land (round 4, first
    // version: two stages and vmcnt(0) in front of every tile — the DMA round trip was exposed once per tile, 5 ms of a 25 ms pass in the
    // timing ablations).  Full tiles travel by DMA (five instructions per wave and tile: the counted waits below); a ragged last tile is
    // staged synchronously through registers.
    const int NT = (int)((p.LY + YB - 1) / YB), NF = (int)(p.LY / YB);      // tiles, full (DMA) tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
        if (j < NF) dma_tile(j, (int64_t)j * YB);
    int stage = 0;
    for (int it = 0; it < NT; ++it) {
        const int64_t y0 = (int64_t)it * YB;
        if (it < NF) {
            const int younger = NF - 1 - it;          // requested tiles behind this one: min(younger, NST - 2) x 5 instructions may stay in flight
            if (younger >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(5 * (NST - 2)) : "memory");
            else if (younger == 1 && NST > 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else { reg_tile(stage, y0); reg_stat(stage, y0); }
        __syncthreads();                                                       // stage landed; stage it - 1 and the mailbox are free
        const bool next_dma = it + NST - 1 < NF;
        if (next_dma) dma_prepare(stage == 0 ? NST - 1 : stage - 1, y0 + (int64_t)(NST - 1) * YB);
        const float* st = reinterpret_cast<const float*>(smem + stage * STAGE + STAT_OFF) + (ds_role ? 64 : 0);
        bf16x8 fb0, fb1, fb2, fb3;
        bf16x8 pf[4];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define BKV_MMA0(Bf, ACC) do { if constexpr (!(BKV_ABL & 2)) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf, xf[0], zero, 0, 0, 0); else ACC = zero; } while (0)
#define BKV_MMA(Bf, KK, ACC) do { if constexpr (!(BKV_ABL & 2)) mma32(Bf, xf[KK], ACC); } while (0)
        // 8 MFMAs of one 32-query half (rows OFF.. of the row-major tile), fragments four k-steps ahead, first k-step with C = 0
#define BKV_DMA(N, ON) do { if (ON) { if (next_dma) dma_one(N); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define BKV_HALF(ACC, OFF, ON)                                                                                         \
    do {                                                                                                               \
        bkv_dsr<OFF>(fb0, ka[0]); bkv_dsr<OFF>(fb1, ka[1]); bkv_dsr<OFF>(fb2, ka[2]); bkv_dsr<OFF>(fb3, ka[3]);         \
        bkv_lgkm<3>(); BKV_MMA0(fb0, ACC); bkv_dsr<OFF>(fb0, ka[4]); BKV_DMA(0, ON);                                    \
        bkv_lgkm<3>(); BKV_MMA(fb1, 1, ACC); bkv_dsr<OFF>(fb1, ka[5]); BKV_DMA(1, ON);                                  \
        bkv_lgkm<3>(); BKV_MMA(fb2, 2, ACC); bkv_dsr<OFF>(fb2, ka[6]); BKV_DMA(2, ON);                                  \
        bkv_lgkm<3>(); BKV_MMA(fb3, 3, ACC); bkv_dsr<OFF>(fb3, ka[7]); BKV_DMA(3, ON);                                  \
        bkv_lgkm<3>(); BKV_MMA(fb0, 4, ACC); BKV_DMA(4, ON); bkv_lgkm<2>(); BKV_MMA(fb1, 5, ACC);                       \
        bkv_lgkm<1>(); BKV_MMA(fb2, 6, ACC); bkv_lgkm<0>(); BKV_MMA(fb3, 7, ACC);                                       \
    } while (0)
        // acc^T += (Y tile)^T . pf for the two 16-query chunks C0, C0 + 1 of one half: 8 MFMAs, each fragment = two transposing reads
        // (bwd_tr_addr: read jj of d-block DD; the chunk is the immediate offset), four fragments = eight reads ahead
        bf16x4 t0l, t0h, t1l, t1h, t2l, t2h, t3l, t3h;
#define BKV_TR(F, C, DD) do { if constexpr (!(BKV_ABL & 32)) { bwd_tr_read<(C) * 4096>(F##l, ta[0][DD]); bwd_tr_read<(C) * 4096>(F##h, ta[1][DD]); } } while (0)
#define BKV_PV(F, C, DD, W) do { bkv_lgkm<W>(); if constexpr (!(BKV_ABL & 4)) mma32(bwd_tr_join(F##l, F##h), pf[C], acc[DD]); } while (0)
#define BKV_ACC_HALF(C0)                                                                                               \
    do {                                                                                                               \
        BKV_TR(t0, C0, 0); BKV_TR(t1, C0, 1); BKV_TR(t2, C0, 2); BKV_TR(t3, C0, 3);                                     \
        BKV_PV(t0, C0, 0, 6); BKV_TR(t0, C0 + 1, 0);                                                                    \
        BKV_PV(t1, C0, 1, 6); BKV_TR(t1, C0 + 1, 1);                                                                    \
        BKV_PV(t2, C0, 2, 6); BKV_TR(t2, C0 + 1, 2);                                                                    \
        BKV_PV(t3, C0, 3, 6); BKV_TR(t3, C0 + 1, 3);                                                                    \
        BKV_PV(t0, C0 + 1, 0, 6); BKV_PV(t1, C0 + 1, 1, 4); BKV_PV(t2, C0 + 1, 2, 2); BKV_PV(t3, C0 + 1, 3, 0);         \
    } while (0)
        if (!ds_role) {
            // ================= P-wave: S(half) -> P(half) -> mailbox + flag, twice; then dV^T += dO^T P^T =================
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x16 sv;
                if (half == 0) BKV_HALF(sv, 0, true); else BKV_HALF(sv, 8192, false);
                // (hipcc's hazard recogniser does not look inside inline asm: a 16-pass MFMA result needs 18 wait states before a VALU reads it)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 15\n\ts_nop 3");
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = half * 2 + q, rb = q * 8, yb = half * 32 + q * 16 + 8 * hi;
                    if constexpr (BKV_ABL & 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[c][j] = (T)sv[rb + j];
                    } else {
                        // P = exp2(S sc - lse_y): single-issue v_fma_f32 / v_exp_f32 as one volatile stream (packed fp32 VALU costs more than
                        // its two halves beside the partner wave's MFMAs); a VALU consuming a v_exp_f32 result sits eight instructions behind it
                        const f32x4 l0 = *reinterpret_cast<const f32x4*>(st + yb), l1 = *reinterpret_cast<const f32x4*>(st + yb + 4);
                        float lv[8], x[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { lv[j] = l0[j]; lv[4 + j] = l1[j]; }
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(x[j]) : "v"(sv[rb + j]), "s"(p.sc), "v"(lv[j]));
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %1" : "=v"(x[j]) : "v"(x[j]));
                        asm volatile("s_nop 1");
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[c][j] = (T)(xvalid ? x[j] : 0.f);
                    }
                }
                if constexpr (!(BKV_ABL & 8)) {
                    // mailbox: lane-private 2 x 16 bytes per half, [chunk][lane] so that a wave's store covers 1 KiB contiguously; then the flag
                    if (half == 0) {
                        asm volatile("ds_write_b128 %0, %1" :: "v"(mail), "v"(pf[0]) : "memory");
                        asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(mail), "v"(pf[1]) : "memory");
                    } else {
                        asm volatile("ds_write_b128 %0, %1 offset:2048" :: "v"(mail), "v"(pf[2]) : "memory");
                        asm volatile("ds_write_b128 %0, %1 offset:3072" :: "v"(mail), "v"(pf[3]) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned seq = (unsigned)it + 1u;
                    if (half == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(flag), "v"(seq) : "memory");
                    else asm volatile("ds_write_b32 %0, %1 offset:4" :: "v"(flag), "v"(seq) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            BKV_ACC_HALF(0);
            BKV_ACC_HALF(2);
        } else {
            // ================= dS-wave: G (both halves), then per half: wait for P, dS = P (G scale - delta scale), dK^T += Q^T dS^T =================
            f32x16 g0, g1;
            BKV_HALF(g0, 0, true);
            BKV_HALF(g1, 8192, false);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 3");
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x16& gv = half == 0 ? g0 : g1;
                bf16x8 pm[2];
                wait_flag(half, (unsigned)it + 1u);
                if constexpr (!(BKV_ABL & 8)) {
                    if (half == 0) { bkv_dsr<0>(pm[0], mail); bkv_dsr<1024>(pm[1], mail); }
                    else { bkv_dsr<2048>(pm[0], mail); bkv_dsr<3072>(pm[1], mail); }
                } else { pm[0] = xf[0]; pm[1] = xf[1]; }
                bkv_lgkm<0>();
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = half * 2 + q, rb = q * 8, yb = half * 32 + q * 16 + 8 * hi;
                    if constexpr (BKV_ABL & 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[c][j] = (T)gv[rb + j];
                    } else {
                        const f32x4 d0 = *reinterpret_cast<const f32x4*>(st + yb), d1 = *reinterpret_cast<const f32x4*>(st + yb + 4);
                        float dv[8], tt[8], x[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { dv[j] = d0[j]; dv[4 + j] = d1[j]; }
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(tt[j]) : "v"(gv[rb + j]), "v"(dv[j]));      // (the softmax scale is applied to dK once, at the end)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float pj = (float)pm[q][j];
                            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[j]) : "v"(pj), "v"(tt[j]));
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[c][j] = (T)x[j];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (half == 0) BKV_ACC_HALF(0); else BKV_ACC_HALF(2);
            }
        }
#undef BKV_ACC_HALF
#undef BKV_PV
#undef BKV_TR
#undef BKV_HALF
#undef BKV_DMA
#undef BKV_MMA
#undef BKV_MMA0
        const unsigned dl = stage == NST - 1 ? (unsigned)(-(NST - 1) * STAGE) : (unsigned)STAGE;
        stage = stage == NST - 1 ? 0 : stage + 1;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] += dl;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) ta[jj][dd] += dl;
    }
#undef BKV_GLDS

    if (xrow < p.LXs) {
        T* oa = ds_role ? (T*)p.out_a + b * p.oa_bs + xrow * p.oa_ls : (T*)p.out_b + b * p.ob_bs + xrow * p.ob_ls;
        oa += (int64_t)h * D + hi * 4;
        const float osc = ds_role ? p.scale : 1.f;       // dK = scale * sum (P (G - delta))^T Q
        // accumulate mode: ALL sixteen previous quads are requested before the first store (interleaved load / add / store, every
        // load is waited for alone behind the store in front of it: sixteen serial round trips per workgroup)
        f32x4 prev[4][4];
        if (p.accumulate) {
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
                for (int e = 0; e < 4; ++e) v[e] = acc[d][rq * 4 + e] * osc;
                T* dst = oa + d * 32 + rq * 8;
                if (p.accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += prev[d][rq][e];
                }
                store4(dst, v);
            }
    }
}

// (A wide form — four 64-row waves, one per SIMD, accumulators in AGPRs, every fragment feeding two MFMAs — was built and measured in
// round 4: 43.6 ms for the whole backward against 38.5 ms with eight 32-row waves, profiles/r04_ab_attn_bwd.log; a lone wave per SIMD runs
// its fragment waits, MFMAs and elementwise stream strictly one after the other.  Removed.)
inline int launch_bwd_kv128(const BwdArgs& p, hipStream_t st) {
    constexpr int LDS = BKV_NST * (32768 + 512) + 16384 + 64;
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)attn_bwd_kv128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        configured.mark();
    }
    dim3 grid((unsigned)((int64_t)p.nx_tiles * p.heads * p.B));
    hipLaunchKernelGGL(attn_bwd_kv128_kernel, grid, dim3(512), LDS, st, p);
    return 0;
}

template <int MODE>
int launch_bwd128(const BwdArgs& p, hipStream_t st, int nsplit) {
    M4D_ENV_ONCE(smx, "M4D_ATTN_BWD_SMX", 1);
    constexpr int LDS = 4 * (32768 + (MODE == BWD_DQ ? 0 : 512));
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)attn_bwd128_kernel<MODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return -3;
        if (hipFuncSetAttribute((const void*)attn_bwd128_kernel<MODE, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        configured.mark();
    }
    dim3 grid((unsigned)((int64_t)p.nx_tiles * p.heads * p.B), (unsigned)nsplit), block(512);
    if (smx == 1) hipLaunchKernelGGL((attn_bwd128_kernel<MODE, 1>), grid, block, LDS, st, p);
    else hipLaunchKernelGGL((attn_bwd128_kernel<MODE, 0>), grid, block, LDS, st, p);
    return 0;
}
