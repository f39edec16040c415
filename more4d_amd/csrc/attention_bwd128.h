// Production flash-attention backward for bf16, head_dim 128 (the Wan2.1 DiT's only training shape).
// Same math as attn_bwd_kernel (attention_bwd.hip) split into THREE single-accumulator passes so that every pass has
// the forward kernel's shape (attention.hip: attn128_kernel) and inherits its measured structure:
//   * 8 waves x 32 X-rows per workgroup share each Y tile (two waves per SIMD, <= 256 VGPRs each);
//   * Y tiles arrive by global->LDS DMA into two stages, ONE barrier per tile, swizzle applied on the source address;
//   * LDS fragment reads are hand-pipelined (inline asm ds_read_b128 + counted lgkmcnt), exp2 is the raw v_exp_f32.
//   DQ:  X = (Q, dO)  Y = (K, V, K^T)     dS = P (G - delta_x) scale        dQ^T += K^T  dS^T      3 matmuls
//   DK:  X = (K, V)   Y = (Q, dO, Q^T)    dS = P (G - delta_y) scale        dK^T += Q^T  dS^T      3 matmuls
//   DV:  X = (K)      Y = (Q, dO^T)       P  = exp2(S sc - lse_y)           dV^T += dO^T P^T       2 matmuls
// (8 matmul units instead of the fused two-pass kernel's 7, for twice the occupancy and no 512-VGPR waves.)
#pragma once

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

enum { BWD_DQ = 0, BWD_DK = 1, BWD_DV = 2 };

template <int MODE, int SMX>   // SMX: 1 = single-issue fp32 VALU forms in the elementwise step (default), 0 = packed v_pk_*_f32 (A/B)
__global__ __launch_bounds__(512, 2) void attn_bwd128_kernel(BwdArgs p) {
    typedef bf16_t T;
    constexpr int D = 128, YB = 64, XB = 256;
    constexpr bool HAS_G = MODE != BWD_DV;
    constexpr bool STAT_Y = MODE != BWD_DQ;
    constexpr int T1 = 16384;                          // second row-major tile (G operand)
    constexpr int T2 = HAS_G ? 32768 : 16384;          // transposed tile (accumulate operand)
    constexpr int STAT_OFF = T2 + 16384;
    constexpr int STAGE = STAT_OFF + (STAT_Y ? 512 : 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * STAGE

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
    unsigned ka[8], va[4];
    {
        const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
        const int kr = perm23(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] = lds0 + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) va[c] = lds0 + T2 + li * 128 + (((c * 2 + hi) ^ ((li >> 1) & 7)) << 4);
    }
    const int k_r = lane >> 4, k_lc0 = lane & 15;     // row-major tile DMA: 4 rows x 256 B per instruction
    const int v_r = lane >> 3, v_pc = lane & 7;       // transposed tile DMA: 8 rows x 128 B per instruction

    const T* gya = (const T*)p.ya + b * p.ya_bs + (int64_t)h * D;
    const T* gyb = HAS_G ? (const T*)p.yb + b * p.yb_bs + (int64_t)h * D : nullptr;
    const T* gyt = (const T*)p.yat + b * p.yat_bs + (int64_t)h * D * p.yat_ls;
    const float* glse = p.lse + ((int64_t)b * p.heads + h) * p.Lq;
    const float* gdel = p.delta + ((int64_t)b * p.heads + h) * p.Lq;

    auto dma_tile = [&](int stage, int64_t y0) {
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = wave * 2 + i;
            const int row = blk * 4 + k_r;
            const int lc = k_lc0 ^ (row & 15);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gya + (y0 + row) * p.ya_ls + lc * 8),
                                             (LDS_AS void*)(base + blk * 1024), 16, 0, 0);
            if constexpr (HAS_G)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gyb + (y0 + row) * p.yb_ls + lc * 8),
                                                 (LDS_AS void*)(base + T1 + blk * 1024), 16, 0, 0);
            const int trow = blk * 8 + v_r;
            const int tlc = v_pc ^ ((trow >> 1) & 7);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gyt + trow * p.yat_ls + y0 + tlc * 8),
                                             (LDS_AS void*)(base + T2 + blk * 1024), 16, 0, 0);
        }
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
                *reinterpret_cast<uint4*>(base + swz_off<256>(row, ch)) =
                    ok ? *reinterpret_cast<const uint4*>(gya + y * p.ya_ls + ch * 8) : make_uint4(0, 0, 0, 0);
                if constexpr (HAS_G)
                    *reinterpret_cast<uint4*>(base + T1 + swz_off<256>(row, ch)) =
                        ok ? *reinterpret_cast<const uint4*>(gyb + y * p.yb_ls + ch * 8) : make_uint4(0, 0, 0, 0);
            }
            {
                const int row = c >> 3, ch = c & 7;
                const int64_t y = y0 + ch * 8;
                const T* src = gyt + row * p.yat_ls + y;
                union { uint4 u; T e[8]; } tmp;
                tmp.u = make_uint4(0, 0, 0, 0);
                if (y + 8 <= p.LY) tmp.u = *reinterpret_cast<const uint4*>(src);
                else if (y < p.LY) {
                    for (int j = 0; j < 8; ++j)
                        if (y + j < p.LY) tmp.e[j] = src[j];
                }
                *reinterpret_cast<uint4*>(base + T2 + swz_off<128>(row, ch)) = tmp.u;
            }
        }
    };
    auto load_stat = [&](int64_t y0) -> float {       // threads 0..63: lse, 64..127: delta of row y0 + (t & 63)
        if (!STAT_Y || t >= 128) return 0.f;
        const int64_t y = y0 + (t & 63);
        if (y >= p.LY) return t < 64 ? INFINITY : 0.f;   // lse = +inf => probability exactly 0
        return t < 64 ? glse[y] : gdel[y];
    };

    const int64_t y_begin = p.ws ? (int64_t)blockIdx.y * p.y_chunk : 0;
    const int64_t y_end = p.ws ? (y_begin + p.y_chunk < p.LY ? y_begin + p.y_chunk : p.LY) : p.LY;
    bool cur_dma = y_begin + YB <= p.LY;
    if (cur_dma && y_begin < y_end) dma_tile(0, y_begin);
    float rstat = load_stat(y_begin);
    int it = 0;
    for (int64_t y0 = y_begin; y0 < y_end; y0 += YB, ++it) {
        const int stage = it & 1;
        if (STAT_Y && t < 128) reinterpret_cast<float*>(smem + stage * STAGE + STAT_OFF)[t] = rstat;
        if (cur_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else reg_tile(stage, y0);
        __syncthreads();
        const int64_t ny0 = y0 + YB;
        bool next_dma = false;
        if (ny0 < y_end) {
            next_dma = ny0 + YB <= p.LY;
            if (next_dma) dma_tile(stage ^ 1, ny0);
            rstat = load_stat(ny0);
        }

        bf16x8 fb0, fb1, fb2, fb3;
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
                        if (y0 + yb + j >= p.LY) pv = 0.f;
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
                    if (y0 + yb + j >= p.LY) pv[0] = 0.f;
                    if (y0 + yb + j + 1 >= p.LY) pv[1] = 0.f;
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
#define M4D_SUB(SO, GO)                                                                                              \
    do {                                                                                                             \
        M4D_DSR(fb0, ka[0], SO); M4D_DSR(fb1, ka[0], GO); M4D_DSR(fb2, ka[1], SO); M4D_DSR(fb3, ka[1], GO);          \
        M4D_SGS(fb0, 0, SO, 3); M4D_SGG(fb1, 0, GO, 3); M4D_SGS(fb2, 1, SO, 3); M4D_SGG(fb3, 1, GO, 3);              \
        M4D_SGS(fb0, 2, SO, 3); M4D_SGG(fb1, 2, GO, 3); M4D_SGS(fb2, 3, SO, 3); M4D_SGG(fb3, 3, GO, 3);              \
        M4D_SGS(fb0, 4, SO, 3); M4D_SGG(fb1, 4, GO, 3); M4D_SGS(fb2, 5, SO, 3); M4D_SGG(fb3, 5, GO, 3);              \
        M4D_SGS(fb0, 6, SO, 3); M4D_SGG(fb1, 6, GO, 2); M4D_SGS(fb2, 7, SO, 1); M4D_SGG(fb3, 7, GO, 0);              \
    } while (0)
            {
                f32x16 s1, g1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s1[r] = 0.f; g1[r] = 0.f; }
                M4D_SUB(0, 16384);
                pf[0] = elementwise(s1, g1, 0, 0);
                pf[1] = elementwise(s1, g1, 0, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                f32x16 s1, g1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s1[r] = 0.f; g1[r] = 0.f; }
                M4D_SUB(8192, 24576);
                // the first four transposed-tile fragments land under the elementwise arithmetic
                M4D_DSR(fb0, va[0], 0); M4D_DSR(fb1, va[0], 4096); M4D_DSR(fb2, va[0], 8192); M4D_DSR(fb3, va[0], 12288);
                pf[2] = elementwise(s1, g1, 1, 0);
                pf[3] = elementwise(s1, g1, 1, 1);
            }
#undef M4D_SUB
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
            M4D_QK(fb0, 0, 0, 0, 3); M4D_QK(fb1, 0, 1, 8192, 3); M4D_QK(fb2, 1, 0, 0, 3); M4D_QK(fb3, 1, 1, 8192, 3);
            M4D_QK(fb0, 2, 0, 0, 3); M4D_QK(fb1, 2, 1, 8192, 3); M4D_QK(fb2, 3, 0, 0, 3); M4D_QK(fb3, 3, 1, 8192, 3);
            M4D_QK(fb0, 4, 0, 0, 3); M4D_QK(fb1, 4, 1, 8192, 3); M4D_QK(fb2, 5, 0, 0, 3); M4D_QK(fb3, 5, 1, 8192, 3);
            M4D_QK(fb0, 6, 0, 0, 3); M4D_QK(fb1, 6, 1, 8192, 2); M4D_QK(fb2, 7, 0, 0, 1); M4D_QK(fb3, 7, 1, 8192, 0);
#undef M4D_QK
            M4D_DSR(fb0, va[0], 0); M4D_DSR(fb1, va[0], 4096); M4D_DSR(fb2, va[0], 8192); M4D_DSR(fb3, va[0], 12288);
#pragma unroll
            for (int c = 0; c < 4; ++c) pf[c] = elementwise(s[c >> 1], s[c >> 1], c >> 1, c & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- acc^T += (transposed Y tile) . pf ----  16 (c, d) steps, fragments 4 steps ahead
#define M4D_PV(B, C, DD, OFF, W) do { M4D_LGKM(W); mma32(B, pf[C], acc[DD]); if ((C) + 1 < 4) M4D_DSR(B, va[((C) + 1) & 3], OFF); } while (0)
        M4D_PV(fb0, 0, 0, 0, 3); M4D_PV(fb1, 0, 1, 4096, 3); M4D_PV(fb2, 0, 2, 8192, 3); M4D_PV(fb3, 0, 3, 12288, 3);
        M4D_PV(fb0, 1, 0, 0, 3); M4D_PV(fb1, 1, 1, 4096, 3); M4D_PV(fb2, 1, 2, 8192, 3); M4D_PV(fb3, 1, 3, 12288, 3);
        M4D_PV(fb0, 2, 0, 0, 3); M4D_PV(fb1, 2, 1, 4096, 3); M4D_PV(fb2, 2, 2, 8192, 3); M4D_PV(fb3, 2, 3, 12288, 3);
        M4D_PV(fb0, 3, 0, 0, 3); M4D_PV(fb1, 3, 1, 4096, 2); M4D_PV(fb2, 3, 2, 8192, 1); M4D_PV(fb3, 3, 3, 12288, 0);
#undef M4D_PV
#undef M4D_LGKM
#undef M4D_DSR
        cur_dma = next_dma;
        const unsigned dl = stage ? (unsigned)-STAGE : (unsigned)STAGE;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] += dl;
#pragma unroll
        for (int c = 0; c < 4; ++c) va[c] += dl;
    }

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
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[d][rq * 4 + e];
                T* dst = oa + d * 32 + rq * 8;
                if (p.accumulate) {
                    f32x4 prev = load4(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += prev[e];
                }
                store4(dst, v);
            }
    }
}

template <int MODE>
int launch_bwd128(const BwdArgs& p, hipStream_t st, int nsplit) {
    M4D_ENV_ONCE(smx, "M4D_ATTN_BWD_SMX", 1);
    constexpr int STAGE = (MODE == BWD_DV ? 32768 : 49152) + (MODE == BWD_DQ ? 0 : 512);
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute((const void*)attn_bwd128_kernel<MODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE) != hipSuccess)
            return -3;
        hipFuncSetAttribute((const void*)attn_bwd128_kernel<MODE, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        configured = true;
    }
    dim3 grid((unsigned)((int64_t)p.nx_tiles * p.heads * p.B), (unsigned)nsplit), block(512);
    if (smx == 1) hipLaunchKernelGGL((attn_bwd128_kernel<MODE, 1>), grid, block, 2 * STAGE, st, p);
    else hipLaunchKernelGGL((attn_bwd128_kernel<MODE, 0>), grid, block, 2 * STAGE, st, p);
    return 0;
}
