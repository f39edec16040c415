// gemm_bt256p_kernel: 256x256 tile, 8 waves (2 x 4, wave tile 128(m) x 64(n)), K-tile 64, FOUR PHASES per K-tile with
// the two wave groups (wm = 0 / 1, one wave of each per SIMD) running half a phase apart: while one group issues its
// LDS fragment reads and the next global->LDS DMA, the other runs its 8 MFMAs at raised priority, and they swap at
// every barrier (the guide's 8-phase 256^2 structure, rebuilt for 32x32x16 MFMAs and this library's operand layout).
//
// Staging units (16 KiB = 128 rows x 128 B, two DMA instructions per lane), chosen so that each unit is read from LDS
// in exactly one phase:     U0 = activation rows {wm*128 + [0,64)}      U1 = {wm*128 + 64 + [0,64)}
//                           V0 = weight rows     {wn*64  + [0,32)}      V1 = {wn*64  + 32 + [0,32)}
// Phase plan of K-tile k (fragments: M = 2 m-blocks x 4 k-steps = 8 regs x4; N0 / N1 = 4 each, their two register sets
// swap roles every K-tile so that the next tile's N0 can be fetched while this tile's N0 is still in use):
//   P1: read U0 -> M               MFMA (m 0-1, n 0)     stage V1(k+1)      wait vmcnt(8)
//   P2: read V1 -> N1              MFMA (m 0-1, n 1)     stage U1(k+1)      wait vmcnt(8)
//   P3: read U1 -> M               MFMA (m 2-3, n 1)     stage U0(k+2)      wait vmcnt(6)
//   P4: read V0(k+1) -> next N0    MFMA (m 2-3, n 0)     stage V0(k+2)
// (8 / 4 / 8 / 4 fragment reads: the four waves of a group move at most 32 KiB through the LDS port per phase, the
// time of the other group's 8 MFMAs.)  Eight 16 KiB slots (2 K-tile buffers x 4 units = 128 KiB).  A slot is
// re-staged >= 3 phases after its only read phase (WAR) and every unit is waited for one phase before its read phase,
// in front of a barrier both groups pass before reading (RAW; vmcnt(8) = four younger units may still be in flight).  K-tile indices past the end are clamped:
// the redundant stagings land in slots that are never read again, which keeps the wait counts uniform.
#pragma once

constexpr int P_UNIT = 16384;
constexpr int P_U0 = 0, P_V0 = P_UNIT, P_V1 = 2 * P_UNIT, P_U1 = 3 * P_UNIT, P_BUF = 4 * P_UNIT;   // 64 KiB per K-tile

__global__ __launch_bounds__(512, 2) void gemm_bt256p_kernel(GemmArgs p) {
    typedef bf16_t T;
    int tm, tn;
    tile_coords(p, tm, tn);
    // edge tiles are shifted inwards (origin clamped to M-256 / N-256) so that every DMA row is in bounds and the
    // per-lane source offset is the same for all tiles; rows / columns below (m_lo, n_lo) are left to the neighbour
    const int64_t m_lo = (int64_t)tm * BM2, n_lo = (int64_t)tn * BN2;
    const int64_t m0 = min(m_lo, p.M - BM2), n0 = min(n_lo, p.N - BN2);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- DMA sources: unit row u = (i*8 + wave)*8 + lrow, i = 0..1; lane -> (lrow, physical chunk).  Per-lane 32-bit
    // byte offsets from the (wave-uniform) tile origin keep the eight source addresses in 8 VGPRs + 2 SGPR pairs ----
    const int lrow = lane >> 3, pc = lane & 7;
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS char*)dyn_smem;
    // per-lane source offsets (bytes): unit row u = wave*8 + lrow of instruction 0; instruction 1 and the second unit
    // of each operand differ by whole rows, which go into the (scalar) base address
    unsigned oa, ow;
    {
        const int u = wave * 8 + lrow;
        const int lc = pc ^ ((u >> 1) & 7);
        oa = (unsigned)((((u >> 6) * 128 + (u & 63)) * p.lda + lc * 8) * 2);
        ow = (unsigned)((((u >> 5) * 64 + (u & 31)) * p.ldw + lc * 8) * 2);
    }
    int nk = (M4D_ABL(p) & 128) ? 2 : (int)(p.K / 64);   // ablation 128: two K-tiles only (per-tile fixed cost)
    int64_t k0 = 0;
    if (p.ksplit > 0) {
        // split-K tail: this workgroup owns K-tiles [kb, ke) of the tile and leaves its unrounded float32 partial sums in its own
        // 256 x 256 slab of the workspace (summed, biased, activated and stored by gemm_tail_fixup_kernel)
        const int sp = blockIdx.x % p.ksplit;
        const int kb = (int)((int64_t)nk * sp / p.ksplit), ke = (int)((int64_t)nk * (sp + 1) / p.ksplit);
        k0 = (int64_t)kb * 64;
        nk = ke - kb;
        p.out = p.ws + (int64_t)blockIdx.x * (BM2 * BN2) - (m0 * BN2 + n0);     // element (m, n) -> slab[(m - m0) * 256 + (n - n0)]
        p.ldc = BN2;
        p.epilogue = M4D_EPI_STORE_F32; p.bias = nullptr; p.nb1 = 1;             // nb1 != 0: keep the accumulators unrounded
    }
    const char* baseA = uniform_ptr((const char*)p.A + (m0 * p.lda + k0) * 2);
    const char* baseW = uniform_ptr((const char*)p.W + (n0 * p.ldw + k0) * 2);
    // one unit = two DMA instructions per lane: rows r0 + (lane's row) and r1 + (lane's row) of the operand tile
    auto stage = [&](const char* base, int64_t ld, unsigned voff, int r0, int r1, int slot_off, int kt) {
        const int kc = kt < nk ? kt : nk - 1;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (kt & 1) * P_BUF + slot_off + wave * 1024);
        const char* s0 = uniform_ptr(base + (r0 * ld + kc * 64) * 2);
        const char* s1 = uniform_ptr(base + (r1 * ld + kc * 64) * 2);
        // explicit  vgpr_offset + sgpr_base  form (the builtin falls back to 64-bit VGPR addresses here)
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(s0) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + 8192), "v"(voff), "s"(s1) : "memory", "m0");
    };
#define P_STAGE_U0(KT) stage(baseA, p.lda, oa, 0, 128, P_U0, KT)
#define P_STAGE_U1(KT) stage(baseA, p.lda, oa, 64, 192, P_U1, KT)
#define P_STAGE_V0(KT) stage(baseW, p.ldw, ow, 0, 128, P_V0, KT)
#define P_STAGE_V1(KT) stage(baseW, p.ldw, ow, 32, 160, P_V1, KT)

    f32x16 acc[2][4];  // [ni][mi]

    // per-lane fragment addresses inside the CURRENT K-tile buffer (toggled by +-P_BUF after every K-tile)
    unsigned am[4], an[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned x = li * ROWB + (((kk * 2 + hi) ^ ((li >> 1) & 7)) << 4);
        am[kk] = lds_base + wm * 8192 + x;     // unit row wm*64 + mi*32 + li   (mi: +4096)
        an[kk] = lds_base + wn * 4096 + x;     // unit row wn*32 + li
    }

    // ---- prologue: K-tile 0 complete, U0 / V0 of K-tile 1 in flight ----
    P_STAGE_U0(0); P_STAGE_V0(0); P_STAGE_V1(0); P_STAGE_U1(0);
    P_STAGE_U0(1); P_STAGE_V0(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();      // group 1 runs half a phase behind group 0

    bf16x8 fm[2][4], fna[4], fnb[4];
#define P_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define P_READ_M(O0, O1)                                                                                   \
    do {                                                                                                   \
        P_DSR(fm[0][0], am[0], O0); P_DSR(fm[1][0], am[0], O1);                                            \
        P_DSR(fm[0][1], am[1], O0); P_DSR(fm[1][1], am[1], O1);                                            \
        P_DSR(fm[0][2], am[2], O0); P_DSR(fm[1][2], am[2], O1);                                            \
        P_DSR(fm[0][3], am[3], O0); P_DSR(fm[1][3], am[3], O1);                                            \
    } while (0)
#define P_READ_N(F, VOFF) do { P_DSR(F[0], an[0], VOFF); P_DSR(F[1], an[1], VOFF); P_DSR(F[2], an[2], VOFF); P_DSR(F[3], an[3], VOFF); } while (0)
#define P_SYNC_IN()                                                  \
    do {                                                             \
        __builtin_amdgcn_s_barrier();                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);                           \
        __builtin_amdgcn_s_setprio(1);                               \
    } while (0)
#define P_SYNC_OUT()                                                 \
    do {                                                             \
        __builtin_amdgcn_s_setprio(0);                               \
        __builtin_amdgcn_sched_barrier(0);                           \
        __builtin_amdgcn_s_barrier();                                \
    } while (0)
#define P_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define P_MFMA(FN, NI, MB)                                                                                 \
    do {                                                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                 \
            mma32(FN[kk], fm[0][kk], acc[NI][MB]);                                                         \
            mma32(FN[kk], fm[1][kk], acc[NI][MB + 1]);                                                     \
        }                                                                                                  \
    } while (0)
    // one K-tile; F0 holds this tile's N0 fragments on entry, F1 receives N1 and then the NEXT tile's N0
#define P_KTILE(KT, F0, F1)                                                                                \
    do {                                                                                                   \
        P_READ_M(0, 4096);               /* U0 */                                                          \
        P_STAGE_V1((KT) + 1);                                                                       \
        P_VM(8);                                                                                           \
        P_SYNC_IN(); P_MFMA(F0, 0, 0); P_SYNC_OUT();                                                       \
        P_READ_N(F1, 32768);             /* V1 */                                                          \
        P_STAGE_U1((KT) + 1);                                                                       \
        P_VM(8);                                                                                           \
        P_SYNC_IN(); P_MFMA(F1, 1, 0); P_SYNC_OUT();                                                       \
        P_READ_M(49152, 53248);          /* U1 */                                                          \
        P_STAGE_U0((KT) + 2);                                                                       \
        P_VM(6);                                                                                           \
        P_SYNC_IN(); P_MFMA(F1, 1, 2); P_SYNC_OUT();                                                       \
        {                                                                                                  \
            const unsigned dl = ((KT) & 1) ? (unsigned)-P_BUF : (unsigned)P_BUF;                           \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { am[kk] += dl; an[kk] += dl; }               \
        }                                                                                                  \
        P_READ_N(F1, 16384);             /* V0 of the next K-tile (other buffer) */                        \
        P_STAGE_V0((KT) + 2);                                                                       \
        P_SYNC_IN(); P_MFMA(F0, 0, 2); P_SYNC_OUT();                                                       \
    } while (0)

    P_READ_N(fna, 16384);                // V0 of K-tile 0
    const T* bias = (const T*)p.bias;
#define P_ZERO_ACC()                                                                                       \
    do {                                                                                                   \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                      \
            _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                  \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;                         \
    } while (0)
#define P_STORE_TILE(TM0, TN0, MLO, NLO)                                                                   \
    do {                                                                                                   \
        if (M4D_ABL(p) & 256) break;       /* ablation 256: no epilogue */                                        \
        char* wl = dyn_smem + wave * 16384;                                                                \
        if ((p.ldc & 7) == 0 || p.epilogue == M4D_EPI_RESID_GATE || p.epilogue == M4D_EPI_STORE_F32) {     \
            epilogue_half_lds<T>(p, wl, acc[0][0], acc[0][1], acc[1][0], acc[1][1], (TM0) + wm * 128, (TN0) + wn * 64, (MLO), (NLO), lane); \
            epilogue_half_lds<T>(p, wl, acc[0][2], acc[0][3], acc[1][2], acc[1][3], (TM0) + wm * 128 + 64, (TN0) + wn * 64, (MLO), (NLO), lane); \
        } else {                                                                                           \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                             \
                const int64_t m = (TM0) + wm * 128 + mi * 32 + li;                                         \
                if (m < (MLO)) continue;                                                                   \
                const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;                             \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                           \
                    epilogue_tile<T>(p, acc[ni][mi], m, (TN0) + wn * 64 + ni * 32, hi, bm, nullptr, (NLO)); \
            }                                                                                              \
        }                                                                                                  \
    } while (0)
    P_ZERO_ACC();
    {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            P_KTILE(kt, fna, fnb);
            P_KTILE(kt + 1, fnb, fna);
        }
        if (kt < nk) P_KTILE(kt, fna, fnb);
    }
#undef P_KTILE
#undef P_MFMA
#undef P_VM
#undef P_SYNC_OUT
#undef P_SYNC_IN
#undef P_READ_N
#undef P_READ_M
#undef P_DSR
    if (wm == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count of the two groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant tail stagings must land before the LDS is re-used
    __builtin_amdgcn_s_barrier();                      // ... by anyone: the epilogue transposes through the same LDS
    if (p.ksplit > 0) { P_STORE_TILE(m0, n0, m0, n0); }     // the whole slab; the fixup applies the edge masks
    else { P_STORE_TILE(m0, n0, m_lo, n_lo); }
}
