// gemm_bt256w_kernel: 256x256 tile, FOUR waves (2 x 2, wave tile 128(m) x 128(n)), ONE wave per SIMD, K-tile 64.
//
// Why a second production structure next to gemm_bt256p_kernel (gemm_phased.h): at the board's power limit the phased kernel
// gains from fewer bytes per MFMA, not from schedule changes.  With 128 x 128 per wave every LDS fragment
// is used by FOUR MFMAs instead of two / four (8 fragment reads per 16 MFMAs = 0.5 reads per MFMA instead of 0.75), and a wave
// that owns its SIMD has 512 registers: the 128 x 128 fp32 accumulators (256 registers) live in AGPRs, four fragment sets
// (128 VGPRs) in VGPRs.  There is no partner wave to fill bubbles, so the wave overlaps its own work: MFMAs are asynchronous
// (16 passes each), the LDS reads / DMA issues / waits / the barrier of a phase sit in the issue shadow of the MFMAs.
//
// Staging units (8 KiB = 64 rows x 128 B).  Wave (wm, wn) reads activation units A0 = rows wm*128 + [0,64), A1 = +64 and weight
// units B0 = rows wn*128 + [0,64), B1 = +64.  A K-tile is four 16 KiB GROUPS, each the same-named unit of both wave rows / columns
// ({A0 of wm 0, A0 of wm 1}, ...); a group is staged by 4 DMA instructions per lane (16 per K-tile) and read in exactly ONE phase.
// Phase plan of K-tile k (FA0 / FA1 / FBx / FBy = fragment sets of 2 blocks x 4 k-steps; FBx / FBy swap roles every K-tile):
//   Q1: MFMA A0 x B0 (16)    read B1(k)   -> FBy       stage the group read 7 phases later
//   Q2: MFMA A0 x B1         read A1(k)   -> FA1
//   Q3: MFMA A1 x B1         read A0(k+1) -> FA0
//   Q4: MFMA A1 x B0         read B0(k+1) -> FBy' (= this tile's B1 set, dead after Q3)
// Two K-tile buffers (128 KiB).  Every phase ends  lgkmcnt(0) . vmcnt(24) . s_barrier : the group read in the NEXT phase was
// staged 7 phases ago (six younger groups x 4 instructions may stay in flight), and the group just read is re-staged right after
// the barrier (WAR).  The barrier itself is issued after the first MFMA of the next phase, so its latency and the waves' skew hide
// behind that MFMA's 16 passes.
#pragma once
#include "gemm_common.h"

namespace {
constexpr int BM2 = 256, BN2 = 256;
extern __shared__ __attribute__((aligned(16))) char dyn_smem[];

constexpr int W_GRP = 16384, W_BUF = 4 * W_GRP;      // group order inside a K-tile buffer: A0, A1, B0, B1
constexpr int W_A0 = 0, W_A1 = W_GRP, W_B0 = 2 * W_GRP, W_B1 = 3 * W_GRP;


// Epilogue of one 64(m) x 64(n) block of a wave's sub-tile, lean form: the fp32 accumulators go to a wave-private 16 KiB LDS region
// exactly as they are (accumulator layout: lane = output ROW m, 4 consecutive n per register quad), and are read back ROW-WISE:
// lane -> (row it*8 + (lane >> 3), columns (lane & 7)*8 + [0, 8)).  Bias, activation, rounding and the gated-residual update run
// in that second layout, where a lane's columns are the same in every iteration (bias: one 8-element load per block, no
// per-element branch, no dependent load in front of a store) and 8 lanes cover one contiguous 128-byte (bf16) / 256-byte (fp32)
// row segment.  EPI is a template parameter: the arithmetic is the shared epilogue's (gemm_common.h: epilogue_half_lds), in the
// same order, so the results are bit-identical.
template <typename T, int EPI>
M4D_DEV void epilogue_block64(const GemmArgs& p, char* wl, const f32x16& a00, const f32x16& a01, const f32x16& a10, const f32x16& a11,
                              int64_t m_base, int64_t n_base, int64_t m_lo, int64_t n_lo, int lane) {
    // a[ni][mi2]: a00 = (ni 0, mi2 0), a01 = (ni 0, mi2 1), a10 = (ni 1, mi2 0), a11 = (ni 1, mi2 1)
    const int li = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int mi2 = 0; mi2 < 2; ++mi2) {
        const int r = mi2 * 32 + li;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const f32x16& acc = ni == 0 ? (mi2 == 0 ? a00 : a01) : (mi2 == 0 ? a10 : a11);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int ch = (ni * 32 + rq * 8 + hi * 4) >> 2;               // 16 chunks of 4 floats per 256-byte row
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[rq * 4 + e];
                *reinterpret_cast<f32x4*>(wl + r * 256 + ((ch ^ (r & 15)) << 4)) = v;
            }
        }
    }
    // wave-private region: program order + the compiler's lgkmcnt wait order the reads after the writes
    constexpr bool F32OUT = EPI == M4D_EPI_RESID_GATE || EPI == M4D_EPI_STORE_F32;
    const int c8 = lane & 7;
    const int64_t nb = n_base + c8 * 8;
    const T* bias = (const T*)p.bias;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if (bias && !p.bias_on_m) { b0 = load4(bias + nb); b1 = load4(bias + nb + 4); }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + (lane >> 3);
        const int64_t m = m_base + r;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((2 * c8) ^ (r & 15)) << 4));
        f32x4 v1 = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((2 * c8 + 1) ^ (r & 15)) << 4));
        if (bias) {
            if (p.bias_on_m) { const float bm = (float)bias[m]; v0 += bm; v1 += bm; }
            else { v0 += b0; v1 += b1; }
        }
        if constexpr (EPI == M4D_EPI_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = gelu_tanh_f(v0[e]); v1[e] = gelu_tanh_f(v1[e]); }
        }
        if constexpr (F32OUT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = round_through<T>(v0[e]); v1[e] = round_through<T>(v1[e]); }
            float* dst = (float*)p.out + m * p.ldc + nb;
            if constexpr (EPI == M4D_EPI_RESID_GATE) {
                if (p.gate) {
                    const float* grow = p.gate + (m / p.rows_per_sample) * p.gate_stride + nb;
                    v0 = v0 * load4(grow); v1 = v1 * load4(grow + 4);
                }
                if (m >= m_lo && nb >= n_lo) { v0 += load4(dst); store4(dst, v0); }
                if (m >= m_lo && nb + 4 >= n_lo) { v1 += load4(dst + 4); store4(dst + 4, v1); }
            } else {
                if (m >= m_lo && nb >= n_lo) store4(dst, v0);
                if (m >= m_lo && nb + 4 >= n_lo) store4(dst + 4, v1);
            }
        } else {
            union { uint4 u; bf16x4 h[2]; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o.h[0][e] = (bf16_t)v0[e]; o.h[1][e] = (bf16_t)v1[e]; }
            // a chunk straddling n_lo rewrites the neighbour's identical values (same K order): benign
            if (m >= m_lo && nb + 8 > n_lo) *reinterpret_cast<uint4*>((T*)p.out + m * p.ldc + nb) = o.u;
        }
    }
}

// ABL: compile-time timing ablations (tool builds only; results wrong when != 0): 1 no DMA, 2 no fragment reads, 4 no barriers, 8 no MFMA,
// 64 = correct kernel + per-workgroup timestamps written over the first output row of every tile (tools/gemm_timeline.py)
template <int ABL, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bt256w_kernel(GemmArgs p) {
    typedef bf16_t T;
    unsigned long long ts[6];
    if constexpr (ABL & 64) { ts[0] = __builtin_readcyclecounter(); ts[5] = __builtin_amdgcn_s_memrealtime(); }
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m_lo = (int64_t)tm * BM2, n_lo = (int64_t)tn * BN2;
    const int64_t m0 = min(m_lo, p.M - BM2), n0 = min(n_lo, p.N - BN2);     // edge tiles shifted inwards (see gemm_phased.h)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS char*)dyn_smem;

    // ---- DMA: instruction i of a group covers group rows i*32 + wave*8 + (lane >> 3); group row u = unit (u >> 6), row (u & 63)
    // of the unit = operand row (u >> 6)*128 + half*64 + (u & 63).  The per-lane part of the source offset is the same for all
    // instructions of an operand (one VGPR each); everything else is wave-uniform and goes into the scalar base.
    const int lrow = lane >> 3, pc = lane & 7;
    unsigned oa, ow;
    {
        const int u = wave * 8 + lrow;
        const int lc = pc ^ ((u >> 1) & 7);
        oa = (unsigned)((u * p.lda + lc * 8) * 2);
        ow = (unsigned)((u * p.ldw + lc * 8) * 2);
    }
    const int nk = (int)(p.K / 64);
    // scalar row bases: group instruction i of half h reads operand rows (i >> 1)*128 + h*64 + (i & 1)*32 + (lane's row); the
    // K position of a staging is carried by the per-group VGPR offset (advanced by one K-tile = 128 B after every use), so a DMA
    // is two instructions: s_add m0 + global_load_lds
    const char* ra[8];
    const char* rw[8];
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) {
        const int row = (jx >> 2) * 128 + ((jx >> 1) & 1) * 64 + (jx & 1) * 32;     // jx = unit*4 + half*2 + (i & 1)
        ra[jx] = uniform_ptr((const char*)p.A + (m0 + row) * p.lda * 2);
        rw[jx] = uniform_ptr((const char*)p.W + (n0 + row) * p.ldw * 2);
    }
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
    unsigned va0 = oa, va1 = oa, vb0 = ow, vb1 = ow;        // + K offset of the NEXT staging of the group
    int ka0 = 0, ka1 = 0, kb0 = 0, kb1 = 0;                  // K-tile index of that staging (scalar; clamps the advance at nk - 1)
#define W_GLDS(VOFF, SRC, DSTOFF)                                                                            \
    if constexpr (!(ABL & 1)) asm volatile("s_add_u32 m0, %0, %3\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsw), "v"(VOFF), "s"(SRC), "i"(DSTOFF) : "memory", "m0", "scc")
    // instruction I (0..3) of the group (operand rows R, half H, group offset G) for the K-tile buffer BUFB (0 / W_BUF)
#define W_STG_I(R, VOFF, H, G, BUFB, I) W_GLDS(VOFF, R[((I) >> 1) * 4 + (H) * 2 + ((I) & 1)], (BUFB) + (G) + (I) * 4096)
#define W_ADV(VOFF, KC) do { KC += 1; VOFF += (KC < nk) ? 128u : 0u; } while (0)
    f32x16 acc[4][4];   // [ni][mi]: rows n, column m = lane (W is the MFMA's A operand, see gemm.hip)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-lane fragment addresses (block 0 of the wave's unit; block 1 = +4096) in K-tile buffer 0 and 1
    unsigned am0[4], an0[4], am1[4], an1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned x = li * ROWB + (((kk * 2 + hi) ^ ((li >> 1) & 7)) << 4);
        am0[kk] = lds_base + wm * 8192 + x;
        an0[kk] = lds_base + wn * 8192 + x;
        am1[kk] = am0[kk] + W_BUF;
        an1[kk] = an0[kk] + W_BUF;
    }
    bf16x8 fa0[2][4], fa1[2][4], fbx[2][4], fby[2][4];
#define W_DSR(dst, addr, OFF) if constexpr (!(ABL & 2)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define W_PIN() __builtin_amdgcn_sched_barrier(0)
#define W_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define W_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    // read one unit (2 blocks x 4 k-steps) from group offset G of the buffer addressed by AD
#define W_READ(F, AD, G)                                                                                   \
    do {                                                                                                   \
        W_DSR(F[0][0], AD[0], G); W_DSR(F[1][0], AD[0], G + 4096);                                          \
        W_DSR(F[0][1], AD[1], G); W_DSR(F[1][1], AD[1], G + 4096);                                          \
        W_DSR(F[0][2], AD[2], G); W_DSR(F[1][2], AD[2], G + 4096);                                          \
        W_DSR(F[0][3], AD[3], G); W_DSR(F[1][3], AD[3], G + 4096);                                          \
    } while (0)
#define W_MMA(FN, FM, NB, MB, KK, NI, MI) if constexpr (!(ABL & 8)) mma32(FN[NI][KK], FM[MI][KK], acc[(NB) + (NI)][(MB) + (MI)])
    // One phase: 16 MFMAs (FN x FM into the accumulator quadrant NB, MB); ONE other instruction in the issue shadow of each MFMA
    // (an MFMA holds the pipe for 8 passes = 32 cycles): after the first MFMA the wait for the group read in this phase and the
    // barrier, then the 8 fragment reads of the phase, then the 4 DMA instructions of the group that was read in the last phase.
#define W_PHASE(FN, FM, NB, MB, RF, RAD, RG, SR, SV, SK, SH, SG, SB)                                        \
    do {                                                                                                   \
        W_LGKM0(); W_PIN();                                                                                \
        W_MMA(FN, FM, NB, MB, 0, 0, 0); W_PIN();                                                           \
        W_VM(24);                                                                                          \
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                               \
        W_PIN();                                                                                           \
        W_MMA(FN, FM, NB, MB, 0, 0, 1); W_PIN(); W_DSR(RF[0][0], RAD[0], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 0, 1, 0); W_PIN(); W_DSR(RF[1][0], RAD[0], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 0, 1, 1); W_PIN(); W_DSR(RF[0][1], RAD[1], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 1, 0, 0); W_PIN(); W_DSR(RF[1][1], RAD[1], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 1, 0, 1); W_PIN(); W_DSR(RF[0][2], RAD[2], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 1, 1, 0); W_PIN(); W_DSR(RF[1][2], RAD[2], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 1, 1, 1); W_PIN(); W_DSR(RF[0][3], RAD[3], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 2, 0, 0); W_PIN(); W_DSR(RF[1][3], RAD[3], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 2, 0, 1); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 0); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 2, 1, 0); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 1); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 2, 1, 1); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 2); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 3, 0, 0); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 3); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 3, 0, 1); W_PIN(); W_ADV(SV, SK); W_PIN();                                     \
        W_MMA(FN, FM, NB, MB, 3, 1, 0); W_MMA(FN, FM, NB, MB, 3, 1, 1); W_PIN();                            \
    } while (0)
    // K-tile KT in buffer (AMC, ANC) = byte offset BC, next K-tile's buffer (AMN, ANN) = BN; FX holds B0(KT) on entry, FY receives
    // B1(KT) and then B0(KT + 1).  Staging issued in phase h = the group read in phase h + 7, into the slot of the group read in
    // phase h - 1: Q1 of tile k is phase 4k -> B0(k + 2) (read in phase 4(k + 2) - 1); Q2 -> B1(k + 2); Q3 -> A1(k + 2);
    // Q4 -> A0(k + 3) (read in phase 4(k + 3) - 2, next buffer).
#define W_KTILE(FX, FY, AMC, ANC, AMN, ANN, BC, BN)                                                        \
    do {                                                                                                   \
        W_PHASE(FX, fa0, 0, 0, FY, ANC, W_B1, rw, vb0, kb0, 0, W_B0, BC);                                   \
        W_PHASE(FY, fa0, 2, 0, fa1, AMC, W_A1, rw, vb1, kb1, 1, W_B1, BC);                                  \
        W_PHASE(FY, fa1, 2, 2, fa0, AMN, W_A0, ra, va1, ka1, 1, W_A1, BC);                                  \
        W_PHASE(FX, fa1, 0, 2, FY, ANN, W_B0, ra, va0, ka0, 0, W_A0, BN);                                   \
    } while (0)
#define W_STAGE(R, VOFF, KC, H, G, BUFB)                                                                   \
    do {                                                                                                   \
        W_STG_I(R, VOFF, H, G, BUFB, 0); W_STG_I(R, VOFF, H, G, BUFB, 1);                                   \
        W_STG_I(R, VOFF, H, G, BUFB, 2); W_STG_I(R, VOFF, H, G, BUFB, 3); W_ADV(VOFF, KC);                  \
    } while (0)

    // ---- prologue: both K-tile buffers in flight in consumption order, then A0(2) ----
    W_STAGE(ra, va0, ka0, 0, W_A0, 0); W_STAGE(rw, vb0, kb0, 0, W_B0, 0); W_STAGE(rw, vb1, kb1, 1, W_B1, 0); W_STAGE(ra, va1, ka1, 1, W_A1, 0);
    W_STAGE(ra, va0, ka0, 0, W_A0, W_BUF); W_STAGE(rw, vb0, kb0, 0, W_B0, W_BUF); W_STAGE(rw, vb1, kb1, 1, W_B1, W_BUF); W_STAGE(ra, va1, ka1, 1, W_A1, W_BUF);
    W_VM(24);                                  // A0(0), B0(0) of this wave have landed
    __builtin_amdgcn_s_barrier();
    W_READ(fa0, am0, W_A0);
    W_READ(fbx, an0, W_B0);
    W_LGKM0();
    __builtin_amdgcn_s_barrier();              // everybody has read A0(0): its slot may be re-staged
    W_STAGE(ra, va0, ka0, 0, W_A0, 0);         // A0(2)
    // entering phase 0 the wait is vmcnt(24): B1(0) is followed by A1(0), A0(1), B0(1), B1(1), A1(1), A0(2) = 24 instructions
    if constexpr (ABL & 64) ts[1] = __builtin_readcyclecounter();
    {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            W_KTILE(fbx, fby, am0, an0, am1, an1, 0, W_BUF);
            W_KTILE(fby, fbx, am1, an1, am0, an0, W_BUF, 0);
        }
        if (kt < nk) W_KTILE(fbx, fby, am0, an0, am1, an1, 0, W_BUF);
    }
#undef W_KTILE
#undef W_PHASE
#undef W_MMA
#undef W_READ
    W_LGKM0();
    if constexpr (ABL & 64) ts[2] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant tail stagings must land before the LDS is re-used
    __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 64) ts[3] = __builtin_readcyclecounter();
    char* wl = dyn_smem + wave * 16384;
    // (the host sends bf16 outputs whose rows are not 16-byte aligned, and the erf-GELU / SiLU epilogues of a few small GEMMs, to
    // gemm_bt256p_kernel)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
            epilogue_block64<T, EPI>(p, wl, acc[nh * 2][mh * 2], acc[nh * 2][mh * 2 + 1], acc[nh * 2 + 1][mh * 2], acc[nh * 2 + 1][mh * 2 + 1],
                                 m0 + wm * 128 + mh * 64, n0 + wn * 128 + nh * 64, m_lo, n_lo, lane);
    if constexpr (ABL & 64) {       // timestamps (shader cycles) + 100 MHz wall clock + hardware id into the tile's first output row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[4] = __builtin_readcyclecounter();
        if (t == 0) {
            unsigned long long* d = (unsigned long long*)((T*)p.out + m0 * p.ldc + n0);
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[0] = ts[0]; d[1] = ts[1]; d[2] = ts[2]; d[3] = ts[3]; d[4] = ts[4]; d[5] = ts[5];
            d[6] = __builtin_amdgcn_s_memrealtime(); d[7] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#undef W_STAGE
#undef W_STG_I
#undef W_GLDS
#undef W_ADV
#undef W_DSR
#undef W_PIN
#undef W_LGKM0
#undef W_VM
}

// launcher body shared by the per-epilogue translation units (gemm_wide_*.hip: one instantiation each, compiled in parallel)
template <int EPI>
int launch_gemm_wide(const GemmArgs& p, unsigned nwg, hipStream_t st) {
#define W_LAUNCH(A)                                                                                                    \
    do {                                                                                                               \
        static bool configured = false;                                                                                \
        if (!configured) {                                                                                             \
            if (hipFuncSetAttribute((const void*)gemm_bt256w_kernel<A, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W_BUF) != hipSuccess) \
                return -3;                                                                                             \
            configured = true;                                                                                         \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_bt256w_kernel<A, EPI>), dim3(nwg), dim3(256), 2 * W_BUF, st, p);                       \
    } while (0)
#ifdef M4D_ABLATIONS
    if constexpr (EPI == M4D_EPI_STORE) {
        if (p.abl & 64) {                       // timeline stamps, alone or on top of a timing ablation (clock = power proxy)
            switch (p.abl & 15) {
                case 1: W_LAUNCH(65); return 0;
                case 2: W_LAUNCH(66); return 0;
                case 3: W_LAUNCH(67); return 0;
                case 7: W_LAUNCH(71); return 0;
                default: W_LAUNCH(64); return 0;
            }
        }
        switch (p.abl & 15) {                  // (bit 16 = every workgroup on tile (0, 0): handled by tile_coords at run time)
            case 1: W_LAUNCH(1); return 0;
            case 2: W_LAUNCH(2); return 0;
            case 4: W_LAUNCH(4); return 0;
            case 8: W_LAUNCH(8); return 0;
            case 3: W_LAUNCH(3); return 0;
            case 7: W_LAUNCH(7); return 0;
            case 10: W_LAUNCH(10); return 0;
            default: break;
        }
    }
#endif
    W_LAUNCH(0);
#undef W_LAUNCH
    return 0;
}

}  // namespace
