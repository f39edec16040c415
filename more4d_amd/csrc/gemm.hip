// m4d_gemm_bt: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues, gfx950 MFMA (32x32x16 bf16 / 32x32x2 f32).
//
// Two kernels share the fragment convention, LDS image and epilogue:
//  * gemm_bt256_kernel (bf16, K % 64 == 0): the production kernel.  256x256 output tile per 512-thread
//    workgroup (8 waves as 2(M) x 4(N), 128x64 per wave = 4x2 MFMA 32x32 tiles, 128 accumulator VGPRs),
//    K-tile 64, two LDS stages of 64 KiB filled by direct global->LDS DMA (global_load_lds_dwordx4: no
//    VGPR round trip, no ds_write), next tile in flight during the current tile's 32 MFMAs per wave,
//    ONE barrier per K-tile.  The DMA writes LDS lane-linearly, so the bank swizzle is applied on the
//    per-lane SOURCE address and again on the fragment read (both-sides-or-neither, cdna guide rule 21).
//  * gemm_bt_kernel<T> (bf16 / fp32, any K with 16-byte rows): 128x128 tile, 4 waves, register-staged
//    copies with zero fill — ragged K, small problems and the exact-fp32 parity mode.
// LDS rows are 128 B; 16-B chunk c of row r lives at chunk (c ^ ((r >> 1) & 7)): the ds_read_b128 of an
// MFMA fragment column is bank-conflict free (guide §6 G4 / T2).
// The MFMA is issued with W as the "A" operand and A as the "B" operand, so every lane ends up with
// 4 CONSECUTIVE n for one m: epilogue loads/stores are 8-16 B per lane and bias/gate are vector loads.
// Tiles are ordered XCD-aware (block b runs on XCD b % 8): each XCD walks a contiguous range of tiles in
// bands of 8 tile-rows, so concurrently resident tiles share A / W panels in that XCD's L2 (guide T1).
#include <stdlib.h>

#include "gemm_common.h"

// gemm_wide_*.hip: the 4-wave 128 x 128-per-wave kernel, one translation unit per epilogue
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_store(const void* args, unsigned nwg, hipStream_t st);
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_gelu(const void* args, unsigned nwg, hipStream_t st);
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_resid(const void* args, unsigned nwg, hipStream_t st);
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_f32(const void* args, unsigned nwg, hipStream_t st);
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_f32_batched(const void* args, unsigned nwg, unsigned nby, hipStream_t st);
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_store_persistent(const void* args, unsigned nwg, unsigned ncu, hipStream_t st);
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_gelu_persistent(const void* args, unsigned nwg, unsigned ncu, hipStream_t st);

namespace {

// ============================================================================ 128x128, register staged
constexpr int BM = 128, BN = 128;
constexpr int STAGE_BYTES = (BM + BN) * ROWB;  // 32 KiB

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_bt_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    constexpr int ES = sizeof(T);
    constexpr int KT = ROWB / ES;        // K elements per tile
    constexpr int KSTEPS = KT / 16;      // MFMA K=16 steps per tile
    typedef typename Frag8<T>::type frag_t;

    if (p.nb1 > 0) {   // batched: blockIdx.y = i2 * nb1 + i1 picks the operand panels and the output slab
        const int b = blockIdx.y, i1 = b % p.nb1, i2 = b / p.nb1;
        p.A = (const char*)p.A + (i1 * p.a_bs1 + i2 * p.a_bs2) * ES;
        p.W = (const char*)p.W + (i1 * p.w_bs1 + i2 * p.w_bs2) * ES;
        p.out = (float*)p.out + (int64_t)b * p.M * p.ldc;
    }
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- global->register staging: 4 x 16 B of A and of W per thread per K-tile ----
    const int srow = t >> 3, schunk = t & 7;
    const char* pa = (const char*)p.A + ((m0 + srow) * p.lda) * ES + schunk * 16;
    const char* pw = (const char*)p.W + ((n0 + srow) * p.ldw) * ES + schunk * 16;
    const int64_t sa = 32 * p.lda * ES, sw = 32 * p.ldw * ES;
    unsigned amask = 0, wmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (m0 + srow + 32 * i < p.M) amask |= 1u << i;
        if (n0 + srow + 32 * i < p.N) wmask |= 1u << i;
    }
    const int64_t kbytes = p.K * ES;
    uint4 ra[4], rw[4];
    auto gload = [&](int kt) {
        const int64_t kb = (int64_t)kt * ROWB + schunk * 16;
        const bool kin = kb < kbytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (kin && ((amask >> i) & 1)) ? *reinterpret_cast<const uint4*>(pa + i * sa + (int64_t)kt * ROWB)
                                                : make_uint4(0, 0, 0, 0);
            rw[i] = (kin && ((wmask >> i) & 1)) ? *reinterpret_cast<const uint4*>(pw + i * sw + (int64_t)kt * ROWB)
                                                : make_uint4(0, 0, 0, 0);
        }
    };
    auto swrite = [&](int stage) {
        char* sA = smem + stage * STAGE_BYTES;
        char* sW = sA + BM * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = lds_off(srow + 32 * i, schunk);
            *reinterpret_cast<uint4*>(sA + off) = ra[i];
            *reinterpret_cast<uint4*>(sW + off) = rw[i];
        }
    };

    f32x16 acc[2][2];  // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto compute = [&](int stage) {
        const char* sA = smem + stage * STAGE_BYTES;
        const char* sW = sA + BM * ROWB;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            frag_t fa[2], fw[2];
            const int c0 = (kk * 16 + hi * 8) * ES / 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rowa = wm * 64 + i * 32 + li, roww = wn * 64 + i * 32 + li;
                if constexpr (ES == 2) {
                    fa[i] = *reinterpret_cast<const frag_t*>(sA + lds_off(rowa, c0));
                    fw[i] = *reinterpret_cast<const frag_t*>(sW + lds_off(roww, c0));
                } else {
                    f32x4 lo = *reinterpret_cast<const f32x4*>(sA + lds_off(rowa, c0));
                    f32x4 hi4 = *reinterpret_cast<const f32x4*>(sA + lds_off(rowa, c0 + 1));
                    fa[i] = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                    lo = *reinterpret_cast<const f32x4*>(sW + lds_off(roww, c0));
                    hi4 = *reinterpret_cast<const f32x4*>(sW + lds_off(roww, c0 + 1));
                    fw[i] = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) mma32(fw[ni], fa[mi], acc[ni][mi]);
        }
    };

    const int nk = (int)((p.K + KT - 1) / KT);
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        compute(kt & 1);
        if (kt + 1 < nk) swrite((kt + 1) & 1);
        __syncthreads();
    }

    const T* bias = (const T*)p.bias;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + li;
        if (m >= p.M) continue;
        const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;
        const float* grow = (p.epilogue == M4D_EPI_RESID_GATE && p.gate) ? p.gate + (m / p.rows_per_sample) * p.gate_stride : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) epilogue_tile<T>(p, acc[ni][mi], m, n0 + wn * 64 + ni * 32, hi, bm, grow);
    }
}

// ============================================================================ 256x256, direct-to-LDS DMA
constexpr int BM2 = 256, BN2 = 256;
constexpr int STAGE2_BYTES = (BM2 + BN2) * ROWB;  // 64 KiB
extern __shared__ __attribute__((aligned(16))) char dyn_smem[];

__global__ __launch_bounds__(512, 2) void gemm_bt256_kernel(GemmArgs p) {
    typedef bf16_t T;
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;   // 2 x 4 waves, wave tile 128(m) x 64(n)

    // ---- DMA addressing: one wave instruction = 8 rows x 128 B = 1 KiB, lane -> (row, physical chunk) ----
    // row block rb = i*8 + wave (i = 0..3) covers rows rb*8 .. rb*8+7 of the 256-row operand tile.
    const int lrow = lane >> 3, pc = lane & 7;
    const T* ga[4];
    const T* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + lrow;
        const int lc = pc ^ ((row >> 1) & 7);          // logical chunk that must land in physical chunk pc
        const int64_t ra_ = min(m0 + row, p.M - 1);    // clamp: rows past the edge are computed, never stored
        const int64_t rw_ = min(n0 + row, p.N - 1);
        ga[i] = (const T*)p.A + ra_ * p.lda + lc * 8;
        gw[i] = (const T*)p.W + rw_ * p.ldw + lc * 8;
    }
    auto issue = [&](int stage, int kt) {
        char* sA = dyn_smem + stage * STAGE2_BYTES;
        char* sW = sA + BM2 * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rb = i * 8 + wave;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ga[i] + kt * 64), (LDS_AS void*)(sA + rb * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gw[i] + kt * 64), (LDS_AS void*)(sW + rb * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][4];  // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-lane LDS byte offsets of the first fragment row of each K=16 step (the XOR swizzle is not additive in kk);
    // further fragments of the same operand are +32 rows = +4096 B (the swizzle only depends on row & 15)
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS char*)dyn_smem;
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        aoff[kk] = lds_off(wm * 128 + li, kk * 2 + hi);
        woff[kk] = BM2 * ROWB + lds_off(wn * 64 + li, kk * 2 + hi);
    }

    const int nk = (int)(p.K / 64);
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        // my DMA for tile kt has landed; after the barrier everyone's has, and everyone is done reading stage^1
        if (M4D_ABL(p) & 64) {   // ablation: DMA stream with two batches in flight, no consumer
            if (kt + 1 < nk) issue(stage ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (!(M4D_ABL(p) & 4)) __builtin_amdgcn_s_barrier();
            continue;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk && !(M4D_ABL(p) & 1)) issue(stage ^ 1, kt + 1);
        if (M4D_ABL(p) & 8) continue;
        // Fragment double buffer.  hipcc's waitcnt pass drains lgkmcnt to 0 in front of every MFMA group here, so the
        // reads are issued from inline asm (invisible to that pass) and waited for with COUNTED lgkmcnt: the 6 reads
        // of step kk+1 are in flight while the 8 MFMAs of step kk run (cdna guide §5.7: own your waits).
        const unsigned sbase = lds_base + stage * STAGE2_BYTES;
        bf16x8 fa[2][4], fw[2][2];
#define M4D_LDFRAG(buf, kk)                                                                                          \
        do {                                                                                                         \
            const unsigned aw_ = sbase + woff[kk], aa_ = sbase + aoff[kk];                                           \
            asm volatile("ds_read_b128 %0, %1" : "=v"(fw[buf][0]) : "v"(aw_));                                       \
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fw[buf][1]) : "v"(aw_));                          \
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[buf][0]) : "v"(aa_));                                       \
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[buf][1]) : "v"(aa_));                          \
            asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fa[buf][2]) : "v"(aa_));                          \
            asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(fa[buf][3]) : "v"(aa_));                         \
        } while (0)
        M4D_LDFRAG(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk == 0) { M4D_LDFRAG(1, 1); }
            else if (kk == 1) { M4D_LDFRAG(0, 2); }
            else if (kk == 2) { M4D_LDFRAG(1, 3); }
            if (kk < 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) mma32(fw[kk & 1][ni], fa[kk & 1][mi], acc[ni][mi]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef M4D_LDFRAG
    }

    const T* bias = (const T*)p.bias;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 128 + mi * 32 + li;
        if (m >= p.M) continue;
        const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;
        const float* grow = (p.epilogue == M4D_EPI_RESID_GATE && p.gate) ? p.gate + (m / p.rows_per_sample) * p.gate_stride : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) epilogue_tile<T>(p, acc[ni][mi], m, n0 + wn * 64 + ni * 32, hi, bm, grow);
    }
}

#include "gemm_phased.h"

// ============================================================================ 256x256, staggered operand rings
// Same tile / waves / MFMA loop as gemm_bt256_kernel, but the two operands use SEPARATE LDS rings of different depth:
// A (activations) 3 slots, W 2 slots = 5 x 32 KiB = all 160 KiB of the CU's LDS.  Per iteration a wave issues the DMA of
// W(t+1) and A(t+2): at the next barrier only W(t+1) (32 KiB per CU, not 64) must have drained; the A tile always has a
// whole extra iteration in flight.  vmcnt(4) retires everything except the newest A batch (in-order counter).
constexpr int SLOT3 = 256 * ROWB;      // 32 KiB per operand slot

__global__ __launch_bounds__(512, 2) void gemm_bt256s_kernel(GemmArgs p) {
    typedef bf16_t T;
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = (int)(p.K / 64);
    const int lrow = lane >> 3, pc = lane & 7;
    const T* ga[4];
    const T* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + lrow;
        const int lc = pc ^ ((row >> 1) & 7);
        ga[i] = (const T*)p.A + min(m0 + row, p.M - 1) * p.lda + lc * 8;
        gw[i] = (const T*)p.W + min(n0 + row, p.N - 1) * p.ldw + lc * 8;
    }
    auto issue_a = [&](int kt, int slot) {
        const int kc = min(kt, nk - 1);
        char* s_ = dyn_smem + slot * SLOT3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ga[i] + kc * 64), (LDS_AS void*)(s_ + (i * 8 + wave) * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int kt) {
        const int kc = min(kt, nk - 1);
        char* s_ = dyn_smem + (3 + (kt & 1)) * SLOT3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gw[i] + kc * 64), (LDS_AS void*)(s_ + (i * 8 + wave) * 1024), 16, 0, 0);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    int aoff[4], woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        aoff[kk] = lds_off(wm * 128 + li, kk * 2 + hi);
        woff[kk] = lds_off(wn * 64 + li, kk * 2 + hi);
    }
    issue_a(0, 0);
    issue_w(0);
    issue_a(1, 1);
    int aslot = 0;                       // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_w(kt + 1);
        issue_a(kt + 2, aslot == 0 ? 2 : aslot - 1);      // (kt + 2) % 3
        const char* sA = dyn_smem + aslot * SLOT3;
        const char* sW = dyn_smem + (3 + (kt & 1)) * SLOT3;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[4], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(sW + woff[kk] + i * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sA + aoff[kk] + i * 4096);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) mma32(fw[ni], fa[mi], acc[ni][mi]);
        }
        aslot = aslot == 2 ? 0 : aslot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const T* bias = (const T*)p.bias;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 128 + mi * 32 + li;
        if (m >= p.M) continue;
        const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;
        const float* grow = (p.epilogue == M4D_EPI_RESID_GATE && p.gate) ? p.gate + (m / p.rows_per_sample) * p.gate_stride : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) epilogue_tile<T>(p, acc[ni][mi], m, n0 + wn * 64 + ni * 32, hi, bm, grow);
    }
}

// ============================================================================ 256x256, ping-pong wave groups
// Same tile and wave layout, but the K loop is cut into K=32 slices held in a 4-slot LDS ring (4 x 32 KiB) and the
// two wave groups (waves 0-3 = upper 128 rows, waves 4-7 = lower 128 rows; one wave of each group per SIMD) run ONE
// INTERVAL APART: while a group issues its 16 MFMAs on slice s, the other group is in its load phase (12 ds_read_b128
// for its next slice + 4 DMA instructions for the slice three ahead).  Every SIMD therefore always has one wave feeding
// the matrix pipe and one wave doing memory work (cdna guide T3/T5: role-split schedule, s_setprio on the MFMA
// cluster pays).  One s_barrier per interval; DMA is waited for with a COUNTED vmcnt(8) (two slices stay in flight
// across barriers) and a staged slice is first read one barrier after the wait that retired it.
constexpr int SLOT_BYTES = 512 * 64;   // (256 A rows + 256 W rows) x 64 B

M4D_DEV int lds_off64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(512, 2) void gemm_bt256pp_kernel(GemmArgs p) {
    const int ABL = M4D_ABL(p);   // timing ablations (tools/abl.sh): results are wrong when != 0
    typedef bf16_t T;
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int ns = (int)(p.K / 32);

    // DMA share of this wave: 4 x (16 rows x 64 B) of every slice; waves 0-3 carry A, waves 4-7 carry W
    const T* gp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = (wave * 4 + i) * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((R >> 2) & 3);
        if (R < 256) gp[i] = (const T*)p.A + min(m0 + R, p.M - 1) * p.lda + lc * 8;
        else gp[i] = (const T*)p.W + min(n0 + (R - 256), p.N - 1) * p.ldw + lc * 8;
    }
    auto issue = [&](int s) {
        const int sc = min(s, ns - 1);   // past the end: harmless re-load into a free slot keeps the vmcnt arithmetic uniform
        char* slot = dyn_smem + (s & 3) * SLOT_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gp[i] + sc * 32), (LDS_AS void*)(slot + i * 1024), 16, 0, 0);
    };
    int ao[2], wo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        ao[kk] = lds_off64(wm * 128 + li, kk * 2 + hi);
        wo[kk] = 256 * 64 + lds_off64(wn * 64 + li, kk * 2 + hi);
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    issue(0);
    issue(1);
    issue(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // lower group starts one interval late
    bf16x8 fa[2][4], fw[2][2];
    for (int s = 0; s < ns; ++s) {
        // ---- load phase: DMA three slices ahead, fragments of slice s
        if (!(ABL & 1)) issue(s + 3);
        const char* slot = dyn_smem + (s & 3) * SLOT_BYTES;
        if (!(ABL & 2) || s == 0)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[kk][i] = *reinterpret_cast<const bf16x8*>(slot + wo[kk] + i * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][i] = *reinterpret_cast<const bf16x8*>(slot + ao[kk] + i * 2048);
        }
        if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // my share of slice s+1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of slice s are complete (slot may be refilled)
        if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        // ---- compute phase
        __builtin_amdgcn_s_setprio(1);
        if (!(M4D_ABL(p) & 8))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) mma32(fw[kk][ni], fa[kk][mi], acc[ni][mi]);
        __builtin_amdgcn_s_setprio(0);
        if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();   // match the lower group's barrier count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const T* bias = (const T*)p.bias;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 128 + mi * 32 + li;
        if (m >= p.M) continue;
        const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;
        const float* grow = (p.epilogue == M4D_EPI_RESID_GATE && p.gate) ? p.gate + (m / p.rows_per_sample) * p.gate_stride : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) epilogue_tile<T>(p, acc[ni][mi], m, n0 + wn * 64 + ni * 32, hi, bm, grow);
    }
}

}  // namespace

namespace {

// Sums the K-slices of the tail tiles and applies the real epilogue (same semantics as epilogue_tile): one workgroup per
// (tile, 16-row strip); thread -> 4 consecutive columns.
template <typename T>
__global__ __launch_bounds__(256) void gemm_tail_fixup_kernel(GemmArgs p) {
    const int tile = blockIdx.x >> 4, strip = blockIdx.x & 15;
    int tm, tn;
    {
        GemmArgs q = p;
        q.ksplit = 1; q.tile_base = p.tile_base;
        tile_coords(q, tm, tn, tile);
    }
    const int64_t m_lo = (int64_t)tm * 256, n_lo = (int64_t)tn * 256;
    const int64_t m0 = min(m_lo, p.M - 256), n0 = min(n_lo, p.N - 256);
    const T* bias = (const T*)p.bias;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int r = strip * 16 + i / 64, c4 = (i % 64) * 4;
        const int64_t m = m0 + r, nb = n0 + c4;
        if (m < m_lo || nb < n_lo) continue;               // rows / columns of the inward-shifted edge tile that belong to the neighbour
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < p.ksplit; ++sp) v += load4(p.ws + ((int64_t)tile * p.ksplit + sp) * 65536 + r * 256 + c4);
        if (bias) { if (p.bias_on_m) v += (float)bias[m]; else v += load4(bias + nb); }
        if (p.epilogue == M4D_EPI_GELU_TANH) { for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]); }
        else if (p.epilogue == M4D_EPI_GELU_ERF) { for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]); }
        else if (p.epilogue == M4D_EPI_SILU) { for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]); }
        if (p.epilogue == M4D_EPI_RESID_GATE) {
            float* dst = (float*)p.out + m * p.ldc + nb;
            f32x4 x = load4(dst);
            f32x4 g = {1.f, 1.f, 1.f, 1.f};
            if (p.gate) g = load4(p.gate + (m / p.rows_per_sample) * p.gate_stride + nb);
            for (int e = 0; e < 4; ++e) x[e] += round_through<T>(v[e]) * g[e];
            store4(dst, x);
        } else if (p.epilogue == M4D_EPI_STORE_F32) {
            for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]);
            store4((float*)p.out + m * p.ldc + nb, v);
        } else {
            store4((T*)p.out + m * p.ldc + nb, v);
        }
    }
}

// (full tiles, K-slices per tail tile) minimising  full/256 + ceil(tail*S/256)/S  (+ 6 % per split for the slab traffic and the
// shorter K loops); S = 1 means a single ordinary launch
inline int tail_split(int64_t nwg, int64_t nk, int ncu) {
    const int64_t tail = nwg % ncu;
    if (nwg < ncu || tail == 0) return 1;
    int best = 1;
    double best_t = 1.0;
    for (int S = 2; S <= 8; ++S) {
        if (nk / S < 8) break;
        const double t = (double)((tail * S + ncu - 1) / ncu) / S * 1.06 + 0.03;
        if (t < best_t - 0.05) { best_t = t; best = S; }
    }
    return best;
}

}  // namespace

// compute units of the current device (the tile-round arithmetic of the split-K tail); 256 on an MI355X
// (a process may drive several GPUs — one per emulated rank or per stream owner: everything device-bound is kept per device)
constexpr int M4D_MAX_DEVICES = 64;
static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev >= 0 && dev < M4D_MAX_DEVICES ? dev : 0;
}
static int device_cus() {
    static int n[M4D_MAX_DEVICES] = {0};
    const int dev = current_device();
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n[dev] = v;
        else n[dev] = 256;
    }
    return n[dev];
}
// production structure for big bf16 problems: 5 (default) = 4-wave 128 x 128-per-wave kernel (gemm_wide.h), 4 = phased two-group
// kernel (gemm_phased.h); 1 two-stage, 2 ping-pong, 3 staggered rings (older A/B structures)
static int gemm_variant() { M4D_ENV_ONCE(v, "M4D_GEMM_VARIANT", 5); return v; }
// M4D_GEMM_TAIL=1 (phased kernel only): split-K over the partial last tile round
static bool gemm_tail_enabled() { M4D_ENV_ONCE(t, "M4D_GEMM_TAIL", 0); return t != 0 && gemm_variant() == 4; }

extern "C" int64_t m4d_gemm_bt_workspace_bytes(m4d_dtype dt, int64_t M, int64_t N, int64_t K) {
    if (!gemm_tail_enabled()) return 0;             // (callers then use the plain entry point: no workspace, no extra call)
    if (dt != M4D_BF16 || K % 64 || M < 512 || N < 512) return 0;
    const int64_t nwg = ((M + 255) / 256) * ((N + 255) / 256);
    const int ncu = device_cus();
    const int S = tail_split(nwg, K / 64, ncu);
    return S > 1 ? (nwg % ncu) * S * 65536 * 4 : 0;
}

static int gemm_bt_impl(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                        int bias_on_m, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                        const float* gate, int64_t gate_stride, int64_t rows_per_sample, void* ws, int64_t ws_bytes, m4d_stream stream) {
    const int es = dt == M4D_BF16 ? 2 : 4;
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "gemm_bt: bad dtype %d", (int)dt);
    M4D_CHECK_ARG(A && W && out, "gemm_bt: null pointer");
    M4D_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_bt: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    M4D_CHECK_ARG((K * es) % 16 == 0, "gemm_bt: K*sizeof(T) must be a multiple of 16 (K=%lld)", (long long)K);
    M4D_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "gemm_bt: N and ldc must be multiples of 4 (N=%lld ldc=%lld)", (long long)N, (long long)ldc);
    M4D_CHECK_ARG((lda * es) % 16 == 0 && (ldw * es) % 16 == 0, "gemm_bt: lda/ldw rows must be 16-byte aligned");
    M4D_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bt: pointers must be 16-byte aligned");
    M4D_CHECK_ARG(epilogue >= 0 && epilogue <= 5, "gemm_bt: bad epilogue %d", epilogue);
    M4D_CHECK_ARG(epilogue != M4D_EPI_RESID_GATE || gate == nullptr || rows_per_sample > 0, "gemm_bt: rows_per_sample must be > 0 with a gate");
    GemmArgs p;
    p.A = A; p.W = W; p.bias = bias; p.out = out; p.gate = gate;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.gate_stride = gate_stride; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : M;
    p.epilogue = epilogue; p.bias_on_m = bias_on_m;
    p.nb1 = 0; p.a_bs1 = p.a_bs2 = p.w_bs1 = p.w_bs2 = 0; p.tap_rows = p.tap_kh = 0; p.tap_s1 = p.tap_s2 = 0;
    p.remap_n = 0; p.tile_base = 0; p.ksplit = 0; p.ws = nullptr; p.tile_off = 0;
    p.abl = 0; p.sync = nullptr;
#ifdef M4D_ABLATIONS
    { M4D_ENV_ONCE(abl_env, "M4D_GEMM_ABL", 0); p.abl = abl_env; }
#endif
    hipStream_t st = (hipStream_t)stream;
    // production kernel: big bf16 problems with K a multiple of the 64-wide K-tile
    const bool big = dt == M4D_BF16 && K % 64 == 0 && M >= 512 && N >= 512;
    int kclass = M4D_KC_GEMM_GENERIC;
    if (big) {
        const int variant = gemm_variant();
        static PerDeviceOnce configured;
        if (configured.pending()) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bt256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE2_BYTES);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)gemm_bt256pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SLOT_BYTES);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)gemm_bt256s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * SLOT3);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)gemm_bt256p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_BUF);
            if (e != hipSuccess) { m4d_set_error("gemm_bt: cannot enable 128 KiB LDS: %s", hipGetErrorString(e)); return -3; }
            configured.mark();
        }
        p.tiles_m = (int)((M + BM2 - 1) / BM2); p.tiles_n = (int)((N + BN2 - 1) / BN2);
        const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
        M4D_CHECK_ARG(nwg < (1ll << 31), "gemm_bt: too many tiles");
        // M4D_GEMM_TAIL=1: split-K over the partial last tile round (needs the workspace).  OFF by default: same-box A/B inside bench.py
        // showed NO gain (1 663 / 1 660 vs 1 663 / 1 665 ms per step) although the 5120-wide GEMMs idle 4.6 % of their CU-rounds —
        // the step runs at the 1.4 kW power limit, so an idle partial round is paid back as clock on the full
        // ones — and it costs the bit-identical results of equal samples in one batch (split tiles sum K in a different order).
        // wide kernel: 16-byte row-aligned bf16 outputs (or fp32 outputs), the four epilogues the DiT's big GEMMs use
        const bool wide_ok = (epilogue == M4D_EPI_RESID_GATE || epilogue == M4D_EPI_STORE_F32 ||
                              ((ldc & 7) == 0 && (epilogue == M4D_EPI_STORE || epilogue == M4D_EPI_GELU_TANH))) &&
                             ((uintptr_t)out % 32) == 0;
        const int ncu = device_cus();
        const int S = (ws && gemm_tail_enabled()) ? tail_split(nwg, K / 64, ncu) : 1;
        if (S > 1 && ws_bytes >= (nwg % ncu) * S * 65536 * 4) {
            const int tail = (int)(nwg % ncu), full = (int)(nwg - tail);
            p.remap_n = full;
            kclass = M4D_KC_GEMM_PHASED;
            hipLaunchKernelGGL(gemm_bt256p_kernel, dim3((unsigned)full), dim3(512), 2 * P_BUF, st, p);
            GemmArgs q = p;
            q.remap_n = 0; q.tile_base = full; q.ksplit = S; q.ws = (float*)ws;
            hipLaunchKernelGGL(gemm_bt256p_kernel, dim3((unsigned)(tail * S)), dim3(512), 2 * P_BUF, st, q);
            hipLaunchKernelGGL(gemm_tail_fixup_kernel<bf16_t>, dim3((unsigned)(tail * 16)), dim3(256), 0, st, q);
        } else if (variant == 4 || (variant == 5 && !wide_ok)) {
            // M4D_GEMM_CHUNK=n: the tile grid in launches of n tiles (whole rounds of the 256 CUs): every launch boundary re-aligns the
            // workgroups that share A / W panels, whose K loops otherwise drift apart and stop hitting each other's lines in L2
            kclass = M4D_KC_GEMM_PHASED;
            M4D_ENV_ONCE(chunk, "M4D_GEMM_CHUNK", 0);
            if (chunk > 0 && nwg > chunk) {
                for (int64_t t0 = 0; t0 < nwg; t0 += chunk) {
                    GemmArgs q = p;
                    q.remap_n = (int)std::min<int64_t>(chunk, nwg - t0); q.tile_off = (int)t0;
                    hipLaunchKernelGGL(gemm_bt256p_kernel, dim3((unsigned)q.remap_n), dim3(512), 2 * P_BUF, st, q);
                }
            } else hipLaunchKernelGGL(gemm_bt256p_kernel, dim3((unsigned)nwg), dim3(512), 2 * P_BUF, st, p);
        }
        else if (variant == 5) {
            kclass = M4D_KC_GEMM_WIDE;
            // persistent form (M4D_GEMM_PERSIST, default on): bf16 epilogues, K/64 even and >= 4, operands below 4 GiB,
            // more tiles than CUs
            M4D_ENV_ONCE(persist, "M4D_GEMM_PERSIST", 1);
            const int64_t nkt = K / 64;
#ifdef M4D_ABLATIONS
            M4D_ENV_ONCE(pgrid, "M4D_GEMM_PERSIST_GRID", 0);     // tool builds: workgroups of the persistent launch (any count >= 1)
            const int ncu_p = pgrid > 0 ? (int)std::min<int64_t>(pgrid, nwg > 1 ? nwg - 1 : 1) : ncu;
#else
            const int ncu_p = ncu;
#endif
            // (tool builds: the timing ablations / timeline stamps exist for the one-tile form only; 128.. = persistent epilogue debug bits)
            M4D_ENV_ONCE(persist_bm, "M4D_GEMM_PERSIST_BIAS_M", 1);      // A/B: 0 = the V^T projection (bias along m) on the one-tile form (rounds 3-4)
            const bool pers_ok = persist && (p.abl & 127) == 0 && (epilogue == M4D_EPI_STORE || epilogue == M4D_EPI_GELU_TANH) && (!(bias && bias_on_m) || (persist_bm && M % 2 == 0)) &&
                                 nkt >= 4 && (nkt & 1) == 0 && M * lda * 2 < (1ll << 32) && N * ldw * 2 < (1ll << 32) && nwg > ncu_p;
            // XCD-wide tile rounds of the persistent kernel (M4D_GEMM_SYNC): 8 arrival counters in a library-owned buffer, zeroed in front
            // of every launch (two persistent GEMMs running at once on different streams would only lose the hint)
            p.sync = nullptr;
            M4D_ENV_ONCE(psync, "M4D_GEMM_SYNC", 1);
            // (only from four rounds on: at the 1.7 rounds of a rank's M = 5 460 shard the poll costs 3 % and there is hardly a panel to share)
            if (pers_ok && psync && ncu_p % 8 == 0 && nwg >= 4 * (int64_t)ncu_p) {
                // one counter buffer PER DEVICE, allocated on the device that is current at its first qualifying launch (a
                // process-wide buffer made launches on a second GPU poll and memset device-0 memory)
                static unsigned* g_sync[M4D_MAX_DEVICES] = {nullptr};
                static bool tried[M4D_MAX_DEVICES] = {false};
                const int dev = current_device();
                if (!tried[dev]) {       // (never allocate inside a stream capture: the hint simply starts with the first launch outside one)
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { cs = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
                    if (cs == hipStreamCaptureStatusNone) {
                        tried[dev] = true;
                        if (hipMalloc((void**)&g_sync[dev], 1024) != hipSuccess) { g_sync[dev] = nullptr; (void)hipGetLastError(); }
                    }
                }
                if (g_sync[dev] && hipMemsetAsync(g_sync[dev], 0, 1024, st) == hipSuccess) p.sync = g_sync[dev];
            }
            const int rc = pers_ok ? (epilogue == M4D_EPI_STORE ? m4d_launch_gemm_wide_store_persistent(&p, (unsigned)nwg, (unsigned)ncu_p, st)
                                                                : m4d_launch_gemm_wide_gelu_persistent(&p, (unsigned)nwg, (unsigned)ncu_p, st))
                         : epilogue == M4D_EPI_STORE ? m4d_launch_gemm_wide_store(&p, (unsigned)nwg, st)
                         : epilogue == M4D_EPI_GELU_TANH ? m4d_launch_gemm_wide_gelu(&p, (unsigned)nwg, st)
                         : epilogue == M4D_EPI_RESID_GATE ? m4d_launch_gemm_wide_resid(&p, (unsigned)nwg, st)
                         : m4d_launch_gemm_wide_f32(&p, (unsigned)nwg, st);
            if (rc != 0) { m4d_set_error("gemm_bt: cannot enable 128 KiB LDS (wide kernel)"); return -3; }
        }
        else if (variant == 3) hipLaunchKernelGGL(gemm_bt256s_kernel, dim3((unsigned)nwg), dim3(512), 5 * SLOT3, st, p);
        else if (variant == 1) hipLaunchKernelGGL(gemm_bt256_kernel, dim3((unsigned)nwg), dim3(512), 2 * STAGE2_BYTES, st, p);
        else hipLaunchKernelGGL(gemm_bt256pp_kernel, dim3((unsigned)nwg), dim3(512), 4 * SLOT_BYTES, st, p);
    } else {
        p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
        const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
        M4D_CHECK_ARG(nwg < (1ll << 31), "gemm_bt: too many tiles");
        dim3 grid((unsigned)nwg), block(256);
        if (dt == M4D_BF16) hipLaunchKernelGGL(gemm_bt_kernel<bf16_t>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(gemm_bt_kernel<float>, grid, block, 0, st, p);
    }
    M4D_CHECK_LAUNCH("gemm_bt");
    m4d_count_launch(kclass);
    return 0;
}

extern "C" int m4d_gemm_bt(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                           int bias_on_m, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                           const float* gate, int64_t gate_stride, int64_t rows_per_sample, m4d_stream stream) {
    return gemm_bt_impl(dt, A, lda, W, ldw, bias, bias_on_m, out, ldc, M, N, K, epilogue, gate, gate_stride, rows_per_sample, nullptr, 0, stream);
}

// m4d_gemm_bt with a caller-owned workspace (m4d_gemm_bt_workspace_bytes; may be NULL / too small: plain single launch): when the
// 256 x 256 tile grid leaves a partial last round on the 256 CUs, the tiles of that round are split along K so that the whole
// chip works on them, and a small kernel sums the float32 slabs and applies the epilogue.
extern "C" int m4d_gemm_bt_ws(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                              int bias_on_m, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                              const float* gate, int64_t gate_stride, int64_t rows_per_sample, void* ws, int64_t ws_bytes, m4d_stream stream) {
    M4D_CHECK_ARG(ws == nullptr || ((uintptr_t)ws % 16) == 0, "gemm_bt_ws: workspace must be 16-byte aligned");
    return gemm_bt_impl(dt, A, lda, W, ldw, bias, bias_on_m, out, ldc, M, N, K, epilogue, gate, gate_stride, rows_per_sample, ws, ws_bytes, stream);
}


// Batched A.W^T with float32 (unrounded) outputs: the split-K partial products of the VAE conv weight gradients
// (more4d_amd/vae_autograd.py:conv_wgrad): batch (i1, i2) reads A + i1*a_bs1 + i2*a_bs2 and W + i1*w_bs1 + i2*w_bs2.
extern "C" int m4d_gemm_bt_batched(m4d_dtype dt, const void* A, int64_t lda, int64_t a_bs1, int64_t a_bs2, const void* W, int64_t ldw,
                                   int64_t w_bs1, int64_t w_bs2, float* out, int64_t M, int64_t N, int64_t K, int nb1, int nb2,
                                   m4d_stream stream) {
    const int es = dt == M4D_BF16 ? 2 : 4;
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "gemm_bt_batched: bad dtype %d", (int)dt);
    M4D_CHECK_ARG(A && W && out && M > 0 && N > 0 && K > 0 && nb1 > 0 && nb2 > 0, "gemm_bt_batched: null/empty");
    M4D_CHECK_ARG((K * es) % 16 == 0 && N % 4 == 0, "gemm_bt_batched: K*sizeof(T) %% 16 and N %% 4 must be 0");
    M4D_CHECK_ARG((lda * es) % 16 == 0 && (ldw * es) % 16 == 0 && (a_bs1 * es) % 16 == 0 && (a_bs2 * es) % 16 == 0 &&
                  (w_bs1 * es) % 16 == 0 && (w_bs2 * es) % 16 == 0, "gemm_bt_batched: strides must keep rows 16-byte aligned");
    M4D_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bt_batched: pointers must be 16-byte aligned");
    M4D_CHECK_ARG((int64_t)nb1 * nb2 <= 65535, "gemm_bt_batched: too many batches");
    GemmArgs p;
    p.A = A; p.W = W; p.bias = nullptr; p.out = out; p.gate = nullptr;
    p.lda = lda; p.ldw = ldw; p.ldc = N; p.M = M; p.N = N; p.K = K;
    p.gate_stride = 0; p.rows_per_sample = M; p.epilogue = M4D_EPI_STORE_F32; p.bias_on_m = 0;
    p.nb1 = nb1; p.a_bs1 = a_bs1; p.a_bs2 = a_bs2; p.w_bs1 = w_bs1; p.w_bs2 = w_bs2; p.abl = 0; p.sync = nullptr;
    p.tap_rows = p.tap_kh = 0; p.tap_s1 = p.tap_s2 = 0;
    p.remap_n = 0; p.tile_base = 0; p.ksplit = 0; p.ws = nullptr; p.tile_off = 0;
    p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)(nb1 * nb2)), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL(gemm_bt_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_bt_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("gemm_bt_batched");
    return 0;
}

// The conv weight gradient on the production kernel: out[s] (float32 [M, N], unrounded) = A_s [M, K] . W_s [N, K]^T for K-slice s of nb1
// (A_s = A + s * a_bs1, W_s = W + s * w_bs1 elements), where the M = taps * tap_rows rows of A are STACKED TAPS: rows [t * tap_rows,
// (t + 1) * tap_rows) are rows [0, tap_rows) of the matrix at A, read tap_s1 * (t / tap_kh) + tap_s2 * (t % tap_kh) elements further
// along K.  dW[(dt, dh, co), (dw, ci)] = sum_p dy[p, co] x[p + tap(dt, dh) + dw, ci] contracts over pixels; with dy as the shifted
// operand all kt * kh taps of a layer are ONE launch whose M (864 .. 3456) fills 256-row tiles — the generic 128 x 128 kernel ran one
// batch element per tap at M = Cout (96 .. 384) and 290 TF (more4d_amd/vae_autograd.py:conv_wgrad, train_vae.py:173-187).
extern "C" int m4d_gemm_bt_taps(m4d_dtype dt, const void* A, int64_t lda, int64_t a_bs1, const void* W, int64_t ldw, int64_t w_bs1, float* out,
                                int64_t M, int64_t N, int64_t K, int nb1, int tap_rows, int tap_kh, int64_t tap_s1, int64_t tap_s2,
                                m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16, "gemm_bt_taps: bf16 only");
    M4D_CHECK_ARG(A && W && out && nb1 > 0 && nb1 <= 65535, "gemm_bt_taps: null/empty");
    M4D_CHECK_ARG(M >= 256 && N >= 256 && K >= 128 && K % 64 == 0 && N % 4 == 0, "gemm_bt_taps: M, N >= 256, K %% 64 == 0 (M=%lld N=%lld K=%lld)",
                  (long long)M, (long long)N, (long long)K);
    M4D_CHECK_ARG(tap_rows > 0 && tap_rows % 32 == 0 && M % tap_rows == 0 && tap_kh > 0 && (M / tap_rows) % tap_kh == 0,
                  "gemm_bt_taps: M must be whole taps of tap_rows (a multiple of 32) rows");
    M4D_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && a_bs1 % 8 == 0 && w_bs1 % 8 == 0 && tap_s1 % 8 == 0 && tap_s2 % 8 == 0,
                  "gemm_bt_taps: strides and tap offsets must keep rows 16-byte aligned");
    M4D_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 32) == 0, "gemm_bt_taps: pointer alignment");
    GemmArgs p;
    p.A = A; p.W = W; p.bias = nullptr; p.out = out; p.gate = nullptr;
    p.lda = lda; p.ldw = ldw; p.ldc = N; p.M = M; p.N = N; p.K = K;
    p.gate_stride = 0; p.rows_per_sample = M; p.epilogue = M4D_EPI_STORE_F32; p.bias_on_m = 0;
    p.nb1 = nb1; p.a_bs1 = a_bs1; p.a_bs2 = 0; p.w_bs1 = w_bs1; p.w_bs2 = 0; p.abl = 0; p.sync = nullptr;
    p.tap_rows = tap_rows; p.tap_kh = tap_kh; p.tap_s1 = tap_s1; p.tap_s2 = tap_s2;
    p.remap_n = 0; p.tile_base = 0; p.ksplit = 0; p.ws = nullptr; p.tile_off = 0;
    p.tiles_m = (int)((M + BM2 - 1) / BM2); p.tiles_n = (int)((N + BN2 - 1) / BN2);
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    if (m4d_launch_gemm_wide_f32_batched(&p, (unsigned)nwg, (unsigned)nb1, (hipStream_t)stream) != 0) {
        m4d_set_error("gemm_bt_taps: cannot enable 128 KiB LDS (wide kernel)");
        return -3;
    }
    M4D_CHECK_LAUNCH("gemm_bt_taps");
    m4d_count_launch(M4D_KC_GEMM_WIDE);
    return 0;
}
