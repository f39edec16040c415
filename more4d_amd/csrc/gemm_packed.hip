// m4d_gemm_bt_packed: C[M,N] = A[M,K] * W[N,K]^T where ONE operand (the weight) has been re-laid-out once into MFMA
// fragment order (m4d_pack_frag) and is streamed from L2/HBM straight into VGPRs, while the other operand (the
// activation) goes through LDS by DMA.
//
// Why (tools/abl.sh): the 256x256 DMA kernel of gemm.hip is bound by the global->LDS DMA path
// (LDS write side, ~23-35 B/clk/CU: both operands = 64 KiB per K-tile per CU), not by MFMA, L2 misses or bank conflicts.
// Here only the activation tile (32 KiB per K-tile) crosses the DMA path; the weight fragments arrive through the
// ordinary VMEM return path as fully coalesced 1-KiB wave loads (the packed layout makes a fragment 64 x 16 B
// contiguous).  LDS then holds FOUR activation stages (128 KiB): three tiles in flight, DMA never drains.
//
// Packed layout: Wp[rb = row/32][kb = k/16][lane 0..63][8 elements], lane (li = lane&31, hi = lane>>5) holding
// W[rb*32 + li][kb*16 + hi*8 + 0..7] — exactly the 32x32x16 MFMA operand fragment (common.h).  Rows are padded to 32.
//
// Tile 256 (D = DMA/activation side) x 256 (P = packed/weight side), 8 waves as 2 (D) x 4 (P): 128 D rows x 64 P rows per
// wave, 4 x 2 MFMA tiles, 32 MFMAs per wave per K-tile (K = 64).  PM = false: P is the N side (ordinary Linear, P
// fragments are the MFMA "A" operand so lanes own 4 consecutive n);  PM = true: P is the M side (V^T = W_v x^T: the
// D fragments are the MFMA "A" operand, lanes own 4 consecutive tokens of one output feature).
// VMEM ordering per iteration: [8 packed-fragment loads for tile t+1 (inline asm, counted by hand)] then [4 DMA
// instructions for activation tile t+3]; `s_waitcnt vmcnt(4)` at the top of the next iteration retires everything but
// the newest DMA batch (vmcnt retires in order), then one s_barrier publishes activation tile t+1.
#include <stdlib.h>
#include "gemm_common.h"

namespace {

constexpr int PT = 256;                 // tile edge on both sides
constexpr int ASTAGE = PT * ROWB;       // 32 KiB activation stage (256 rows x 128 B)
constexpr int NSTAGE = 4;
extern __shared__ __attribute__((aligned(16))) char dsm[];

#define M4D_PLOAD(dst, ptr, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr) : "memory")

template <bool PM>
__global__ __launch_bounds__(512, 2) void gemm_packed_kernel(GemmArgs p) {
    typedef bf16_t T;
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * PT, n0 = (int64_t)tn * PT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wd = wave >> 2, wp = wave & 3;
    const int nk = (int)(p.K / 64);
    const int64_t KB = p.K / 16;                        // 16-wide k blocks per packed row block

    // D side = row-major activations, P side = packed weights
    const T* Dmat = (const T*)(PM ? p.W : p.A);
    const int64_t ldd = PM ? p.ldw : p.lda;
    const int64_t Drows = PM ? p.N : p.M, Prows = PM ? p.M : p.N;
    const int64_t d0 = PM ? n0 : m0, p0 = PM ? m0 : n0;
    const T* Pmat = (const T*)(PM ? p.A : p.W);

    // ---- DMA of the activation tile: 32 wave instructions of 8 rows x 128 B; wave w issues i*8 + w, i = 0..3
    const T* ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        ga[i] = Dmat + min(d0 + row, Drows - 1) * ldd + lc * 8;
    }
    auto issue_a = [&](int kt) {
        const int kc = min(kt, nk - 1);          // past the end: harmless re-load keeps the vmcnt arithmetic uniform
        char* st = dsm + (kt & (NSTAGE - 1)) * ASTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ga[i] + kc * 64), (LDS_AS void*)(st + (i * 8 + wave) * 1024), 16, 0, 0);
    };
    // ---- packed fragments of this wave: row blocks p0/32 + wp*2 + {0,1}; one K-tile = 4 consecutive 1-KiB blocks
    const int64_t prb_last = (Prows + 31) / 32 - 1;
    const T* pb[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int64_t rb = min(p0 / 32 + wp * 2 + f, prb_last);
        pb[f] = Pmat + (rb * KB * 64 + lane) * 8;
    }

    f32x16 acc[2][4];   // [pi][di]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    bf16x8 pA[4][2], pB[4][2];   // packed fragments [kk][f] of the current / next K-tile (roles alternate)

#define M4D_LOAD_P(BUF, KT)                                                              \
    do {                                                                                 \
        const int kc_ = min((KT), nk - 1);                                               \
        const T* q0_ = pb[0] + (int64_t)kc_ * 2048;                                      \
        const T* q1_ = pb[1] + (int64_t)kc_ * 2048;                                      \
        M4D_PLOAD(BUF[0][0], q0_, 0);    M4D_PLOAD(BUF[0][1], q1_, 0);                   \
        M4D_PLOAD(BUF[1][0], q0_, 1024); M4D_PLOAD(BUF[1][1], q1_, 1024);                \
        M4D_PLOAD(BUF[2][0], q0_, 2048); M4D_PLOAD(BUF[2][1], q1_, 2048);                \
        M4D_PLOAD(BUF[3][0], q0_, 3072); M4D_PLOAD(BUF[3][1], q1_, 3072);                \
    } while (0)

#define M4D_ITER(CUR, NXT, KT)                                                                               \
    do {                                                                                                     \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     /* CUR fragments + activation tiles <= KT+1 landed */ \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        M4D_LOAD_P(NXT, (KT) + 1);                                                                           \
        issue_a((KT) + 3);                                                                                   \
        const char* sA_ = dsm + ((KT) & (NSTAGE - 1)) * ASTAGE;                                              \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                   \
            bf16x8 fd_[4];                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                fd_[i] = *reinterpret_cast<const bf16x8*>(sA_ + lds_off(wd * 128 + i * 32 + li, kk * 2 + hi)); \
            _Pragma("unroll") for (int pi = 0; pi < 2; ++pi)                                                 \
                _Pragma("unroll") for (int di = 0; di < 4; ++di) {                                           \
                    if (PM) mma32(fd_[di], CUR[kk][pi], acc[pi][di]);                                        \
                    else mma32(CUR[kk][pi], fd_[di], acc[pi][di]);                                           \
                }                                                                                            \
        }                                                                                                    \
    } while (0)

    // prologue: fragments of tile 0, then activation tiles 0..2 (issue order matters for the counted waits)
    M4D_LOAD_P(pA, 0);
    issue_a(0);
    issue_a(1);
    issue_a(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // pA + activation tile 0
    // (the first M4D_ITER waits vmcnt(4): tiles 0 and 1 — slightly conservative, once)
    for (int kt = 0; kt < nk; kt += 2) {
        M4D_ITER(pA, pB, kt);
        if (kt + 1 < nk) M4D_ITER(pB, pA, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef M4D_ITER
#undef M4D_LOAD_P

    // ---- epilogue (shared): regs run along the MFMA "A" side
    const T* bias = (const T*)p.bias;
    if (!PM) {       // lanes = m (D rows), regs = n (P rows)
#pragma unroll
        for (int di = 0; di < 4; ++di) {
            const int64_t m = m0 + wd * 128 + di * 32 + li;
            if (m >= p.M) continue;
            const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;
            const float* grow = (p.epilogue == M4D_EPI_RESID_GATE && p.gate) ? p.gate + (m / p.rows_per_sample) * p.gate_stride : nullptr;
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) epilogue_tile<T>(p, acc[pi][di], m, n0 + wp * 64 + pi * 32, hi, bm, grow);
        }
    } else {         // lanes = m (P rows), regs = n (D rows)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int64_t m = m0 + wp * 64 + pi * 32 + li;
            if (m >= p.M) continue;
            const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;
            const float* grow = (p.epilogue == M4D_EPI_RESID_GATE && p.gate) ? p.gate + (m / p.rows_per_sample) * p.gate_stride : nullptr;
#pragma unroll
            for (int di = 0; di < 4; ++di) epilogue_tile<T>(p, acc[pi][di], m, n0 + wd * 128 + di * 32, hi, bm, grow);
        }
    }
}

// W [rows, K] row-major (row stride ld) -> packed fragments, rows padded to 32 with zeros
__global__ __launch_bounds__(256) void pack_frag_kernel(const bf16_t* W, int64_t ld, bf16_t* out, int64_t rows, int64_t K) {
    const int64_t KB = K / 16, RB = (rows + 31) / 32;
    const int64_t total = RB * KB * 64;          // 16-byte chunks
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63);
        const int64_t blk = c >> 6, kb = blk % KB, rb = blk / KB;
        const int64_t row = rb * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < rows) v = *reinterpret_cast<const uint4*>(W + row * ld + k);
        *reinterpret_cast<uint4*>(out + c * 8) = v;
    }
}

}  // namespace

extern "C" int64_t m4d_pack_frag_elems(int64_t rows, int64_t K) { return (rows + 31) / 32 * 32 * K; }

extern "C" int m4d_pack_frag(m4d_dtype dt, const void* W, int64_t ld, void* out, int64_t rows, int64_t K, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16, "pack_frag: bf16 only");
    M4D_CHECK_ARG(W && out && rows > 0 && K > 0 && K % 16 == 0 && ld % 8 == 0, "pack_frag: K must be a multiple of 16, ld of 8");
    M4D_CHECK_ARG(((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "pack_frag: 16-byte alignment");
    const int64_t chunks = (rows + 31) / 32 * (K / 16) * 64;
    int64_t g = (chunks + 255) / 256;
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ld, (bf16_t*)out, rows, K);
    M4D_CHECK_LAUNCH("pack_frag");
    return 0;
}

extern "C" int m4d_gemm_bt_packed(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, int packed_side,
                                  const void* bias, int bias_on_m, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                  int epilogue, const float* gate, int64_t gate_stride, int64_t rows_per_sample,
                                  m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16, "gemm_bt_packed: bf16 only");
    M4D_CHECK_ARG(A && W && out && M > 0 && N > 0 && K > 0, "gemm_bt_packed: null/empty");
    M4D_CHECK_ARG(K % 64 == 0, "gemm_bt_packed: K must be a multiple of 64 (K=%lld)", (long long)K);
    M4D_CHECK_ARG(packed_side == 0 || packed_side == 1, "gemm_bt_packed: packed_side 0 (W / N side) or 1 (A / M side)");
    M4D_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "gemm_bt_packed: N and ldc must be multiples of 4");
    M4D_CHECK_ARG(((packed_side ? ldw : lda) % 8) == 0, "gemm_bt_packed: activation rows must be 16-byte aligned");
    M4D_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bt_packed: 16-byte alignment");
    M4D_CHECK_ARG(epilogue >= 0 && epilogue <= 5, "gemm_bt_packed: bad epilogue %d", epilogue);
    GemmArgs p;
    p.A = A; p.W = W; p.bias = bias; p.out = out; p.gate = gate;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.gate_stride = gate_stride; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : M;
    p.epilogue = epilogue; p.bias_on_m = bias_on_m; p.abl = 0; p.sync = nullptr; p.nb1 = 0; p.a_bs1 = p.a_bs2 = p.w_bs1 = p.w_bs2 = 0; p.tap_rows = p.tap_kh = 0; p.tap_s1 = p.tap_s2 = 0; p.remap_n = 0; p.tile_base = 0; p.ksplit = 0; p.ws = nullptr; p.tile_off = 0;
    p.tiles_m = (int)((M + PT - 1) / PT); p.tiles_n = (int)((N + PT - 1) / PT);
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    M4D_CHECK_ARG(nwg < (1ll << 31), "gemm_bt_packed: too many tiles");
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_packed_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * ASTAGE);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_packed_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * ASTAGE);
        if (e != hipSuccess) { m4d_set_error("gemm_bt_packed: cannot enable 128 KiB LDS: %s", hipGetErrorString(e)); return -3; }
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (packed_side == 0) hipLaunchKernelGGL(gemm_packed_kernel<false>, dim3((unsigned)nwg), dim3(512), NSTAGE * ASTAGE, st, p);
    else hipLaunchKernelGGL(gemm_packed_kernel<true>, dim3((unsigned)nwg), dim3(512), NSTAGE * ASTAGE, st, p);
    M4D_CHECK_LAUNCH("gemm_bt_packed");
    return 0;
}
