// m4d_gemm_bt: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues, gfx950 MFMA.
//
// Structure (v1): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave as
// 2x2 MFMA 32x32 tiles), K-tile of 128 BYTES per row (64 bf16 / 32 fp32), two LDS stages (64 KiB ->
// 2 workgroups per CU), register-staged global->LDS copies issued one tile ahead, one barrier per
// K-tile.  LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 so that the
// ds_read_b128 of an MFMA fragment column is bank-conflict free (cdna guide §6 G4).
// The MFMA is issued with W as the "A" operand and A as the "B" operand, so every lane ends up with
// 4 CONSECUTIVE n for one m: epilogue loads/stores are 8-16 B per lane and bias/gate are vector loads.
// Tiles are ordered XCD-aware (block b runs on XCD b%8): each XCD walks a contiguous band of tiles so
// that neighbouring tiles share A/W panels in that XCD's L2.
#include "common.h"
#include "more4d_hip.h"

namespace {

struct GemmArgs {
    const void* A; const void* W; const void* bias; void* out;
    const float* gate;
    int64_t lda, ldw, ldc, M, N, K, gate_stride, rows_per_sample;
    int epilogue, bias_on_m;
    int tiles_m, tiles_n;
};

constexpr int BM = 128, BN = 128, ROWB = 128;          // tile rows, bytes of K per LDS row
constexpr int STAGE_BYTES = (BM + BN) * ROWB;          // 32 KiB

M4D_DEV int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_bt_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    constexpr int ES = sizeof(T);
    constexpr int KT = ROWB / ES;        // K elements per tile
    constexpr int KSTEPS = KT / 16;      // MFMA K=16 steps per tile
    typedef typename Frag8<T>::type frag_t;

    // ---- XCD-aware tile order ----
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GM = 8;  // tiles_m band height: 8 x tiles_n tiles share 8 A panels
    const int band = bid / (GM * p.tiles_n);
    const int band_rows = min(GM, p.tiles_m - band * GM);
    const int in_band = bid - band * GM * p.tiles_n;
    const int tm = band * GM + in_band % band_rows;
    const int tn = in_band / band_rows;
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

    // ---- epilogue: lane holds n = nb + [0..4), m fixed, per (ni, mi, rq) ----
    const T* bias = (const T*)p.bias;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + li;
        if (m >= p.M) continue;
        float bm = 0.f;
        if (bias && p.bias_on_m) bm = (float)bias[m];
        const float* grow = nullptr;
        if (p.epilogue == M4D_EPI_RESID_GATE && p.gate) grow = p.gate + (m / p.rows_per_sample) * p.gate_stride;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int64_t nb = n0 + wn * 64 + ni * 32 + rq * 8 + hi * 4;
                if (nb >= p.N) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][rq * 4 + e];
                if (bias) {
                    if (p.bias_on_m) { v += bm; }
                    else { v += load4(bias + nb); }
                }
                switch (p.epilogue) {
                    case M4D_EPI_GELU_TANH:
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
                        break;
                    case M4D_EPI_GELU_ERF:
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                        break;
                    case M4D_EPI_SILU:
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                        break;
                    default: break;
                }
                if (p.epilogue == M4D_EPI_RESID_GATE) {
                    float* r = (float*)p.out + m * p.ldc + nb;
                    f32x4 x = load4(r);
                    f32x4 g = {1.f, 1.f, 1.f, 1.f};
                    if (grow) g = load4(grow + nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] += round_through<T>(v[e]) * g[e];
                    store4(r, x);
                } else if (p.epilogue == M4D_EPI_STORE_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]);
                    store4((float*)p.out + m * p.ldc + nb, v);
                } else {
                    store4((T*)p.out + m * p.ldc + nb, v);
                }
            }
        }
    }
}

}  // namespace

extern "C" int m4d_gemm_bt(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                           int bias_on_m, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                           const float* gate, int64_t gate_stride, int64_t rows_per_sample, m4d_stream stream) {
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
    p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    M4D_CHECK_ARG(nwg < (1ll << 31), "gemm_bt: too many tiles");
    dim3 grid((unsigned)nwg), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL(gemm_bt_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_bt_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("gemm_bt");
    return 0;
}
