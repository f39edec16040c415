// Shared pieces of the GEMM kernels (gemm.hip, gemm_packed.hip): argument block, LDS swizzle, XCD-aware tile order,
// fused epilogue.  See gemm.hip for the design notes.
#pragma once
#include "common.h"
#include "more4d_hip.h"

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

struct GemmArgs {
    const void* A; const void* W; const void* bias; void* out;
    const float* gate;
    int64_t lda, ldw, ldc, M, N, K, gate_stride, rows_per_sample;
    int epilogue, bias_on_m;
    int tiles_m, tiles_n;
    int64_t a_bs1, a_bs2, w_bs1, w_bs2;   // batched launches (gridDim.y = nb1 * nb2): element offsets of batch (i1, i2); out += batch * M * ldc
    int nb1;                               // 0 = not batched
    // stacked taps (m4d_gemm_bt_taps, the conv weight gradient on the wide kernel): the M rows are tap_rows-row groups t = dt * tap_kh + dh
    // of the SAME tap_rows rows of A, read tap_s1 * dt + tap_s2 * dh elements further along K (0 = off)
    int tap_rows, tap_kh;
    int64_t tap_s1, tap_s2;
    // split-K tail (m4d_gemm_bt_ws): the launch over the full tile rounds uses `remap_n` (< tiles_m*tiles_n) logical tiles; the tail
    // launch has ksplit > 0: block b computes K-slice b % ksplit of logical tile tile_base + b / ksplit into its float32 slab of `ws`
    int remap_n, tile_base, ksplit;
    int tile_off;   // chunked launches (M4D_GEMM_CHUNK): this launch covers logical tiles [tile_off, tile_off + remap_n)
    float* ws;
    int abl;   // timing ablations (tools only; results wrong when != 0): 1 no DMA, 2 frags once, 4 no barriers, 8 no MFMA
    unsigned* sync;   // persistent wide kernel: 8 arrival counters (one per XCD, 128 bytes apart, zeroed before the launch) or nullptr
};

constexpr int ROWB = 128;  // bytes of K per LDS row

M4D_DEV int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// XCD-aware, banded tile order -> (tm, tn)
M4D_DEV void tile_coords(const GemmArgs& p, int& tm, int& tn, int bid) {
    if (p.ksplit > 0) bid = p.tile_base + bid / p.ksplit;      // tail launch: logical tile id, no XCD remap
    else {
        const int nwg = p.remap_n > 0 ? p.remap_n : p.tiles_m * p.tiles_n;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx + p.tile_off;
    }
    constexpr int GM = 8;
    const int band = bid / (GM * p.tiles_n);
    const int band_rows = min(GM, p.tiles_m - band * GM);
    const int in_band = bid - band * GM * p.tiles_n;
    tm = band * GM + in_band % band_rows;
    tn = in_band / band_rows;
    if (M4D_ABL(p) & 16) { tm = 0; tn = 0; }          // ablation: every workgroup reads the same panels (all L2 hits)
    if (M4D_ABL(p) & 32) { tm = blockIdx.x % p.tiles_m; tn = blockIdx.x / p.tiles_m; }   // ablation: naive order
}
M4D_DEV void tile_coords(const GemmArgs& p, int& tm, int& tn) { tile_coords(p, tm, tn, (int)blockIdx.x); }

// Epilogue for one 32(n) x 32(m) accumulator tile: this lane holds column m, rows nb0 + 8*rq + 4*hi + [0,4).
// bf16 outputs are widened to 16-byte stores: a v_permlane32_swap per dword exchanges the 4-column groups of the two
// half-waves so lanes 0-31 own columns 16j..16j+7 and lanes 32-63 columns 16j+8..16j+15 of their row (cdna guide T21:
// the store tail is issue-bound, half the instructions = half the tail).
template <typename T>
M4D_DEV void epilogue_tile(const GemmArgs& p, const f32x16& acc, int64_t m, int64_t nb0, int hi, float bias_m,
                           const float* grow, int64_t n_lo = 0) {   // columns below n_lo belong to the neighbouring tile
    const T* bias = (const T*)p.bias;
    f32x4 v[4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const int64_t nb = nb0 + rq * 8 + hi * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[rq][e] = acc[rq * 4 + e];
        if (bias) {
            if (p.bias_on_m) { v[rq] += bias_m; }
            else if (nb < p.N) { v[rq] += load4(bias + nb); }
        }
        switch (p.epilogue) {
            case M4D_EPI_GELU_TANH:
#pragma unroll
                for (int e = 0; e < 4; ++e) v[rq][e] = gelu_tanh_f(v[rq][e]);
                break;
            case M4D_EPI_GELU_ERF:
#pragma unroll
                for (int e = 0; e < 4; ++e) v[rq][e] = gelu_erf_f(v[rq][e]);
                break;
            case M4D_EPI_SILU:
#pragma unroll
                for (int e = 0; e < 4; ++e) v[rq][e] = silu_f(v[rq][e]);
                break;
            default: break;
        }
    }
    if (p.epilogue == M4D_EPI_RESID_GATE) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int64_t nb = nb0 + rq * 8 + hi * 4;
            if (nb >= p.N || nb < n_lo) continue;
            float* r = (float*)p.out + m * p.ldc + nb;
            f32x4 x = load4(r);
            f32x4 g = {1.f, 1.f, 1.f, 1.f};
            if (grow) g = load4(grow + nb);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] += round_through<T>(v[rq][e]) * g[e];
            store4(r, x);
        }
    } else if (p.epilogue == M4D_EPI_STORE_F32) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int64_t nb = nb0 + rq * 8 + hi * 4;
            if (nb >= p.N) continue;
            if (!p.nb1) {      // (batched split-K partial sums keep their float32 accumulators)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[rq][e] = round_through<T>(v[rq][e]);
            }
            store4((float*)p.out + m * p.ldc + nb, v[rq]);
        }
    } else {
        if constexpr (sizeof(T) == 2) {
            if ((p.N & 7) == 0 && (p.ldc & 7) == 0) {     // both half-waves of a row take the same branch: N % 8 == 0
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { bf16x4 h; unsigned u[2]; } a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a.h[e] = (bf16_t)v[2 * j][e]; b.h[e] = (bf16_t)v[2 * j + 1][e]; }
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a.u[0], b.u[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a.u[1], b.u[1], false, false);
                    const int64_t nb = nb0 + j * 16 + hi * 8;
                    if (nb < p.N)
                        *reinterpret_cast<uint4*>((T*)p.out + m * p.ldc + nb) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
                return;
            }
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int64_t nb = nb0 + rq * 8 + hi * 4;
            if (nb >= p.N) continue;
            store4((T*)p.out + m * p.ldc + nb, v[rq]);
        }
    }
}

// wave-uniform 64-bit value forced into SGPRs (keeps the DMA in its  vgpr_offset + sgpr_base  addressing form)
M4D_DEV const char* uniform_ptr(const char* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// Epilogue of one 64(m) x 64(n) half of a wave's sub-tile THROUGH LDS.  The accumulator layout gives every lane its own
// output ROW (lane = m), so direct stores touch 64 different rows per instruction (16-byte pieces of 64 cache lines:
// measured 1.4 TB/s, 16 % of a K = 5120 GEMM).  Here the half-tile is written to a wave-private LDS region with bias /
// activation / gate applied, then read back row-wise so that 8 (bf16) or 16 (fp32) consecutive lanes cover one
// contiguous 128 / 256-byte row segment of the output.  16-byte chunks are XOR-swizzled by the row index.
template <typename T>
M4D_DEV void epilogue_half_lds(const GemmArgs& p, char* wl, const f32x16& a00, const f32x16& a01, const f32x16& a10,
                               const f32x16& a11, int64_t m_base, int64_t n_base, int64_t m_lo, int64_t n_lo, int lane) {
    // a[ni][mi2]: a00 = (ni 0, mi2 0), a01 = (ni 0, mi2 1), a10 = (ni 1, mi2 0), a11 = (ni 1, mi2 1)
    const int li = lane & 31, hi = lane >> 5;
    const T* bias = (const T*)p.bias;
    const bool f32out = p.epilogue == M4D_EPI_RESID_GATE || p.epilogue == M4D_EPI_STORE_F32;
#pragma unroll
    for (int mi2 = 0; mi2 < 2; ++mi2) {
        const int r = mi2 * 32 + li;
        const int64_t m = m_base + r;
        const float bm = (bias && p.bias_on_m) ? (float)bias[m] : 0.f;
        const float* grow = (p.epilogue == M4D_EPI_RESID_GATE && p.gate) ? p.gate + (m / p.rows_per_sample) * p.gate_stride : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const f32x16& acc = ni == 0 ? (mi2 == 0 ? a00 : a01) : (mi2 == 0 ? a10 : a11);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int nl = ni * 32 + rq * 8 + hi * 4;
                const int64_t nb = n_base + nl;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[rq * 4 + e];
                if (bias) { if (p.bias_on_m) v += bm; else v += load4(bias + nb); }
                if (p.epilogue == M4D_EPI_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
                } else if (p.epilogue == M4D_EPI_GELU_ERF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                } else if (p.epilogue == M4D_EPI_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                }
                if (f32out) {
                    if (!p.nb1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]);
                    }
                    if (grow) v = v * load4(grow + nb);
                    const int ch = nl >> 2;                                   // 16 chunks of 4 floats per 256-byte row
                    *reinterpret_cast<f32x4*>(wl + r * 256 + ((ch ^ (r & 15)) << 4)) = v;
                } else {
                    const int ch = nl >> 3;                                   // 8 chunks of 8 bf16 per 128-byte row
                    store4(reinterpret_cast<T*>(wl + r * 128 + ((ch ^ (r & 7)) << 4) + (nl & 4) * 2), v);
                }
            }
        }
    }
    // wave-private region: program order + the compiler's lgkmcnt wait order the reads after the writes
    if (f32out) {
        const int ch = lane & 15;
        const int64_t nb = n_base + ch * 4;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 4 + (lane >> 4);
            const int64_t m = m_base + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(wl + r * 256 + ((ch ^ (r & 15)) << 4));
            if (m < m_lo || nb < n_lo) continue;
            float* dst = (float*)p.out + m * p.ldc + nb;
            if (p.epilogue == M4D_EPI_RESID_GATE) v += load4(dst);
            store4(dst, v);
        }
    } else {
        const int ch = lane & 7;
        const int64_t nb = n_base + ch * 8;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 8 + (lane >> 3);
            const int64_t m = m_base + r;
            const uint4 v = *reinterpret_cast<const uint4*>(wl + r * 128 + ((ch ^ (r & 7)) << 4));
            if (m < m_lo || nb + 8 <= n_lo) continue;      // a chunk straddling n_lo rewrites identical values (benign)
            *reinterpret_cast<uint4*>((T*)p.out + m * p.ldc + nb) = v;
        }
    }
}

}  // namespace
