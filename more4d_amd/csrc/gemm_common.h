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
    int abl;   // timing ablations (tools only; results wrong when != 0): 1 no DMA, 2 frags once, 4 no barriers, 8 no MFMA
};

constexpr int ROWB = 128;  // bytes of K per LDS row

M4D_DEV int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// XCD-aware, banded tile order -> (tm, tn)
M4D_DEV void tile_coords(const GemmArgs& p, int& tm, int& tn) {
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int GM = 8;
    const int band = bid / (GM * p.tiles_n);
    const int band_rows = min(GM, p.tiles_m - band * GM);
    const int in_band = bid - band * GM * p.tiles_n;
    tm = band * GM + in_band % band_rows;
    tn = in_band / band_rows;
    if (p.abl & 16) { tm = 0; tn = 0; }          // ablation: every workgroup reads the same panels (all L2 hits)
    if (p.abl & 32) { tm = blockIdx.x % p.tiles_m; tn = blockIdx.x / p.tiles_m; }   // ablation: naive order
}

// Epilogue for one 32(n) x 32(m) accumulator tile: this lane holds column m, rows nb0 + 8*rq + 4*hi + [0,4).
template <typename T>
M4D_DEV void epilogue_tile(const GemmArgs& p, const f32x16& acc, int64_t m, int64_t nb0, int hi, float bias_m,
                           const float* grow) {
    const T* bias = (const T*)p.bias;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const int64_t nb = nb0 + rq * 8 + hi * 4;
        if (nb >= p.N) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[rq * 4 + e];
        if (bias) {
            if (p.bias_on_m) { v += bias_m; }
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

}  // namespace
