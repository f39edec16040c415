// Training-step kernels of the DiT path (train_wan.py:1891-2015): the HBM-bound halves of the
// backward pass (LayerNorm+modulate, RMSNorm+RoPE, activation derivatives, column reductions for bias /
// modulation / norm-weight gradients, tile transpose for the wgrad/dgrad GEMM operands) and the fused
// clip+AdamW update.  The GEMM-shaped halves reuse m4d_gemm_bt; attention has m4d_attention_bwd.
#include "common.h"
#include "more4d_hip.h"

namespace {

// ------------------------------------------------------------------ transpose: out[c, r] = in[r, c]
struct TrArgs { const void* in; void* out; int64_t R, C, ld_in, ld_out; };

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(TrArgs p) {
    constexpr int PAD = sizeof(T) == 2 ? 2 : 1;      // odd dword row stride: conflict-free column reads
    __shared__ T tile[64][64 + PAD];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const bool inner = r0 + 64 <= p.R && c0 + 64 <= p.C;      // (workgroup-uniform) tile entirely inside: no per-access predicates —
    if (inner) {                                              // predicated, every load sits in its own basic block and is waited for alone
        const int cx = (t & 15) * 4, ry = t >> 4;
        const T* src = (const T*)p.in + (r0 + ry) * p.ld_in + c0 + cx;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = load4(src + i * 16 * p.ld_in);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[ry + i * 16][cx + e] = (T)v[i][e];
    } else {
        const int cx = (t & 15) * 4, ry = t >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = ry + i * 16;
            const int64_t r = r0 + rr, c = c0 + cx;
            if (r < p.R) {
                const T* src = (const T*)p.in + r * p.ld_in + c;
                if (c + 3 < p.C) {
                    const f32x4 v = load4(src);
#pragma unroll
                    for (int e = 0; e < 4; ++e) tile[rr][cx + e] = (T)v[e];
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (c + e < p.C) tile[rr][cx + e] = src[e];
                }
            }
        }
    }
    __syncthreads();
    if (inner) {
        const int rx = (t & 15) * 4, cy = t >> 4;
        T* dst = (T*)p.out + (c0 + cy) * p.ld_out + r0 + rx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)tile[rx + e][cy + i * 16];
            store4(dst + i * 16 * p.ld_out, v);
        }
    } else {
        const int rx = (t & 15) * 4, cy = t >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = cy + i * 16;
            const int64_t c = c0 + cc, r = r0 + rx;
            if (c < p.C) {
                T* dst = (T*)p.out + c * p.ld_out + r;
                if (r + 3 < p.R) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)tile[rx + e][cc];
                    store4(dst, v);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (r + e < p.R) dst[e] = tile[rx + e][cc];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ grouped column sums
// out[g, c] += sum over the rows r of group g of a[r, c] (* b[r, c]);  g = r / rows_per_group
struct ColArgs { const void *a, *b; float* out; int64_t R, C, lda, ldb, rows_per_group; int chunks; };

template <typename TA, typename TB, bool HASB>
__global__ __launch_bounds__(256) void colsum_kernel(ColArgs p) {
    constexpr int RC = 256;                          // rows per workgroup
    __shared__ f32x4 part[4][64];
    const int t = threadIdx.x, lx = t & 63, ly = t >> 6;
    const int64_t c = ((int64_t)blockIdx.x * 64 + lx) * 4;
    const int64_t by = (int64_t)blockIdx.z * gridDim.y + blockIdx.y;      // (group, chunk) pairs beyond 65 535 continue in grid.z
    const int64_t g = by / p.chunks, ch = by % p.chunks;
    if (g * p.rows_per_group >= p.R) return;
    const int64_t rbeg = g * p.rows_per_group + ch * RC;
    int64_t rend = rbeg + RC;
    const int64_t gend = (g + 1) * p.rows_per_group < p.R ? (g + 1) * p.rows_per_group : p.R;
    if (rend > gend) rend = gend;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < p.C) {
        for (int64_t r = rbeg + ly; r < rend; r += 4) {
            f32x4 v = load4((const TA*)p.a + r * p.lda + c);
            if constexpr (HASB) v = v * load4((const TB*)p.b + r * p.ldb + c);
            acc += v;
        }
    }
    part[ly][lx] = acc;
    __syncthreads();
    if (ly == 0 && c < p.C) {
        acc = part[0][lx] + part[1][lx] + part[2][lx] + part[3][lx];
        float* o = p.out + g * p.C + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(o + e, acc[e]);
    }
}

// rows_per_group == 1 (per-token modulation, autograd.py: one gate / modulation vector per row): the "sum" is the row itself —
// out[r, c] += a[r, c] (* b[r, c]), every element owned by one thread: no grid.y of R groups, no atomics
template <typename TA, typename TB, bool HASB>
__global__ __launch_bounds__(256) void rowprod_kernel(ColArgs p) {
    const int64_t nv = p.C >> 2, total = p.R * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nv, c = (i % nv) * 4;
        f32x4 v = load4((const TA*)p.a + r * p.lda + c);
        if constexpr (HASB) v = v * load4((const TB*)p.b + r * p.ldb + c);
        float* o = p.out + r * p.C + c;
        store4(o, load4(o) + v);
    }
}

// ------------------------------------------------------------------ out T [R, C] = in f32 [R, C] * gate[sample, c]
struct ScArgs { const float* in; void* out; const float* gate; int64_t R, C, rows_per_sample, gate_stride; };

template <typename T>
__global__ __launch_bounds__(256) void scale_cast_kernel(ScArgs p) {
    const int64_t nv = p.C >> 2, total = p.R * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nv, c = (i % nv) * 4;
        f32x4 v = load4(p.in + r * p.C + c);
        if (p.gate) v = v * load4(p.gate + (r / p.rows_per_sample) * p.gate_stride + c);
        store4((T*)p.out + r * p.C + c, v);
    }
}

// ------------------------------------------------------------------ out f32 = x f32 + y T * gate[sample, c]   (gated residual)
struct RgArgs { const float* x; const void* y; const float* gate; float* out; int64_t R, C, rows_per_sample, gate_stride; };

template <typename T>
__global__ __launch_bounds__(256) void resid_gate_kernel(RgArgs p) {
    const int64_t nv = p.C >> 2, total = p.R * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nv, c = (i % nv) * 4;
        f32x4 v = load4((const T*)p.y + r * p.C + c);
        if (p.gate) v = v * load4(p.gate + (r / p.rows_per_sample) * p.gate_stride + c);
        store4(p.out + r * p.C + c, load4(p.x + r * p.C + c) + v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* a, const T* b, T* out, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4)
        store4(out + i, load4(a + i) + load4(b + i));
}

// ------------------------------------------------------------------ dy *= act'(pre)   (act: 1 silu, 2 gelu_tanh, 3 gelu_erf, 4 sigmoid,
// 5 sigmoid given its output, 6 clamp(-1,1))
M4D_DEV float dact(float x, int act) {
    if (act == 1) { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
    if (act == 2) {
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        const float th = tanhf(k0 * (x + k1 * x * x * x));
        return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * k0 * (1.f + 3.f * k1 * x * x);
    }
    if (act == 3) return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    if (act == 4) { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f - s); }     // sigmoid
    if (act == 5) return x * (1.f - x);                                                    // sigmoid, x = its OUTPUT
    if (act == 6) return (x >= -1.f && x <= 1.f) ? 1.f : 0.f;                              // clamp(-1, 1)
    return 1.f;
}
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(T* dy, const T* pre, int64_t n, int act) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        f32x4 d = load4(dy + i);
        const f32x4 x = load4(pre + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] *= dact(x[e], act);
        store4(dy + i, d);
    }
}

// ------------------------------------------------------------------ LayerNorm(+modulate / affine) backward
// y = xhat * m + s,  m = 1 + scale[sample] (modulated) or ln_w (affine) or 1:
//   dx += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * m
//   dshift[grp, c] += dy,  dscale[grp, c] += dy * xhat      (grp = sample for modulation, 0 for the affine weights)
// One wave walks rows l = first, first + stride, ... of one sample keeping its column partials in registers.
struct LnBwdArgs {
    const float* x; const void* dy; float* dx;
    const float *scale, *ln_w;
    float *dshift, *dscale;
    int64_t rows_per_sample, mod_stride, red_stride;   // red_stride: elements between samples in dshift/dscale (0 => shared)
    int C, B; float eps;
};

// column partials of the 4 waves of a workgroup are summed in LDS (wave after wave), then ONE wave issues the global
// atomics: 4x fewer atomics than per-wave flushing (they dominated the first version of these kernels)
template <int MAXV>
M4D_DEV void flush_partials(const f32x4 (&part)[MAXV], float* lds, float* out, int nv, int lt, int wv) {
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c4 = lt + i * 64;
                if (c4 < nv) {
                    f32x4* q = reinterpret_cast<f32x4*>(lds) + c4;
                    *q = w == 0 ? part[i] : *q + part[i];
                }
            }
        }
        __syncthreads();
    }
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * 64;
            if (c4 < nv) {
                const f32x4 v = reinterpret_cast<const f32x4*>(lds)[c4];
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(out + c4 * 4 + e, v[e]);
            }
        }
    }
    __syncthreads();
}

// FULL: C == MAXV * 256 (host-checked) — every lane owns all MAXV pieces of its row, no per-piece lane predicates.  With them each piece
// sits in a basic block of its own and hipcc waits for its loads before it requests the next piece's (elementwise.hip: the same finding
// on the forward kernels); the wave index is made scalar so that rows are addressed as scalar base + lane offset.
// PERROW: rows_per_sample == 1 (per-token modulation: every row has its own scale vector and its own dshift / dscale rows) — the waves
// of a persistent grid walk ROWS, the "column partials" of a row are its own values and are added to dshift / dscale directly: no
// grid.y of B * Lp samples (which overflows 65 535 at B * Lp beyond that, e.g. a 720p clip), no one-row workgroups, no atomics
template <typename TD, int MAXV, bool FULL = false, bool PERROW = false>
__global__ __launch_bounds__(256, 1) void ln_bwd_kernel(LnBwdArgs p) {
    constexpr int G = 64;
    extern __shared__ __attribute__((aligned(16))) float red_lds[];   // C floats
    const int lt = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t sample0 = PERROW ? 0 : (int64_t)blockIdx.z * gridDim.y + blockIdx.y;
    if (!PERROW && sample0 >= p.B) return;
    const int C = p.C, nv = C >> 2;
    const float* sc0 = p.scale ? p.scale + sample0 * p.mod_stride : nullptr;
    f32x4 ps[MAXV], pq[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { ps[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pq[i] = ps[i]; }
    const int64_t l_end = PERROW ? (int64_t)p.B : p.rows_per_sample;
    for (int64_t l = (int64_t)blockIdx.x * 4 + wv; l < l_end; l += (int64_t)gridDim.x * 4) {
        const int64_t row = PERROW ? l : sample0 * p.rows_per_sample + l;
        const float* sc = PERROW ? (p.scale ? p.scale + row * p.mod_stride : nullptr) : sc0;
        if (PERROW) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) { ps[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pq[i] = ps[i]; }
        }
        const float* xr = p.x + row * C;
        const TD* dr = (const TD*)p.dy + row * C;
        float* dxr = p.dx + row * C;
        f32x4 v[MAXV], g[MAXV], dxo[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * G;
            if (FULL || c4 < nv) {
                v[i] = load4(xr + c4 * 4);
                g[i] = load4(dr + c4 * 4);
                dxo[i] = load4(dxr + c4 * 4);     // issued with the other loads: one memory round trip per row
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * G;
            if (FULL || c4 < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + p.eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * G;
            if (FULL || c4 < nv) {
                const int c = c4 * 4;
                f32x4 xh = (v[i] - mean) * rstd;
                ps[i] += g[i];
                pq[i] += g[i] * xh;
                if (sc) g[i] = g[i] * (1.f + load4(sc + c));
                else if (p.ln_w) g[i] = g[i] * load4(p.ln_w + c);
                v[i] = xh;
#pragma unroll
                for (int e = 0; e < 4; ++e) { m1 += g[i][e]; m2 += g[i][e] * xh[e]; }
            }
        }
        m1 = wave_sum(m1) / C;
        m2 = wave_sum(m2) / C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * G;
            if (FULL || c4 < nv) store4(dxr + c4 * 4, dxo[i] + (g[i] - m1 - v[i] * m2) * rstd);
        }
        if (PERROW && p.dshift) {
            float* ds = p.dshift + row * p.red_stride;
            float* dq = p.dscale + row * p.red_stride;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c4 = lt + i * G;
                if (FULL || c4 < nv) {
                    store4(ds + c4 * 4, load4(ds + c4 * 4) + ps[i]);
                    store4(dq + c4 * 4, load4(dq + c4 * 4) + pq[i]);
                }
            }
        }
    }
    if (!PERROW && p.dshift) {
        flush_partials<MAXV>(ps, red_lds, p.dshift + sample0 * p.red_stride, nv, lt, wv);
        flush_partials<MAXV>(pq, red_lds, p.dscale + sample0 * p.red_stride, nv, lt, wv);
    }
}

// ------------------------------------------------------------------ spatial-guidance backward (wan_transformer4d.py:781)
// forward (m4d_ln_modulate with g_ss): z = u * (1 + S*g) + H*g,  u = LN(x) * (1 + scale) + shift,  (S | H) = g_ss[sample,
// l % period] for l < g_len.  Given dz this kernel rewrites it IN PLACE to du = dz * (1 + S*g) (the gradient the plain
// LayerNorm backward then takes) and writes, per spatial position, A = sum_f dz*u and Bm = sum_f dz over the frames f that
// share the position: dS = A*g, dH = Bm*g, dgate = sum (A*S + Bm*H) are cheap host-side combinations of (A | Bm).
// One workgroup per (position, sample): it owns every row of its position, so no atomics.
struct GuidBwdArgs {
    const float* x; void* dz; const float *shift, *scale, *ss, *gate; float* ab;
    int64_t rows_per_sample, mod_stride, period, glen, mod_rows;      // mod_rows: rows that share one (shift, scale) vector
    int C; float eps;
};

M4D_DEV float block_sum256(float v, float* lds4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

template <typename TD, int MAXV>
__global__ __launch_bounds__(256) void guid_bwd_kernel(GuidBwdArgs p) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int64_t pos = blockIdx.x, sample = blockIdx.y;
    const int C = p.C, nv = C >> 2;
    const float* ssr = p.ss + (sample * p.period + pos) * 2 * C;
    f32x4 A[MAXV], Bm[MAXV], m[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        A[i] = f32x4{0.f, 0.f, 0.f, 0.f}; Bm[i] = A[i]; m[i] = A[i];
        const int c4 = tid + i * 256;
        if (c4 < nv) m[i] = 1.f + load4(ssr + c4 * 4) * load4(p.gate + c4 * 4);
    }
    for (int64_t l = pos; l < p.glen; l += p.period) {
        const int64_t row = sample * p.rows_per_sample + l;
        const float* sh = p.shift + (row / p.mod_rows) * p.mod_stride;      // (per sample, or per token: mod_rows == 1)
        const float* sc = p.scale + (row / p.mod_rows) * p.mod_stride;
        const float* xr = p.x + row * C;
        TD* dr = (TD*)p.dz + row * C;
        f32x4 v[MAXV], g[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = tid + i * 256;
            if (c4 < nv) {
                v[i] = load4(xr + c4 * 4);
                g[i] = load4(dr + c4 * 4);
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
        const float mean = block_sum256(s, red) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = tid + i * 256;
            if (c4 < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(block_sum256(q, red) / C + p.eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = tid + i * 256;
            if (c4 < nv) {
                const int c = c4 * 4;
                const f32x4 u = (v[i] - mean) * rstd * (1.f + load4(sc + c)) + load4(sh + c);
                A[i] += g[i] * u;
                Bm[i] += g[i];
                store4(dr + c, g[i] * m[i]);
            }
        }
    }
    float* abr = p.ab + (sample * p.period + pos) * 2 * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = tid + i * 256;
        if (c4 < nv) { store4(abr + c4 * 4, A[i]); store4(abr + C + c4 * 4, Bm[i]); }
    }
}

// ------------------------------------------------------------------ RMSNorm(+RoPE) backward, in place on dy
// forward (m4d_rmsnorm_rope): y = rot(xhat * w), xhat = x * rsqrt(mean(x^2) + eps)
//   g = rot^T(dy);  dw[c] += g * xhat;  dx = rstd * (g*w - xhat * mean(g*w*xhat))
struct RmsBwdArgs {
    void* dy[2]; const void* x[2]; const float* w[2]; float* dw[2];
    const float *cos_t, *sin_t;
    int64_t ld_dy, ld_x, rows, rows_per_sample, rope_len, pos_offset;
    int C, head_dim; float eps;
};

template <typename T, int MAXV, bool FULL = false>      // FULL: C == MAXV * 256 and 256 % head_dim == 0 (host-checked), see ln_bwd_kernel
__global__ __launch_bounds__(256, 1) void rms_bwd_kernel(RmsBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red_lds[];   // C floats
    const int lt = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int which = blockIdx.y;
    const int C = p.C, nv = C >> 2;
    const float* w = p.w[which];
    const int half = p.head_dim >> 1;
    f32x4 pw[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) pw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < p.rows; row += (int64_t)gridDim.x * 4) {
        T* dr = (T*)p.dy[which] + row * p.ld_dy;
        const T* xr = (const T*)p.x[which] + row * p.ld_x;
        const int64_t l = row % p.rows_per_sample;
        const bool rot = p.cos_t && l < p.rope_len;
        const float* ct = rot ? p.cos_t + (p.pos_offset + l) * half : nullptr;
        const float* st = rot ? p.sin_t + (p.pos_offset + l) * half : nullptr;
        f32x4 v[MAXV], g[MAXV];
        float s = 0.f;
        f32x2 cs0 = {0.f, 0.f}, sn0 = {0.f, 0.f};      // FULL: a piece step (256 elements) is a multiple of head_dim — one pair of loads per row
        if (FULL && rot) {
            const int pi0 = ((lt * 4) % p.head_dim) >> 1;
            cs0 = *reinterpret_cast<const f32x2*>(ct + pi0);
            sn0 = *reinterpret_cast<const f32x2*>(st + pi0);
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * 64;
            if (FULL || c4 < nv) {
                const int c = c4 * 4;
                v[i] = load4(xr + c);
                g[i] = load4(dr + c);
                if (rot) {   // transpose of the forward rotation on pairs (c, c+1), (c+2, c+3)
                    const int pi = (c % p.head_dim) >> 1;
                    const f32x2 cs = FULL ? cs0 : *reinterpret_cast<const f32x2*>(ct + pi);
                    const f32x2 sn = FULL ? sn0 : *reinterpret_cast<const f32x2*>(st + pi);
                    const float a0 = g[i][0], b0 = g[i][1], a1 = g[i][2], b1 = g[i][3];
                    g[i][0] = a0 * cs[0] + b0 * sn[0];
                    g[i][1] = -a0 * sn[0] + b0 * cs[0];
                    g[i][2] = a1 * cs[1] + b1 * sn[1];
                    g[i][3] = -a1 * sn[1] + b1 * cs[1];
                }
                s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            }
        }
        if (w == nullptr) {      // qk_norm=False (reference :431-432: norm_q / norm_k are Identity): the forward only rotated -> dx = rot^T(dy)
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c4 = lt + i * 64;
                if (FULL || c4 < nv) store4(dr + c4 * 4, g[i]);
            }
            continue;
        }
        const float inv = rsqrtf(wave_sum(s) / C + p.eps);
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * 64;
            if (FULL || c4 < nv) {
                const f32x4 xh = v[i] * inv;
                pw[i] += g[i] * xh;
                g[i] = g[i] * load4(w + c4 * 4);
                v[i] = xh;
#pragma unroll
                for (int e = 0; e < 4; ++e) m2 += g[i][e] * xh[e];
            }
        }
        m2 = wave_sum(m2) / C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * 64;
            if (FULL || c4 < nv) store4(dr + c4 * 4, (g[i] - v[i] * m2) * inv);
        }
    }
    if (w != nullptr) flush_partials<MAXV>(pw, red_lds, p.dw[which], nv, lt, wv);
}

// ------------------------------------------------------------------ sum of squares (global gradient norm)
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* x, int64_t n, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n & ~(int64_t)3;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n4; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const f32x4 v = load4(x + i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4)) { const float v = (float)x[n4 + threadIdx.x]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------ fused clip + AdamW (torch.optim.AdamW semantics)
struct AdamArgs {
    void *p, *m, *v; const void* g;
    int64_t n;
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
    const float* grad_scale;     // optional device scalar multiplied into the gradient (clip coefficient)
};
// one element's update; the contraction is pinned (no fma) so that the element-wise and the eight-per-thread kernel round alike
M4D_DEV void adamw_update(const AdamArgs& a, float g, float& w, float& m, float& v) {
#pragma clang fp contract(off)
    w *= 1.f - a.lr * a.wd;
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    w -= (a.lr / a.bc1) * (m / denom);
}
template <typename T, typename TS>
__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    const float gs = a.grad_scale ? *a.grad_scale : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = (float)((const T*)a.g)[i] * gs;
        float w = (float)((T*)a.p)[i];
        float m = (float)((TS*)a.m)[i], v = (float)((TS*)a.v)[i];
        adamw_update(a, g, w, m, v);
        ((T*)a.p)[i] = (T)w;
        ((TS*)a.m)[i] = (TS)m;
        ((TS*)a.v)[i] = (TS)v;
    }
}

// eight elements per thread and iteration, 16-byte (bf16) / 2 x 16-byte (fp32) accesses; same arithmetic per element as adamw_kernel
template <typename T> M4D_DEV void ld8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
}
template <typename T> M4D_DEV void st8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x8*>(p) = r;
    } else {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
}
template <typename T, typename TS>
__global__ __launch_bounds__(256) void adamw_vec8_kernel(AdamArgs a) {
    const float gs = a.grad_scale ? *a.grad_scale : 1.f;
    const int64_t n8 = a.n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float g[8], w[8], m[8], v[8];
        ld8((const T*)a.g + i * 8, g); ld8((const T*)a.p + i * 8, w); ld8((const TS*)a.m + i * 8, m); ld8((const TS*)a.v + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) adamw_update(a, g[e] * gs, w[e], m[e], v[e]);
        st8((T*)a.p + i * 8, w); st8((TS*)a.m + i * 8, m); st8((TS*)a.v + i * 8, v);
    }
}

inline unsigned grid_for(int64_t work, int per_block, unsigned cap = 65535u * 4) {
    int64_t g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

}  // namespace

#define DT_OK(dt) ((dt) == M4D_BF16 || (dt) == M4D_F32)

extern "C" int m4d_transpose(m4d_dtype dt, const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C,
                             m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dt), "transpose: bad dtype");
    M4D_CHECK_ARG(in && out && R > 0 && C > 0, "transpose: bad arguments");
    M4D_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= C && ld_out >= R, "transpose: leading dims must be multiples of 4 and cover the rows");
    const uintptr_t al = dt == M4D_BF16 ? 8 : 16;
    M4D_CHECK_ARG(((uintptr_t)in % al) == 0 && ((uintptr_t)out % al) == 0, "transpose: tensors must be aligned to 4 elements");
    TrArgs p{in, out, R, C, ld_in, ld_out};
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64)), block(256);
    M4D_CHECK_ARG(grid.y <= 65535u, "transpose: too many rows (%lld)", (long long)R);
    if (dt == M4D_BF16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("transpose");
    return 0;
}

extern "C" int m4d_colsum(m4d_dtype a_dt, const void* a, int64_t lda, m4d_dtype b_dt, const void* b, int64_t ldb, float* out,
                          int64_t R, int64_t C, int64_t rows_per_group, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(a_dt) && (!b || DT_OK(b_dt)), "colsum: bad dtype");
    M4D_CHECK_ARG(a && out && R > 0 && C > 0 && rows_per_group > 0, "colsum: bad arguments");
    M4D_CHECK_ARG(C % 4 == 0 && lda % 4 == 0 && (!b || ldb % 4 == 0), "colsum: C and leading dims must be multiples of 4");
    const int64_t G = (R + rows_per_group - 1) / rows_per_group;
    ColArgs p{a, b, out, R, C, lda, ldb, rows_per_group, (int)((rows_per_group + 255) / 256)};
    const int64_t gy = G * p.chunks;
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)(gy < 65535 ? gy : 65535), (unsigned)((gy + 65534) / 65535)), block(256);
    M4D_CHECK_ARG(grid.z <= 65535u, "colsum: too many row chunks");
    hipStream_t st = (hipStream_t)stream;
    if (rows_per_group == 1) {      // one group per row: elementwise
        dim3 g1(grid_for(R * (C / 4), 256, 16384));
#define ROWPROD(TA, TB, HB) hipLaunchKernelGGL((rowprod_kernel<TA, TB, HB>), g1, block, 0, st, p)
        if (!b) { if (a_dt == M4D_BF16) ROWPROD(bf16_t, bf16_t, false); else ROWPROD(float, float, false); }
        else if (a_dt == M4D_BF16 && b_dt == M4D_BF16) ROWPROD(bf16_t, bf16_t, true);
        else if (a_dt == M4D_F32 && b_dt == M4D_BF16) ROWPROD(float, bf16_t, true);
        else if (a_dt == M4D_BF16 && b_dt == M4D_F32) ROWPROD(bf16_t, float, true);
        else ROWPROD(float, float, true);
#undef ROWPROD
        M4D_CHECK_LAUNCH("colsum(rows)");
        return 0;
    }
#define COLSUM(TA, TB, HB) hipLaunchKernelGGL((colsum_kernel<TA, TB, HB>), grid, block, 0, st, p)
    if (!b) { if (a_dt == M4D_BF16) COLSUM(bf16_t, bf16_t, false); else COLSUM(float, float, false); }
    else if (a_dt == M4D_BF16 && b_dt == M4D_BF16) COLSUM(bf16_t, bf16_t, true);
    else if (a_dt == M4D_F32 && b_dt == M4D_BF16) COLSUM(float, bf16_t, true);
    else if (a_dt == M4D_BF16 && b_dt == M4D_F32) COLSUM(bf16_t, float, true);
    else COLSUM(float, float, true);
#undef COLSUM
    M4D_CHECK_LAUNCH("colsum");
    return 0;
}

extern "C" int m4d_scale_cast(const float* in, const float* gate, int64_t gate_stride, int64_t rows_per_sample,
                              m4d_dtype out_dt, void* out, int64_t R, int64_t C, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(out_dt), "scale_cast: bad dtype");
    M4D_CHECK_ARG(in && out && R > 0 && C > 0 && C % 4 == 0 && rows_per_sample > 0, "scale_cast: bad arguments");
    ScArgs p{in, out, gate, R, C, rows_per_sample, gate_stride};
    dim3 grid(grid_for(R * (C / 4), 256, 16384)), block(256);
    if (out_dt == M4D_BF16) hipLaunchKernelGGL(scale_cast_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(scale_cast_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("scale_cast");
    return 0;
}

extern "C" int m4d_resid_gate(const float* x, m4d_dtype y_dt, const void* y, const float* gate, int64_t gate_stride,
                              int64_t rows_per_sample, float* out, int64_t R, int64_t C, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(y_dt), "resid_gate: bad dtype");
    M4D_CHECK_ARG(x && y && out && R > 0 && C > 0 && C % 4 == 0 && rows_per_sample > 0, "resid_gate: bad arguments");
    RgArgs p{x, y, gate, out, R, C, rows_per_sample, gate_stride};
    dim3 grid(grid_for(R * (C / 4), 256, 16384)), block(256);
    if (y_dt == M4D_BF16) hipLaunchKernelGGL(resid_gate_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(resid_gate_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("resid_gate");
    return 0;
}

extern "C" int m4d_add(m4d_dtype dt, const void* a, const void* b, void* out, int64_t n, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dt), "add: bad dtype");
    M4D_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0, "add: bad arguments");
    dim3 grid(grid_for(n / 4, 256, 16384)), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    else hipLaunchKernelGGL(add_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)out, n);
    M4D_CHECK_LAUNCH("add");
    return 0;
}

extern "C" int m4d_act_bwd(m4d_dtype dt, void* dy, const void* pre, int64_t n, int act, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dt), "act_bwd: bad dtype");
    M4D_CHECK_ARG(dy && pre && n > 0 && n % 4 == 0 && act >= 1 && act <= 6, "act_bwd: bad arguments");
    dim3 grid(grid_for(n / 4, 256, 16384)), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (bf16_t*)dy, (const bf16_t*)pre, n, act);
    else hipLaunchKernelGGL(act_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (float*)dy, (const float*)pre, n, act);
    M4D_CHECK_LAUNCH("act_bwd");
    return 0;
}

extern "C" int m4d_ln_modulate_bwd(const float* x, m4d_dtype dy_dt, const void* dy, float* dx, int B, int64_t rows_per_sample,
                                   int C, const float* scale, int64_t mod_stride, const float* ln_w, float eps, float* dshift,
                                   float* dscale, int64_t red_stride, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dy_dt), "ln_modulate_bwd: bad dtype");
    M4D_CHECK_ARG(x && dy && dx && B > 0 && rows_per_sample > 0, "ln_modulate_bwd: bad arguments");
    M4D_CHECK_ARG(C % 4 == 0 && C <= 8192, "ln_modulate_bwd: C=%d must be a multiple of 4 and <= 8192", C);
    M4D_CHECK_ARG(!(scale && ln_w), "ln_modulate_bwd: scale and ln_w are exclusive");
    M4D_CHECK_ARG((dshift == nullptr) == (dscale == nullptr), "ln_modulate_bwd: dshift and dscale come together");
    LnBwdArgs p{x, dy, dx, scale, ln_w, dshift, dscale, rows_per_sample, mod_stride, red_stride, C, B, eps};
    int64_t nb = (rows_per_sample + 3) / 4;
    const int64_t cap = (512 + B - 1) / B;      // two workgroups per CU chip-wide; more only adds atomics
    if (nb > cap) nb = cap;
    dim3 grid((unsigned)nb, (unsigned)(B < 65535 ? B : 65535), (unsigned)((B + 65534) / 65535)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)C * sizeof(float);
    const bool bf = dy_dt == M4D_BF16;
    if (rows_per_sample == 1 && red_stride != 0) {      // one modulation vector per row (and per-row dshift / dscale): the per-row form
        dim3 gr((unsigned)((B + 3) / 4 < 1024 ? (B + 3) / 4 : 1024));
#define LNR(TD, MV, FL) hipLaunchKernelGGL((ln_bwd_kernel<TD, MV, FL, true>), gr, block, lds, st, p)
        if (C == 5120) { if (bf) LNR(bf16_t, 20, true); else LNR(float, 20, true); }
        else if (C <= 2048) { if (bf) LNR(bf16_t, 8, false); else LNR(float, 8, false); }
        else if (C <= 5120) { if (bf) LNR(bf16_t, 20, false); else LNR(float, 20, false); }
        else { if (bf) LNR(bf16_t, 32, false); else LNR(float, 32, false); }
#undef LNR
        M4D_CHECK_LAUNCH("ln_modulate_bwd(rows)");
        return 0;
    }
#define LNB(TD, MV) hipLaunchKernelGGL((ln_bwd_kernel<TD, MV>), grid, block, lds, st, p)
    M4D_ENV_ONCE(full_ok, "M4D_TRAIN_ROWS_FULL", 1);      // 0: the predicated kernels (A/B)
    if (C == 5120 && full_ok) { if (bf) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 20, true>), grid, block, lds, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<float, 20, true>), grid, block, lds, st, p); }
    else if (C <= 2048) { if (bf) LNB(bf16_t, 8); else LNB(float, 8); }
    else if (C <= 5120) { if (bf) LNB(bf16_t, 20); else LNB(float, 20); }
    else { if (bf) LNB(bf16_t, 32); else LNB(float, 32); }
#undef LNB
    M4D_CHECK_LAUNCH("ln_modulate_bwd");
    return 0;
}

extern "C" int m4d_guidance_bwd(const float* x, m4d_dtype dz_dt, void* dz, int B, int64_t rows_per_sample, int C,
                                const float* shift, const float* scale, int64_t mod_stride, float eps, const float* g_ss,
                                const float* g_gate, int64_t g_period, int64_t g_len, float* ab, m4d_stream stream) {
    return m4d_guidance_bwd_m(x, dz_dt, dz, B, rows_per_sample, C, shift, scale, mod_stride, 0, eps, g_ss, g_gate, g_period, g_len, ab, stream);
}

extern "C" int m4d_guidance_bwd_m(const float* x, m4d_dtype dz_dt, void* dz, int B, int64_t rows_per_sample, int C,
                                  const float* shift, const float* scale, int64_t mod_stride, int64_t mod_rows, float eps, const float* g_ss,
                                  const float* g_gate, int64_t g_period, int64_t g_len, float* ab, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dz_dt), "guidance_bwd: bad dtype");
    M4D_CHECK_ARG(x && dz && shift && scale && g_ss && g_gate && ab && B > 0 && rows_per_sample > 0, "guidance_bwd: bad arguments");
    M4D_CHECK_ARG(C % 4 == 0 && C <= 8192, "guidance_bwd: C=%d must be a multiple of 4 and <= 8192", C);
    M4D_CHECK_ARG(g_period > 0 && g_len >= 0 && g_len <= rows_per_sample, "guidance_bwd: bad period / length");
    GuidBwdArgs p{x, dz, shift, scale, g_ss, g_gate, ab, rows_per_sample, mod_stride, g_period, g_len, mod_rows > 0 ? mod_rows : rows_per_sample, C, eps};
    dim3 grid((unsigned)g_period, (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define GDB(TD, MV) hipLaunchKernelGGL((guid_bwd_kernel<TD, MV>), grid, block, 0, st, p)
    const bool bf = dz_dt == M4D_BF16;
    if (C <= 2048) { if (bf) GDB(bf16_t, 2); else GDB(float, 2); }
    else if (C <= 5120) { if (bf) GDB(bf16_t, 5); else GDB(float, 5); }
    else { if (bf) GDB(bf16_t, 8); else GDB(float, 8); }
#undef GDB
    M4D_CHECK_LAUNCH("guidance_bwd");
    return 0;
}

extern "C" int m4d_rmsnorm_rope_bwd(m4d_dtype dt, void* dy0, void* dy1, int64_t ld_dy, const void* x0, const void* x1,
                                    int64_t ld_x, const float* w0, const float* w1, float* dw0, float* dw1, int64_t rows,
                                    int C, int head_dim, float eps, const float* cos_t, const float* sin_t,
                                    int64_t rows_per_sample, int64_t rope_len, int64_t pos_offset, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dt), "rmsnorm_rope_bwd: bad dtype");
    M4D_CHECK_ARG(dy0 && x0 && rows > 0 && (w0 == nullptr) == (dw0 == nullptr), "rmsnorm_rope_bwd: bad arguments");      // w == NULL: rotation only
    M4D_CHECK_ARG((dy1 == nullptr) == (x1 == nullptr) && (!dy1 || ((w1 == nullptr) == (dw1 == nullptr))), "rmsnorm_rope_bwd: second tensor incomplete");
    M4D_CHECK_ARG(C % 4 == 0 && C <= 8192 && ld_dy % 4 == 0 && ld_x % 4 == 0, "rmsnorm_rope_bwd: C / leading dims must be multiples of 4, C <= 8192");
    M4D_CHECK_ARG(!cos_t || (head_dim % 4 == 0 && C % head_dim == 0 && sin_t), "rmsnorm_rope_bwd: bad rope configuration");
    RmsBwdArgs p;
    p.dy[0] = dy0; p.dy[1] = dy1; p.x[0] = x0; p.x[1] = x1; p.w[0] = w0; p.w[1] = w1; p.dw[0] = dw0; p.dw[1] = dw1;
    p.cos_t = cos_t; p.sin_t = sin_t;
    p.ld_dy = ld_dy; p.ld_x = ld_x; p.rows = rows; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : rows;
    p.rope_len = rope_len; p.pos_offset = pos_offset; p.C = C; p.head_dim = head_dim > 0 ? head_dim : 4; p.eps = eps;
    int64_t nb = (rows + 3) / 4;
    if (nb > 256) nb = 256;
    dim3 grid((unsigned)nb, dy1 ? 2u : 1u), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)C * sizeof(float);
#define RMB(T, MV) hipLaunchKernelGGL((rms_bwd_kernel<T, MV>), grid, block, lds, st, p)
    const bool bf = dt == M4D_BF16;
    M4D_ENV_ONCE(full_ok, "M4D_TRAIN_ROWS_FULL", 1);      // 0: the predicated kernels (A/B)
    if (C == 5120 && full_ok && 256 % p.head_dim == 0) { if (bf) hipLaunchKernelGGL((rms_bwd_kernel<bf16_t, 20, true>), grid, block, lds, st, p); else hipLaunchKernelGGL((rms_bwd_kernel<float, 20, true>), grid, block, lds, st, p); }
    else if (C <= 2048) { if (bf) RMB(bf16_t, 8); else RMB(float, 8); }
    else if (C <= 5120) { if (bf) RMB(bf16_t, 20); else RMB(float, 20); }
    else { if (bf) RMB(bf16_t, 32); else RMB(float, 32); }
#undef RMB
    M4D_CHECK_LAUNCH("rmsnorm_rope_bwd");
    return 0;
}

extern "C" int m4d_sumsq(m4d_dtype dt, const void* x, int64_t n, float* out, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dt), "sumsq: bad dtype");
    M4D_CHECK_ARG(x && out && n > 0, "sumsq: bad arguments");
    M4D_CHECK_ARG(((uintptr_t)x % 16) == 0, "sumsq: tensor must be 16-byte aligned");
    dim3 grid(grid_for(n / 4 + 1, 256, 2048)), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL(sumsq_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, n, out);
    else hipLaunchKernelGGL(sumsq_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, n, out);
    M4D_CHECK_LAUNCH("sumsq");
    return 0;
}

extern "C" int m4d_adamw(m4d_dtype dt, void* param, const void* grad, m4d_dtype state_dt, void* exp_avg, void* exp_avg_sq,
                         int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                         const float* grad_scale, m4d_stream stream) {
    M4D_CHECK_ARG(DT_OK(dt) && DT_OK(state_dt), "adamw: bad dtype");
    M4D_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw: bad arguments");
    M4D_CHECK_ARG(state_dt == M4D_F32 || state_dt == dt, "adamw: optimizer state must be float32 or the parameter dtype");
    AdamArgs a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.grad_scale = grad_scale;
    dim3 grid(grid_for(n, 256, 8192)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (n % 8 == 0 && n >= (1 << 16) && (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0) {
        const dim3 g8(grid_for(n / 8, 256, 8192));
        if (dt == M4D_BF16 && state_dt == M4D_BF16) hipLaunchKernelGGL((adamw_vec8_kernel<bf16_t, bf16_t>), g8, block, 0, st, a);
        else if (dt == M4D_BF16) hipLaunchKernelGGL((adamw_vec8_kernel<bf16_t, float>), g8, block, 0, st, a);
        else hipLaunchKernelGGL((adamw_vec8_kernel<float, float>), g8, block, 0, st, a);
        M4D_CHECK_LAUNCH("adamw");
        return 0;
    }
    if (dt == M4D_BF16 && state_dt == M4D_BF16) hipLaunchKernelGGL((adamw_kernel<bf16_t, bf16_t>), grid, block, 0, st, a);
    else if (dt == M4D_BF16) hipLaunchKernelGGL((adamw_kernel<bf16_t, float>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((adamw_kernel<float, float>), grid, block, 0, st, a);
    M4D_CHECK_LAUNCH("adamw");
    return 0;
}
