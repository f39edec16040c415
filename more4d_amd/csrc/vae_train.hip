// Backward-side kernels of the Motion-Sensitive VAE and its trajectory adaptors (more4d_amd/vae_autograd.py; reference
// wan_vae.py:549-676 `encode_full` / `decode_full`, trajectory_module.py:101-279, train_vae.py:434-495), channels-last:
//   pad_transpose     channels-last frames -> zero-padded pixel-major panels (operands of the conv weight-gradient GEMMs)
//   wgrad_reduce      split-K partial products -> packed-layout weight gradient
//   rmsnorm_silu_bwd  RMS_norm (+SiLU) backward with the gamma gradient
//   softmax_rows_bwd  dS = scale * P * (dP - rowsum(P dP))   (mid-block attention)
//   upsample2x        nearest-exact 2x (+ channel-half -> frame interleave) and its transpose
//   groupnorm_bwd     GroupNorm(32)(+swish) backward: statistics pass, reduction pass, apply pass
// All HBM-bound: 16-byte vector accesses along the channel axis, grid-stride loops, column reductions kept in registers / LDS with
// one atomic per channel per workgroup.
#include "common.h"
#include "more4d_hip.h"

namespace {

inline unsigned grid_for(int64_t n, int per_block = 256, int64_t cap = 8192) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}

// ------------------------------------------------------------------ pad + shift + transpose
// out[(s*C + c) * ld_out + q] = P[q + s][c], P = frames zero-padded to [T, Hp, Wp] (image at (pad_top, pad_left)), flattened.
// One workgroup: 64 consecutive q x 64 channels x one shift, through a padded LDS tile.
struct PadTArgs {
    const void* src; void* out;
    int64_t ps, ld_out, cols;
    int C, T, H, W, Hp, Wp, pad_top, pad_left;
};

template <typename T>
__global__ __launch_bounds__(256) void pad_transpose_kernel(PadTArgs p) {
    constexpr int EPV = 16 / sizeof(T);                 // elements per 16-byte vector
    __shared__ __attribute__((aligned(16))) T tile[64][64 + EPV];
    const int s = blockIdx.z, c0 = blockIdx.y * 64;
    const int64_t q0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x;
    {   // load: pixel t/4, 16 channels starting at (t%4)*16, as 16-byte vectors when the source allows it
        const int px = t >> 2, cpart = (t & 3) * 16;
        const int64_t q = q0 + px + s;
        const int64_t hw = (int64_t)p.Hp * p.Wp;
        const int64_t fr = q / hw;
        const int r = (int)(q - fr * hw);
        const int h = r / p.Wp - p.pad_top, w = r % p.Wp - p.pad_left;
        const bool ok = fr < p.T && h >= 0 && h < p.H && w >= 0 && w < p.W;
        const T* sp = (const T*)p.src + ((fr * p.H + h) * (int64_t)p.W + w) * p.ps + c0 + cpart;
        const bool vec = (p.ps % EPV) == 0 && (((uintptr_t)p.src) & 15) == 0;
#pragma unroll
        for (int v = 0; v < 16 / EPV; ++v) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            const int c = c0 + cpart + v * EPV;
            if (ok && c + EPV <= p.C && vec) raw = *reinterpret_cast<const uint4*>(sp + v * EPV);
            else if (ok && c < p.C) {
                T tmp[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) tmp[e] = (c + e < p.C) ? sp[v * EPV + e] : (T)0.f;
                raw = *reinterpret_cast<const uint4*>(tmp);
            }
            *reinterpret_cast<uint4*>(&tile[px][cpart + v * EPV]) = raw;
        }
    }
    __syncthreads();
    {   // store: channel t/4, 16 consecutive q starting at (t%4)*16 (16-byte aligned: q0 and ld_out are multiples of 8)
        const int ch = t >> 2, qpart = (t & 3) * 16;
        const int c = c0 + ch;
        if (c < p.C) {
            T* op = (T*)p.out + ((int64_t)s * p.C + c) * p.ld_out + q0 + qpart;
            const bool vec = (p.ld_out % EPV) == 0 && (((uintptr_t)p.out) & 15) == 0;
#pragma unroll
            for (int v = 0; v < 16 / EPV; ++v) {
                T tmp[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) tmp[e] = tile[qpart + v * EPV + e][ch];
                if (vec && q0 + qpart + v * EPV + EPV <= p.cols) *reinterpret_cast<uint4*>(op + v * EPV) = *reinterpret_cast<const uint4*>(tmp);
                else
                    for (int e = 0; e < EPV; ++e)
                        if (q0 + qpart + v * EPV + e < p.cols) op[v * EPV + e] = tmp[e];
            }
        }
    }
}

// ------------------------------------------------------------------ dW[co, dt, dh, dw, ci] += sum_s part[dh, s, co, dw*cip + ci]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dw, int S, int Mp, int cop, int kt, int kh, int kw,
                                                           int cip, int dt) {
    const int64_t N = (int64_t)kw * cip, total = (int64_t)cop * kh * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i % N;
        const int dh = (int)((i / N) % kh);
        const int co = (int)(i / (N * kh));
        const float* src = part + (((int64_t)dh * S) * Mp + co) * N + n;
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += src[(int64_t)s * Mp * N];
        dw[(((int64_t)co * kt + dt) * kh + dh) * N + n] += acc;
    }
}

// part [S, kt*kh, cop, N] (m4d_gemm_bt_taps: every tap of the layer in one launch) -> dw[co][dt][dh][..] += sum over the K-slices
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float* part, float* dw, int S, int cop, int taps, int64_t N) {
    const int64_t total = (int64_t)cop * taps * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i % N;
        const int t = (int)((i / N) % taps);
        const int co = (int)(i / (N * taps));
        const float* src = part + ((int64_t)t * cop + co) * N + n;
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += src[(int64_t)s * taps * cop * N];
        dw[((int64_t)co * taps + t) * N + n] += acc;
    }
}

// ------------------------------------------------------------------ RMS_norm (+SiLU) backward
// y = silu?(round_T(x * sc * gamma)), sc = sqrt(C) / max(|x|, 1e-12):   a = du * gamma * sc,  dx = a - x * sum(a x) / |x|^2,
// dgamma[c] += du[c] * x[c] * sc.   SW lanes per pixel, VPL 16-byte vectors per lane (as the forward kernel).
struct RmsBwdArgs {
    const void* x; const void* dy; void* dx; const float* gamma; float* dgamma;
    int64_t P, x_ld, dy_ld, dx_ld;
    int C, silu;
};

template <typename T, int EPV> M4D_DEV void ldvec(const T* p, float (&v)[EPV]) {
    if constexpr (sizeof(T) == 2) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
    } else {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r[e];
    }
}
template <typename T, int EPV> M4D_DEV void stvec(T* p, const float (&v)[EPV]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x8*>(p) = r;
    } else {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = v[e];
        *reinterpret_cast<f32x4*>(p) = r;
    }
}

template <typename T, int SW, int VPL>
__global__ __launch_bounds__(256) void rmsnorm_silu_bwd_kernel(RmsBwdArgs p) {
    constexpr int EPV = 16 / sizeof(T);
    extern __shared__ float sdg[];                     // [C] per-workgroup gamma-gradient accumulator
    const int sub = threadIdx.x / SW, ls = threadIdx.x % SW, nsub = 256 / SW;
    const int nvec = p.C / EPV;
    for (int c = threadIdx.x; c < p.C; c += 256) sdg[c] = 0.f;
    __syncthreads();
    float dg[VPL][EPV];
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int e = 0; e < EPV; ++e) dg[i][e] = 0.f;
    const float sqc = sqrtf((float)p.C);
    for (int64_t pix0 = (int64_t)blockIdx.x * nsub; pix0 < p.P; pix0 += (int64_t)gridDim.x * nsub) {
        const int64_t pix = pix0 + sub;
        const bool active = pix < p.P;
        const T* xr = (const T*)p.x + (active ? pix : 0) * p.x_ld;
        const T* dr = (const T*)p.dy + (active ? pix : 0) * p.dy_ld;
        float x[VPL][EPV], a[VPL][EPV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int cv = ls + i * SW;
            if (cv < nvec) {
                ldvec<T, EPV>(xr + cv * EPV, x[i]);
#pragma unroll
                for (int e = 0; e < EPV; ++e) ss += x[i][e] * x[i][e];
            }
        }
#pragma unroll
        for (int o = SW / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        const float sc = sqc / nrm;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int cv = ls + i * SW;
            if (cv < nvec) {
                float d[EPV];
                ldvec<T, EPV>(dr + cv * EPV, d);
#pragma unroll
                for (int e = 0; e < EPV; e += 4) {
                    const f32x4 g = load4(p.gamma + cv * EPV + e);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float du = d[e + j];
                        if (p.silu) {
                            const float u = round_through<T>(x[i][e + j] * sc * g[j]);
                            const float sg = 1.f / (1.f + __expf(-u));
                            du *= sg * (1.f + u * (1.f - sg));
                        }
                        if (active) dg[i][e + j] += du * x[i][e + j] * sc;
                        a[i][e + j] = du * g[j] * sc;
                        dot += a[i][e + j] * x[i][e + j];
                    }
                }
            }
        }
#pragma unroll
        for (int o = SW / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        const float k = dot / (nrm * nrm);
        if (active) {
            T* orow = (T*)p.dx + pix * p.dx_ld;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int cv = ls + i * SW;
                if (cv < nvec) {
                    float o[EPV];
#pragma unroll
                    for (int e = 0; e < EPV; ++e) o[e] = a[i][e] - x[i][e] * k;
                    stvec<T, EPV>(orow + cv * EPV, o);
                }
            }
        }
    }
    // column partials: the nsub sub-groups of the workgroup add into LDS, then one atomic per channel
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int cv = ls + i * SW;
        if (cv < nvec)
#pragma unroll
            for (int e = 0; e < EPV; ++e) atomicAdd(&sdg[cv * EPV + e], dg[i][e]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256) atomicAdd(p.dgamma + c, sdg[c]);
}

// ------------------------------------------------------------------ softmax backward, one row per workgroup
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* pm, const float* dp, T* out, int C, int Cpad, float scale) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T* pr = pm + row * Cpad;
    const float* dr = dp + row * Cpad;
    T* orow = out + row * Cpad;
    const int t = threadIdx.x;
    float s = 0.f;
    for (int c = t; c < C; c += 256) s += (float)pr[c] * dr[c];
    s = wave_sum(s);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    const float dot = red[0] + red[1] + red[2] + red[3];
    for (int c = t; c < Cpad; c += 256) orow[c] = c < C ? (T)(scale * (float)pr[c] * (dr[c] - dot)) : (T)0.f;
}

// ------------------------------------------------------------------ nearest-exact 2x (+ tsplit), forward and transpose
// forward:  out[tt, y, x, :] = in[tt >> ts, y/2, x/2, (tt & ts) * c + :]           (ts = tsplit)
// backward: out[t, h, w, half*c + :] = sum_{dy,dx} in[2t + half (or t), 2h+dy, 2w+dx, :]
template <typename T, int BWD>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* in, T* out, int t, int h, int w, int c, int ts) {
    constexpr int EPV = 16 / sizeof(T);
    const int cv = c / EPV;
    if constexpr (!BWD) {
        const int tt = t * (ts ? 2 : 1);
        const int64_t total = (int64_t)tt * 4 * h * w * cv;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int v = (int)(i % cv);
            int64_t r = i / cv;
            const int x = (int)(r % (2 * w)); r /= 2 * w;
            const int y = (int)(r % (2 * h));
            const int f = (int)(r / (2 * h));
            const int fi = ts ? f >> 1 : f, half = ts ? (f & 1) : 0;
            const T* sp = in + (((int64_t)fi * h + (y >> 1)) * w + (x >> 1)) * (int64_t)(c * (ts ? 2 : 1)) + half * c + v * EPV;
            *reinterpret_cast<uint4*>(out + i * EPV) = *reinterpret_cast<const uint4*>(sp);
        }
    } else {
        const int halves = ts ? 2 : 1;
        const int64_t total = (int64_t)t * h * w * halves * cv;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int v = (int)(i % cv);
            int64_t r = i / cv;
            const int half = (int)(r % halves); r /= halves;
            const int x = (int)(r % w); r /= w;
            const int y = (int)(r % h);
            const int f = (int)(r / h);
            const int fi = ts ? 2 * f + half : f;
            float acc[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float u[EPV];
                    ldvec<T, EPV>(in + ((((int64_t)fi * 2 * h + 2 * y + dy) * 2 * w) + 2 * x + dx) * (int64_t)c + v * EPV, u);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) acc[e] += u[e];
                }
            stvec<T, EPV>(out + (((int64_t)f * h + y) * w + x) * (int64_t)(c * halves) + half * c + v * EPV, acc);
        }
    }
}

// ------------------------------------------------------------------ GroupNorm(+swish) backward, channels-last [F, HW, C]
// y = (x - m) r w + b, z = swish(round_T(y)):  g = dz * swish'(y);  dw[c] += g xhat, db[c] += g;  gh = g w;
// dx = r (gh - mean_grp(gh) - xhat mean_grp(gh xhat)).  Three passes: (1) sum / sumsq partials (as the forward), (2) per block
// sum(gh), sum(gh xhat) per group + dw / db, (3) apply.
struct GnBwdArgs {
    const void* x; const void* dy; void* dx; const float* weight; const float* bias;
    float* stat;        // [F, nblk, G, 2]  (sum, sumsq)
    float* red;         // [F, nblk, G, 2]  (sum gh, sum gh*xhat)
    float* dweight; float* dbias;
    int64_t HW;
    int F, C, G, nblk, ppb, silu; float eps;
};

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(GnBwdArgs p) {
    __shared__ float red[256][2];
    const int NV = p.C >> 2, slots = 256 / NV;
    const int t = threadIdx.x, v = t % NV, slot = t / NV;
    const int f = blockIdx.y, blk = blockIdx.x;
    const int64_t p0 = (int64_t)blk * p.ppb, p1 = min(p0 + p.ppb, p.HW);
    const T* xf = (const T*)p.x + (int64_t)f * p.HW * p.C;
    float s = 0.f, q = 0.f;
    if (slot < slots)
        for (int64_t px = p0 + slot; px < p1; px += slots) {
            const f32x4 u = load4(xf + px * p.C + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s += u[e]; q += u[e] * u[e]; }
        }
    red[t][0] = s; red[t][1] = q;
    __syncthreads();
    const int vpg = (p.C / p.G) >> 2;
    if (t < p.G) {
        float ss = 0.f, qq = 0.f;
        for (int j = 0; j < vpg; ++j)
            for (int sl = 0; sl < slots; ++sl) { ss += red[sl * NV + t * vpg + j][0]; qq += red[sl * NV + t * vpg + j][1]; }
        float* dst = p.stat + (((int64_t)f * p.nblk + blk) * p.G + t) * 2;
        dst[0] = ss; dst[1] = qq;
    }
}

template <typename T> M4D_DEV void gn_group_stats(const GnBwdArgs& p, int f, float* mean, float* rstd) {
    const int t = threadIdx.x;
    if (t < p.G) {
        float ss = 0.f, qq = 0.f;
        const float* src = p.stat + ((int64_t)f * p.nblk * p.G + t) * 2;
        for (int b = 0; b < p.nblk; ++b) { ss += src[(int64_t)b * p.G * 2]; qq += src[(int64_t)b * p.G * 2 + 1]; }
        const float n = (float)p.HW * (p.C / p.G);
        const float m = ss / n;
        mean[t] = m;
        rstd[t] = rsqrtf(fmaxf(qq / n - m * m, 0.f) + p.eps);
    }
    __syncthreads();
}

template <typename T> M4D_DEV float gn_g(const GnBwdArgs& p, float xh, float dz, float w, float b) {
    if (!p.silu) return dz;
    const float y = round_through<T>(xh * w + b);
    const float sg = 1.f / (1.f + __expf(-y));
    return dz * sg * (1.f + y * (1.f - sg));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(GnBwdArgs p) {
    __shared__ float mean[64], rstd[64];
    __shared__ float red[256][2];
    const int t = threadIdx.x, f = blockIdx.y, blk = blockIdx.x;
    gn_group_stats<T>(p, f, mean, rstd);
    const int NV = p.C >> 2, slots = 256 / NV;
    const int v = t % NV, slot = t / NV;
    const int cpg = p.C / p.G;
    float s1 = 0.f, s2 = 0.f;
    f32x4 dwv = {0.f, 0.f, 0.f, 0.f}, dbv = {0.f, 0.f, 0.f, 0.f};
    if (slot < slots) {
        const int g = (v * 4) / cpg;
        const float m = mean[g], r = rstd[g];
        const f32x4 w = load4(p.weight + v * 4), bb = load4(p.bias + v * 4);
        const int64_t p0 = (int64_t)blk * p.ppb, p1 = min(p0 + p.ppb, p.HW);
        const T* xf = (const T*)p.x + (int64_t)f * p.HW * p.C;
        const T* df = (const T*)p.dy + (int64_t)f * p.HW * p.C;
        for (int64_t px = p0 + slot; px < p1; px += slots) {
            const f32x4 u = load4(xf + px * p.C + v * 4), d = load4(df + px * p.C + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (u[e] - m) * r;
                const float g_ = gn_g<T>(p, xh, d[e], w[e], bb[e]);
                dwv[e] += g_ * xh; dbv[e] += g_;
                s1 += g_ * w[e]; s2 += g_ * w[e] * xh;
            }
        }
    }
    red[t][0] = s1; red[t][1] = s2;
    __syncthreads();
    const int vpg = cpg >> 2;
    if (t < p.G) {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < vpg; ++j)
            for (int sl = 0; sl < slots; ++sl) { a += red[sl * NV + t * vpg + j][0]; b += red[sl * NV + t * vpg + j][1]; }
        float* dst = p.red + (((int64_t)f * p.nblk + blk) * p.G + t) * 2;
        dst[0] = a; dst[1] = b;
    }
    __syncthreads();
    // per-channel weight / bias gradients: combine the pixel slots through LDS, one atomic per channel per workgroup
    float* cw = &red[0][0];          // reuse: [2][C] floats (C <= 256 -> 512 floats = the whole array)
    for (int i = t; i < 2 * p.C; i += 256) cw[i] = 0.f;
    __syncthreads();
    if (slot < slots)
#pragma unroll
        for (int e = 0; e < 4; ++e) { atomicAdd(&cw[v * 4 + e], dwv[e]); atomicAdd(&cw[p.C + v * 4 + e], dbv[e]); }
    __syncthreads();
    for (int c = t; c < p.C; c += 256) { atomicAdd(p.dweight + c, cw[c]); atomicAdd(p.dbias + c, cw[p.C + c]); }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdArgs p) {
    __shared__ float mean[64], rstd[64], m1[64], m2[64];
    const int t = threadIdx.x, f = blockIdx.y, blk = blockIdx.x;
    gn_group_stats<T>(p, f, mean, rstd);
    const int cpg = p.C / p.G;
    if (t < p.G) {
        float a = 0.f, b = 0.f;
        const float* src = p.red + ((int64_t)f * p.nblk * p.G + t) * 2;
        for (int bb = 0; bb < p.nblk; ++bb) { a += src[(int64_t)bb * p.G * 2]; b += src[(int64_t)bb * p.G * 2 + 1]; }
        const float n = (float)p.HW * cpg;
        m1[t] = a / n; m2[t] = b / n;
    }
    __syncthreads();
    const int NV = p.C >> 2, slots = 256 / NV;
    const int v = t % NV, slot = t / NV;
    if (slot >= slots) return;
    const int g = (v * 4) / cpg;
    const float m = mean[g], r = rstd[g], a1 = m1[g], a2 = m2[g];
    const f32x4 w = load4(p.weight + v * 4), bb = load4(p.bias + v * 4);
    const int64_t p0 = (int64_t)blk * p.ppb, p1 = min(p0 + p.ppb, p.HW);
    const T* xf = (const T*)p.x + (int64_t)f * p.HW * p.C;
    const T* df = (const T*)p.dy + (int64_t)f * p.HW * p.C;
    T* of = (T*)p.dx + (int64_t)f * p.HW * p.C;
    for (int64_t px = p0 + slot; px < p1; px += slots) {
        const f32x4 u = load4(xf + px * p.C + v * 4), d = load4(df + px * p.C + v * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (u[e] - m) * r;
            const float gh = gn_g<T>(p, xh, d[e], w[e], bb[e]) * w[e];
            o[e] = r * (gh - a1 - xh * a2);
        }
        store4(of + px * p.C + v * 4, o);
    }
}

}  // namespace

namespace {
// All `nshift` (<= 4) shifted copies of one 64-pixel x 64-channel tile from ONE load of its 64 + nshift - 1 pixels: the shifted panels of
// a conv's weight-gradient operand differ by one pixel each, and the per-pixel index arithmetic (two 64-bit divisions) costs more than the
// 32 bytes it fetches.  Same output as pad_transpose_kernel launched once per shift.
template <typename T>
__global__ __launch_bounds__(256) void pad_transpose_ms_kernel(PadTArgs p, int nshift) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ __attribute__((aligned(16))) T tile[64 + 3][64 + EPV];
    const int c0 = blockIdx.y * 64;
    const int64_t q0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int64_t hw = (int64_t)p.Hp * p.Wp;
    const bool vec_in = (p.ps % EPV) == 0 && (((uintptr_t)p.src) & 15) == 0;
    for (int px = t >> 2; px < 64 + nshift - 1; px += 64) {      // pixel px of the tile, 16 channels starting at (t % 4) * 16
        const int cpart = (t & 3) * 16;
        const int64_t q = q0 + px;
        const int64_t fr = q / hw;
        const int r = (int)(q - fr * hw);
        const int h = r / p.Wp - p.pad_top, w = r % p.Wp - p.pad_left;
        const bool ok = fr < p.T && h >= 0 && h < p.H && w >= 0 && w < p.W;
        const T* sp = (const T*)p.src + ((fr * p.H + h) * (int64_t)p.W + w) * p.ps + c0 + cpart;
#pragma unroll
        for (int v = 0; v < 16 / EPV; ++v) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            const int c = c0 + cpart + v * EPV;
            if (ok && c + EPV <= p.C && vec_in) raw = *reinterpret_cast<const uint4*>(sp + v * EPV);
            else if (ok && c < p.C) {
                T tmp[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) tmp[e] = (c + e < p.C) ? sp[v * EPV + e] : (T)0.f;
                raw = *reinterpret_cast<const uint4*>(tmp);
            }
            *reinterpret_cast<uint4*>(&tile[px][cpart + v * EPV]) = raw;
        }
    }
    __syncthreads();
    const int ch = t >> 2, qpart = (t & 3) * 16;
    const int c = c0 + ch;
    if (c >= p.C) return;
    T col[16 + 3];
#pragma unroll
    for (int k = 0; k < 16 + 3; ++k) col[k] = tile[qpart + k][ch];
    const bool vec_out = (p.ld_out % EPV) == 0 && (((uintptr_t)p.out) & 15) == 0;
#pragma unroll
    for (int sft = 0; sft < 4; ++sft) {
        if (sft >= nshift) break;
        T* op = (T*)p.out + ((int64_t)sft * p.C + c) * p.ld_out + q0 + qpart;
#pragma unroll
        for (int v = 0; v < 16 / EPV; ++v) {
            T tmp[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) tmp[e] = col[sft + v * EPV + e];
            if (vec_out && q0 + qpart + v * EPV + EPV <= p.cols) *reinterpret_cast<uint4*>(op + v * EPV) = *reinterpret_cast<const uint4*>(tmp);
            else
                for (int e = 0; e < EPV; ++e)
                    if (q0 + qpart + v * EPV + e < p.cols) op[v * EPV + e] = tmp[e];
        }
    }
}
}  // namespace

extern "C" int m4d_pad_transpose(m4d_dtype dt, const void* src, int64_t pixel_stride, int C, int T, int H, int W, int Hp, int Wp,
                                 int pad_top, int pad_left, int nshift, void* out, int64_t cols, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "pad_transpose: bad dtype");
    M4D_CHECK_ARG(src && out && C > 0 && T > 0 && H > 0 && W > 0 && nshift > 0 && cols > 0, "pad_transpose: null/empty");
    M4D_CHECK_ARG(Hp >= H + pad_top && Wp >= W + pad_left && pixel_stride >= C, "pad_transpose: padded geometry smaller than the image");
    PadTArgs p{src, out, pixel_stride, cols, cols, C, T, H, W, Hp, Wp, pad_top, pad_left};
    M4D_ENV_ONCE(ms, "M4D_PADT_MS", 1);      // 0: one workgroup per shift (A/B)
    if (ms && nshift >= 2 && nshift <= 4) {
        dim3 gm((unsigned)((cols + 63) / 64), (unsigned)((C + 63) / 64), 1), bm(256);
        if (dt == M4D_BF16) hipLaunchKernelGGL(pad_transpose_ms_kernel<bf16_t>, gm, bm, 0, (hipStream_t)stream, p, nshift);
        else hipLaunchKernelGGL(pad_transpose_ms_kernel<float>, gm, bm, 0, (hipStream_t)stream, p, nshift);
        M4D_CHECK_LAUNCH("pad_transpose");
        return 0;
    }
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)nshift), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL(pad_transpose_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(pad_transpose_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("pad_transpose");
    return 0;
}

extern "C" int m4d_wgrad_reduce(const float* part, float* dw, int S, int Mp, int cop, int kt, int kh, int kw, int cip, int dt,
                                m4d_stream stream) {
    M4D_CHECK_ARG(part && dw && S > 0 && Mp >= cop && cop > 0 && dt >= 0 && dt < kt && kh > 0 && kw > 0 && cip > 0, "wgrad_reduce: bad arguments");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for((int64_t)cop * kh * kw * cip)), dim3(256), 0, (hipStream_t)stream, part, dw, S, Mp,
                       cop, kt, kh, kw, cip, dt);
    M4D_CHECK_LAUNCH("wgrad_reduce");
    return 0;
}

extern "C" int m4d_wgrad_reduce_taps(const float* part, float* dw, int S, int cop, int kt, int kh, int kw, int cip, m4d_stream stream) {
    M4D_CHECK_ARG(part && dw && S > 0 && cop > 0 && kt > 0 && kh > 0 && kw > 0 && cip > 0, "wgrad_reduce_taps: bad arguments");
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3(grid_for((int64_t)cop * kt * kh * kw * cip)), dim3(256), 0, (hipStream_t)stream, part, dw, S,
                       cop, kt * kh, (int64_t)kw * cip);
    M4D_CHECK_LAUNCH("wgrad_reduce_taps");
    return 0;
}

extern "C" int m4d_rmsnorm_silu_cl_bwd(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, const void* dy, int64_t dy_ld,
                                       void* dx, int64_t dx_ld, float* dgamma, int64_t P, int C, int silu, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "rmsnorm_silu_cl_bwd: bad dtype");
    M4D_CHECK_ARG(x && gamma && dy && dx && dgamma && P > 0, "rmsnorm_silu_cl_bwd: null/empty");
    const int epv = dt == M4D_BF16 ? 8 : 4;
    M4D_CHECK_ARG(C % epv == 0 && C / epv <= 128, "rmsnorm_silu_cl_bwd: C=%d must be a multiple of %d and <= %d", C, epv, 128 * epv);
    M4D_CHECK_ARG(x_ld % epv == 0 && dy_ld % epv == 0 && dx_ld % epv == 0 && x_ld >= C && dy_ld >= C && dx_ld >= C, "rmsnorm_silu_cl_bwd: bad row strides");
    RmsBwdArgs p{x, dy, dx, gamma, dgamma, P, x_ld, dy_ld, dx_ld, C, silu};
    const int nvec = C / epv;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)C * sizeof(float);
#define RB(T, SW, VPL) hipLaunchKernelGGL((rmsnorm_silu_bwd_kernel<T, SW, VPL>), dim3(grid_for(P, 256 / SW, 1024)), dim3(256), lds, st, p)
    if (dt == M4D_BF16) {
        if (nvec <= 16) RB(bf16_t, 16, 1); else if (nvec <= 32) RB(bf16_t, 32, 1); else if (nvec <= 64) RB(bf16_t, 64, 1); else RB(bf16_t, 64, 2);
    } else {
        if (nvec <= 16) RB(float, 16, 1); else if (nvec <= 32) RB(float, 32, 1); else if (nvec <= 64) RB(float, 64, 1); else RB(float, 64, 2);
    }
#undef RB
    M4D_CHECK_LAUNCH("rmsnorm_silu_cl_bwd");
    return 0;
}

extern "C" int m4d_softmax_rows_bwd(m4d_dtype dt, const void* p, const float* dp, void* out, int64_t rows, int C, int Cpad, float scale,
                                    m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "softmax_rows_bwd: bad dtype");
    M4D_CHECK_ARG(p && dp && out && rows > 0 && C > 0 && Cpad >= C, "softmax_rows_bwd: bad arguments");
    if (dt == M4D_BF16) hipLaunchKernelGGL(softmax_rows_bwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p, dp, (bf16_t*)out, C, Cpad, scale);
    else hipLaunchKernelGGL(softmax_rows_bwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const float*)p, dp, (float*)out, C, Cpad, scale);
    M4D_CHECK_LAUNCH("softmax_rows_bwd");
    return 0;
}

extern "C" int m4d_upsample2x_cl(m4d_dtype dt, const void* in, void* out, int t, int h, int w, int c, int tsplit, int backward,
                                 m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "upsample2x_cl: bad dtype");
    const int epv = dt == M4D_BF16 ? 8 : 4;
    M4D_CHECK_ARG(in && out && t > 0 && h > 0 && w > 0 && c > 0 && c % epv == 0, "upsample2x_cl: c must be a multiple of %d", epv);
    const int64_t n = (int64_t)t * h * w * (c / epv) * (backward ? 1 : 4) * (tsplit ? 2 : 1);
    const dim3 grid(grid_for(n)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dt == M4D_BF16) {
        if (backward) hipLaunchKernelGGL((upsample2x_kernel<bf16_t, 1>), grid, block, 0, st, (const bf16_t*)in, (bf16_t*)out, t, h, w, c, tsplit);
        else hipLaunchKernelGGL((upsample2x_kernel<bf16_t, 0>), grid, block, 0, st, (const bf16_t*)in, (bf16_t*)out, t, h, w, c, tsplit);
    } else {
        if (backward) hipLaunchKernelGGL((upsample2x_kernel<float, 1>), grid, block, 0, st, (const float*)in, (float*)out, t, h, w, c, tsplit);
        else hipLaunchKernelGGL((upsample2x_kernel<float, 0>), grid, block, 0, st, (const float*)in, (float*)out, t, h, w, c, tsplit);
    }
    M4D_CHECK_LAUNCH("upsample2x_cl");
    return 0;
}

extern "C" int64_t m4d_groupnorm_cl_bwd_workspace(int F, int64_t HW, int G) { return (int64_t)F * ((HW + 2047) / 2048) * G * 4; }

extern "C" int m4d_groupnorm_cl_bwd(m4d_dtype dt, const void* x, const float* weight, const float* bias, const void* dy, void* dx,
                                    float* dweight, float* dbias, float* ws, int64_t ws_floats, int F, int64_t HW, int C, int G, float eps,
                                    int silu, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "groupnorm_cl_bwd: bad dtype");
    M4D_CHECK_ARG(x && weight && bias && dy && dx && dweight && dbias && ws && F > 0 && HW > 0, "groupnorm_cl_bwd: null/empty");
    M4D_CHECK_ARG(C % 4 == 0 && C <= 256 && 256 % (C / 4) == 0, "groupnorm_cl_bwd: C=%d unsupported (C/4 must divide 256, C <= 256)", C);
    M4D_CHECK_ARG(G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0, "groupnorm_cl_bwd: channels per group must be a multiple of 4");
    const int ppb = 2048;
    const int nblk = (int)((HW + ppb - 1) / ppb);
    const int64_t half = (int64_t)F * nblk * G * 2;
    M4D_CHECK_ARG(ws_floats >= 2 * half, "groupnorm_cl_bwd: workspace too small (need %lld floats)", (long long)(2 * half));
    GnBwdArgs p{x, dy, dx, weight, bias, ws, ws + half, dweight, dbias, HW, F, C, G, nblk, ppb, silu, eps};
    dim3 grid(nblk, F), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dt == M4D_BF16) {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<bf16_t>, grid, block, 0, st, p);
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, grid, block, 0, st, p);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, grid, block, 0, st, p);
    } else {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<float>, grid, block, 0, st, p);
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, grid, block, 0, st, p);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, block, 0, st, p);
    }
    M4D_CHECK_LAUNCH("groupnorm_cl_bwd");
    return 0;
}
