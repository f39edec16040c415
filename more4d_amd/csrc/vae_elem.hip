// HBM-bound kernels of the Motion-Sensitive 3D-VAE path on CHANNELS-LAST activations [T, H, W, C]:
// per-pixel RMS norm (+SiLU), GroupNorm(+swish) for the trajectory adaptors, row softmax for the mid attention,
// and the NCTHW <-> channels-last boundary conversions with their fused pointwise epilogues.
// Channels-last makes every per-pixel reduction a contiguous 16-byte-vector read (the reference's NCTHW layout
// makes RMS_norm a strided reduction, wan_vae.py :55-58).
#include "common.h"
#include "more4d_hip.h"

namespace {

template <typename T, int EPV> M4D_DEV void ldv(const T* p, float (&v)[EPV]) {
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
template <typename T, int EPV> M4D_DEV void stv(T* p, const float (&v)[EPV]) {
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

inline unsigned grid_for(int64_t n, int per_block = 256) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------ RMS norm (+SiLU), channels-last
// SW lanes cooperate on one pixel (SW = 16/32/64 chosen from C), each lane holds VPL 16-byte vectors.
struct RmsClArgs {
    const void* x; void* out; const float* gamma;
    int64_t P, x_ld, out_ld;
    int C, silu;
    int64_t out_plane;      // != 0: planar-16 output [C/16][rows][16] with `out_plane` elements between planes (bf16; conv_halo_kernel's input)
};

template <typename T, int SW, int VPL>
__global__ __launch_bounds__(256) void rmsnorm_silu_cl_kernel(RmsClArgs p) {
    constexpr int EPV = 16 / sizeof(T);
    const int sub = threadIdx.x / SW, ls = threadIdx.x % SW;
    const int64_t pix = (int64_t)blockIdx.x * (256 / SW) + sub;
    const bool active = pix < p.P;
    const int nvec = p.C / EPV;
    const T* xr = (const T*)p.x + (active ? pix : 0) * p.x_ld;
    float v[VPL][EPV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int cv = ls + i * SW;
        if (cv < nvec) {
            ldv<T, EPV>(xr + cv * EPV, v[i]);
#pragma unroll
            for (int e = 0; e < EPV; ++e) s += v[i][e] * v[i][e];
        }
    }
#pragma unroll
    for (int o = SW / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    // F.normalize: x / max(||x||, 1e-12), then * sqrt(C) * gamma   (wan_vae.py :55-58)
    const float sc = rms_scale_f(sqrtf((float)p.C), s);
    if (!active) return;
    T* orow = (T*)p.out + pix * p.out_ld;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int cv = ls + i * SW;
        if (cv < nvec) {
            float y[EPV];
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                const f32x4 g = load4(p.gamma + cv * EPV + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float u = v[i][e + j] * sc * g[j];
                    if (p.silu) u = silu_f(round_through<T>(u));
                    y[e + j] = u;
                }
            }
            // planar-16: this lane's 8 channels are half (cv & 1) of pixel `pix` in plane cv >> 1; the 4 pixels of a wave make one
            // 128-byte line per plane
            if (p.out_plane) stv<T, EPV>((T*)p.out + (int64_t)(cv >> 1) * p.out_plane + pix * 16 + (cv & 1) * 8, y);
            else stv<T, EPV>(orow + cv * EPV, y);
        }
    }
}

// ------------------------------------------------------------------ GroupNorm (+swish), channels-last, two deterministic passes
struct GnArgs {
    const void* x; void* out; float* partial; float* stat; const float* weight; const float* bias;
    int64_t HW;
    int F, C, G, nblk, ppb, silu; float eps;
    // planar-16 output (out_plane != 0): frames in groups of `fpg`, group g = [C/16][fpg*HW][16] with out_plane elements between planes
    // and out_group between groups (each group stays below the 2 GiB m4d_conv_cl_planar addresses)
    int64_t out_plane, out_group; int fpg;
};

// A thread owns VEC = 16 bytes of channels of a pixel (8 bf16 / 4 float; 4 when C % 8 != 0) = VEC/4 sub-vectors of 4 channels, each
// inside one group (channels per group % 4 == 0).  16-byte accesses, 8 of them in flight per thread.
template <typename T, int VEC> struct GnVec;
template <typename T> struct GnVec<T, 4> {
    f32x4 h[1];
    M4D_DEV void load(const T* p) { h[0] = load4(p); }
    M4D_DEV void store(T* p) const { store4(p, h[0]); }
};
template <> struct GnVec<bf16_t, 8> {
    f32x4 h[2];
    M4D_DEV void load(const bf16_t* p) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e >> 2][e & 3] = (float)v[e];
    }
    M4D_DEV void store(bf16_t* p) const {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)h[e >> 2][e & 3];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};

// pass 1: partial[f][blk][g] = (sum, sumsq) over this block's pixels
template <typename T, int VEC>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(GnArgs p) {
    constexpr int NH = VEC / 4;
    __shared__ float red[256 * NH][2];
    const int NV = p.C / VEC, slots = 256 / NV;
    const int t = threadIdx.x, v = t % NV, slot = t / NV;
    const int f = blockIdx.y, blk = blockIdx.x;
    const int64_t p0 = (int64_t)blk * p.ppb, p1 = min(p0 + p.ppb, p.HW);
    const T* xf = (const T*)p.x + (int64_t)f * p.HW * p.C + v * VEC;
    float s[NH] = {}, q[NH] = {};
    auto add = [&](const GnVec<T, VEC>& u) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[h] += u.h[h][e]; q[h] += u.h[h][e] * u.h[h][e]; }
    };
    if (slot < slots) {
        int64_t px = p0 + slot;
        for (; px + 7 * slots < p1; px += 8 * slots) {
            GnVec<T, VEC> u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j].load(xf + (px + j * slots) * p.C);
#pragma unroll
            for (int j = 0; j < 8; ++j) add(u[j]);
        }
        for (; px < p1; px += slots) { GnVec<T, VEC> u; u.load(xf + px * p.C); add(u); }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) { red[t * NH + h][0] = s[h]; red[t * NH + h][1] = q[h]; }
    __syncthreads();
    const int vpg = p.C / p.G / 4;                  // 4-channel sub-vectors per group; sub-vector i of slot sl = red[(sl * NV) * NH + i]
    if (t < p.G) {
        float ss = 0.f, qq = 0.f;
        for (int j = 0; j < vpg; ++j)
            for (int sl = 0; sl < slots; ++sl) { ss += red[sl * NV * NH + t * vpg + j][0]; qq += red[sl * NV * NH + t * vpg + j][1]; }
        float* dst = p.partial + (((int64_t)f * p.nblk + blk) * p.G + t) * 2;
        dst[0] = ss; dst[1] = qq;
    }
}

// pass 2 (one small workgroup per frame): stat[f][g] = (mean, rstd) from the partials in a fixed order (the apply kernel's workgroups
// each used to re-reduce all nblk partials with one dependent load after the other: ~0.4 ms per workgroup at 195 partials)
__global__ __launch_bounds__(256) void groupnorm_finalize_kernel(GnArgs p, int cpg) {
    __shared__ float red[256][2];
    const int f = blockIdx.x, t = threadIdx.x, g = t % p.G, lane = t / p.G, nl = 256 / p.G;
    float ss = 0.f, qq = 0.f;
    const float* src = p.partial + ((int64_t)f * p.nblk * p.G + g) * 2;
    if (lane < nl)
        for (int b = lane; b < p.nblk; b += nl) { ss += src[(int64_t)b * p.G * 2]; qq += src[(int64_t)b * p.G * 2 + 1]; }
    red[t][0] = ss; red[t][1] = qq;
    __syncthreads();
    if (t < p.G) {
        ss = 0.f; qq = 0.f;
        for (int l = 0; l < nl; ++l) { ss += red[l * p.G + t][0]; qq += red[l * p.G + t][1]; }
        const float n = (float)p.HW * cpg;
        const float m = ss / n;
        p.stat[((int64_t)f * p.G + t) * 2] = m;
        p.stat[((int64_t)f * p.G + t) * 2 + 1] = rsqrtf(fmaxf(qq / n - m * m, 0.f) + p.eps);
    }
}

// pass 3: normalise + affine (+ x*sigmoid(x))
template <typename T, int VEC>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(GnArgs p) {
    constexpr int NH = VEC / 4;
    const int t = threadIdx.x, f = blockIdx.y, blk = blockIdx.x;
    const int cpg = p.C / p.G;
    const int NV = p.C / VEC, slots = 256 / NV;
    const int v = t % NV, slot = t / NV;
    if (slot >= slots) return;
    float m[NH], r[NH];
    f32x4 w[NH], bb[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int c0 = v * VEC + h * 4, g = c0 / cpg;
        m[h] = p.stat[((int64_t)f * p.G + g) * 2]; r[h] = p.stat[((int64_t)f * p.G + g) * 2 + 1];
        w[h] = load4(p.weight + c0); bb[h] = load4(p.bias + c0);
    }
    const int64_t p0 = (int64_t)blk * p.ppb, p1 = min(p0 + p.ppb, p.HW);
    const T* xf = (const T*)p.x + (int64_t)f * p.HW * p.C + v * VEC;
    // channels-last: pixel stride C; planar-16 (VEC = 8): this thread's 8 channels are half (v & 1) of plane v >> 1, pixel stride 16
    const int64_t opx = p.out_plane ? 16 : p.C;
    T* of = p.out_plane ? (T*)p.out + (int64_t)(f / p.fpg) * p.out_group + (int64_t)(v >> 1) * p.out_plane + (int64_t)(f % p.fpg) * p.HW * 16 + (v & 1) * 8
                        : (T*)p.out + (int64_t)f * p.HW * p.C + v * VEC;
    auto apply = [&](GnVec<T, VEC>& u) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = (u.h[h][e] - m[h]) * r[h] * w[h][e] + bb[h][e];
                if (p.silu) y = silu_f(round_through<T>(y));
                u.h[h][e] = y;
            }
    };
    int64_t px = p0 + slot;
    for (; px + 7 * slots < p1; px += 8 * slots) {       // 8 loads in flight per thread (x and out may be the same buffer: load all first)
        GnVec<T, VEC> u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j].load(xf + (px + j * slots) * p.C);
#pragma unroll
        for (int j = 0; j < 8; ++j) { apply(u[j]); u[j].store(of + (px + j * slots) * opx); }
    }
    for (; px < p1; px += slots) { GnVec<T, VEC> u; u.load(xf + px * p.C); apply(u); u.store(of + px * opx); }
}

// ------------------------------------------------------------------ row softmax (VAE mid attention scores)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const TI* x, int64_t ldx, TO* out, int64_t ldo, int C, int Cpad, float scale) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const TI* xr = x + row * ldx;
    TO* orow = out + row * ldo;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float mx = -INFINITY;
    for (int c = t; c < C; c += 256) mx = fmaxf(mx, (float)xr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    __syncthreads();
    float s = 0.f;
    for (int c = t; c < C; c += 256) s += __expf((float)xr[c] * scale - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = t; c < Cpad; c += 256) orow[c] = c < C ? (TO)(__expf((float)xr[c] * scale - mx) * inv) : (TO)0.f;
}

// ------------------------------------------------------------------ layout boundaries
struct LayoutArgs {
    const void* src; void* dst; const void* aux;
    const float* ch_scale; const float* ch_shift;
    int64_t ld;                 // channels-last pixel stride (elements)
    int C, Cp, T, H, W, act;
    float scale, shift;
};

// [C,T,H,W] -> [T,H,W,ld]: channels c < C copied (v*scale+shift, then per-channel affine), C <= c < Cp zero-filled
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void ncthw_to_cl_kernel(LayoutArgs p) {
    const int64_t npix = (int64_t)p.T * p.H * p.W, total = npix * p.Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.Cp);
        const int64_t pix = i / p.Cp;
        float v = 0.f;
        if (c < p.C) {
            v = (float)((const TS*)p.src)[(int64_t)c * npix + pix] * p.scale + p.shift;
            if (p.ch_scale) v = v * p.ch_scale[c] + p.ch_shift[c];
        }
        ((TD*)p.dst)[pix * p.ld + c] = (TD)v;
    }
}

// [T,H,W,ld] -> [C,T,H,W] with per-channel affine and act: 0 none, 1 clamp(-1,1), 2 sigmoid(v + aux[c,t,h,w])
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cl_to_ncthw_kernel(LayoutArgs p) {
    const int64_t npix = (int64_t)p.T * p.H * p.W, total = npix * p.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % npix;
        const int c = (int)(i / npix);
        float v = (float)((const TS*)p.src)[pix * p.ld + c] * p.scale + p.shift;
        if (p.ch_scale) v = v * p.ch_scale[c] + p.ch_shift[c];
        if (p.act == 1) v = fminf(fmaxf(v, -1.f), 1.f);
        else if (p.act == 2) { v += (float)((const TD*)p.aux)[i]; v = 1.f / (1.f + __expf(-v)); }
        ((TD*)p.dst)[i] = (TD)v;
    }
}

}  // namespace

static int rmsnorm_silu_launch(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, void* out, int64_t out_ld, int64_t out_plane,
                               int64_t P, int C, int silu, m4d_stream stream);

extern "C" int m4d_rmsnorm_silu_cl(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, void* out, int64_t out_ld,
                                   int64_t P, int C, int silu, m4d_stream stream) {
    M4D_CHECK_ARG(x && gamma && out && P > 0, "rmsnorm_silu_cl: null/empty");
    const int epv = dt == M4D_BF16 ? 8 : 4;
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "rmsnorm_silu_cl: bad dtype");
    M4D_CHECK_ARG(C % epv == 0 && C / epv <= 128, "rmsnorm_silu_cl: C=%d must be a multiple of %d and <= %d", C, epv, 128 * epv);
    M4D_CHECK_ARG(x_ld % epv == 0 && out_ld % epv == 0 && x_ld >= C && out_ld >= C, "rmsnorm_silu_cl: bad row strides");
    return rmsnorm_silu_launch(dt, x, x_ld, gamma, out, out_ld, 0, P, C, silu, stream);
}

extern "C" int m4d_rmsnorm_silu_cl_planar(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, void* out, int64_t out_plane_stride,
                                          int64_t P, int C, int silu, m4d_stream stream) {
    M4D_CHECK_ARG(x && gamma && out && P > 0, "rmsnorm_silu_cl_planar: null/empty");
    M4D_CHECK_ARG(dt == M4D_BF16 && C % 16 == 0 && C <= 1024 && x_ld % 8 == 0 && x_ld >= C, "rmsnorm_silu_cl_planar: bf16, C a multiple of 16");
    M4D_CHECK_ARG(out_plane_stride >= P * 16 && out_plane_stride % 8 == 0, "rmsnorm_silu_cl_planar: plane stride %lld too small for %lld rows",
                  (long long)out_plane_stride, (long long)P);
    return rmsnorm_silu_launch(dt, x, x_ld, gamma, out, C, out_plane_stride, P, C, silu, stream);
}

static int rmsnorm_silu_launch(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, void* out, int64_t out_ld, int64_t out_plane,
                               int64_t P, int C, int silu, m4d_stream stream) {
    const int epv = dt == M4D_BF16 ? 8 : 4;
    RmsClArgs p{x, out, gamma, P, x_ld, out_ld, C, silu, out_plane};
    const int nvec = C / epv;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256);
#define RL(T, SW, VPL) hipLaunchKernelGGL((rmsnorm_silu_cl_kernel<T, SW, VPL>), dim3((unsigned)((P + 256 / SW - 1) / (256 / SW))), block, 0, st, p)
    if (dt == M4D_BF16) {
        if (nvec <= 16) RL(bf16_t, 16, 1); else if (nvec <= 32) RL(bf16_t, 32, 1); else if (nvec <= 64) RL(bf16_t, 64, 1); else RL(bf16_t, 64, 2);
    } else {
        if (nvec <= 16) RL(float, 16, 1); else if (nvec <= 32) RL(float, 32, 1); else if (nvec <= 64) RL(float, 64, 1); else RL(float, 64, 2);
    }
#undef RL
    M4D_CHECK_LAUNCH("rmsnorm_silu_cl");
    return 0;
}

constexpr int GN_PPB = 512;        // pixels per workgroup of the stats / apply passes

static int groupnorm_impl(m4d_dtype dt, const void* x, void* out, float* partial, int64_t partial_floats, const float* weight, const float* bias,
                          int F, int64_t HW, int C, int G, float eps, int silu, int fpg, int64_t out_plane, int64_t out_group, m4d_stream stream);

extern "C" int m4d_groupnorm_cl(m4d_dtype dt, const void* x, void* out, float* partial, int64_t partial_floats,
                                const float* weight, const float* bias, int F, int64_t HW, int C, int G, float eps, int silu,
                                m4d_stream stream) {
    return groupnorm_impl(dt, x, out, partial, partial_floats, weight, bias, F, HW, C, G, eps, silu, 0, 0, 0, stream);
}

extern "C" int m4d_groupnorm_cl_planar(m4d_dtype dt, const void* x, void* out, float* partial, int64_t partial_floats, const float* weight,
                                       const float* bias, int F, int64_t HW, int C, int G, float eps, int silu, int frames_per_group,
                                       int64_t out_plane_stride, int64_t out_group_stride, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 && C % 16 == 0, "groupnorm_cl_planar: bf16, C %% 16");
    M4D_CHECK_ARG(frames_per_group > 0 && out_plane_stride >= (int64_t)frames_per_group * HW * 16 && out_plane_stride % 8 == 0 &&
                  out_group_stride >= (int64_t)(C / 16) * out_plane_stride, "groupnorm_cl_planar: bad group / plane strides");
    return groupnorm_impl(dt, x, out, partial, partial_floats, weight, bias, F, HW, C, G, eps, silu, frames_per_group, out_plane_stride,
                          out_group_stride, stream);
}

// the planar-16 GroupNorm with the statistics already reduced per block by the producer (m4d_conv_cl_planar_gnstats): finalize + apply
extern "C" int m4d_groupnorm_cl_planar_apply(m4d_dtype dt, const void* x, void* out, const float* partial, int partial_blocks, float* stat,
                                             const float* weight, const float* bias, int F, int64_t HW, int C, int G, float eps, int silu,
                                             int frames_per_group, int64_t out_plane_stride, int64_t out_group_stride, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 && C % 16 == 0 && x && out && partial && stat && weight && bias && F > 0 && HW > 0 && partial_blocks > 0,
                  "groupnorm_cl_planar_apply: bf16, C %% 16, non-null arguments");
    M4D_CHECK_ARG(G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0 && 256 % (C / 8) == 0, "groupnorm_cl_planar_apply: unsupported C / G");
    M4D_CHECK_ARG(frames_per_group > 0 && out_plane_stride >= (int64_t)frames_per_group * HW * 16 && out_plane_stride % 8 == 0 &&
                  out_group_stride >= (int64_t)(C / 16) * out_plane_stride, "groupnorm_cl_planar_apply: bad group / plane strides");
    const int nblk = (int)((HW + GN_PPB - 1) / GN_PPB);
    GnArgs p{x, out, const_cast<float*>(partial), stat, weight, bias, HW, F, C, G, partial_blocks, GN_PPB, silu, eps, out_plane_stride,
             out_group_stride, frames_per_group};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(F), dim3(256), 0, st, p, C / G);
    p.nblk = nblk;
    hipLaunchKernelGGL((groupnorm_apply_kernel<bf16_t, 8>), dim3(nblk, F), dim3(256), 0, st, p);
    M4D_CHECK_LAUNCH("groupnorm_cl_planar_apply");
    return 0;
}

static int groupnorm_impl(m4d_dtype dt, const void* x, void* out, float* partial, int64_t partial_floats, const float* weight, const float* bias,
                          int F, int64_t HW, int C, int G, float eps, int silu, int fpg, int64_t out_plane, int64_t out_group, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "groupnorm_cl: bad dtype");
    M4D_CHECK_ARG(x && out && partial && weight && bias && F > 0 && HW > 0, "groupnorm_cl: null/empty");
    M4D_CHECK_ARG(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, "groupnorm_cl: C=%d unsupported (C/4 must divide 256)", C);
    M4D_CHECK_ARG(G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0, "groupnorm_cl: channels per group must be a multiple of 4");
    const int nblk = (int)((HW + GN_PPB - 1) / GN_PPB);
    M4D_CHECK_ARG(partial_floats >= m4d_groupnorm_cl_workspace(F, HW, G), "groupnorm_cl: workspace too small (need %lld floats)",
                  (long long)m4d_groupnorm_cl_workspace(F, HW, G));
    GnArgs p{x, out, partial, partial + (int64_t)F * nblk * G * 2, weight, bias, HW, F, C, G, nblk, GN_PPB, silu, eps, out_plane, out_group, fpg > 0 ? fpg : 1};
    dim3 grid(nblk, F), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dt == M4D_BF16 && C % 8 == 0) {
        hipLaunchKernelGGL((groupnorm_stats_kernel<bf16_t, 8>), grid, block, 0, st, p);
        hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(F), block, 0, st, p, C / G);
        hipLaunchKernelGGL((groupnorm_apply_kernel<bf16_t, 8>), grid, block, 0, st, p);
    } else if (dt == M4D_BF16) {
        hipLaunchKernelGGL((groupnorm_stats_kernel<bf16_t, 4>), grid, block, 0, st, p);
        hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(F), block, 0, st, p, C / G);
        hipLaunchKernelGGL((groupnorm_apply_kernel<bf16_t, 4>), grid, block, 0, st, p);
    } else {
        hipLaunchKernelGGL((groupnorm_stats_kernel<float, 4>), grid, block, 0, st, p);
        hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(F), block, 0, st, p, C / G);
        hipLaunchKernelGGL((groupnorm_apply_kernel<float, 4>), grid, block, 0, st, p);
    }
    M4D_CHECK_LAUNCH("groupnorm_cl");
    return 0;
}

extern "C" int64_t m4d_groupnorm_cl_workspace(int F, int64_t HW, int G) {      // per-workgroup partial sums + the per-frame (mean, rstd)
    return (int64_t)F * ((HW + GN_PPB - 1) / GN_PPB) * G * 2 + (int64_t)F * G * 2;
}

extern "C" int m4d_softmax_rows(m4d_dtype in_dt, const void* x, int64_t ldx, m4d_dtype out_dt, void* out, int64_t ldo,
                                int64_t rows, int C, int Cpad, float scale, m4d_stream stream) {
    M4D_CHECK_ARG(x && out && rows > 0 && C > 0 && Cpad >= C && ldo >= Cpad && ldx >= C, "softmax_rows: bad arguments");
    dim3 grid((unsigned)rows), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (in_dt == M4D_F32 && out_dt == M4D_F32) hipLaunchKernelGGL((softmax_rows_kernel<float, float>), grid, block, 0, st, (const float*)x, ldx, (float*)out, ldo, C, Cpad, scale);
    else if (in_dt == M4D_F32 && out_dt == M4D_BF16) hipLaunchKernelGGL((softmax_rows_kernel<float, bf16_t>), grid, block, 0, st, (const float*)x, ldx, (bf16_t*)out, ldo, C, Cpad, scale);
    else if (in_dt == M4D_BF16 && out_dt == M4D_BF16) hipLaunchKernelGGL((softmax_rows_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, C, Cpad, scale);
    else { m4d_set_error("softmax_rows: unsupported dtype pair"); return -1; }
    M4D_CHECK_LAUNCH("softmax_rows");
    return 0;
}

extern "C" int m4d_ncthw_to_cl(m4d_dtype src_dt, const void* src, m4d_dtype dst_dt, void* dst, int64_t dst_pixel_stride, int C,
                               int Cp, int T, int H, int W, float scale, float shift, const float* ch_scale,
                               const float* ch_shift, m4d_stream stream) {
    M4D_CHECK_ARG(src && dst && C > 0 && Cp >= C && dst_pixel_stride >= Cp && T > 0 && H > 0 && W > 0, "ncthw_to_cl: bad arguments");
    M4D_CHECK_ARG((ch_scale == nullptr) == (ch_shift == nullptr), "ncthw_to_cl: ch_scale and ch_shift go together");
    LayoutArgs p{src, dst, nullptr, ch_scale, ch_shift, dst_pixel_stride, C, Cp, T, H, W, 0, scale, shift};
    dim3 grid(grid_for((int64_t)T * H * W * Cp)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (src_dt == M4D_F32 && dst_dt == M4D_F32) hipLaunchKernelGGL((ncthw_to_cl_kernel<float, float>), grid, block, 0, st, p);
    else if (src_dt == M4D_F32 && dst_dt == M4D_BF16) hipLaunchKernelGGL((ncthw_to_cl_kernel<float, bf16_t>), grid, block, 0, st, p);
    else if (src_dt == M4D_BF16 && dst_dt == M4D_BF16) hipLaunchKernelGGL((ncthw_to_cl_kernel<bf16_t, bf16_t>), grid, block, 0, st, p);
    else if (src_dt == M4D_BF16 && dst_dt == M4D_F32) hipLaunchKernelGGL((ncthw_to_cl_kernel<bf16_t, float>), grid, block, 0, st, p);
    else { m4d_set_error("ncthw_to_cl: bad dtypes"); return -1; }
    M4D_CHECK_LAUNCH("ncthw_to_cl");
    return 0;
}

extern "C" int m4d_cl_to_ncthw(m4d_dtype src_dt, const void* src, int64_t src_pixel_stride, m4d_dtype dst_dt, void* dst, int C,
                               int T, int H, int W, float scale, float shift, const float* ch_scale, const float* ch_shift,
                               int act, const void* aux, m4d_stream stream) {
    M4D_CHECK_ARG(src && dst && C > 0 && src_pixel_stride >= C && T > 0 && H > 0 && W > 0, "cl_to_ncthw: bad arguments");
    M4D_CHECK_ARG(act >= 0 && act <= 2 && (act != 2 || aux), "cl_to_ncthw: act 2 needs aux");
    M4D_CHECK_ARG((ch_scale == nullptr) == (ch_shift == nullptr), "cl_to_ncthw: ch_scale and ch_shift go together");
    LayoutArgs p{src, dst, aux, ch_scale, ch_shift, src_pixel_stride, C, C, T, H, W, act, scale, shift};
    dim3 grid(grid_for((int64_t)T * H * W * C)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (src_dt == M4D_F32 && dst_dt == M4D_F32) hipLaunchKernelGGL((cl_to_ncthw_kernel<float, float>), grid, block, 0, st, p);
    else if (src_dt == M4D_BF16 && dst_dt == M4D_BF16) hipLaunchKernelGGL((cl_to_ncthw_kernel<bf16_t, bf16_t>), grid, block, 0, st, p);
    else if (src_dt == M4D_BF16 && dst_dt == M4D_F32) hipLaunchKernelGGL((cl_to_ncthw_kernel<bf16_t, float>), grid, block, 0, st, p);
    else if (src_dt == M4D_F32 && dst_dt == M4D_BF16) hipLaunchKernelGGL((cl_to_ncthw_kernel<float, bf16_t>), grid, block, 0, st, p);
    else { m4d_set_error("cl_to_ncthw: bad dtypes"); return -1; }
    M4D_CHECK_LAUNCH("cl_to_ncthw");
    return 0;
}
