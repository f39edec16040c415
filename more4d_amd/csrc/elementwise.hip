// HBM-bound kernels of the DiT path: fused LayerNorm+modulate(+guidance), RMSNorm+RoPE, patch gather /
// scatter, CFG+Euler, unary casts.  All are one-pass over their tensor: rows are held in registers
// (16-B vector loads, wave64 shuffle reductions, one LDS hop across the 4 waves of a workgroup).
#include "common.h"
#include "more4d_hip.h"

namespace {

// One WAVE per row (4 rows per 256-thread workgroup): the row lives in registers (MAXV float4 per lane), both
// reductions are wave64 shuffles, no LDS and no barrier, so many rows are in flight per CU (HBM latency hiding).
// MAXV is picked per launch from C: 8 (C <= 2048), 20 (C <= 5120), 32 (C <= 8192).

// ------------------------------------------------------------------ LayerNorm + modulate
struct LnArgs {
    const void* x; void* out;
    const float *shift, *scale, *ln_w, *ln_b, *g_ss, *g_gate;
    int64_t rows, rows_per_sample, mod_stride, g_period, g_len, g_rows;      // g_rows: rows per guidance sample (the modulation may be per row)
    int C; float eps;
};

template <typename TI, typename TO, int MAXV, bool FULL = false>      // FULL: C == MAXV * 256 (host-checked): no per-piece lane predicates (see the rows kernel below)
__global__ __launch_bounds__(256) void ln_modulate_kernel(LnArgs p) {
    constexpr int G = 64;
    const int tid = threadIdx.x;
    const int sub = FULL ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), lt = tid & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + sub;
    const bool active = row < p.rows;   // wave-uniform
    const int C = p.C, nv = C >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
    const TI* xr = (const TI*)p.x + (active ? row : 0) * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lt + i * G;
        if (FULL || c4 < nv) {
            v[i] = load4(xr + c4 * 4);
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
    if (!active) return;
    const int64_t sample = row / p.rows_per_sample, l = row % p.rows_per_sample;
    const float* sh = p.shift ? p.shift + sample * p.mod_stride : nullptr;
    const float* sc = p.scale ? p.scale + sample * p.mod_stride : nullptr;
    const float* gs = nullptr;
    if (p.g_ss) {      // guidance table row: by the row's position inside its GUIDANCE sample (g_rows rows; == rows_per_sample unless the
        const int64_t gsample = row / p.g_rows, gl = row % p.g_rows;      // modulation is per token, wan_transformer4d.py:655-657 with :757-783)
        if (gl < p.g_len) gs = p.g_ss + (gsample * p.g_period + gl % p.g_period) * 2 * C;
    }
    (void)l;
    TO* orow = (TO*)p.out + row * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lt + i * G;
        if (FULL || c4 < nv) {
            const int c = c4 * 4;
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd;
            if (p.ln_w) { y = y * load4(p.ln_w + c) + load4(p.ln_b + c); }
            if (sc) { y = y * (1.f + load4(sc + c)) + load4(sh + c); }
            if (gs) {   // x * (1 + scale*gate) + shift*gate  (wan_transformer4d.py:781)
                const f32x4 g = load4(p.g_gate + c);
                y = y * (1.f + load4(gs + c) * g) + load4(gs + C + c) * g;
            }
            store4(orow + c, y);
        }
    }
}


// The DiT's two uses of ln_modulate — (1 + scale) / shift of one sample, or the affine weight / bias — apply the SAME two C-vectors to
// every row of a sample; read per row from global memory they are 2/3 of what a row moves through L2 and 40 dependent L2 loads behind the
// row reductions.  Here a workgroup keeps them in LDS as A[c] = 1 + scale (or ln_w) and B[c] = shift (or ln_b) and walks row blocks
// b, b + gridDim.x, ... (one row per wave, as above), re-filling the LDS when the walk crosses a sample boundary.  Per-row arithmetic and
// its order are the kernel's above: same bits.  (Four waves per SIMD like the kernel above: without the bound hipcc takes 173 VGPRs
// for the loop and halves the occupancy — the first version of this kernel was 9 % slower for that reason alone.)
template <typename TI, typename TO, int MAXV, int OCC = 3, bool NT = false, bool FULL = false>
__global__ __launch_bounds__(256, OCC) void ln_modulate_rows_kernel(LnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float ln_ab[];      // A[C], B[C]
    constexpr int G = 64;
    const int tid = threadIdx.x;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6), lt = tid & 63;      // (wave-uniform row: scalar base + lane offset addressing)
    const int C = p.C, nv = C >> 2;
    const bool affine = p.ln_w != nullptr;
    const int64_t nblk = (p.rows + 3) / 4;
    int64_t cur_sample = -1;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t sample = affine ? 0 : (blk * 4) / p.rows_per_sample;       // (host: rows_per_sample % 4 == 0, a block never straddles)
        if (sample != cur_sample) {
            if (cur_sample >= 0) __syncthreads();
            const float* a_src = affine ? p.ln_w : p.scale + sample * p.mod_stride;
            const float* b_src = affine ? p.ln_b : p.shift + sample * p.mod_stride;
            for (int c4 = tid; c4 < nv; c4 += 256) {
                f32x4 a = load4(a_src + c4 * 4);
                if (!affine) a = 1.f + a;
                *reinterpret_cast<f32x4*>(ln_ab + c4 * 4) = a;
                *reinterpret_cast<f32x4*>(ln_ab + C + c4 * 4) = load4(b_src + c4 * 4);
            }
            __syncthreads();
            cur_sample = sample;
        }
        const int64_t row = blk * 4 + sub;
        const bool active = row < p.rows;   // wave-uniform
        f32x4 v[MAXV];
        float s = 0.f;
        const TI* xr = (const TI*)p.x + (active ? row : 0) * C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * G;
            if (FULL || c4 < nv) {      // FULL: C == MAXV * 256 (host-checked), no per-piece lane predicates
                // (scalar base stepped per group of four pieces + one lane offset + immediates: not twenty offset registers)
                const TI* xg = xr + (i >> 2) * (4 * G * 4);
                const int cg = lt * 4 + (i & 3) * (G * 4);
                if constexpr (NT && sizeof(TI) == 4) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xg + cg));
                else v[i] = load4(xg + cg);
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * G;
            if (FULL || c4 < nv) {      // FULL: C == MAXV * 256 (host-checked), no per-piece lane predicates
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + p.eps);
        if (!active) continue;
        TO* orow = (TO*)p.out + row * C;
        const float* lnB = (const float*)__builtin_assume_aligned(ln_ab + C, 16);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lt + i * G;
            if (FULL || c4 < nv) {      // FULL: C == MAXV * 256 (host-checked), no per-piece lane predicates
                const int c = c4 * 4;
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd;
                y = y * *reinterpret_cast<const f32x4*>(ln_ab + c) + *reinterpret_cast<const f32x4*>(lnB + c);
                store4(orow + (i >> 2) * (4 * G * 4) + lt * 4 + (i & 3) * (G * 4), y);
            }
            if ((i & (OCC >= 4 ? 1 : 3)) == (OCC >= 4 ? 1 : 3)) __builtin_amdgcn_sched_barrier(0);      // (A / B reads of at most four / two pieces in flight: hoisted further they cost registers)
        }
    }
}

// ------------------------------------------------------------------ RMSNorm (+ RoPE)
struct RmsArgs {
    void* x[2]; const float* w[2];
    const float *cos_t, *sin_t;
    int64_t ld, rows, rows_per_sample, rope_len, pos_offset;
    int C, head_dim; float eps;
};

// 16-byte accesses for both dtypes: EPV = 8 bf16 / 4 fp32 elements per lane per access
template <typename T, int EPV> M4D_DEV void load_vec(const T* p, float (&v)[EPV]) {
    if constexpr (EPV == 8) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
    } else {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r[e];
    }
}
template <typename T, int EPV> M4D_DEV void store_vec(T* p, const float (&v)[EPV]) {
    if constexpr (EPV == 8) {
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

// one rotation (a, b) -> (a c - b s, a s + b c) with the contraction spelled out, so that every kernel that rotates rounds alike
M4D_DEV void rope_pair(float a, float b, float c, float s, float& y0, float& y1) {
#pragma clang fp contract(off)
    const float bs = b * s, bc = b * c;
    y0 = __builtin_fmaf(a, c, -bs);
    y1 = __builtin_fmaf(a, s, bc);
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(RmsArgs p) {
    constexpr int EPV = 16 / sizeof(T);
    constexpr int NVEC = MAXV * 4 / EPV;
    const int tid = threadIdx.x;
    const int sub = tid >> 6, lt = tid & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + sub;
    const bool active = row < p.rows;
    const int which = blockIdx.y;
    const int C = p.C, nvec = C / EPV;     // host guarantees C % 8 == 0 for bf16
    T* xr = (T*)p.x[which] + (active ? row : 0) * p.ld;
    const float* w = p.w[which];
    float v[NVEC][EPV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int cv = lt + i * 64;
        if (cv < nvec) {
            load_vec<T, EPV>(xr + cv * EPV, v[i]);
#pragma unroll
            for (int e = 0; e < EPV; ++e) s += v[i][e] * v[i][e];
        }
    }
    const float inv = rsqrtf(wave_sum(s) / C + p.eps);
    if (!active) return;
    const bool norm = w != nullptr;           // NULL weight: RoPE only (qk_norm=False, reference :431-432: norm_q / norm_k = nn.Identity)
    const int64_t l = row % p.rows_per_sample;
    const bool rot = p.cos_t && l < p.rope_len;
    const int half = p.head_dim >> 1;
    const float* ct = rot ? p.cos_t + (p.pos_offset + l) * half : nullptr;
    const float* st = rot ? p.sin_t + (p.pos_offset + l) * half : nullptr;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int cv = lt + i * 64;
        if (cv < nvec) {
            const int c = cv * EPV;
            float y[EPV];
            // reference: (x * rsqrt(..)).to(x.dtype) * weight  (wan_transformer4d.py:391-394)
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                if (norm) {
                    const f32x4 wv = load4(w + c + e);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[e + j] = round_through<T>(v[i][e + j] * inv) * wv[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[e + j] = v[i][e + j];
                }
            }
            if (rot) {
                const int pi = (c % p.head_dim) >> 1;  // first pair index inside the head; pairs are (c+2j, c+2j+1)
#pragma unroll
                for (int e = 0; e < EPV; e += 4) {
                    const f32x2 cs = *reinterpret_cast<const f32x2*>(ct + pi + (e >> 1));
                    const f32x2 sn = *reinterpret_cast<const f32x2*>(st + pi + (e >> 1));
                    const float a0 = y[e], b0 = y[e + 1], a1 = y[e + 2], b1 = y[e + 3];
                    rope_pair(a0, b0, cs[0], sn[0], y[e], y[e + 1]);
                    rope_pair(a1, b1, cs[1], sn[1], y[e + 2], y[e + 3]);
                }
            }
            store_vec<T, EPV>(xr + c, y);
        }
    }
}

// The DiT's shape of the kernel above (C = 5120: every lane owns exactly NVEC 16-byte pieces of its row; rows >= 4096) as a straight-line,
// persistent form.  What the general kernel costs there (ISA, round 4): its per-piece lane predicates put every row load into a basic
// block of its own, so hipcc waits for each piece before requesting the next (ten serial HBM round trips per row), and the second pass
// loads the weight (20 KB of fp32 per 10 KB row) and eight-byte cos / sin pairs from L2 behind the reduction, piece by piece: 130 vector
// memory instructions per row of which 20 move the row.  Here a workgroup walks row blocks b, b + gridDim.x, ... (one row per wave), all
// pieces of a row are requested back to back, the weight sits in LDS (staged once per workgroup), and the rotation pairs of a lane are the
// SAME for all its pieces (a piece step is 64 lanes x 16 bytes = a multiple of head_dim): two 16-byte loads per row, requested with the
// row.  Arithmetic and its order are the general kernel's: same bits.
template <typename T, int MAXV>
__global__ __launch_bounds__(256, 4) void rmsnorm_rope_rows_kernel(RmsArgs p) {
    constexpr int EPV = 16 / sizeof(T);
    constexpr int NVEC = MAXV * 4 / EPV;
    extern __shared__ __attribute__((aligned(16))) float rms_w[];      // C weights (norm == true)
    const int tid = threadIdx.x;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6), lt = tid & 63;
    const int which = blockIdx.y;
    const int C = p.C;
    const float* w = p.w[which];
    const bool norm = w != nullptr;
    if (norm) {
        for (int c4 = tid; c4 < (C >> 2); c4 += 256) *reinterpret_cast<f32x4*>(rms_w + c4 * 4) = load4(w + c4 * 4);
        __syncthreads();
    }
    const int half = p.head_dim >> 1;
    const int pi = ((lt * EPV) % p.head_dim) >> 1;      // first pair index inside the head of every piece of this lane
    const int64_t nblk = (p.rows + 3) / 4;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row = blk * 4 + sub;
        if (row >= p.rows) continue;                    // wave-uniform
        T* xr = (T*)p.x[which] + row * p.ld;
        const int64_t l = row % p.rows_per_sample;
        const bool rot = p.cos_t && l < p.rope_len;
        float v[NVEC][EPV];
#pragma unroll
        for (int i = 0; i < NVEC; ++i) load_vec<T, EPV>(xr + (i >> 2) * (4 * 64 * EPV) + lt * EPV + (i & 3) * (64 * EPV), v[i]);
        float cs[EPV / 2], sn[EPV / 2];
        if (rot) {
            const float* ct = p.cos_t + (p.pos_offset + l) * half + pi;
            const float* st = p.sin_t + (p.pos_offset + l) * half + pi;
#pragma unroll
            for (int e = 0; e < EPV / 2; e += 2) {
                const f32x2 c2 = *reinterpret_cast<const f32x2*>(ct + e), s2 = *reinterpret_cast<const f32x2*>(st + e);
                cs[e] = c2[0]; cs[e + 1] = c2[1]; sn[e] = s2[0]; sn[e + 1] = s2[1];
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NVEC; ++i)
#pragma unroll
            for (int e = 0; e < EPV; ++e) s += v[i][e] * v[i][e];
        const float inv = rsqrtf(wave_sum(s) / C + p.eps);
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const int c = (i >> 2) * (4 * 64 * EPV) + lt * EPV + (i & 3) * (64 * EPV);
            float y[EPV];
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                if (norm) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(rms_w + c + e);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[e + j] = round_through<T>(v[i][e + j] * inv) * wv[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[e + j] = v[i][e + j];
                }
            }
            if (rot) {
#pragma unroll
                for (int e = 0; e < EPV; e += 4) {
                    const float c0 = cs[e >> 1], c1 = cs[(e >> 1) + 1], s0 = sn[e >> 1], s1 = sn[(e >> 1) + 1];
                    const float a0 = y[e], b0 = y[e + 1], a1 = y[e + 2], b1 = y[e + 3];
                    rope_pair(a0, b0, c0, s0, y[e], y[e + 1]);
                    rope_pair(a1, b1, c1, s1, y[e + 2], y[e + 3]);
                }
            }
            store_vec<T, EPV>(xr + c, y);
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ------------------------------------------------------------------ patchify / unpatchify
struct PatchArgs {
    const void* src0; const void* src1; void* out;
    int c0, c1, B, F, H, W, pt, ph, pw;
};

// one thread per output element group of pw (contiguous in both source row and output K index)
template <typename TS, typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(PatchArgs p) {
    const int f = p.F / p.pt, h = p.H / p.ph, w = p.W / p.pw;
    const int Ctot = p.c0 + p.c1;
    const int Kdim = Ctot * p.pt * p.ph * p.pw;
    const int64_t total = (int64_t)p.B * f * h * w * Kdim;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx;
        const int kq = (int)(r % p.pw); r /= p.pw;
        const int kp = (int)(r % p.ph); r /= p.ph;
        const int kt = (int)(r % p.pt); r /= p.pt;
        const int c = (int)(r % Ctot); r /= Ctot;
        const int ww = (int)(r % w); r /= w;
        const int hh = (int)(r % h); r /= h;
        const int ff = (int)(r % f); r /= f;
        const int b = (int)r;
        const TS* src; int cc, cn;
        if (c < p.c0) { src = (const TS*)p.src0; cc = c; cn = p.c0; }
        else { src = (const TS*)p.src1; cc = c - p.c0; cn = p.c1; }
        const int64_t si = ((((int64_t)b * cn + cc) * p.F + (ff * p.pt + kt)) * p.H + (hh * p.ph + kp)) * p.W + (ww * p.pw + kq);
        ((TO*)p.out)[idx] = (TO)(float)src[si];
    }
}

struct UnpatchArgs {
    const float* tok; void* out;
    int64_t tok_bs, tok_row0;
    int B, c, F, H, W, pt, ph, pw;
};

template <typename TO>
__global__ __launch_bounds__(256) void unpatchify_kernel(UnpatchArgs p) {
    // out[b, ch, f*pt+kt, h*ph+kp, w*pw+kq] = tok[b, row0 + (f,h,w), ((kt*ph+kp)*pw+kq)*c + ch]
    const int Fo = p.F * p.pt, Ho = p.H * p.ph, Wo = p.W * p.pw;
    const int64_t total = (int64_t)p.B * p.c * Fo * Ho * Wo;
    const int vec = p.pt * p.ph * p.pw * p.c;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho); r /= Ho;
        const int z = (int)(r % Fo); r /= Fo;
        const int ch = (int)(r % p.c); r /= p.c;
        const int b = (int)r;
        const int f = z / p.pt, kt = z % p.pt, h = y / p.ph, kp = y % p.ph, w = x / p.pw, kq = x % p.pw;
        const int64_t trow = p.tok_row0 + ((int64_t)f * p.H + h) * p.W + w;
        const float v = p.tok[b * p.tok_bs + trow * vec + ((kt * p.ph + kp) * p.pw + kq) * p.c + ch];
        ((TO*)p.out)[idx] = (TO)v;
    }
}

// ------------------------------------------------------------------ CFG + Euler
template <typename TV, typename TR>
__global__ __launch_bounds__(256) void cfg_euler_kernel(float* x, const TV* v, int64_t n, float g, float ds) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 vu = load4(v + i * 4), vc = load4(v + n + i * 4);
        f32x4 xv = load4(x + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // noise_pred computed in the model dtype (pipeline :820-822), step in fp32 (:760), cast back (:789)
            const float np_ = round_through<TV>(vu[e] + g * (vc[e] - vu[e]));
            xv[e] = round_through<TR>(xv[e] + ds * np_);
        }
        store4(x + i * 4, xv);
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void unary_kernel(const TI* x, TO* out, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = (float)x[i];
        if (act == 1) v = silu_f(v);
        else if (act == 2) v = gelu_tanh_f(v);
        else if (act == 3) v = gelu_erf_f(v);
        out[i] = (TO)v;
    }
}

// eight elements per thread and iteration with 16-byte accesses (n % 8 == 0, both pointers 16-byte aligned: the residual-stream casts of
// the DiT): the element-wise form above moves 4 + 2 bytes per lane and instruction — 1.8 TB/s where this one is bound by HBM
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void unary_vec8_kernel(const TI* x, TO* out, int64_t n8, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        if constexpr (sizeof(TI) == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
        } else {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
        }
        if (act != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act == 1 ? silu_f(v[e]) : act == 2 ? gelu_tanh_f(v[e]) : gelu_erf_f(v[e]);
        }
        if constexpr (sizeof(TO) == 4) {
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
            *reinterpret_cast<f32x4*>(out + i * 8) = a;
            *reinterpret_cast<f32x4*>(out + i * 8 + 4) = b;
        } else {
            bf16x8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x8*>(out + i * 8) = a;
        }
    }
}

__global__ __launch_bounds__(256) void add_bcast_kernel(const float* a, const float* bias, float* out, int64_t total, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + bias[i % n];
}

// out = a*x + b*y (float32): TeaCache residual add / subtract (wan_transformer4d.py:1226, 1268-1270)
__global__ __launch_bounds__(256) void axpby_kernel(const float* x, const float* y, float* out, int64_t n, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a * x[i] + b * y[i];
}

// out = a0*x0 + a1*x1 + a2*x2 + a3*x3 (float32, unused terms have a null pointer): the multistep solver updates
struct LinArgs { const float* x[4]; float a[4]; float* out; int64_t n; };
__global__ __launch_bounds__(256) void lincomb_kernel(LinArgs p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (p.x[k]) v = fmaf(p.a[k], p.x[k][i], v);
        p.out[i] = v;
    }
}

// Merge two partial attention results over disjoint key sets (same queries): o_a <- wa*o_a + wb*o_b, lse_a <- lse with
// lse = log2(2^lse_a + 2^lse_b) (log2 domain, as m4d_attention_lse writes it), wa = 2^(lse_a - lse), wb = 2^(lse_b - lse).
// A side with no valid key has lse = -inf and weight 0.  16 bytes per thread; the two weights of a (row, head) are recomputed
// by the 16 / 32 threads that share them (broadcast loads).
struct MergeArgs { void* oa; const void* ob; float* la; const float* lb; int64_t oa_bs, oa_ls, ob_bs, ob_ls, L; int B, heads, D; };
template <typename T, int EPV>
__global__ __launch_bounds__(256) void attn_merge_kernel(MergeArgs p) {
    const int vpr = p.heads * p.D / EPV;                    // vectors per row
    const int64_t total = (int64_t)p.B * p.L * vpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpr);
        const int64_t r = i / vpr;
        const int64_t l = r % p.L;
        const int b = (int)(r / p.L);
        const int c = v * EPV, h = c / p.D;
        const int64_t si = ((int64_t)b * p.heads + h) * p.L + l;
        const float la = p.la[si], lb = p.lb[si];
        const float mx = fmaxf(la, lb);
        float wa = 0.f, wb = 0.f, lt = -INFINITY;
        if (mx > -INFINITY) {
            const float ea = __builtin_amdgcn_exp2f(la - mx), eb = __builtin_amdgcn_exp2f(lb - mx);
            const float inv = 1.f / (ea + eb);
            wa = ea * inv; wb = eb * inv;
            lt = mx + log2f(ea + eb);
        }
        T* pa = (T*)p.oa + b * p.oa_bs + l * p.oa_ls + c;
        const T* pb = (const T*)p.ob + b * p.ob_bs + l * p.ob_ls + c;
        float xa[EPV], xb[EPV];
        load_vec<T, EPV>(pa, xa);
        load_vec<T, EPV>(pb, xb);
#pragma unroll
        for (int e = 0; e < EPV; ++e) xa[e] = wa * xa[e] + wb * xb[e];
        store_vec<T, EPV>(pa, xa);
        if (c % p.D == 0) p.la[si] = lt;     // one writer per (row, head); every reader of la[si] belongs to this wave-front
    }
}

// out[0] = sum|cur - prev|, out[1] = sum|prev| (one workgroup; TeaCache's relative-L1 on the [B,6,C] modulation, cache_utils.py)
__global__ __launch_bounds__(1024) void rel_l1_kernel(const float* prev, const float* cur, float* out, int64_t n) {
    __shared__ float rd[16], rp[16];
    float d = 0.f, pp = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) { d += fabsf(cur[i] - prev[i]); pp += fabsf(prev[i]); }
    d = wave_sum(d); pp = wave_sum(pp);
    if ((threadIdx.x & 63) == 0) { rd[threadIdx.x >> 6] = d; rp[threadIdx.x >> 6] = pp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < 16; ++i) { a += rd[i]; b += rp[i]; }
        out[0] = a; out[1] = b;
    }
}

// bilinear resize (align_corners = False, torch semantics) of a channels-last map [B, Hi, Wi, C] -> [B, Ho, Wo, C]
template <typename T>
__global__ __launch_bounds__(256) void bilinear_cl_kernel(const T* x, T* out, int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int64_t total = (int64_t)B * Ho * Wo * C;
    const float sh = (float)Hi / Ho, sw = (float)Wi / Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float fy = fmaxf((ho + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf((wo + 0.5f) * sw - 0.5f, 0.f);
        const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
        const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
        const float ly = fy - y0, lx = fx - x0;
        const T* xb = x + (int64_t)b * Hi * Wi * C + c;
        const float v00 = (float)xb[((int64_t)y0 * Wi + x0) * C], v01 = (float)xb[((int64_t)y0 * Wi + x1) * C];
        const float v10 = (float)xb[((int64_t)y1 * Wi + x0) * C], v11 = (float)xb[((int64_t)y1 * Wi + x1) * C];
        out[i] = (T)((1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11));
    }
}

inline unsigned grid_for(int64_t n, int per_block = 256) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > 2048 * 4) g = 2048 * 4;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int m4d_ln_modulate(m4d_dtype x_dt, const void* x, m4d_dtype out_dt, void* out, int64_t rows, int C,
                               int64_t rows_per_sample, const float* shift, const float* scale, int64_t mod_stride,
                               const float* ln_w, const float* ln_b, float eps, const float* g_ss,
                               const float* g_gate, int64_t g_period, int64_t g_len, m4d_stream stream) {
    return m4d_ln_modulate_g(x_dt, x, out_dt, out, rows, C, rows_per_sample, shift, scale, mod_stride, ln_w, ln_b, eps, g_ss, g_gate, g_period,
                             g_len, 0, stream);
}

extern "C" int m4d_ln_modulate_g(m4d_dtype x_dt, const void* x, m4d_dtype out_dt, void* out, int64_t rows, int C,
                                 int64_t rows_per_sample, const float* shift, const float* scale, int64_t mod_stride,
                                 const float* ln_w, const float* ln_b, float eps, const float* g_ss,
                                 const float* g_gate, int64_t g_period, int64_t g_len, int64_t g_rows, m4d_stream stream) {
    M4D_CHECK_ARG(x && out && rows > 0, "ln_modulate: null/empty");
    M4D_CHECK_ARG(C % 4 == 0 && C > 0 && C <= 8192, "ln_modulate: C=%d must be a multiple of 4 and <= 8192", C);
    M4D_CHECK_ARG((shift == nullptr) == (scale == nullptr), "ln_modulate: shift and scale must both be set or both NULL");
    M4D_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr), "ln_modulate: ln_w and ln_b must both be set or both NULL");
    M4D_CHECK_ARG(g_ss == nullptr || (g_gate && g_period > 0), "ln_modulate: guidance needs gate and period");
    LnArgs p;
    p.x = x; p.out = out; p.shift = shift; p.scale = scale; p.ln_w = ln_w; p.ln_b = ln_b; p.g_ss = g_ss; p.g_gate = g_gate;
    p.rows = rows; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : rows; p.mod_stride = mod_stride;
    p.g_period = g_period > 0 ? g_period : 1; p.g_len = g_len; p.C = C; p.eps = eps;
    p.g_rows = g_rows > 0 ? g_rows : p.rows_per_sample;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256), grid((unsigned)((rows + 3) / 4));
    // LDS-staged, persistent form: exactly one of (scale, shift) / (ln_w, ln_b), no guidance terms, blocks of 4 rows inside one sample
    M4D_ENV_ONCE(ln_rows, "M4D_LN_ROWS", 1);
    const bool rows_form = ln_rows && !g_ss && ((scale != nullptr) != (ln_w != nullptr)) && (ln_w || p.rows_per_sample % 4 == 0) &&
                           rows >= 4096 && C > 2048 && C <= 5120;
    if (rows_form) {
        static int ncu_dev[64] = {0};       // per device: a process may drive several GPUs
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
        if (!ncu_dev[dev]) {
            int v = 0;
            ncu_dev[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        }
        const int ncu = ncu_dev[dev];
        const size_t lds = (size_t)C * 8;
        // C == 5120: every lane owns exactly twenty pieces of its row — no per-piece lane predicates (FULL).  With them each piece sits in a
        // basic block of its own: 106 SGPRs of masks, 168 VGPRs, twenty offset registers; without: 67 / 134 and 4.5 -> 5.2 TB/s
        // (M4D_LN_VAR=5: the predicated form, A/B; four waves per SIMD and non-temporal row loads measured no different: tools/ab_elem.sh)
        M4D_ENV_ONCE(ln_var, "M4D_LN_VAR", 0);
        const int per_cu = 3;          // three waves per SIMD
        grid = dim3((unsigned)std::min<int64_t>((rows + 3) / 4, (int64_t)ncu * per_cu));
#define LN_ROWS_LAUNCH(TI, TO)                                                                                        \
    do {                                                                                                             \
        if (C == 5120 && ln_var != 5) hipLaunchKernelGGL((ln_modulate_rows_kernel<TI, TO, 20, 3, false, true>), grid, block, lds, st, p); \
        else hipLaunchKernelGGL((ln_modulate_rows_kernel<TI, TO, 20>), grid, block, lds, st, p);                     \
    } while (0)
        if (x_dt == M4D_F32 && out_dt == M4D_F32) LN_ROWS_LAUNCH(float, float);
        else if (x_dt == M4D_F32 && out_dt == M4D_BF16) LN_ROWS_LAUNCH(float, bf16_t);
        else if (x_dt == M4D_BF16 && out_dt == M4D_BF16) LN_ROWS_LAUNCH(bf16_t, bf16_t);
        else if (x_dt == M4D_BF16 && out_dt == M4D_F32) LN_ROWS_LAUNCH(bf16_t, float);
        else { m4d_set_error("ln_modulate: bad dtypes"); return -1; }
#undef LN_ROWS_LAUNCH
        M4D_CHECK_LAUNCH("ln_modulate");
        return 0;
    }
#define LN_LAUNCH(TI, TO)                                                                             \
    do {                                                                                              \
        if (C == 5120) hipLaunchKernelGGL((ln_modulate_kernel<TI, TO, 20, true>), grid, block, 0, st, p); \
        else if (C <= 2048) hipLaunchKernelGGL((ln_modulate_kernel<TI, TO, 8>), grid, block, 0, st, p);    \
        else if (C <= 5120) hipLaunchKernelGGL((ln_modulate_kernel<TI, TO, 20>), grid, block, 0, st, p); \
        else hipLaunchKernelGGL((ln_modulate_kernel<TI, TO, 32>), grid, block, 0, st, p);             \
    } while (0)
    if (x_dt == M4D_F32 && out_dt == M4D_F32) LN_LAUNCH(float, float);
    else if (x_dt == M4D_F32 && out_dt == M4D_BF16) LN_LAUNCH(float, bf16_t);
    else if (x_dt == M4D_BF16 && out_dt == M4D_BF16) LN_LAUNCH(bf16_t, bf16_t);
    else if (x_dt == M4D_BF16 && out_dt == M4D_F32) LN_LAUNCH(bf16_t, float);
    else { m4d_set_error("ln_modulate: bad dtypes"); return -1; }
#undef LN_LAUNCH
    M4D_CHECK_LAUNCH("ln_modulate");
    return 0;
}

extern "C" int m4d_rmsnorm_rope(m4d_dtype dt, void* x0, void* x1, int64_t ld, const float* w0, const float* w1,
                                int64_t rows, int C, int head_dim, float eps, const float* cos_t, const float* sin_t,
                                int64_t rows_per_sample, int64_t rope_len, int64_t pos_offset, m4d_stream stream) {
    M4D_CHECK_ARG(x0 && rows > 0, "rmsnorm_rope: null/empty");
    M4D_CHECK_ARG(w0 || cos_t, "rmsnorm_rope: neither a norm weight nor rope tables: nothing to do");
    M4D_CHECK_ARG(x1 == nullptr || (w1 != nullptr) == (w0 != nullptr), "rmsnorm_rope: x1 must be normalised like x0 (both weights or neither)");
    M4D_CHECK_ARG(C % 4 == 0 && C > 0 && C <= 8192, "rmsnorm_rope: C=%d must be a multiple of 4 and <= 8192", C);
    M4D_CHECK_ARG(head_dim > 0 && head_dim % 8 == 0 && C % head_dim == 0, "rmsnorm_rope: head_dim=%d must divide C and be a multiple of 8", head_dim);
    M4D_CHECK_ARG(dt != M4D_BF16 || (ld % 8 == 0 && ((uintptr_t)x0 % 16) == 0 && (x1 == nullptr || ((uintptr_t)x1 % 16) == 0)), "rmsnorm_rope: bf16 rows must be 16-byte aligned");
    M4D_CHECK_ARG((cos_t == nullptr) == (sin_t == nullptr), "rmsnorm_rope: cos and sin must both be set or both NULL");
    M4D_CHECK_ARG(ld % 4 == 0 && ld >= C, "rmsnorm_rope: bad ld");
    RmsArgs p;
    p.x[0] = x0; p.x[1] = x1; p.w[0] = w0; p.w[1] = w1; p.cos_t = cos_t; p.sin_t = sin_t;
    p.ld = ld; p.rows = rows; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : rows;
    p.rope_len = rope_len; p.pos_offset = pos_offset; p.C = C; p.head_dim = head_dim; p.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256), grid((unsigned)((rows + 3) / 4), x1 ? 2 : 1);
    // straight-line persistent form for the DiT's width (M4D_RMS_ROWS=0: the general kernel, same bits)
    M4D_ENV_ONCE(rms_rows, "M4D_RMS_ROWS", 1);
    const int epv = dt == M4D_BF16 ? 8 : 4;
    if (rms_rows && C == 5120 && rows >= 4096 && (64 * epv) % head_dim == 0 && (dt == M4D_BF16 || dt == M4D_F32) &&
        (cos_t == nullptr || (((uintptr_t)cos_t | (uintptr_t)sin_t) % 8 == 0 && head_dim % 4 == 0))) {
        static int ncu_dev[64] = {0};       // per device: a process may drive several GPUs
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
        if (!ncu_dev[dev]) {
            int v = 0;
            ncu_dev[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        }
        const int64_t per_y = std::max<int64_t>(1, (int64_t)ncu_dev[dev] * 4 / (x1 ? 2 : 1));
        const dim3 gr((unsigned)std::min<int64_t>((rows + 3) / 4, per_y), x1 ? 2 : 1);
        const size_t lds = w0 ? (size_t)C * 4 : 0;
        if (dt == M4D_BF16) hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<bf16_t, 20>), gr, block, lds, st, p);
        else hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<float, 20>), gr, block, lds, st, p);
        M4D_CHECK_LAUNCH("rmsnorm_rope");
        return 0;
    }
#define RMS_LAUNCH(T)                                                                                 \
    do {                                                                                              \
        if (C <= 2048) hipLaunchKernelGGL((rmsnorm_rope_kernel<T, 8>), grid, block, 0, st, p);        \
        else if (C <= 5120) hipLaunchKernelGGL((rmsnorm_rope_kernel<T, 20>), grid, block, 0, st, p);  \
        else hipLaunchKernelGGL((rmsnorm_rope_kernel<T, 32>), grid, block, 0, st, p);                 \
    } while (0)
    if (dt == M4D_BF16) RMS_LAUNCH(bf16_t);
    else if (dt == M4D_F32) RMS_LAUNCH(float);
    else { m4d_set_error("rmsnorm_rope: bad dtype"); return -1; }
#undef RMS_LAUNCH
    M4D_CHECK_LAUNCH("rmsnorm_rope");
    return 0;
}

extern "C" int m4d_patchify(m4d_dtype src_dt, const void* src0, int c0, const void* src1, int c1, m4d_dtype out_dt,
                            void* out, int B, int F, int H, int W, int pt, int ph, int pw, m4d_stream stream) {
    M4D_CHECK_ARG(src0 && out && c0 > 0 && c1 >= 0 && (c1 == 0 || src1), "patchify: null/empty");
    M4D_CHECK_ARG(pt > 0 && ph > 0 && pw > 0 && F % pt == 0 && H % ph == 0 && W % pw == 0, "patchify: grid %dx%dx%d not divisible by patch %dx%dx%d", F, H, W, pt, ph, pw);
    PatchArgs p{src0, src1, out, c0, c1, B, F, H, W, pt, ph, pw};
    const int64_t total = (int64_t)B * F * H * W * (c0 + c1);
    dim3 block(256), grid(grid_for(total));
    hipStream_t st = (hipStream_t)stream;
    if (src_dt == M4D_F32 && out_dt == M4D_F32) hipLaunchKernelGGL((patchify_kernel<float, float>), grid, block, 0, st, p);
    else if (src_dt == M4D_F32 && out_dt == M4D_BF16) hipLaunchKernelGGL((patchify_kernel<float, bf16_t>), grid, block, 0, st, p);
    else if (src_dt == M4D_BF16 && out_dt == M4D_BF16) hipLaunchKernelGGL((patchify_kernel<bf16_t, bf16_t>), grid, block, 0, st, p);
    else if (src_dt == M4D_BF16 && out_dt == M4D_F32) hipLaunchKernelGGL((patchify_kernel<bf16_t, float>), grid, block, 0, st, p);
    else { m4d_set_error("patchify: bad dtypes"); return -1; }
    M4D_CHECK_LAUNCH("patchify");
    return 0;
}

extern "C" int m4d_unpatchify(const float* tok, int64_t tok_bs, int64_t tok_row0, m4d_dtype out_dt, void* out, int B,
                              int c, int F, int H, int W, int pt, int ph, int pw, m4d_stream stream) {
    M4D_CHECK_ARG(tok && out && B > 0 && c > 0 && F > 0 && H > 0 && W > 0, "unpatchify: null/empty");
    UnpatchArgs p{tok, out, tok_bs, tok_row0, B, c, F, H, W, pt, ph, pw};
    const int64_t total = (int64_t)B * c * F * pt * H * ph * W * pw;
    dim3 block(256), grid(grid_for(total));
    hipStream_t st = (hipStream_t)stream;
    if (out_dt == M4D_F32) hipLaunchKernelGGL((unpatchify_kernel<float>), grid, block, 0, st, p);
    else if (out_dt == M4D_BF16) hipLaunchKernelGGL((unpatchify_kernel<bf16_t>), grid, block, 0, st, p);
    else { m4d_set_error("unpatchify: bad dtype"); return -1; }
    M4D_CHECK_LAUNCH("unpatchify");
    return 0;
}

extern "C" int m4d_cfg_euler(float* x, m4d_dtype v_dt, const void* v, int64_t n, float guidance, float dsigma,
                             m4d_dtype round_dt, m4d_stream stream) {
    M4D_CHECK_ARG(x && v && n > 0 && n % 4 == 0, "cfg_euler: n must be a positive multiple of 4");
    dim3 block(256), grid(grid_for(n / 4));
    hipStream_t st = (hipStream_t)stream;
    if (v_dt == M4D_F32 && round_dt == M4D_F32) hipLaunchKernelGGL((cfg_euler_kernel<float, float>), grid, block, 0, st, x, (const float*)v, n, guidance, dsigma);
    else if (v_dt == M4D_BF16 && round_dt == M4D_BF16) hipLaunchKernelGGL((cfg_euler_kernel<bf16_t, bf16_t>), grid, block, 0, st, x, (const bf16_t*)v, n, guidance, dsigma);
    else if (v_dt == M4D_BF16 && round_dt == M4D_F32) hipLaunchKernelGGL((cfg_euler_kernel<bf16_t, float>), grid, block, 0, st, x, (const bf16_t*)v, n, guidance, dsigma);
    else if (v_dt == M4D_F32 && round_dt == M4D_BF16) hipLaunchKernelGGL((cfg_euler_kernel<float, bf16_t>), grid, block, 0, st, x, (const float*)v, n, guidance, dsigma);
    else { m4d_set_error("cfg_euler: bad dtypes"); return -1; }
    M4D_CHECK_LAUNCH("cfg_euler");
    return 0;
}

extern "C" int m4d_unary(m4d_dtype in_dt, const void* x, m4d_dtype out_dt, void* out, int64_t n, int act,
                         m4d_stream stream) {
    M4D_CHECK_ARG(x && out && n > 0, "unary: null/empty");
    M4D_CHECK_ARG(act >= 0 && act <= 3, "unary: bad act %d", act);
    dim3 block(256), grid(grid_for(n));
    hipStream_t st = (hipStream_t)stream;
    if (n % 8 == 0 && n >= (1 << 16) && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        const dim3 g8(grid_for(n / 8));
        if (in_dt == M4D_F32 && out_dt == M4D_F32) hipLaunchKernelGGL((unary_vec8_kernel<float, float>), g8, block, 0, st, (const float*)x, (float*)out, n / 8, act);
        else if (in_dt == M4D_F32 && out_dt == M4D_BF16) hipLaunchKernelGGL((unary_vec8_kernel<float, bf16_t>), g8, block, 0, st, (const float*)x, (bf16_t*)out, n / 8, act);
        else if (in_dt == M4D_BF16 && out_dt == M4D_BF16) hipLaunchKernelGGL((unary_vec8_kernel<bf16_t, bf16_t>), g8, block, 0, st, (const bf16_t*)x, (bf16_t*)out, n / 8, act);
        else if (in_dt == M4D_BF16 && out_dt == M4D_F32) hipLaunchKernelGGL((unary_vec8_kernel<bf16_t, float>), g8, block, 0, st, (const bf16_t*)x, (float*)out, n / 8, act);
        else { m4d_set_error("unary: bad dtypes"); return -1; }
        M4D_CHECK_LAUNCH("unary");
        return 0;
    }
    if (in_dt == M4D_F32 && out_dt == M4D_F32) hipLaunchKernelGGL((unary_kernel<float, float>), grid, block, 0, st, (const float*)x, (float*)out, n, act);
    else if (in_dt == M4D_F32 && out_dt == M4D_BF16) hipLaunchKernelGGL((unary_kernel<float, bf16_t>), grid, block, 0, st, (const float*)x, (bf16_t*)out, n, act);
    else if (in_dt == M4D_BF16 && out_dt == M4D_BF16) hipLaunchKernelGGL((unary_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)out, n, act);
    else if (in_dt == M4D_BF16 && out_dt == M4D_F32) hipLaunchKernelGGL((unary_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)x, (float*)out, n, act);
    else { m4d_set_error("unary: bad dtypes"); return -1; }
    M4D_CHECK_LAUNCH("unary");
    return 0;
}

extern "C" int m4d_add_bcast(const float* a, const float* bias, float* out, int64_t B, int64_t n, m4d_stream stream) {
    M4D_CHECK_ARG(a && bias && out && B > 0 && n > 0, "add_bcast: null/empty");
    dim3 block(256), grid(grid_for(B * n));
    hipLaunchKernelGGL(add_bcast_kernel, grid, block, 0, (hipStream_t)stream, a, bias, out, B * n, n);
    M4D_CHECK_LAUNCH("add_bcast");
    return 0;
}

extern "C" int m4d_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, m4d_stream stream) {
    M4D_CHECK_ARG(x && y && out && n > 0, "axpby: null/empty");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, out, n, a, b);
    M4D_CHECK_LAUNCH("axpby");
    return 0;
}

extern "C" int m4d_lincomb(const float* x0, float a0, const float* x1, float a1, const float* x2, float a2, const float* x3,
                           float a3, float* out, int64_t n, m4d_stream stream) {
    M4D_CHECK_ARG(x0 && out && n > 0, "lincomb: null/empty");
    LinArgs p;
    p.x[0] = x0; p.x[1] = x1; p.x[2] = x2; p.x[3] = x3;
    p.a[0] = a0; p.a[1] = a1; p.a[2] = a2; p.a[3] = a3;
    p.out = out; p.n = n;
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("lincomb");
    return 0;
}

extern "C" int m4d_attn_merge(m4d_dtype dt, void* o_a, int64_t oa_bs, int64_t oa_ls, float* lse_a, const void* o_b, int64_t ob_bs,
                              int64_t ob_ls, const float* lse_b, int B, int64_t L, int heads, int head_dim, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "attn_merge: bad dtype");
    M4D_CHECK_ARG(o_a && o_b && lse_a && lse_b && B > 0 && L > 0 && heads > 0, "attn_merge: bad arguments");
    const int epv = dt == M4D_BF16 ? 8 : 4;
    M4D_CHECK_ARG(head_dim == 32 || head_dim == 64 || head_dim == 128, "attn_merge: head_dim must be 32, 64 or 128");   // one head's vectors sit in one wave
    M4D_CHECK_ARG(head_dim % epv == 0 && oa_ls % epv == 0 && ob_ls % epv == 0 && oa_bs % epv == 0 && ob_bs % epv == 0 &&
                  ((uintptr_t)o_a % 16) == 0 && ((uintptr_t)o_b % 16) == 0, "attn_merge: rows must be 16-byte aligned");
    MergeArgs p{o_a, o_b, lse_a, lse_b, oa_bs, oa_ls, ob_bs, ob_ls, L, B, heads, head_dim};
    dim3 grid(grid_for((int64_t)B * L * heads * head_dim / epv)), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL((attn_merge_kernel<bf16_t, 8>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_merge_kernel<float, 4>), grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("attn_merge");
    return 0;
}

extern "C" int m4d_rel_l1(const float* prev, const float* cur, float* out2, int64_t n, m4d_stream stream) {
    M4D_CHECK_ARG(prev && cur && out2 && n > 0, "rel_l1: null/empty");
    hipLaunchKernelGGL(rel_l1_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, prev, cur, out2, n);
    M4D_CHECK_LAUNCH("rel_l1");
    return 0;
}

extern "C" int m4d_bilinear_cl(m4d_dtype dt, const void* x, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C,
                               m4d_stream stream) {
    M4D_CHECK_ARG(x && out && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_cl: null/empty");
    dim3 grid(grid_for((int64_t)B * Ho * Wo * C)), block(256);
    if (dt == M4D_BF16) hipLaunchKernelGGL(bilinear_cl_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, B, Hi, Wi, Ho, Wo, C);
    else if (dt == M4D_F32) hipLaunchKernelGGL(bilinear_cl_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)out, B, Hi, Wi, Ho, Wo, C);
    else { m4d_set_error("bilinear_cl: bad dtype"); return -1; }
    M4D_CHECK_LAUNCH("bilinear_cl");
    return 0;
}
