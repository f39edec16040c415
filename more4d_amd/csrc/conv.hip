// m4d_conv_cl: causal 3-D / 2-D convolution on channels-last activations as an IMPLICIT GEMM on the gfx950 MFMA
// (replaces nn.Conv3d / nn.Conv2d of wan_vae.py :21-40, :83-100, :238-239 and trajectory_module.py :73-100).
//
//   out[(to,ho,wo), co] = bias[co] + sum_{dt,dh,dw,c} x[ti, hi, wi, c] * w[co, (dt,dh,dw,c)]  (+ resid[(to,ho,wo), co])
//   ti = to*st + dt - pad_t,  hi = ho*sh + dh - pad_h,  wi = wo*sw + dw - pad_w;   out-of-range taps read zero.
// The GEMM view is M = To*Ho*Wo output pixels, N = Cout, K = kt*kh*kw*Cin with K ordered (dt, dh, dw, c), so a
// 16-byte chunk of K is 8 (bf16) / 4 (fp32) CONSECUTIVE CHANNELS of one tap: the A-operand loader gathers those
// chunks straight from the [T,H,W,C] tensor — no im2col buffer, no torch.cat / F.pad copies (the reference pads and
// concatenates before every conv, wan_vae.py :33-38).  Causality: the caller keeps each conv's last two input frames
// in front of the chunk in the same buffer (a 2-frame tail), so the time axis is a plain "valid" convolution.
// Fused into the loader: nearest-exact 2x spatial up-sampling (`ups`, wan_vae.py :61-67, :81-87) and the temporal
// de-interleave of upsample3d's time_conv output (`tsplit`: logical frame f = physical frame f>>1, channel half f&1,
// wan_vae.py :138-141).  Fused into the epilogue: bias and the residual / shortcut add (:224).
// Tiling, LDS image, fragment convention and epilogue lane layout are those of gemm_bt_kernel (gemm.hip): 128x128
// tile, 4 waves, K-tile of 128 bytes per row, register-staged copies one tile ahead, XOR-swizzled LDS rows.
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "more4d_hip.h"

namespace {

struct ConvArgs {
    const void* x; const void* w; const void* bias; const void* resid; void* out;
    int64_t xs;            // elements between consecutive input pixels (>= Cin; 2*Cin with tsplit)
    int64_t ldo, ldr;      // row strides (elements) of out / resid
    int Tin, Hin, Win, Cin, Cout;
    int kt, kh, kw, st, sh, sw, pad_t, pad_h, pad_w;
    int To, Ho, Wo;
    int ups, tsplit;
    int64_t M, K;
    int tiles_m, tiles_n;
    int abl;               // timing ablations (tool builds only)
    int64_t xplane;        // != 0: input is planar-16, [Cin/16][rows][16] with xplane elements between planes (conv_halo_kernel only)
    const float* post_gamma; void* post_out; int64_t post_plane; int post_silu;      // fused RMS_norm(+SiLU) of the next layer (conv_halo.h)
    unsigned long long* dbg;   // tool builds only (M4D_CONV_ABL & 64): per-workgroup timestamps, 8 words each (tools/conv_timeline.py)
    float* gn_partial;     // per-patch GroupNorm(32 x 4 channels) statistics of the result, [To][patches][32][2] (Cout = 128)
    const void* wt;        // the weights again in the tiled order of m4d_conv_pack_weights, or nullptr (LDS-halo kernels only)
};

constexpr int ROWB = 128, BM = 128, BN = 128;
constexpr int STAGE_BYTES = (BM + BN) * ROWB;

M4D_DEV int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_cl_kernel(ConvArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    constexpr int ES = sizeof(T);
    constexpr int KT = ROWB / ES;
    constexpr int KSTEPS = KT / 16;
    constexpr int EPC = 16 / ES;
    typedef typename Frag8<T>::type frag_t;

    // consecutive workgroups walk tiles_n fastest inside 8-row bands (A rows are re-read by every n tile)
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int srow = t >> 3, schunk = t & 7;

    // ---- per-thread output pixels of the 4 staged A rows ----
    int ti0[4], hi0[4], wi0[4];
    unsigned amask = 0, wmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + srow + 32 * i;
        if (m < p.M) {
            amask |= 1u << i;
            const int wo = (int)(m % p.Wo);
            const int64_t r = m / p.Wo;
            const int ho = (int)(r % p.Ho), to = (int)(r / p.Ho);
            ti0[i] = to * p.st - p.pad_t;
            hi0[i] = ho * p.sh - p.pad_h;
            wi0[i] = wo * p.sw - p.pad_w;
        } else { ti0[i] = hi0[i] = wi0[i] = 0; }
        if (n0 + srow + 32 * i < p.Cout) wmask |= 1u << i;
    }
    const char* pw = (const char*)p.w + ((n0 + srow) * p.K) * ES + schunk * 16;
    const int64_t sw_ = 32 * p.K * ES;
    const int khw = p.kh * p.kw;
    const int Hl = p.Hin << p.ups, Wl = p.Win << p.ups;     // logical (up-sampled) input extent
    const int Tl = p.Tin << p.tsplit;

    uint4 ra[4], rw[4];
    // This thread always stages the same 16-byte chunk column: its K index advances by KT per tile, so (tap, channel)
    // and the tap's (dt, dh, dw) are carried incrementally — the per-tile integer divisions of the first version cost more
    // VALU time than the tile's MFMAs (gload is called with ktile = 0, 1, 2, ... in order).
    int g_c = schunk * EPC, g_dt = 0, g_dh = 0, g_dw = 0;
    int64_t g_k = schunk * EPC;
    auto tap_norm = [&]() {
        while (g_c >= p.Cin) {
            g_c -= p.Cin;
            if (++g_dw == p.kw) { g_dw = 0; if (++g_dh == p.kh) { g_dh = 0; ++g_dt; } }
        }
    };
    tap_norm();
    auto gload = [&](int ktile) {
        const bool kin = g_k < p.K;
        const int dt = g_dt, dh = g_dh, dw = g_dw, c = g_c;
        g_k += KT; g_c += KT;
        tap_norm();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kin && ((amask >> i) & 1)) {
                int ti = ti0[i] + dt, hh = hi0[i] + dh, ww = wi0[i] + dw;
                if (ti >= 0 && ti < Tl && hh >= 0 && hh < Hl && ww >= 0 && ww < Wl) {
                    int cc = c;
                    if (p.tsplit) { cc += (ti & 1) * p.Cin; ti >>= 1; }
                    if (p.ups) { hh >>= 1; ww >>= 1; }
                    const int64_t pix = ((int64_t)ti * p.Hin + hh) * p.Win + ww;
                    v = *reinterpret_cast<const uint4*>((const T*)p.x + pix * p.xs + cc);
                }
            }
            ra[i] = v;
            rw[i] = (kin && ((wmask >> i) & 1)) ? *reinterpret_cast<const uint4*>(pw + i * sw_ + (int64_t)ktile * ROWB)
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

    f32x16 acc[2][2];
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

    // ---- epilogue: lane holds 4 consecutive output channels of one pixel ----
    const T* bias = (const T*)p.bias;
    const T* resid = (const T*)p.resid;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int64_t nb = n0 + wn * 64 + ni * 32 + rq * 8 + hi * 4;
                if (nb >= p.Cout) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][rq * 4 + e];
                if (bias) v += load4(bias + nb);
                if (resid) {
                    const f32x4 r = load4(resid + m * p.ldr + nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]) + r[e];   // conv output is T, then x + h (:224)
                }
                store4((T*)p.out + m * p.ldo + nb, v);
            }
    }
}

// ============================================================================ production path: bf16, stride 1, plain taps
// 256(pixels) x 128(channels) tile, 8 waves (4 x 2, wave tile 64 x 64), K-tile 64, two 48 KiB LDS stages filled by
// global->LDS DMA with PER-LANE source addresses: a lane's 16-byte chunk is 8 channels of one tap of one pixel, or — for
// taps that fall outside the input, rows past M and K past the end — a 16-byte read of a zero page, so padding costs no
// branches in the consumer.  Per K-tile a lane only advances its (tap, channel) position and looks the tap up in a
// per-row 27-bit validity mask computed once (the first version recomputed and bounds-checked every coordinate per tile:
// more VALU time than the tile's MFMAs).  Covers the ResidualBlock / attention / shortcut convolutions (about 85 % of the
// VAE FLOPs); strided, up-sampling and time-split convolutions keep conv_cl_kernel.
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
__device__ __attribute__((aligned(256))) char m4d_zero_page[256];
extern __shared__ __attribute__((aligned(16))) char conv_dyn_smem[];

__global__ __launch_bounds__(512, 2) void conv_cl256_kernel(ConvArgs p) {
    typedef bf16_t T;
    constexpr int BM2 = 256, BN2 = 128, STAGE2 = (BM2 + BN2) * ROWB;
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- DMA lane roles: one instruction = 8 rows x 8 chunks; A instruction i covers rows i*64 + wave*8 + lrow ----
    const int lrow = lane >> 3, pc = lane & 7;
    const int rsub = wave * 8 + lrow;                       // row inside a 64-row group
    const int lc = pc ^ ((rsub >> 1) & 7);                  // logical chunk landing in physical chunk pc (same for all groups)
    int rowoff[4];                                          // byte offset of (ti0, hi0, wi0) from x (may be "negative")
    unsigned vmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + i * 64 + rsub;
        vmask[i] = 0;
        rowoff[i] = 0;
        if (m < p.M) {
            const int wo = (int)(m % p.Wo);
            const int64_t r = m / p.Wo;
            const int ho = (int)(r % p.Ho), to = (int)(r / p.Ho);
            const int ti0 = to - p.pad_t, hi0 = ho - p.pad_h, wi0 = wo - p.pad_w;
            rowoff[i] = (int)((((int64_t)ti0 * p.Hin + hi0) * p.Win + wi0) * p.xs * 2);
            // separable validity: bits 0-7 dt, 8-15 dh, 16-23 dw (kernel extents <= 8)
            for (int d = 0; d < p.kt; ++d) if (ti0 + d >= 0 && ti0 + d < p.Tin) vmask[i] |= 1u << d;
            for (int d = 0; d < p.kh; ++d) if (hi0 + d >= 0 && hi0 + d < p.Hin) vmask[i] |= 1u << (8 + d);
            for (int d = 0; d < p.kw; ++d) if (wi0 + d >= 0 && wi0 + d < p.Win) vmask[i] |= 1u << (16 + d);
        }
    }
    const char* wrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t n = min(n0 + i * 64 + rsub, (int64_t)p.Cout - 1);
        wrow[i] = (const char*)p.w + n * p.K * 2;
    }
    // K position of this lane's chunk: channel c inside tap (dt, dh, dw); advances by 64 per K-tile
    int g_c = lc * 8, g_dt = 0, g_dh = 0, g_dw = 0;
    int64_t g_k = lc * 8;
    auto tap_norm = [&]() {
        while (g_c >= p.Cin) {
            g_c -= p.Cin;
            if (++g_dw == p.kw) { g_dw = 0; if (++g_dh == p.kh) { g_dh = 0; ++g_dt; } }
        }
    };
    tap_norm();
    const char* zero = m4d_zero_page;
    auto issue = [&](int stage) {
        char* sA = conv_dyn_smem + stage * STAGE2;
        char* sW = sA + BM2 * ROWB;
        const bool kin = g_k < p.K;
        const int tapoff = (int)((((int64_t)g_dt * p.Hin + g_dh) * p.Win + g_dw) * p.xs * 2) + g_c * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = kin && (((vmask[i] >> g_dt) & (vmask[i] >> (8 + g_dh)) & (vmask[i] >> (16 + g_dw))) & 1u);
            const char* src = ok ? (const char*)p.x + (int64_t)(rowoff[i] + tapoff) : zero;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)(sA + (i * 8 + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* src = kin ? wrow[i] + g_k * 2 : zero;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)(sW + (i * 8 + wave) * 1024), 16, 0, 0);
        }
        g_k += 64; g_c += 64;
        tap_norm();
    };

    f32x16 acc[2][2];   // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-lane LDS byte offsets of the first fragment row of each K = 16 step (the next 32 rows are +4096 B)
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS char*)conv_dyn_smem;
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        aoff[kk] = lds_off(wm * 64 + li, kk * 2 + hi);
        woff[kk] = BM2 * ROWB + lds_off(wn * 64 + li, kk * 2 + hi);
    }
    const int nk = (int)((p.K + 63) / 64);
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue(stage ^ 1);
        // fragment double buffer from inline asm with counted lgkmcnt (hipcc drains to 0 in front of every MFMA group otherwise)
        const unsigned sb = lds_base + stage * STAGE2;
        bf16x8 fa[2][2], fw[2][2];
#define CV_LDFRAG(buf, kk)                                                                                           \
        do {                                                                                                         \
            const unsigned aa_ = sb + aoff[kk], aw_ = sb + woff[kk];                                                 \
            asm volatile("ds_read_b128 %0, %1" : "=v"(fw[buf][0]) : "v"(aw_));                                       \
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fw[buf][1]) : "v"(aw_));                          \
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[buf][0]) : "v"(aa_));                                       \
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[buf][1]) : "v"(aa_));                          \
        } while (0)
        CV_LDFRAG(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk == 0) { CV_LDFRAG(1, 1); }
            else if (kk == 1) { CV_LDFRAG(0, 2); }
            else if (kk == 2) { CV_LDFRAG(1, 3); }
            if (kk < 3) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) mma32(fw[kk & 1][ni], fa[kk & 1][mi], acc[ni][mi]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef CV_LDFRAG
    }

    const T* bias = (const T*)p.bias;
    const T* resid = (const T*)p.resid;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int64_t nb = n0 + wn * 64 + ni * 32 + rq * 8 + hi * 4;
                if (nb >= p.Cout) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][rq * 4 + e];
                if (bias) v += load4(bias + nb);
                if (resid) {
                    const f32x4 r = load4(resid + m * p.ldr + nb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]) + r[e];
                }
                store4((T*)p.out + m * p.ldo + nb, v);
            }
    }
}

#include "conv_halo.h"
#include "conv_halo64.h"

template <int KT, int KH, int TH, int TW, int NT, int MT, int SD = 1>
int launch_halo(ConvArgs& p, hipStream_t st) {
    constexpr int LDS = halo::Cfg<KT, KH, TH, TW, NT, MT, SD>::LDS_BYTES;
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)conv_halo_kernel<KT, KH, TH, TW, NT, MT, SD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            m4d_set_error("conv_cl: cannot enable %d bytes of LDS", LDS);
            return -3;
        }
        configured.mark();
    }
    p.tiles_m = p.To * ((p.Ho + TH - 1) / TH) * ((p.Wo + TW - 1) / TW);
    p.tiles_n = (p.Cout + NT * 32 - 1) / (NT * 32);
    m4d_count_launch(MT == 3 ? (TH == 12 ? M4D_KC_CONV_HALO_MT3_12X32 : M4D_KC_CONV_HALO_MT3_24X16) : M4D_KC_CONV_HALO);
    if (p.post_out) m4d_count_launch(p.resid ? M4D_KC_CONV_FUSED_NORM_RESID : M4D_KC_CONV_FUSED_NORM);
    if (p.gn_partial) m4d_count_launch(M4D_KC_CONV_GNSTATS);
    hipLaunchKernelGGL((conv_halo_kernel<KT, KH, TH, TW, NT, MT, SD>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(64 * halo::Cfg<KT, KH, TH, TW, NT, MT, SD>::NWAVE), LDS, st, p);
    return 0;
}

// one wave per SIMD (conv_halo64.h): KT = 3: 10 x 32 patches, 96-channel tiles; KT = 1: 8 x 32 patches, 128-channel tiles
template <int KT, int MT, int NT>
int launch_halo64(ConvArgs& p, hipStream_t st) {
    using C64 = Halo64Cfg<KT, MT, NT>;
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)conv_halo64_kernel<KT, MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, C64::LDS_BYTES) != hipSuccess) {
            m4d_set_error("conv_cl: cannot enable %d bytes of LDS", C64::LDS_BYTES);
            return -3;
        }
        configured.mark();
    }
    p.tiles_m = p.To * ((p.Ho + C64::TH - 1) / C64::TH) * ((p.Wo + 31) / 32);
    p.tiles_n = p.Cout / (NT * 32);
    m4d_count_launch(M4D_KC_CONV_HALO64);
    if (p.post_out) m4d_count_launch(p.resid ? M4D_KC_CONV_FUSED_NORM_RESID : M4D_KC_CONV_FUSED_NORM);
    if (p.gn_partial) m4d_count_launch(M4D_KC_CONV_GNSTATS);
    hipLaunchKernelGGL((conv_halo64_kernel<KT, MT, NT>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(128), C64::LDS_BYTES, st, p);
    return 0;
}

// output channels per workgroup = NT x 32: 96 / 192 / 384 -> 1 / 2 / 4 tiles of 96, <= 32 / <= 64 (the 3-channel heads) one tile of
// 32 / 64, everything else tiles of 128
template <int KT, int KH, int TH, int TW>
int launch_halo_nt(ConvArgs& p, hipStream_t st) {
    if constexpr (KT == 1 && KH == 3 && TH == 8 && TW == 32) {
        // the 3 x 3 conv on 128-channel tiles (the adaptors' convs, trajectory_module.py:54-71) as one wave per SIMD: the same 8 x 32 patches
        // (GroupNorm-statistics blocks, fused norm) and the same accumulation order as conv_halo_kernel<1, 3, 8, 32, 4, 2> — bit-identical
        // Built, bit-identical, and NOT faster (tools/check_conv64k1.py --time, same box: 1.488 vs 1.481 ms at 128 -> 128 channels, 480 x 832, 12 frames:
        // nine taps per chunk make a 72-tap main loop, the per-workgroup prologue and epilogue dominate both kernels) — off unless M4D_CONV_HALO64K1=1
        M4D_ENV_ONCE(h64k1, "M4D_CONV_HALO64K1", 0);
        if (h64k1 && p.wt && p.Cout % 128 == 0 && !p.ups && !p.tsplit && p.Cin % 16 == 0 && (int64_t)p.To * ((p.Ho + 7) / 8) * ((p.Wo + 31) / 32) * (p.Cout / 128) > 256)
            return launch_halo64<1, 4, 4>(p, st);
    }
    if (p.gn_partial) return launch_halo<KT, KH, TH, TW, 4, 2>(p, st);       // (Cout = 128: one 128-channel tile per patch)
    if (p.post_out) {        // fused next-layer norm: one workgroup must own all channels of a pixel
        switch (p.Cout) {
            case 32: return launch_halo<KT, KH, TH, TW, 1, 2>(p, st);
            case 64: return launch_halo<KT, KH, TH, TW, 2, 2>(p, st);
            case 96: return launch_halo<KT, KH, TH, TW, 3, 2>(p, st);
            case 128: return launch_halo<KT, KH, TH, TW, 4, 2>(p, st);
            default: m4d_set_error("conv_cl_planar_norm: Cout must be 32, 64, 96 or 128 (got %d)", p.Cout); return -1;
        }
    }
    if (p.Cout <= 32) return launch_halo<KT, KH, TH, TW, 1, 2>(p, st);
    if (p.Cout <= 64) return launch_halo<KT, KH, TH, TW, 2, 2>(p, st);
    // small maps (60 x 104 x 1 frame x 384 channels = 112 workgroups of 96 channels on 512 slots): narrower channel tiles, 3x the
    // workgroups; the extra halo copies come out of L2 (the whole input is 5 MB)
    const int64_t patches = (int64_t)p.To * ((p.Ho + TH - 1) / TH) * ((p.Wo + TW - 1) / TW);
    M4D_ENV_ONCE(small, "M4D_CONV_SMALL", 1);
    if (small && p.Cout % 32 == 0 && patches * ((p.Cout + 95) / 96) <= 256) return launch_halo<KT, KH, TH, TW, 1, 2>(p, st);
    return ((p.Cout + 31) / 32) % 3 == 0 ? launch_halo<KT, KH, TH, TW, 3, 2>(p, st) : launch_halo<KT, KH, TH, TW, 4, 2>(p, st);
}

// patch shape + tile shape of the LDS-halo kernel for this problem
int launch_halo_auto(ConvArgs& p, hipStream_t st) {
    const bool wide = (p.Wo % 32 == 0) || p.Wo >= 256;        // 32-column patches; narrow maps (104, 208 columns) use 16 columns
    // 96-channel tiles on maps that divide into 12 x 32 / 24 x 16 patches: THREE pixel tiles per wave (4 waves, 384 pixels): 6 fragment
    // reads feed 9 MFMAs (8 x 32 patches: 5 per 6), a third less weight traffic from L2, 7 % less halo: +5 % on the 96- / 192-channel layers
    M4D_ENV_ONCE(mt3, "M4D_CONV_MT3", 1);
    if (mt3 && ((p.Cout + 31) / 32) % 3 == 0 && p.Cout > 64 && (!p.post_out || p.Cout == 96) && !p.gn_partial) {
        const int th = wide ? 12 : 24, tw = wide ? 32 : 16;
        const int nh = (p.Ho + th - 1) / th;
        const int64_t patches = (int64_t)p.To * nh * ((p.Wo + tw - 1) / tw);
        const bool fits = (nh * th - p.Ho) * 20 <= p.Ho;      // at most 5 % of the rows are padding
        if (fits && (p.post_out || patches * ((p.Cout + 95) / 96) > 256)) {      // (small maps: launch_halo_nt's 32-channel tiles)
            // M4D_CONV_HALO64=1 (default): the one-wave-per-SIMD kernel (conv_halo64.h) on 32-column maps whose rows divide into 10-row patches
            M4D_ENV_ONCE(h64, "M4D_CONV_HALO64", 2);
            // (2, default: planar-16 and channels-last inputs; 1: planar-16 only — what the residual blocks' fused norms write; 0: off.  On
            //  channels-last inputs, whose halo pieces are 32 bytes out of every Cin * 2, the first version (plain weights) was 5 % SLOWER than
            //  the 12 x 32 kernel; on tiled weights it is 8 % faster there too: tools/check_conv64.py --time.  The inference path has few such
            //  launches, the training path — channels-last staging — many.)
            // (M4D_CONV_HALO64_NARROW: maps that are no multiple of 32 columns wide but lose at most 1 / 13 of a 32-column patch row to
            //  padding — the 208-column maps — take it as well instead of the 24 x 16 kernel)
            M4D_ENV_ONCE(narrow64, "M4D_CONV_HALO64_NARROW", 1);
            const bool wide64 = wide || (narrow64 && (((p.Wo + 31) / 32) * 32 - p.Wo) * 12 <= p.Wo);
            if (h64 && p.wt && (p.xplane || h64 == 2) && p.kt == 3 && wide64 && !p.ups && !p.tsplit && p.Cout % 96 == 0 &&
                (((p.Ho + 9) / 10) * 10 - p.Ho) * 20 <= p.Ho)
                return launch_halo64<3, 5, 3>(p, st);
            if (p.kt == 3) return wide ? launch_halo<3, 3, 12, 32, 3, 3>(p, st) : launch_halo<3, 3, 24, 16, 3, 3>(p, st);
            return wide ? launch_halo<1, 3, 12, 32, 3, 3>(p, st) : launch_halo<1, 3, 24, 16, 3, 3>(p, st);
        }
    }
    if (p.kt == 3) return wide ? launch_halo_nt<3, 3, 8, 32>(p, st) : launch_halo_nt<3, 3, 16, 16>(p, st);
    return wide ? launch_halo_nt<1, 3, 8, 32>(p, st) : launch_halo_nt<1, 3, 16, 16>(p, st);
}

#ifdef M4D_ABLATIONS
// tool builds: timing ablations (M4D_CONV_ABL) and the timeline buffer (M4D_CONV_DBG_PTR = device address, tools/conv_timeline.py)
inline void conv_tool_switches(ConvArgs& p) {
    M4D_ENV_ONCE(conv_abl, "M4D_CONV_ABL", 0);
    p.abl = conv_abl;
    static unsigned long long* dbgp = nullptr;
    static bool rd = false;
    if (!rd) { rd = true; const char* v = getenv("M4D_CONV_DBG_PTR"); if (v) dbgp = (unsigned long long*)strtoull(v, nullptr, 0); }
    p.dbg = dbgp;
}
#endif

}  // namespace

static int conv_cl_impl(m4d_dtype dt, const void* x, int64_t x_pixel_stride, const void* w, const void* bias,
                        const void* resid, int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win,
                        int Cin, int Cout, int kt, int kh, int kw, int st, int sh, int sw, int pad_t, int pad_h,
                        int pad_w, int To, int Ho, int Wo, int ups, int tsplit, const void* wt, m4d_stream stream) {
    const int es = dt == M4D_BF16 ? 2 : 4;
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "conv_cl: bad dtype %d", (int)dt);
    M4D_CHECK_ARG(x && w && out, "conv_cl: null pointer");
    M4D_CHECK_ARG(Tin > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0 && To > 0 && Ho > 0 && Wo > 0, "conv_cl: empty problem");
    M4D_CHECK_ARG(kt >= 1 && kh >= 1 && kw >= 1 && st >= 1 && sh >= 1 && sw >= 1, "conv_cl: bad kernel/stride");
    M4D_CHECK_ARG((Cin * es) % 16 == 0, "conv_cl: Cin*sizeof(T) must be a multiple of 16 (pad the channels): Cin=%d", Cin);
    M4D_CHECK_ARG(Cout % 4 == 0, "conv_cl: Cout must be a multiple of 4 (pad the filters): Cout=%d", Cout);
    M4D_CHECK_ARG((x_pixel_stride * es) % 16 == 0 && x_pixel_stride >= (int64_t)Cin * (tsplit ? 2 : 1), "conv_cl: bad input pixel stride");
    M4D_CHECK_ARG(out_ld % 4 == 0 && out_ld >= Cout && (!resid || (resid_ld % 4 == 0 && resid_ld >= Cout)), "conv_cl: bad out/resid stride");
    M4D_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0, "conv_cl: pointers must be 16-byte aligned");
    M4D_CHECK_ARG((ups == 0 || ups == 1) && (tsplit == 0 || tsplit == 1), "conv_cl: ups/tsplit are flags");
    M4D_CHECK_ARG(!wt || (dt == M4D_BF16 && Cin % 16 == 0 && ((uintptr_t)wt % 16) == 0), "conv_cl: tiled weights are bf16, Cin %% 16, 16-byte aligned");
    ConvArgs p;
    p.wt = wt;
    p.x = x; p.w = w; p.bias = bias; p.resid = resid; p.out = out;
    p.xs = x_pixel_stride; p.ldo = out_ld; p.ldr = resid_ld;
    p.Tin = Tin; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout;
    p.kt = kt; p.kh = kh; p.kw = kw; p.st = st; p.sh = sh; p.sw = sw; p.pad_t = pad_t; p.pad_h = pad_h; p.pad_w = pad_w;
    p.To = To; p.Ho = Ho; p.Wo = Wo; p.ups = ups; p.tsplit = tsplit;
    p.M = (int64_t)To * Ho * Wo;
    p.K = (int64_t)kt * kh * kw * Cin;
    p.abl = 0; p.xplane = 0; p.post_gamma = nullptr; p.post_out = nullptr; p.post_plane = 0; p.post_silu = 0; p.gn_partial = nullptr; p.dbg = nullptr;
#ifdef M4D_ABLATIONS
    conv_tool_switches(p);
    { M4D_ENV_ONCE(conv_planar, "M4D_CONV_PLANAR", 0); if (conv_planar) p.xplane = (int64_t)Tin * Hin * Win * 16; }     // timing experiment: same bytes read as planar-16
#endif
    // production kernel: bf16, unit stride, no fused up-sampling / time split, <= 32 taps, input extent addressable in 31 bits
    const int64_t xbytes = (int64_t)Tin * Hin * Win * x_pixel_stride * 2;
    M4D_ENV_ONCE(conv_variant, "M4D_CONV_VARIANT", 3);     // 3 LDS-halo kernel where it applies (default) | 2 DMA-gather implicit GEMM | 1 register-staged
    const bool v2_base = conv_variant >= 2 && dt == M4D_BF16 && st == 1 && sh == 1 && sw == 1 && kt <= 8 && kh <= 8 && kw <= 8 && p.M >= 1024;
    const bool v2_shape = v2_base && !ups && !tsplit;
    // the halo kernel also reads through the up-sampled / time-split views (its DMA addresses are per halo pixel anyway)
    const bool halo_shape = conv_variant == 3 && v2_base && kw == 3 && kh == 3 && (kt == 3 || kt == 1) && pad_t == 0 && pad_h == 1 &&
                            pad_w == 1 && Ho == (Hin << ups) && Wo == (Win << ups) && To == (Tin << tsplit) - kt + 1 && Cin % 16 == 0 &&
                            (!tsplit || kt == 1);
    // byte extent the kernel this problem lands on can address: the LDS-halo kernel takes unsigned 32-bit offsets into a raw buffer
    // descriptor (just under 2 GiB), the DMA-gather implicit GEMM (1x1 shortcut convs, M4D_CONV_VARIANT=2) signed 31-bit element
    // arithmetic on 2-byte elements (1 GiB)
    const int64_t xlimit = halo_shape ? (1ll << 31) - (1ll << 20) : (1ll << 30);
    const int64_t frame_bytes = (int64_t)Hin * Win * x_pixel_stride * 2;
    // (Tin > 1 and one frame below the limit: a single oversized frame must NOT re-enter this branch — it falls through to the
    // register-staged kernel, which addresses 64 bits — ADVICE r3)
    if (v2_shape && xbytes >= xlimit && kt == 1 && pad_t == 0 && To == Tin && Tin > 1 && frame_bytes < xlimit) {
        // 2-D convolution over many frames (the adaptors: 49 x 480 x 832 x 128): frames are independent, so launch groups of
        // frames whose input fits the target kernel's offsets (otherwise the group would fall through to the register-staged kernel)
        const int per = (int)std::max<int64_t>(1, (xlimit - 1) / frame_bytes);
        for (int f0 = 0; f0 < Tin; f0 += per) {
            const int nf = std::min(per, Tin - f0);
            const int rc = conv_cl_impl(dt, (const char*)x + (int64_t)f0 * frame_bytes, x_pixel_stride, w, bias,
                                        resid ? (const char*)resid + (int64_t)f0 * Ho * Wo * resid_ld * 2 : nullptr, resid_ld,
                                        (char*)out + (int64_t)f0 * Ho * Wo * out_ld * 2, out_ld, nf, Hin, Win, Cin, Cout, kt, kh, kw,
                                        st, sh, sw, pad_t, pad_h, pad_w, nf, Ho, Wo, ups, tsplit, wt, stream);
            if (rc) return rc;
        }
        return 0;
    }
    // the Resample down-sampling conv (3x3, stride 2 in H and W, zero padding on the bottom / right only = reads past the map are zero)
    const bool s2_shape = conv_variant == 3 && dt == M4D_BF16 && kt == 1 && kh == 3 && kw == 3 && st == 1 && sh == 2 && sw == 2 && pad_t == 0 &&
                          pad_h == 0 && pad_w == 0 && !ups && !tsplit && To == Tin && Ho == (Hin + 1) / 2 && Wo == (Win + 1) / 2 &&
                          Cin % 16 == 0 && Cout % 32 == 0 && p.M >= 1024;
    if (s2_shape && xbytes < (1ll << 31) - (1ll << 20)) {
        const bool wide = (Wo % 32 == 0) || Wo >= 256;
        const bool n3 = (Cout / 32) % 3 == 0;
        int rc;
        if (wide) rc = n3 ? launch_halo<1, 3, 8, 32, 3, 2, 2>(p, (hipStream_t)stream) : launch_halo<1, 3, 8, 32, 4, 2, 2>(p, (hipStream_t)stream);
        else rc = n3 ? launch_halo<1, 3, 16, 16, 3, 2, 2>(p, (hipStream_t)stream) : launch_halo<1, 3, 16, 16, 4, 2, 2>(p, (hipStream_t)stream);
        if (rc) return rc;
        M4D_CHECK_LAUNCH("conv_cl");
        return 0;
    }
    if (halo_shape && xbytes < (1ll << 31) - (1ll << 20)) {        // (unsigned 32-bit byte offsets into a raw buffer descriptor)
        const bool wide = (Wo % 32 == 0) || Wo >= 256;        // 8 x 32 patches; narrow maps (104, 208 columns) use 16 x 16
        int rc;
        rc = launch_halo_auto(p, (hipStream_t)stream);
        if (rc) return rc;
        M4D_CHECK_LAUNCH("conv_cl");
        return 0;
    }
    if (v2_shape && xbytes < (1ll << 30)) {
        static PerDeviceOnce configured;
        if (configured.pending()) {
            if (hipFuncSetAttribute((const void*)conv_cl256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 128) * ROWB) != hipSuccess) {
                m4d_set_error("conv_cl: cannot enable 96 KiB LDS");
                return -3;
            }
            configured.mark();
        }
        p.tiles_m = (int)((p.M + 255) / 256); p.tiles_n = (Cout + 127) / 128;
        const int64_t nwg2 = (int64_t)p.tiles_m * p.tiles_n;
        M4D_CHECK_ARG(nwg2 < (1ll << 31), "conv_cl: too many tiles");
        m4d_count_launch(M4D_KC_CONV_GENERIC);
        hipLaunchKernelGGL(conv_cl256_kernel, dim3((unsigned)nwg2), dim3(512), 2 * (256 + 128) * ROWB, (hipStream_t)stream, p);
        M4D_CHECK_LAUNCH("conv_cl");
        return 0;
    }
    p.tiles_m = (int)((p.M + BM - 1) / BM); p.tiles_n = (Cout + BN - 1) / BN;
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    M4D_CHECK_ARG(nwg < (1ll << 31), "conv_cl: too many tiles");
    dim3 grid((unsigned)nwg), block(256);
    m4d_count_launch(M4D_KC_CONV_GENERIC);
    if (dt == M4D_BF16) hipLaunchKernelGGL(conv_cl_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv_cl_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    M4D_CHECK_LAUNCH("conv_cl");
    return 0;
}

/* m4d_conv_cl for the 3x3(x3), stride-1, pad-(0,1,1) case with the input in planar-16 layout (what m4d_rmsnorm_silu_cl_planar writes
 * into a causal conv's staging buffer): a halo pixel's 16-channel piece is 32 contiguous bytes next to its row neighbours' instead
 * of 32 bytes out of a Cin*2-byte pixel, so the halo DMA fetches whole lines it uses (fabric traffic / 4 at Cin = 96). */
static int conv_cl_planar_impl(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* bias, const void* resid,
                               int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout, int kt, int To,
                               const float* norm_gamma, void* norm_out, int64_t norm_plane, int norm_silu, float* gn_partial, const void* wt,
                               m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16, "conv_cl_planar: bf16 only");
    M4D_CHECK_ARG(!wt || ((uintptr_t)wt % 16) == 0, "conv_cl_planar: tiled weights must be 16-byte aligned");
    M4D_CHECK_ARG(!gn_partial || (Cout == 128 && out && !norm_out && out_ld % 8 == 0 && (!resid || resid_ld % 8 == 0)),
                  "conv_cl_planar_gnstats: Cout = 128 (GroupNorm of 32 groups x 4 channels), row strides %% 8");
    M4D_CHECK_ARG(x && w && (out || norm_out) && Tin > 0 && Hin > 0 && Win > 0 && Cout > 0 && To > 0, "conv_cl_planar: null/empty");
    if (norm_out) {
        M4D_CHECK_ARG(norm_gamma && norm_plane >= (int64_t)To * Hin * Win * 16 && norm_plane % 8 == 0 && ((uintptr_t)norm_out % 16) == 0,
                      "conv_cl_planar_norm: gamma / plane stride / alignment of the normalised output");
        M4D_CHECK_ARG(Cout % 32 == 0 && Cout <= 128 && (!out || out_ld % 8 == 0) && (!resid || resid_ld % 8 == 0),
                      "conv_cl_planar_norm: Cout in {32, 64, 96, 128}, row strides %% 8");
    }
    M4D_CHECK_ARG((kt == 3 || kt == 1) && To == Tin - kt + 1, "conv_cl_planar: kt=%d To=%d Tin=%d", kt, To, Tin);
    M4D_CHECK_ARG(Cin % 16 == 0 && Cout % 4 == 0, "conv_cl_planar: Cin %% 16, Cout %% 4");
    M4D_CHECK_ARG(x_plane_stride >= (int64_t)Tin * Hin * Win * 16 && x_plane_stride % 8 == 0, "conv_cl_planar: plane stride too small");
    M4D_CHECK_ARG((int64_t)(Cin / 16) * x_plane_stride * 2 < (1ll << 31), "conv_cl_planar: input extent must stay below 2 GiB");
    M4D_CHECK_ARG((!out || (out_ld % 4 == 0 && out_ld >= Cout)) && (!resid || (resid_ld % 4 == 0 && resid_ld >= Cout)), "conv_cl_planar: bad out/resid stride");
    M4D_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0, "conv_cl_planar: pointers must be 16-byte aligned");
    ConvArgs p;
    p.wt = wt;
    p.x = x; p.w = w; p.bias = bias; p.resid = resid; p.out = out;
    p.xs = Cin; p.ldo = out_ld; p.ldr = resid_ld;
    p.Tin = Tin; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout;
    p.kt = kt; p.kh = 3; p.kw = 3; p.st = p.sh = p.sw = 1; p.pad_t = 0; p.pad_h = p.pad_w = 1;
    p.To = To; p.Ho = Hin; p.Wo = Win; p.ups = 0; p.tsplit = 0;
    p.M = (int64_t)To * Hin * Win;
    p.K = (int64_t)kt * 9 * Cin;
    p.abl = 0; p.xplane = x_plane_stride; p.dbg = nullptr;
#ifdef M4D_ABLATIONS
    conv_tool_switches(p);
#endif
    p.post_gamma = norm_gamma; p.post_out = norm_out; p.post_plane = norm_plane; p.post_silu = norm_silu; p.gn_partial = gn_partial;
    const bool wide = (Win % 32 == 0) || Win >= 256;
    int rc;
    rc = launch_halo_auto(p, (hipStream_t)stream);
    if (rc) return rc;
    M4D_CHECK_LAUNCH("conv_cl_planar");
    return 0;
}

extern "C" int m4d_conv_cl(m4d_dtype dt, const void* x, int64_t x_pixel_stride, const void* w, const void* bias,
                           const void* resid, int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win,
                           int Cin, int Cout, int kt, int kh, int kw, int st, int sh, int sw, int pad_t, int pad_h,
                           int pad_w, int To, int Ho, int Wo, int ups, int tsplit, m4d_stream stream) {
    return conv_cl_impl(dt, x, x_pixel_stride, w, bias, resid, resid_ld, out, out_ld, Tin, Hin, Win, Cin, Cout, kt, kh, kw, st, sh, sw, pad_t, pad_h,
                        pad_w, To, Ho, Wo, ups, tsplit, nullptr, stream);
}

extern "C" int m4d_conv_cl_tw(m4d_dtype dt, const void* x, int64_t x_pixel_stride, const void* w, const void* w_tiled, const void* bias,
                              const void* resid, int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win,
                              int Cin, int Cout, int kt, int kh, int kw, int st, int sh, int sw, int pad_t, int pad_h,
                              int pad_w, int To, int Ho, int Wo, int ups, int tsplit, m4d_stream stream) {
    return conv_cl_impl(dt, x, x_pixel_stride, w, bias, resid, resid_ld, out, out_ld, Tin, Hin, Win, Cin, Cout, kt, kh, kw, st, sh, sw, pad_t, pad_h,
                        pad_w, To, Ho, Wo, ups, tsplit, w_tiled, stream);
}

extern "C" int m4d_conv_cl_planar(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* bias, const void* resid,
                                  int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout, int kt, int To,
                                  m4d_stream stream) {
    M4D_CHECK_ARG(out, "conv_cl_planar: null output");
    return conv_cl_planar_impl(dt, x, x_plane_stride, w, bias, resid, resid_ld, out, out_ld, Tin, Hin, Win, Cin, Cout, kt, To, nullptr, nullptr, 0, 0,
                               nullptr, nullptr, stream);
}

extern "C" int m4d_conv_cl_planar_gnstats_blocks(int Hin, int Win) {
    const bool wide = (Win % 32 == 0) || Win >= 256;
    return wide ? ((Hin + 7) / 8) * ((Win + 31) / 32) : ((Hin + 15) / 16) * ((Win + 15) / 16);
}

extern "C" int m4d_conv_cl_planar_gnstats(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* bias, const void* resid,
                                          int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout, int kt,
                                          int To, float* gn_partial, m4d_stream stream) {
    M4D_CHECK_ARG(out && gn_partial, "conv_cl_planar_gnstats: null output / statistics");
    return conv_cl_planar_impl(dt, x, x_plane_stride, w, bias, resid, resid_ld, out, out_ld, Tin, Hin, Win, Cin, Cout, kt, To, nullptr, nullptr, 0, 0,
                               gn_partial, nullptr, stream);
}

extern "C" int m4d_conv_cl_planar_norm(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* bias, const void* resid,
                                       int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout, int kt,
                                       int To, const float* norm_gamma, void* norm_out, int64_t norm_out_plane_stride, int silu,
                                       m4d_stream stream) {
    M4D_CHECK_ARG(norm_out, "conv_cl_planar_norm: null normalised output");
    return conv_cl_planar_impl(dt, x, x_plane_stride, w, bias, resid, resid_ld, out, out_ld, Tin, Hin, Win, Cin, Cout, kt, To, norm_gamma, norm_out,
                               norm_out_plane_stride, silu, nullptr, nullptr, stream);
}

// one entry for the three planar forms with tiled weights: norm_out / gn_partial select the fused epilogues as in the entries above
extern "C" int m4d_conv_cl_planar_tw(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* w_tiled, const void* bias,
                                     const void* resid, int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin,
                                     int Cout, int kt, int To, const float* norm_gamma, void* norm_out, int64_t norm_out_plane_stride,
                                     int silu, float* gn_partial, m4d_stream stream) {
    M4D_CHECK_ARG(out || norm_out, "conv_cl_planar_tw: null output");
    M4D_CHECK_ARG(!(norm_out && gn_partial), "conv_cl_planar_tw: fused norm and GroupNorm statistics exclude each other");
    return conv_cl_planar_impl(dt, x, x_plane_stride, w, bias, resid, resid_ld, out, out_ld, Tin, Hin, Win, Cin, Cout, kt, To, norm_gamma, norm_out,
                               norm_out_plane_stride, silu, gn_partial, w_tiled, stream);
}

// ---- tiled weights ----
namespace {
// slot i of the tiled copy: unit u = i / 64 = (row block * chunks + chunk) * taps + tap, slot s = i % 64 -> row s / 2 of the block, physical
// 16-byte half s & 1 holding channels [8 c, 8 c + 8) of the chunk with c = (s & 1) ^ ((row >> 3) & 1) — the LDS image the halo kernels read
__global__ __launch_bounds__(256) void conv_pack_w_kernel(const bf16_t* w, bf16_t* out, int Cout, int Cin, int taps, int64_t nslots) {
    const int nch = Cin / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i & 63);
        int64_t u = i >> 6;
        const int tap = (int)(u % taps); u /= taps;
        const int chunk = (int)(u % nch);
        const int rb = (int)(u / nch);
        const int r = s >> 1;
        const int row = min(rb * 32 + r, Cout - 1);
        const int c = (s & 1) ^ ((r >> 3) & 1);
        const uint4 v = *reinterpret_cast<const uint4*>(w + ((int64_t)row * taps + tap) * Cin + chunk * 16 + c * 8);
        *reinterpret_cast<uint4*>(out + i * 8) = v;
    }
}
}  // namespace

extern "C" int64_t m4d_conv_tiled_weight_bytes(int Cin, int Cout, int taps) {
    if (Cin <= 0 || Cout <= 0 || taps <= 0 || Cin % 16) return 0;
    return (int64_t)((Cout + 31) / 32) * (Cin / 16) * taps * 1024;
}

extern "C" int m4d_conv_pack_weights(m4d_dtype dt, const void* w, void* w_tiled, int Cin, int Cout, int taps, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16, "conv_pack_weights: bf16 only");
    M4D_CHECK_ARG(w && w_tiled && Cin > 0 && Cout > 0 && taps > 0 && Cin % 16 == 0, "conv_pack_weights: null / empty / Cin %% 16");
    M4D_CHECK_ARG(((uintptr_t)w % 16) == 0 && ((uintptr_t)w_tiled % 16) == 0, "conv_pack_weights: pointers must be 16-byte aligned");
    const int64_t nslots = m4d_conv_tiled_weight_bytes(Cin, Cout, taps) / 16;
    M4D_CHECK_ARG(nslots * 16 < (1ll << 31), "conv_pack_weights: the tiled copy must stay below 2 GiB");
    const unsigned grid = (unsigned)std::min<int64_t>((nslots + 255) / 256, 4096);
    hipLaunchKernelGGL(conv_pack_w_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)w_tiled, Cout, Cin, taps, nslots);
    M4D_CHECK_LAUNCH("conv_pack_weights");
    return 0;
}
