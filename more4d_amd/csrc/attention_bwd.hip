// m4d_attention_bwd: flash-attention backward for gfx950, same "doubly swapped" register layout as the
// forward kernel (attention.hip), so no probability / score tile ever leaves registers.
//
// One template covers both passes.  "X" is the lane-local side (32 rows per wave, one row per lane pair),
// "Y" the streamed side (tiles of YB rows through LDS):
//   S^T[y][x] = Ya[y,:] . Xa[x,:]        G^T[y][x] = Yb[y,:] . Xb[x,:]      (A = Y tile rows, B = X fragments)
//   pass Q  (X = queries, Y = keys):   Xa = Q, Xb = dO, Ya = K, Yb = V
//        P = exp2(S*sc - lse[x]);  dS = P * (G - delta[x]) * scale;     dQ^T[d][x] += K^T[d][y] dS^T[y][x]
//   pass KV (X = keys, Y = queries):   Xa = K, Xb = V,  Ya = Q, Yb = dO
//        P = exp2(S*sc - lse[y]);  dV^T[d][x] += dO^T[d][y] P^T[y][x];
//        dS = P * (G - delta[y]) * scale;                               dK^T[d][x] += Q^T[d][y] dS^T[y][x]
// The accumulate MFMAs take the TRANSPOSED Y tiles (K^T, Q^T, dO^T: [d][y], y contiguous) as A operand, the same
// role V^T plays in the forward; the host supplies them (m4d_transpose, HBM-bound, <2 % of the pass).
// lse is the forward's log2-domain log-sum-exp, delta[q] = sum_d dO[q,d] O[q,d] (m4d_attention_bwd computes it).
// Workgroup = 4 waves x 32 X rows; one wave per SIMD (512 VGPRs: two fragment sets + two accumulators).
#include <stdlib.h>
#include "common.h"
#include "attn_common.h"
#include "more4d_hip.h"

namespace {

struct BwdArgs {
    const void *xa, *xb, *ya, *yb, *yat, *ybt;
    int64_t xa_bs, xa_ls, xb_bs, xb_ls, ya_bs, ya_ls, yb_bs, yb_ls, yat_bs, yat_ls, ybt_bs, ybt_ls;
    const float *lse, *delta;     // [B, heads, Lq]
    void *out_a, *out_b;          // pass Q: dQ ; pass KV: dK, dV
    int64_t oa_bs, oa_ls, ob_bs, ob_ls;
    int64_t LX, LXs, LY, Lq;      // X rows valid / stored, Y rows valid, query count (lse/delta row length)
    int B, heads, nx_tiles, accumulate;
    float sc, scale;
    // production kernels only: blockIdx.y walks Y chunks of y_chunk rows and adds into the float workspace `ws`
    // ([B, LXs, heads*D]) instead of storing (short X sides, e.g. 512 text keys against 21 840 queries)
    float* ws;
    int64_t y_chunk;
};

template <typename T, int D, bool KV>
__global__ __launch_bounds__(256, 1) void attn_bwd_kernel(BwdArgs p) {
    constexpr int ES = sizeof(T);
    constexpr int YB = TileCfg<T>::KVB;
    constexpr int NSUB = YB / 32;
    constexpr int RRB = D * ES;       // bytes per row of a row-major Y tile
    constexpr int TRB = YB * ES;      // bytes per row of a transposed Y tile
    constexpr int RCPR = RRB / 16, TCPR = TRB / 16;
    constexpr int TILE_BYTES = YB * D * ES;
    constexpr int NLD = TILE_BYTES / 16 / 256;
    constexpr int EPC = 16 / ES;
    constexpr int NKK = D / 16, NDB = D / 32;
    constexpr int NT = KV ? 4 : 3;    // tiles per stage
    typedef typename Frag8<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NT tiles + 2*YB floats (> 64 KiB for D = 128, pass KV)
    char* sYa = smem;
    char* sYb = smem + TILE_BYTES;
    char* sYaT = smem + 2 * TILE_BYTES;
    char* sYbT = smem + 3 * TILE_BYTES;                 // KV only
    float* sStat = reinterpret_cast<float*>(smem + NT * TILE_BYTES);   // KV only: lse[YB], delta[YB]

    const int hb = blockIdx.x / p.nx_tiles, xt = blockIdx.x % p.nx_tiles;
    const int b = hb / p.heads, h = hb % p.heads;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hi = lane >> 5;
    const int64_t xrow = (int64_t)xt * 128 + wave * 32 + li;
    const bool xvalid = xrow < p.LX;

    // ---- X fragments (B operands) ----
    frag_t xaf[NKK], xbf[NKK];
    {
        const T* pa = (const T*)p.xa + b * p.xa_bs + xrow * p.xa_ls + (int64_t)h * D + hi * 8;
        const T* pb = (const T*)p.xb + b * p.xb_bs + xrow * p.xb_ls + (int64_t)h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            if (xvalid) {
                xaf[kk] = *reinterpret_cast<const frag_t*>(pa + kk * 16);
                xbf[kk] = *reinterpret_cast<const frag_t*>(pb + kk * 16);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { xaf[kk][j] = (T)0.f; xbf[kk][j] = (T)0.f; }
            }
        }
    }
    float lse_x = 0.f, delta_x = 0.f;
    if (!KV && xvalid) {
        const int64_t si = ((int64_t)b * p.heads + h) * p.Lq + xrow;
        lse_x = p.lse[si];
        delta_x = p.delta[si];
    }

    f32x16 acc_a[NDB], acc_b[KV ? NDB : 1];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_a[d][r] = 0.f; if constexpr (KV) acc_b[d][r] = 0.f; }

    const T* gya = (const T*)p.ya + b * p.ya_bs + (int64_t)h * D;
    const T* gyb = (const T*)p.yb + b * p.yb_bs + (int64_t)h * D;
    const T* gyat = (const T*)p.yat + b * p.yat_bs + (int64_t)h * D * p.yat_ls;
    const T* gybt = KV ? (const T*)p.ybt + b * p.ybt_bs + (int64_t)h * D * p.ybt_ls : nullptr;
    const float* glse = p.lse + ((int64_t)b * p.heads + h) * p.Lq;
    const float* gdel = p.delta + ((int64_t)b * p.heads + h) * p.Lq;

    uint4 ra[NLD], rb[NLD], rat[NLD], rbt[KV ? NLD : 1];
    float rstat = 0.f;

    auto load_t = [&](const T* base, int64_t ls, int64_t y0, int row, int ch) -> uint4 {
        const int64_t y = y0 + ch * EPC;
        const T* src = base + row * ls + y;
        if (y + EPC <= p.LY) return *reinterpret_cast<const uint4*>(src);
        if (y >= p.LY) return make_uint4(0, 0, 0, 0);
        union { uint4 u; T e[EPC]; } tmp;
        tmp.u = make_uint4(0, 0, 0, 0);
        for (int j = 0; j < EPC; ++j)
            if (y + j < p.LY) tmp.e[j] = src[j];
        return tmp.u;
    };
    auto gload = [&](int64_t y0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = t + 256 * i;
            {
                const int row = c / RCPR, ch = c % RCPR;
                const int64_t y = y0 + row;
                const bool ok = y < p.LY;
                ra[i] = ok ? *reinterpret_cast<const uint4*>(gya + y * p.ya_ls + ch * EPC) : make_uint4(0, 0, 0, 0);
                rb[i] = ok ? *reinterpret_cast<const uint4*>(gyb + y * p.yb_ls + ch * EPC) : make_uint4(0, 0, 0, 0);
            }
            {
                const int row = c / TCPR, ch = c % TCPR;
                rat[i] = load_t(gyat, p.yat_ls, y0, row, ch);
                if constexpr (KV) rbt[i] = load_t(gybt, p.ybt_ls, y0, row, ch);
            }
        }
        if (KV && t < 2 * YB) {
            const int64_t y = y0 + (t % YB);
            // invalid rows: lse = +inf makes their probabilities exactly zero
            rstat = y < p.LY ? (t < YB ? glse[y] : gdel[y]) : (t < YB ? INFINITY : 0.f);
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = t + 256 * i;
            *reinterpret_cast<uint4*>(sYa + swz_off<RRB>(c / RCPR, c % RCPR)) = ra[i];
            *reinterpret_cast<uint4*>(sYb + swz_off<RRB>(c / RCPR, c % RCPR)) = rb[i];
            *reinterpret_cast<uint4*>(sYaT + swz_off<TRB>(c / TCPR, c % TCPR)) = rat[i];
            if constexpr (KV) *reinterpret_cast<uint4*>(sYbT + swz_off<TRB>(c / TCPR, c % TCPR)) = rbt[i];
        }
        if (KV && t < 2 * YB) sStat[t] = rstat;
    };

    gload(0);
    for (int64_t y0 = 0; y0 < p.LY; y0 += YB) {
        swrite();
        __syncthreads();
        if (y0 + YB < p.LY) gload(y0 + YB);

        f32x16 s[NSUB], g[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[sub][r] = 0.f; g[sub][r] = 0.f; }
            const int yr = sub * 32 + perm23(li);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int c0 = (kk * 16 + hi * 8) / EPC;
                const int o0 = swz_off<RRB>(yr, c0), o1 = swz_off<RRB>(yr, c0 + 1);
                frag_t fa = lds_frag<T>(sYa, o0, o1);
                mma32(fa, xaf[kk], s[sub]);
                frag_t fb = lds_frag<T>(sYb, o0, o1);
                mma32(fb, xbf[kk], g[sub]);
            }
        }
        // ---- probabilities and dS (register r of sub-tile `sub` <-> y = y0 + sub*32 + 16*(r>>3) + 8*hi + (r&7)) ----
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int yi = sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                float pv, ds;
                if (KV) {
                    pv = xvalid ? exp2f(fmaf(s[sub][r], p.sc, -sStat[yi])) : 0.f;
                    ds = pv * (g[sub][r] - sStat[YB + yi]) * p.scale;
                } else {
                    pv = (y0 + yi < p.LY) ? exp2f(fmaf(s[sub][r], p.sc, -lse_x)) : 0.f;
                    ds = pv * (g[sub][r] - delta_x) * p.scale;
                }
                s[sub][r] = pv;
                g[sub][r] = ds;
            }
        // ---- accumulate through the transposed tiles ----
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const frag_t dsf = pack8<T>(g[sub], si * 8);
                const int c0 = (sub * 32 + si * 16 + hi * 8) / EPC;
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    const int row = d * 32 + li;
                    const int o0 = swz_off<TRB>(row, c0), o1 = swz_off<TRB>(row, c0 + 1);
                    frag_t fa = lds_frag<T>(sYaT, o0, o1);
                    mma32(fa, dsf, acc_a[d]);
                    if constexpr (KV) {
                        const frag_t pf = pack8<T>(s[sub], si * 8);
                        frag_t fb = lds_frag<T>(sYbT, o0, o1);
                        mma32(fb, pf, acc_b[d]);
                    }
                }
            }
        __syncthreads();
    }

    // ---- store: lane owns d = dblk*32 + rq*8 + hi*4 + [0,4) of its X row ----
    if (xrow < p.LXs) {
        T* oa = (T*)p.out_a + b * p.oa_bs + xrow * p.oa_ls + (int64_t)h * D + hi * 4;
        T* ob = KV ? (T*)p.out_b + b * p.ob_bs + xrow * p.ob_ls + (int64_t)h * D + hi * 4 : nullptr;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc_a[d][rq * 4 + e];
                T* dst = oa + d * 32 + rq * 8;
                if (p.accumulate) {
                    f32x4 prev = load4(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += prev[e];
                }
                store4(dst, v);
                if constexpr (KV) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc_b[d][rq * 4 + e];
                    T* dstb = ob + d * 32 + rq * 8;
                    if (p.accumulate) {
                        f32x4 prev = load4(dstb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += prev[e];
                    }
                    store4(dstb, v);
                }
            }
    }
}

#include "attention_bwd128.h"
#include "attention_bwd_kvp.h"
#include "attention_bwd_dqp.h"
#include "attention_bwd64.h"

// float workspace [B, rows, C] -> T output rows (b, l) at out + b*bs + l*ls (+= when accumulate)
struct WsArgs { const float* ws; void* out; int64_t bs, ls, rows; int C, B, accumulate; };
__global__ __launch_bounds__(256) void ws_store_kernel(WsArgs p) {
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nv = p.C >> 2, total = (int64_t)p.B * p.rows * nv;
    if (i4 >= total) return;
    const int64_t c = (i4 % nv) * 4, l = (i4 / nv) % p.rows, b = i4 / nv / p.rows;
    f32x4 v = load4(p.ws + i4 * 4);
    bf16_t* dst = (bf16_t*)p.out + b * p.bs + l * p.ls + c;
    if (p.accumulate) v += load4(dst);
    store4(dst, v);
}

// delta[b, h, l] = sum_d dO[b, l, h, d] * O[b, l, h, d];  16 lanes x 8 elements per (row, head) for D = 128
struct DeltaArgs {
    const void *o, *d_o; float* delta;
    int64_t o_bs, o_ls, do_bs, do_ls, Lq;
    int B, heads, D;
};
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(DeltaArgs p) {
    // D/8 lanes x 8 elements per (row, head): a wave reads 64 x 16 B contiguous bytes of each operand
    const int lpr = p.D >> 3;                                   // lanes per (row, head): 4, 8 or 16
    const int64_t total = (int64_t)p.B * p.Lq * p.heads;        // ordered (b, l, h): h fastest => contiguous memory
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = gid / lpr;
    const int sub = (int)(gid % lpr);
    float s = 0.f;
    int h = 0; int64_t l = 0, b = 0;
    if (item < total) {
        h = (int)(item % p.heads);
        l = (item / p.heads) % p.Lq;
        b = item / p.heads / p.Lq;
        const T* po = (const T*)p.o + b * p.o_bs + l * p.o_ls + (int64_t)h * p.D + sub * 8;
        const T* pd = (const T*)p.d_o + b * p.do_bs + l * p.do_ls + (int64_t)h * p.D + sub * 8;
        const f32x4 a0 = load4(po), a1 = load4(po + 4), c0 = load4(pd), c1 = load4(pd + 4);
        s = a0[0] * c0[0] + a0[1] * c0[1] + a0[2] * c0[2] + a0[3] * c0[3] + a1[0] * c1[0] + a1[1] * c1[1] + a1[2] * c1[2] + a1[3] * c1[3];
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (item < total && sub == 0) p.delta[(b * p.heads + h) * p.Lq + l] = s;
}

template <typename T, int D, bool KV>
int launch_one(const BwdArgs& p, hipStream_t st) {
    constexpr int YB = TileCfg<T>::KVB;
    constexpr int LDS = (KV ? 4 : 3) * YB * D * (int)sizeof(T) + 2 * YB * 4;
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)attn_bwd_kernel<T, D, KV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return -3;
        configured.mark();
    }
    dim3 grid((unsigned)((int64_t)p.nx_tiles * p.heads * p.B)), block(256);
    hipLaunchKernelGGL((attn_bwd_kernel<T, D, KV>), grid, block, LDS, st, p);
    return 0;
}

template <typename T, bool KV>
int launch_bwd(const BwdArgs& p, int D, hipStream_t st) {
    switch (D) {
        case 32: return launch_one<T, 32, KV>(p, st);
        case 64: return launch_one<T, 64, KV>(p, st);
        case 128: return launch_one<T, 128, KV>(p, st);
        default: return -2;
    }
}

}  // namespace

extern "C" int m4d_attention_bwd(m4d_dtype dt, const m4d_attn_bwd_args* a, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "attention_bwd: bad dtype %d", (int)dt);
    M4D_CHECK_ARG(a, "attention_bwd: null args");
    M4D_CHECK_ARG(a->q && a->k && a->v && a->o && a->d_o && a->lse && a->delta && a->dq && a->dk && a->dv, "attention_bwd: null pointer");
    M4D_CHECK_ARG(a->B > 0 && a->Lq > 0 && a->Lk > 0 && a->heads > 0 && a->Lk_rows >= a->Lk, "attention_bwd: bad sizes");
    if (!(a->head_dim == 32 || a->head_dim == 64 || a->head_dim == 128)) {
        m4d_set_error("attention_bwd: unsupported head_dim %d (32, 64, 128)", a->head_dim);
        return -2;
    }
    const int64_t strides[] = {a->q_bs, a->q_ls, a->k_bs, a->k_ls, a->v_bs, a->v_ls, a->o_bs, a->o_ls, a->do_bs, a->do_ls,
                               a->dq_bs, a->dq_ls, a->dk_bs, a->dk_ls, a->dv_bs, a->dv_ls};
    for (int64_t s : strides) M4D_CHECK_ARG(s % 8 == 0, "attention_bwd: strides must be multiples of 8 elements");
    if (a->qt || a->kt || a->dot) {      // (transposed copies: only the generic kernels read them)
        const int64_t tstrides[] = {a->qt_bs, a->qt_ls, a->kt_bs, a->kt_ls, a->dot_bs, a->dot_ls};
        for (int64_t s : tstrides) M4D_CHECK_ARG(s % 8 == 0, "attention_bwd: strides of the transposed operands must be multiples of 8 elements");
    }
    const void* ptrs[] = {a->q, a->k, a->v, a->o, a->d_o, a->qt, a->kt, a->dot, a->dq, a->dk, a->dv};      // (null qt / kt / dot pass)
    for (const void* q : ptrs) M4D_CHECK_ARG(((uintptr_t)q % 16) == 0, "attention_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool bf = dt == M4D_BF16;
    {
        DeltaArgs d;
        d.o = a->o; d.d_o = a->d_o; d.delta = a->delta;
        d.o_bs = a->o_bs; d.o_ls = a->o_ls; d.do_bs = a->do_bs; d.do_ls = a->do_ls; d.Lq = a->Lq;
        d.B = a->B; d.heads = a->heads; d.D = a->head_dim;
        const int64_t total = (int64_t)a->B * a->heads * a->Lq * (a->head_dim / 8);
        dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (bf) hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, grid, block, 0, st, d);
        else hipLaunchKernelGGL(attn_delta_kernel<float>, grid, block, 0, st, d);
        M4D_CHECK_LAUNCH("attention_bwd(delta)");
    }
    BwdArgs p;
    p.ws = nullptr; p.y_chunk = 0;
    p.lse = a->lse; p.delta = a->delta; p.Lq = a->Lq;
    p.B = a->B; p.heads = a->heads; p.scale = a->scale; p.sc = a->scale * 1.4426950408889634f;
    M4D_ENV_ONCE(bwd_generic, "M4D_ATTN_BWD_GENERIC", 0);
    if (bf && a->head_dim == 128 && !bwd_generic) {
        // production path: three forward-shaped passes (attention_bwd128.h)
        p.yat = p.ybt = nullptr; p.yat_bs = p.yat_ls = p.ybt_bs = p.ybt_ls = 0; p.out_b = nullptr; p.ob_bs = p.ob_ls = 0;
        // dQ: X = (Q, dO), Y = (K, V)
        p.xa = a->q; p.xa_bs = a->q_bs; p.xa_ls = a->q_ls; p.xb = a->d_o; p.xb_bs = a->do_bs; p.xb_ls = a->do_ls;
        p.ya = a->k; p.ya_bs = a->k_bs; p.ya_ls = a->k_ls; p.yb = a->v; p.yb_bs = a->v_bs; p.yb_ls = a->v_ls;
        p.out_a = a->dq; p.oa_bs = a->dq_bs; p.oa_ls = a->dq_ls;
        p.LX = p.LXs = a->Lq; p.LY = a->Lk; p.nx_tiles = (int)((a->Lq + 255) / 256); p.accumulate = a->accumulate_dq;
        // M4D_ATTN_BWD_DQ_PHASED=1 (default): attn_bwd_dqp_kernel (attention_bwd_dqp.h, the forward kernel's phased schedule); 0: lock-step
        M4D_ENV_ONCE(dq_phased, "M4D_ATTN_BWD_DQ_PHASED", 1);
        // M4D_ATTN_BWD64 (default 3): attn_bwd_dq64_kernel / attn_bwd_kv64_kernel (attention_bwd64.h: one wave per SIMD, generated streams) where eligible; 0: A/B
        M4D_ENV_ONCE(bwd64_on, "M4D_ATTN_BWD64", 3);      // bit 0: the dQ pass, bit 1: the dK / dV pass
        const bool dq64 = (bwd64_on & 1) && bwd_dq64_ok(p);
        if (dq64 ? launch_bwd_dq64(p, st) : dq_phased ? launch_bwd_dqp(p, st) : launch_bwd128<BWD_DQ>(p, st, 1)) {
            m4d_set_error("attention_bwd: cannot configure the dq kernel");
            return -3;
        }
        M4D_CHECK_LAUNCH("attention_bwd(dq128)");
        m4d_count_launch(dq64 ? M4D_KC_ATTN_BWD64 : M4D_KC_ATTN_BWD128);
        // dK: X = (K, V), Y = (Q, dO)
        p.xa = a->k; p.xa_bs = a->k_bs; p.xa_ls = a->k_ls; p.xb = a->v; p.xb_bs = a->v_bs; p.xb_ls = a->v_ls;
        p.ya = a->q; p.ya_bs = a->q_bs; p.ya_ls = a->q_ls; p.yb = a->d_o; p.yb_bs = a->do_bs; p.yb_ls = a->do_ls;
        p.out_a = a->dk; p.oa_bs = a->dk_bs; p.oa_ls = a->dk_ls;
        p.LX = a->Lk; p.LXs = a->Lk_rows; p.LY = a->Lq; p.nx_tiles = (int)((a->Lk_rows + 255) / 256); p.accumulate = a->accumulate_dkv;
        // few key tiles against many queries (cross-attention): split the query loop over blockIdx.y
        const int64_t blocks = (int64_t)p.nx_tiles * a->heads * a->B, ytiles = (a->Lq + 63) / 64;
        const int64_t ws_need = (int64_t)a->B * a->Lk_rows * a->heads * 128;
        int nsplit = 1;
        if (a->ws && a->ws_elems >= ws_need && blocks < 192 && ytiles >= 32) {
            nsplit = (int)((512 + blocks - 1) / blocks);
            if (nsplit > ytiles / 8) nsplit = (int)(ytiles / 8);
            if (nsplit > 16) nsplit = 16;
        }
        const int64_t chunk = nsplit > 1 ? ((ytiles + nsplit - 1) / nsplit) * 64 : 0;
        // fused dK / dV pass (attention_bwd128.h: role-split wave pairs, S and P computed once): the self-attention shape; the split-Y
        // cross-attention case (few keys, many queries, float workspace) keeps the two separate passes.  M4D_ATTN_BWD_FUSED=0: A/B.
        // 2 (default) = attn_bwd_kvp_kernel (attention_bwd_kvp.h: the forward kernel's phased schedule), 1 = the first fused kernel.
        M4D_ENV_ONCE(bwd_fused, "M4D_ATTN_BWD_FUSED", 2);
        // few keys against many queries (cross-attention, nsplit > 1): M4D_ATTN_BWD_FUSED_SHORT=1 sends that case through the fused kernel as
        // well (ceil(Lk / 128) x heads workgroups, each walking all query tiles) instead of the split dK / dV passes + workspace reduction
        M4D_ENV_ONCE(bwd_fused_short, "M4D_ATTN_BWD_FUSED_SHORT", 1);
        if (bwd_fused && (nsplit == 1 || (bwd_fused == 2 && bwd_fused_short))) {
            p.out_b = a->dv; p.ob_bs = a->dv_bs; p.ob_ls = a->dv_ls;
            // M4D_ATTN_BWD64 & 2 (default on): attn_bwd_kv64_kernel (attention_bwd64.h: role-split, one wave per SIMD) where eligible
            if ((bwd64_on & 2) && nsplit == 1 && bwd_kv64_ok(p)) {
                if (launch_bwd_kv64(p, st)) { m4d_set_error("attention_bwd: cannot configure the dk/dv kernel"); return -3; }
                M4D_CHECK_LAUNCH("attention_bwd(dkv64)");
                m4d_count_launch(M4D_KC_ATTN_BWD64);
                return 0;
            }
            p.nx_tiles = (int)((a->Lk_rows + 127) / 128);
            if (bwd_fused == 2 ? launch_bwd_kvp(p, st) : launch_bwd_kv128(p, st)) { m4d_set_error("attention_bwd: cannot configure the fused dk/dv kernel"); return -3; }
            M4D_CHECK_LAUNCH("attention_bwd(dkv128 fused)");
            m4d_count_launch(M4D_KC_ATTN_BWD128);
            return 0;
        }
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {   // dV: X = (K), Y = (Q, dO)
                p.out_a = a->dv; p.oa_bs = a->dv_bs; p.oa_ls = a->dv_ls;
            }
            if (nsplit > 1) {
                if (hipMemsetAsync(a->ws, 0, ws_need * sizeof(float), st) != hipSuccess) { m4d_set_error("attention_bwd: memset failed"); return -3; }
                p.ws = a->ws; p.y_chunk = chunk;
            }
            const int rc2 = pass == 0 ? launch_bwd128<BWD_DK>(p, st, nsplit) : launch_bwd128<BWD_DV>(p, st, nsplit);
            if (rc2) { m4d_set_error("attention_bwd: cannot configure the dk/dv kernel"); return -3; }
            M4D_CHECK_LAUNCH("attention_bwd(dkv128)");
            m4d_count_launch(M4D_KC_ATTN_BWD128);
            if (nsplit > 1) {
                WsArgs w{a->ws, p.out_a, p.oa_bs, p.oa_ls, a->Lk_rows, a->heads * 128, a->B, p.accumulate};
                const int64_t n4 = ws_need / 4;
                hipLaunchKernelGGL(ws_store_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w);
                M4D_CHECK_LAUNCH("attention_bwd(ws_store)");
            }
        }
        return 0;
    }
    // the generic kernels read transposed copies of Q, K, dO (the bf16 / head_dim 128 passes above take them out of the row-major tiles)
    M4D_CHECK_ARG(a->qt && a->kt && a->dot, "attention_bwd: qt / kt / dot are required for this dtype / head_dim");
    // ---- pass Q: X = queries ----
    p.xa = a->q; p.xa_bs = a->q_bs; p.xa_ls = a->q_ls;
    p.xb = a->d_o; p.xb_bs = a->do_bs; p.xb_ls = a->do_ls;
    p.ya = a->k; p.ya_bs = a->k_bs; p.ya_ls = a->k_ls;
    p.yb = a->v; p.yb_bs = a->v_bs; p.yb_ls = a->v_ls;
    p.yat = a->kt; p.yat_bs = a->kt_bs; p.yat_ls = a->kt_ls;
    p.ybt = nullptr; p.ybt_bs = p.ybt_ls = 0;
    p.out_a = a->dq; p.oa_bs = a->dq_bs; p.oa_ls = a->dq_ls;
    p.out_b = nullptr; p.ob_bs = p.ob_ls = 0;
    p.LX = p.LXs = a->Lq; p.LY = a->Lk; p.nx_tiles = (int)((a->Lq + 127) / 128); p.accumulate = a->accumulate_dq;
    int rc = bf ? launch_bwd<bf16_t, false>(p, a->head_dim, st) : launch_bwd<float, false>(p, a->head_dim, st);
    if (rc) { m4d_set_error("attention_bwd: unsupported configuration"); return rc; }
    M4D_CHECK_LAUNCH("attention_bwd(dq)");
    m4d_count_launch(M4D_KC_ATTN_BWD_GENERIC);
    // ---- pass KV: X = keys ----
    p.xa = a->k; p.xa_bs = a->k_bs; p.xa_ls = a->k_ls;
    p.xb = a->v; p.xb_bs = a->v_bs; p.xb_ls = a->v_ls;
    p.ya = a->q; p.ya_bs = a->q_bs; p.ya_ls = a->q_ls;
    p.yb = a->d_o; p.yb_bs = a->do_bs; p.yb_ls = a->do_ls;
    p.yat = a->qt; p.yat_bs = a->qt_bs; p.yat_ls = a->qt_ls;
    p.ybt = a->dot; p.ybt_bs = a->dot_bs; p.ybt_ls = a->dot_ls;
    p.out_a = a->dk; p.oa_bs = a->dk_bs; p.oa_ls = a->dk_ls;
    p.out_b = a->dv; p.ob_bs = a->dv_bs; p.ob_ls = a->dv_ls;
    p.LX = a->Lk; p.LXs = a->Lk_rows; p.LY = a->Lq; p.nx_tiles = (int)((a->Lk_rows + 127) / 128); p.accumulate = a->accumulate_dkv;
    rc = bf ? launch_bwd<bf16_t, true>(p, a->head_dim, st) : launch_bwd<float, true>(p, a->head_dim, st);
    if (rc) { m4d_set_error("attention_bwd: unsupported configuration"); return rc; }
    M4D_CHECK_LAUNCH("attention_bwd(dkv)");
    m4d_count_launch(M4D_KC_ATTN_BWD_GENERIC);
    return 0;
}
