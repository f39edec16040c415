// m4d_attention: non-causal flash attention forward for gfx950 (wave64, MFMA 32x32).
//
// Formulation ("doubly swapped", so every softmax statistic is lane-local):
//   S^T[key][q] = K · Q^T        A-operand = K tile rows (from LDS), B-operand = Q (registers)
//   O^T[d][q]  += V^T · P^T      A-operand = V^T tile rows (from LDS), B-operand = P (registers)
// Lane (q = l&31, hi = l>>5) of a wave owns query row q: its 16 S^T accumulator registers per 32-key
// sub-tile are 16 of that row's 32 scores (the other 16 live in lane l^32), and its O^T registers are
// 64 of the row's 128 outputs.  Row max needs one cross-half exchange per tile; the row sum is kept
// per half and combined once in the epilogue.  K rows are fetched from LDS through the permutation
// "swap index bits 2 and 3", which makes accumulator registers 8s..8s+7 hold 8 CONSECUTIVE keys
// (16s + 8hi + [0,8)), i.e. exactly the B-operand fragment of the P·V MFMA and one 16-B read of a V^T row:
// P never leaves registers and needs no cross-lane shuffle.
// V arrives transposed (V^T[d][key]) — the projection GEMM writes it that way for free.
//
// Workgroup = 4 waves x 32 queries = 128 queries; KV tile = 64 keys (bf16) / 32 keys (fp32 parity
// mode); one LDS stage (K tile + V^T tile, 32 KiB) with the next tile's global loads held in
// registers while the current tile is computed (issue-early / write-late), two barriers per tile.
// LDS images are XOR-swizzled per 16-B chunk so fragment reads (ds_read_b128) are conflict-free.
#include <stdlib.h>
#include "common.h"
#include "attn_common.h"
#include "more4d_hip.h"

namespace {

struct AttnArgs {
    const void* q; void* out;
    m4d_kv_segs kv;
    int64_t q_bs, q_ls, o_bs, o_ls, Lq;
    int B, heads, nq_tiles, accumulate;
    float sc;  // softmax scale * log2(e)
    float* lse;  // optional [B, heads, Lq]: log2-domain log-sum-exp of the scaled scores (training)
    int abl;     // timing ablations of attn128p_kernel (tools only, results wrong): 1 no softmax math, 2 no M-phase stream; 64 = stamps
    unsigned long long* dbg;   // tool builds (abl & 64): per-workgroup shader-clock / wall-clock stamps, 4 words each (tools/attn_clock.py)
};

template <typename T, int D>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void attn_kernel(AttnArgs p) {
    constexpr int ES = sizeof(T);
    constexpr int KVB = TileCfg<T>::KVB;
    constexpr int NSUB = KVB / 32;
    constexpr int KRB = D * ES;      // bytes per K row
    constexpr int VRB = KVB * ES;    // bytes per V^T row
    constexpr int KCPR = KRB / 16, VCPR = VRB / 16;
    constexpr int TILE_BYTES = KVB * D * ES;
    constexpr int NLD = TILE_BYTES / 16 / 256;  // 16-B chunks per thread per operand
    constexpr int EPC = 16 / ES;                 // elements per chunk
    constexpr int NKK = D / 16, NDB = D / 32;
    typedef typename Frag8<T>::type frag_t;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];
    char* sK = smem;
    char* sV = smem + TILE_BYTES;

    // ---- block -> (q tile, head, batch); (b,h) groups pinned per XCD so K/V stay in that L2 ----
    const int HB = p.heads * p.B;
    int qt, hb;
    if ((HB & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        hb = xcd * (HB >> 3) + idx / p.nq_tiles;
        qt = idx % p.nq_tiles;
    } else {
        hb = blockIdx.x / p.nq_tiles;
        qt = blockIdx.x % p.nq_tiles;
    }
    const int b = hb / p.heads, h = hb % p.heads;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hi = lane >> 5;
    const int64_t qrow = (int64_t)qt * 128 + wave * 32 + li;
    const bool qvalid = qrow < p.Lq;

    // ---- Q fragments (B operand of S^T = K Q^T) ----
    frag_t qf[NKK];
    {
        const T* qp = (const T*)p.q + b * p.q_bs + qrow * p.q_ls + (int64_t)h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            if (qvalid) qf[kk] = *reinterpret_cast<const frag_t*>(qp + kk * 16);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[kk][j] = (T)0.f;
            }
        }
    }

    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- KV tile iterator over segments ----
    int seg = 0;
    while (seg < p.kv.nseg && p.kv.len[seg] <= 0) ++seg;
    int64_t key0 = 0;
    uint4 rk[NLD], rv[NLD];

    auto gload = [&](int s, int64_t k0) {
        const int64_t len = p.kv.len[s];
        const T* kp = (const T*)p.kv.k[s] + b * p.kv.k_bs[s] + (int64_t)h * D;
        const T* vp = (const T*)p.kv.vt[s] + b * p.kv.vt_bs[s] + (int64_t)h * D * p.kv.vt_ls[s];
        const int64_t kls = p.kv.k_ls[s], vls = p.kv.vt_ls[s];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = t + 256 * i;
            {   // K: [KVB rows][KCPR chunks]
                const int row = c / KCPR, ch = c % KCPR;
                const int64_t key = k0 + row;
                rk[i] = key < len ? *reinterpret_cast<const uint4*>(kp + key * kls + ch * EPC) : make_uint4(0, 0, 0, 0);
            }
            {   // V^T: [D rows][VCPR chunks]
                const int row = c / VCPR, ch = c % VCPR;
                const int64_t key = k0 + ch * EPC;
                const T* src = vp + row * vls + key;
                if (key + EPC <= len) rv[i] = *reinterpret_cast<const uint4*>(src);
                else if (key >= len) rv[i] = make_uint4(0, 0, 0, 0);
                else {  // ragged last chunk: element-wise, zero fill
                    union { uint4 u; T e[EPC]; } tmp;
                    tmp.u = make_uint4(0, 0, 0, 0);
                    for (int j = 0; j < EPC; ++j)
                        if (key + j < len) tmp.e[j] = src[j];
                    rv[i] = tmp.u;
                }
            }
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = t + 256 * i;
            *reinterpret_cast<uint4*>(sK + swz_off<KRB>(c / KCPR, c % KCPR)) = rk[i];
            *reinterpret_cast<uint4*>(sV + swz_off<VRB>(c / VCPR, c % VCPR)) = rv[i];
        }
    };

    if (seg < p.kv.nseg) gload(seg, key0);
    while (seg < p.kv.nseg) {
        swrite();
        __syncthreads();
        // current tile meta, then advance + prefetch
        const int64_t cur_len = p.kv.len[seg], cur_k0 = key0;
        int nseg_ = seg;
        int64_t nk0 = key0 + KVB;
        if (nk0 >= cur_len) {
            nk0 = 0;
            ++nseg_;
            while (nseg_ < p.kv.nseg && p.kv.len[nseg_] <= 0) ++nseg_;
        }
        if (nseg_ < p.kv.nseg) gload(nseg_, nk0);

        // ---- S^T = K Q^T ----
        f32x16 s[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
            const int krow = sub * 32 + perm23(li);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int c0 = (kk * 16 + hi * 8) / EPC;
                frag_t kf = lds_frag<T>(sK, swz_off<KRB>(krow, c0), swz_off<KRB>(krow, c0 + 1));
                mma32(kf, qf[kk], s[sub]);
            }
        }
        // ---- mask the ragged tail of a segment ----
        if (cur_k0 + KVB > cur_len) {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t key = cur_k0 + sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= cur_len) s[sub][r] = -INFINITY;
                }
        }
        // ---- online softmax (exp2 domain) ----
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * p.sc);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(fmaf(s[sub][r], p.sc, -m_new));
                s[sub][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const frag_t pf = pack8<T>(s[sub], si * 8);
                const int c0 = (sub * 32 + si * 16 + hi * 8) / EPC;
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    const int vrow = d * 32 + li;
                    frag_t vf = lds_frag<T>(sV, swz_off<VRB>(vrow, c0), swz_off<VRB>(vrow, c0 + 1));
                    mma32(vf, pf, o[d]);
                }
            }
        __syncthreads();
        seg = nseg_;
        key0 = nk0;
    }

    // ---- epilogue: normalise, (accumulate), store 4 consecutive d per lane ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (p.lse && qvalid && hi == 0) p.lse[((int64_t)b * p.heads + h) * p.Lq + qrow] = m_run + log2f(l_tot);
    if (qvalid) {
        T* op = (T*)p.out + b * p.o_bs + qrow * p.o_ls + (int64_t)h * D + hi * 4;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[d][rq * 4 + e] * inv;
                T* dst = op + d * 32 + rq * 8;
                if (p.accumulate) {
                    f32x4 prev = load4(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]) + prev[e];
                }
                store4(dst, v);
            }
    }
}

// ============================================================================ production kernel: bf16, D = 128
// Same math and register layout as attn_kernel<bf16,128>, restructured around the measured bottlenecks
// (profiles/r01: VALU:MFMA = 15:1, 46 % of wave cycles parked on waits):
//  * K / V^T tiles arrive by direct global->LDS DMA (global_load_lds_dwordx4) into two LDS stages — no staging VGPRs,
//    no ds_write pass, next tile in flight during the whole current tile, ONE barrier per tile.  The XOR bank swizzle
//    is applied on the per-lane source address (DMA writes LDS lane-linearly) and on the fragment reads.
//  * the ragged last tile of a segment is staged through registers with zero fill (the DMA cannot mask), once per segment;
//  * VALU diet: LDS fragment offsets are per-lane constants + immediates, exp2 is the raw v_exp_f32, the O / l rescale is
//    skipped when no row's running max moved (wave-uniform branch), the cross-half max uses v_permlane32_swap.
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// NW waves (32 queries each) share every K/V^T tile: NW = 8 doubles the flops per DMA'd byte vs two 4-wave workgroups
// per CU (256 flop/B instead of 128; the global->LDS path is the scarce resource).
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn128_kernel(AttnArgs p) {
    typedef bf16_t T;
    constexpr int D = 128, KVB = 64, STAGE = 32768, VOFF = 16384, QB = NW * 32, IPW = 16 / NW;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int HB = p.heads * p.B;
    int qt, hb;
    if ((HB & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        hb = xcd * (HB >> 3) + idx / p.nq_tiles;
        qt = idx % p.nq_tiles;
    } else {
        hb = blockIdx.x / p.nq_tiles;
        qt = blockIdx.x % p.nq_tiles;
    }
    const int b = hb / p.heads, h = hb % p.heads;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hi = lane >> 5;
    const int64_t qrow = (int64_t)qt * QB + wave * 32 + li;
    const bool qvalid = qrow < p.Lq;

    bf16x8 qf[8];
    {
        const T* qp = (const T*)p.q + b * p.q_bs + qrow * p.q_ls + (int64_t)h * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (qvalid) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 16);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[kk][j] = (T)0.f;
            }
        }
    }
    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // per-lane LDS offsets (bytes inside a stage); sub-tile / d-block steps are immediates
    int koff[8], voff[4];
    {
        const int kr = perm23(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) koff[kk] = kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) voff[c] = VOFF + li * 128 + (((c * 2 + hi) ^ ((li >> 1) & 7)) << 4);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    // DMA lane roles: K instr = 4 rows x 256 B, V^T instr = 8 rows x 128 B; wave w issues instr w*4 .. w*4+3 of each
    const int k_r = lane >> 4, k_lc0 = lane & 15;     // row within the 4-row group, physical chunk
    const int v_r = lane >> 3, v_pc = lane & 7;

    int seg = 0;
    while (seg < p.kv.nseg && p.kv.len[seg] <= 0) ++seg;
    int64_t key0 = 0;

    auto dma_tile = [&](int stage, int s, int64_t k0) {
        const T* kp = (const T*)p.kv.k[s] + b * p.kv.k_bs[s] + (int64_t)h * D + k0 * p.kv.k_ls[s];
        const T* vp = (const T*)p.kv.vt[s] + b * p.kv.vt_bs[s] + (int64_t)h * D * p.kv.vt_ls[s] + k0;
        const int64_t kls = p.kv.k_ls[s], vls = p.kv.vt_ls[s];
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int blk = wave * IPW + i;
            const int krow = blk * 4 + k_r;                       // 0..63
            const int klc = k_lc0 ^ (krow & 15);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(kp + krow * kls + klc * 8),
                                             (LDS_AS void*)(base + blk * 1024), 16, 0, 0);
            const int vrow = blk * 8 + v_r;                       // 0..127 (d)
            const int vlc = v_pc ^ ((vrow >> 1) & 7);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(vp + vrow * vls + vlc * 8),
                                             (LDS_AS void*)(base + VOFF + blk * 1024), 16, 0, 0);
        }
    };
    auto reg_tile = [&](int stage, int s, int64_t k0) {   // ragged tile: zero-filled, synchronous
        const int64_t len = p.kv.len[s];
        const T* kp = (const T*)p.kv.k[s] + b * p.kv.k_bs[s] + (int64_t)h * D;
        const T* vp = (const T*)p.kv.vt[s] + b * p.kv.vt_bs[s] + (int64_t)h * D * p.kv.vt_ls[s];
        const int64_t kls = p.kv.k_ls[s], vls = p.kv.vt_ls[s];
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 1024 / (NW * 64); ++i) {
            const int c = t + NW * 64 * i;
            {
                const int row = c >> 4, ch = c & 15;
                const int64_t key = k0 + row;
                const uint4 v = key < len ? *reinterpret_cast<const uint4*>(kp + key * kls + ch * 8) : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(base + swz_off<256>(row, ch)) = v;
            }
            {
                const int row = c >> 3, ch = c & 7;
                const int64_t key = k0 + ch * 8;
                const T* src = vp + row * vls + key;
                union { uint4 u; T e[8]; } tmp;
                tmp.u = make_uint4(0, 0, 0, 0);
                if (key + 8 <= len) tmp.u = *reinterpret_cast<const uint4*>(src);
                else if (key < len) {
                    for (int j = 0; j < 8; ++j)
                        if (key + j < len) tmp.e[j] = src[j];
                }
                *reinterpret_cast<uint4*>(base + VOFF + swz_off<128>(row, ch)) = tmp.u;
            }
        }
    };

    int it = 0;
    bool cur_dma = false;
    if (seg < p.kv.nseg) {
        cur_dma = key0 + KVB <= p.kv.len[seg];
        if (cur_dma) dma_tile(0, seg, key0);
    }
    // store (or add) this query row's normalised output; also closes a softmax group (kv.new_softmax)
    auto flush = [&](bool add) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        if (p.lse && qvalid && hi == 0) p.lse[((int64_t)b * p.heads + h) * p.Lq + qrow] = m_run + log2f(l_tot);
        if (qvalid) {
            T* op = (T*)p.out + b * p.o_bs + qrow * p.o_ls + (int64_t)h * D + hi * 4;
            // accumulate mode (the image branch of the i2v cross-attention on top of the text branch): ALL sixteen previous quads are
            // requested before the first store.  Interleaved (load, add, store per quad) every load's wait is a vmcnt(0) behind the
            // store in front of it: sixteen serial L2 round trips, ~20 of the 45 us a 13-tile cross-attention workgroup lives.
            bf16x4 prev[4][4];
            if (add) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) prev[d][rq] = *reinterpret_cast<const bf16x4*>(op + d * 32 + rq * 8);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[d][rq * 4 + e] * inv;
                    if (add) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = round_through<T>(v[e]) + (float)prev[d][rq][e];
                    }
                    store4(op + d * 32 + rq * 8, v);
                }
        }
    };
    bool group_add = p.accumulate != 0;
    while (seg < p.kv.nseg) {
        const int stage = it & 1;
        const int64_t cur_len = p.kv.len[seg], cur_k0 = key0;
        if (it > 0 && cur_k0 == 0 && ((p.kv.new_softmax >> seg) & 1)) {
            // a new softmax starts with this segment: park the finished one in the output (its tile stream goes on underneath)
            flush(group_add);
            group_add = true;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
            m_run = -INFINITY; l_run = 0.f;
        }
        if (cur_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else reg_tile(stage, seg, cur_k0);
        __syncthreads();
        // advance + prefetch the next tile by DMA (if it is a full tile)
        int nseg_ = seg;
        int64_t nk0 = key0 + KVB;
        if (nk0 >= cur_len) {
            nk0 = 0;
            ++nseg_;
            while (nseg_ < p.kv.nseg && p.kv.len[nseg_] <= 0) ++nseg_;
        }
        bool next_dma = false;
        if (nseg_ < p.kv.nseg) {
            next_dma = nk0 + KVB <= p.kv.len[nseg_];
            if (next_dma) dma_tile(stage ^ 1, nseg_, nk0);
        }

        // ---- S^T = K Q^T ----  16 (kk, sub) steps; K fragments are read 4 steps ahead from inline asm with counted
        // lgkmcnt (hipcc serialises ds_read -> lgkmcnt(0) -> MFMA here, exposing the LDS latency 16 times per phase)
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
        const unsigned sb = lds0 + stage * STAGE;
        unsigned ka[8], va[4];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ka[kk] = sb + koff[kk];
#pragma unroll
        for (int c = 0; c < 4; ++c) va[c] = sb + voff[c];
        bf16x8 fb0, fb1, fb2, fb3;
#define M4D_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define M4D_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define M4D_QK(B, KK, SUB, OFF, W) do { M4D_LGKM(W); mma32(B, qf[KK], s[SUB]); if ((KK) + 2 < 8) M4D_DSR(B, ka[((KK) + 2) & 7], OFF); } while (0)
        M4D_DSR(fb0, ka[0], 0); M4D_DSR(fb1, ka[0], 8192); M4D_DSR(fb2, ka[1], 0); M4D_DSR(fb3, ka[1], 8192);
        M4D_QK(fb0, 0, 0, 0, 3); M4D_QK(fb1, 0, 1, 8192, 3); M4D_QK(fb2, 1, 0, 0, 3); M4D_QK(fb3, 1, 1, 8192, 3);
        M4D_QK(fb0, 2, 0, 0, 3); M4D_QK(fb1, 2, 1, 8192, 3); M4D_QK(fb2, 3, 0, 0, 3); M4D_QK(fb3, 3, 1, 8192, 3);
        M4D_QK(fb0, 4, 0, 0, 3); M4D_QK(fb1, 4, 1, 8192, 3); M4D_QK(fb2, 5, 0, 0, 3); M4D_QK(fb3, 5, 1, 8192, 3);
        M4D_QK(fb0, 6, 0, 0, 3); M4D_QK(fb1, 6, 1, 8192, 2); M4D_QK(fb2, 7, 0, 0, 1); M4D_QK(fb3, 7, 1, 8192, 0);
        // first four V^T fragments start now and land under the softmax arithmetic
        M4D_DSR(fb0, va[0], 0); M4D_DSR(fb1, va[0], 4096); M4D_DSR(fb2, va[0], 8192); M4D_DSR(fb3, va[0], 12288);
        if (cur_k0 + KVB > cur_len) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t key = cur_k0 + sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= cur_len) s[sub][r] = -INFINITY;
                }
        }
        // ---- online softmax (exp2 domain) ----
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
        {
            const unsigned u = __float_as_uint(mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_run, mx * p.sc);
        if (__any(m_new > m_run)) {          // some row's max moved: rescale (exact: alpha == 1 where it did not)
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        // single-issue v_fma_f32 / v_exp_f32 / v_add_f32 as ONE volatile asm stream in a fixed order: hipcc SLP-packs the
        // plain C into v_pk_*_f32, which costs more than its two scalar halves beside the other waves' MFMAs
        // (MI355X_MICROARCH.md price list); its hazard recogniser does not see through inline asm, so every v_add_f32 trails
        // the v_exp_f32 it consumes by four instructions (gfx940+ trans-use hazard needs one)
        float psa = 0.f, psb = 0.f;
        {
            const float nm = -m_run;
#define M4D_SM_A(SUB, R) do { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s[SUB][R]) : "v"(s[SUB][R]), "s"(p.sc), "v"(nm)); \
                              asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s[SUB][(R) + 1]) : "v"(s[SUB][(R) + 1]), "s"(p.sc), "v"(nm)); } while (0)
#define M4D_SM_B(SUB, R) do { asm volatile("v_exp_f32 %0, %1" : "=v"(s[SUB][R]) : "v"(s[SUB][R])); \
                              asm volatile("v_exp_f32 %0, %1" : "=v"(s[SUB][(R) + 1]) : "v"(s[SUB][(R) + 1])); } while (0)
#define M4D_SM_C(SUB, R) do { asm volatile("v_add_f32 %0, %1, %2" : "=v"(psa) : "v"(psa), "v"(s[SUB][R])); \
                              asm volatile("v_add_f32 %0, %1, %2" : "=v"(psb) : "v"(psb), "v"(s[SUB][(R) + 1])); } while (0)
            M4D_SM_A(0, 0); M4D_SM_B(0, 0);
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                M4D_SM_A(i >> 3, 2 * (i & 7));
                M4D_SM_B(i >> 3, 2 * (i & 7));
                M4D_SM_C((i - 1) >> 3, 2 * ((i - 1) & 7));
            }
            asm volatile("s_nop 1");
            M4D_SM_C(1, 14);
#undef M4D_SM_A
#undef M4D_SM_B
#undef M4D_SM_C
        }
        l_run += psa + psb;
        // ---- O^T += V^T P^T ----  16 (c, d) steps, V^T fragments 4 steps ahead
        bf16x8 pf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) pf[c] = pack8<T>(s[c >> 1], (c & 1) * 8);
        __builtin_amdgcn_sched_barrier(0);
#define M4D_PV(B, C, DD, OFF, W) do { M4D_LGKM(W); mma32(B, pf[C], o[DD]); if ((C) + 1 < 4) M4D_DSR(B, va[((C) + 1) & 3], OFF); } while (0)
        M4D_PV(fb0, 0, 0, 0, 3); M4D_PV(fb1, 0, 1, 4096, 3); M4D_PV(fb2, 0, 2, 8192, 3); M4D_PV(fb3, 0, 3, 12288, 3);
        M4D_PV(fb0, 1, 0, 0, 3); M4D_PV(fb1, 1, 1, 4096, 3); M4D_PV(fb2, 1, 2, 8192, 3); M4D_PV(fb3, 1, 3, 12288, 3);
        M4D_PV(fb0, 2, 0, 0, 3); M4D_PV(fb1, 2, 1, 4096, 3); M4D_PV(fb2, 2, 2, 8192, 3); M4D_PV(fb3, 2, 3, 12288, 3);
        M4D_PV(fb0, 3, 0, 0, 3); M4D_PV(fb1, 3, 1, 4096, 2); M4D_PV(fb2, 3, 2, 8192, 1); M4D_PV(fb3, 3, 3, 12288, 0);
#undef M4D_PV
#undef M4D_QK
#undef M4D_LGKM
#undef M4D_DSR
        seg = nseg_;
        key0 = nk0;
        cur_dma = next_dma;
        ++it;
    }

    flush(group_add);
}

#include "attention_phased.h"
#include "attention_wide.h"
#include "attention_xp.h"
#include "attention_q64.h"

int attn_device_cus() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 64) dev = 0;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n[dev] = v;
        else n[dev] = 256;
    }
    return n[dev];
}
// attn128x_kernel (attention_xp.h) takes bf16 / head_dim 128 calls with at least four query tiles of 256 and a key list too short for the
// long-loop kernels (M4D_ATTN_XP=0: the lock-step 4-wave kernel as before)
bool xp_ok(const AttnArgs& p, bool w8) {
    M4D_ENV_ONCE(xp, "M4D_ATTN_XP", 1);
    M4D_ENV_ONCE(force_lockstep, "M4D_ATTN_LOCKSTEP", 0);
    M4D_ENV_ONCE(force_w4, "M4D_ATTN_W4", 0);
    if (!xp || force_lockstep || force_w4 || w8 || p.Lq <= 1024) return false;
    int64_t keys = 0;
    for (int i = 0; i < p.kv.nseg; ++i) keys += p.kv.len[i] > 0 ? p.kv.len[i] : 0;
    return keys > 0;
}

template <typename T>
int launch(const AttnArgs& p, int D, hipStream_t st) {
    dim3 grid((unsigned)((int64_t)p.nq_tiles * p.heads * p.B)), block(256);
    M4D_ENV_ONCE(force_generic, "M4D_ATTN_GENERIC", 0);
    M4D_ENV_ONCE(force_w4, "M4D_ATTN_W4", 0);
    M4D_ENV_ONCE(force_lockstep, "M4D_ATTN_LOCKSTEP", 0);
    if (p.kv.new_softmax != 0 && !(sizeof(T) == 2 && D == 128)) return -2;      // grouped softmaxes: lockstep bf16 kernels only
    if (sizeof(T) == 2 && D == 128 && (!force_generic || p.kv.new_softmax != 0)) {
        int64_t keys = 0;
        for (int i = 0; i < p.kv.nseg; ++i) keys += p.kv.len[i] > 0 ? p.kv.len[i] : 0;
        // 256-query workgroups (8 waves share each K/V tile) for long self-attention; short key loops (cross-attention)
        // keep the 4-wave workgroups (measured: 1000 vs 924 TF at Lk = 21840, 657 vs 678 TF at Lk = 512)
        M4D_ENV_ONCE(min_keys8, "M4D_ATTN_W8_MIN_KEYS", 2048);     // (A/B: the 8-wave kernels on shorter key loops)
        const bool w8 = p.Lq > 1024 && keys >= min_keys8 && !force_w4;
        AttnArgs q = p;
        bool same_strides = true;     // the phased kernel shares one per-lane DMA offset across segments
        for (int i = 1; i < p.kv.nseg; ++i)
            if (p.kv.len[i] > 0 && (p.kv.k_ls[i] != p.kv.k_ls[0] || p.kv.vt_ls[i] != p.kv.vt_ls[0])) same_strides = false;
        if (p.kv.len[0] <= 0) same_strides = p.kv.nseg == 1;
        // attn128q_kernel (attention_q64.h): one wave per SIMD, 4 x 64 query rows, generated instruction stream.  It folds scale * log2(e)
        // into Q with ONE bf16 rounding per element; a caller that has already folded it into q (scale * log2(e) == 1: the DiT's
        // self-attention folds it into the RMSNorm weight of q) gets q's bits unchanged.  With any other scale the log-sum-exp would
        // not match the unrounded scores the backward recomputes, so lse calls take this kernel only with a folded scale.
        M4D_ENV_ONCE(q64_mode, "M4D_ATTN_Q64", 1);     // A/B: 0 = attn128p_kernel
        const bool folded = fabsf(p.sc - 1.f) < 1e-6f;
        if (folded) q.sc = 1.f;
        int q64_tiles = 0, q64_rag = 0;      // full 64-key tiles of all segments / segments with a ragged tail
        for (int i = 0; i < p.kv.nseg; ++i)
            if (p.kv.len[i] > 0) { q64_tiles += (int)(p.kv.len[i] / 64); q64_rag += (p.kv.len[i] % 64) != 0; }
        // (the A/B switches of the older kernels keep their meaning: M4D_ATTN_LOCKSTEP=1 / M4D_ATTN_WIDE=1 select those kernels — ADVICE r5)
        M4D_ENV_ONCE(wide_req, "M4D_ATTN_WIDE", 0);
        if (w8 && q64_mode && !force_lockstep && !wide_req && same_strides && !p.kv.new_softmax && !p.accumulate && q64_tiles >= 4 && q64_rag <= 5 &&
            (folded || !p.lse) && p.kv.k_ls[0] < (1 << 20) && p.kv.vt_ls[0] < (1 << 23) && p.q_ls < (1 << 20) && p.o_ls < (1 << 20)) {
            static PerDeviceOnce configured_q;
            if (configured_q.pending()) {
                if (hipFuncSetAttribute((const void*)attn128q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768) != hipSuccess) return -3;
                configured_q.mark();
            }
            q.nq_tiles = (int)((p.Lq + 255) / 256);
            m4d_count_launch(M4D_KC_ATTN_Q64);
            hipLaunchKernelGGL(attn128q_kernel, dim3((unsigned)((int64_t)q.nq_tiles * p.heads * p.B)), dim3(256), 5 * 32768, st, q);
            return 0;
        }
        M4D_ENV_ONCE(wide_mode, "M4D_ATTN_WIDE", 0);   // 1: 64-queries-per-wave kernel (attention_wide.h); same-box A/B: phased 1048 vs wide 1020 TF sustained
        if (w8 && same_strides && keys >= 4 * 64 + 64 * p.kv.nseg && wide_mode && !p.kv.new_softmax) {
            // 64 queries per wave: half the LDS traffic per MFMA, softmax interleaved into the MFMA stream (attention_wide.h)
            static PerDeviceOnce configured_w;
            if (configured_w.pending()) {
                if (hipFuncSetAttribute((const void*)attn128w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
#ifdef M4D_ABLATIONS
                if (hipFuncSetAttribute((const void*)attn128w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
                if (hipFuncSetAttribute((const void*)attn128w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
                if (hipFuncSetAttribute((const void*)attn128w_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
#endif
                configured_w.mark();
            }
            q.nq_tiles = (int)((p.Lq + 255) / 256);
            m4d_count_launch(M4D_KC_ATTN_OTHER);
            const dim3 gw((unsigned)((int64_t)q.nq_tiles * p.heads * p.B));
#ifdef M4D_ABLATIONS
            switch (p.abl & 3) {   // timing ablations (tool builds only): 1 no softmax, 2 no MFMAs in the main loop
                case 1: hipLaunchKernelGGL(attn128w_kernel<1>, gw, dim3(256), 4 * 32768, st, q); break;
                case 2: hipLaunchKernelGGL(attn128w_kernel<2>, gw, dim3(256), 4 * 32768, st, q); break;
                case 3: hipLaunchKernelGGL(attn128w_kernel<3>, gw, dim3(256), 4 * 32768, st, q); break;
                default: hipLaunchKernelGGL(attn128w_kernel<0>, gw, dim3(256), 4 * 32768, st, q);
            }
#else
            hipLaunchKernelGGL(attn128w_kernel<0>, gw, dim3(256), 4 * 32768, st, q);
#endif
        } else if (w8 && same_strides && !force_lockstep && !p.kv.new_softmax) {
            // two wave groups half a tile apart: softmax of one under the MFMAs of the other (attention_phased.h)
            static PerDeviceOnce configured;
            if (configured.pending()) {
                if (hipFuncSetAttribute((const void*)attn128p_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
                if (hipFuncSetAttribute((const void*)attn128p_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
                if (hipFuncSetAttribute((const void*)attn128p_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
                if (hipFuncSetAttribute((const void*)attn128p_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 8192) != hipSuccess) return -3;
                configured.mark();
            }
            q.nq_tiles = (int)((p.Lq + 255) / 256);
            M4D_ENV_ONCE(smx, "M4D_ATTN_SMX", 1);   // 1 = scalar softmax arithmetic (default: +9 % sustained over the packed form), 0 = packed
            M4D_ENV_ONCE(prio, "M4D_ATTN_PRIO", 1);
            const dim3 gp((unsigned)((int64_t)q.nq_tiles * p.heads * p.B));
            m4d_count_launch(M4D_KC_ATTN_PHASED);
            if (smx == 0) hipLaunchKernelGGL((attn128p_kernel<0, 1>), gp, dim3(512), 4 * 32768 + ((M4D_ABL(q) & 128) ? 8192 : 0), st, q);
            else if (prio == 0) hipLaunchKernelGGL((attn128p_kernel<1, 0>), gp, dim3(512), 4 * 32768 + ((M4D_ABL(q) & 128) ? 8192 : 0), st, q);
            else if (prio == 2) hipLaunchKernelGGL((attn128p_kernel<1, 2>), gp, dim3(512), 4 * 32768 + ((M4D_ABL(q) & 128) ? 8192 : 0), st, q);
            else hipLaunchKernelGGL((attn128p_kernel<1, 1>), gp, dim3(512), 4 * 32768 + ((M4D_ABL(q) & 128) ? 8192 : 0), st, q);
        } else if (xp_ok(p, w8)) {
            // short key lists against many queries (cross-attention): one persistent workgroup per CU walks (query tile, key tile) pairs
            static PerDeviceOnce configured_x;
            if (configured_x.pending()) {
                if (hipFuncSetAttribute((const void*)attn128x_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 32768) != hipSuccess) return -3;
                if (hipFuncSetAttribute((const void*)attn128x_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 32768) != hipSuccess) return -3;
                if (hipFuncSetAttribute((const void*)attn128x_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768 + 32768) != hipSuccess) return -3;
                configured_x.mark();
            }
            q.nq_tiles = (int)((p.Lq + 255) / 256);
            const int64_t items = (int64_t)q.nq_tiles * p.heads * p.B;
            int nwg = attn_device_cus();
            if (((p.heads * p.B) & 7) == 0) nwg &= ~7;
            if (nwg < 8) nwg = 8;
            if (items < nwg && ((p.heads * p.B) & 7) != 0) nwg = (int)items;
            m4d_count_launch(M4D_KC_ATTN_XP);
            M4D_ENV_ONCE(xprio, "M4D_ATTN_XP_PRIO", 1);      // A/B: 0 no s_setprio, 1 raised around every MFMA stream, 2 static (younger group)
            if (xprio == 2) hipLaunchKernelGGL((attn128x_kernel<2>), dim3((unsigned)nwg), dim3(512), 4 * 32768 + 32768, st, q);
            else if (xprio == 0) hipLaunchKernelGGL((attn128x_kernel<0>), dim3((unsigned)nwg), dim3(512), 4 * 32768 + 32768, st, q);
            else hipLaunchKernelGGL((attn128x_kernel<1>), dim3((unsigned)nwg), dim3(512), 4 * 32768 + 32768, st, q);
        } else if (w8) {
            q.nq_tiles = (int)((p.Lq + 255) / 256);
            m4d_count_launch(M4D_KC_ATTN_OTHER);
            hipLaunchKernelGGL(attn128_kernel<8>, dim3((unsigned)((int64_t)q.nq_tiles * p.heads * p.B)), dim3(512), 0, st, q);
        } else {
            m4d_count_launch(M4D_KC_ATTN_OTHER);
            hipLaunchKernelGGL(attn128_kernel<4>, grid, block, 0, st, q);
        }
        return 0;
    }
    m4d_count_launch(M4D_KC_ATTN_OTHER);
    switch (D) {
        case 32: hipLaunchKernelGGL((attn_kernel<T, 32>), grid, block, 0, st, p); break;
        case 64: hipLaunchKernelGGL((attn_kernel<T, 64>), grid, block, 0, st, p); break;
        case 128: hipLaunchKernelGGL((attn_kernel<T, 128>), grid, block, 0, st, p); break;
        default: return -2;
    }
    return 0;
}

}  // namespace

static int attention_impl(m4d_dtype dt, const void* q, int64_t q_bs, int64_t q_ls, const m4d_kv_segs* kv, void* out,
                          int64_t o_bs, int64_t o_ls, int B, int64_t Lq, int heads, int head_dim, float scale,
                          int accumulate, float* lse, m4d_stream stream) {
    M4D_CHECK_ARG(dt == M4D_BF16 || dt == M4D_F32, "attention: bad dtype %d", (int)dt);
    M4D_CHECK_ARG(q && kv && out, "attention: null pointer");
    M4D_CHECK_ARG(B > 0 && Lq > 0 && heads > 0, "attention: empty problem");
    M4D_CHECK_ARG(kv->nseg >= 1 && kv->nseg <= M4D_MAX_KV_SEGS, "attention: nseg=%d out of range", kv->nseg);
    if (!(head_dim == 32 || head_dim == 64 || head_dim == 128)) {
        m4d_set_error("attention: unsupported head_dim %d (32, 64, 128)", head_dim);
        return -2;
    }
    int64_t total = 0;
    for (int s = 0; s < kv->nseg; ++s) {
        M4D_CHECK_ARG(kv->len[s] >= 0, "attention: negative segment length");
        if (kv->len[s] == 0) continue;
        M4D_CHECK_ARG(kv->k[s] && kv->vt[s], "attention: null K/V segment %d", s);
        M4D_CHECK_ARG(kv->k_ls[s] % 8 == 0 && kv->vt_ls[s] % 8 == 0 && kv->k_bs[s] % 8 == 0 && kv->vt_bs[s] % 8 == 0,
                      "attention: K/V strides must be multiples of 8 elements (segment %d)", s);
        M4D_CHECK_ARG(((uintptr_t)kv->k[s] % 16) == 0 && ((uintptr_t)kv->vt[s] % 16) == 0, "attention: K/V must be 16-byte aligned");
        total += kv->len[s];
    }
    M4D_CHECK_ARG(total > 0, "attention: no keys");
    M4D_CHECK_ARG(kv->new_softmax == 0 || (lse == nullptr && (kv->new_softmax & 1) == 0 && kv->new_softmax < (1 << kv->nseg)),
                  "attention: new_softmax marks segments 1.. of a call without lse");
    M4D_CHECK_ARG(q_ls % 8 == 0 && q_bs % 8 == 0 && o_ls % 8 == 0 && o_bs % 8 == 0, "attention: q/out strides must be multiples of 8 elements");
    M4D_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)out % 16) == 0, "attention: q/out must be 16-byte aligned");
    AttnArgs p;
    p.q = q; p.out = out; p.kv = *kv;
    p.q_bs = q_bs; p.q_ls = q_ls; p.o_bs = o_bs; p.o_ls = o_ls; p.Lq = Lq;
    p.B = B; p.heads = heads; p.nq_tiles = (int)((Lq + 127) / 128); p.accumulate = accumulate;
    p.sc = scale * 1.4426950408889634f;
    p.lse = lse;
    p.abl = 0; p.dbg = nullptr;
#ifdef M4D_ABLATIONS
    { M4D_ENV_ONCE(abl, "M4D_ATTN_ABL", 0); p.abl = abl; }
    { static unsigned long long* dbgp = nullptr; static bool rd = false;
      if (!rd) { rd = true; const char* v = getenv("M4D_ATTN_DBG_PTR"); if (v) dbgp = (unsigned long long*)strtoull(v, nullptr, 0); }
      p.dbg = dbgp; }
#endif
    int rc = dt == M4D_BF16 ? launch<bf16_t>(p, head_dim, (hipStream_t)stream) : launch<float>(p, head_dim, (hipStream_t)stream);
    if (rc) { m4d_set_error("attention: unsupported configuration"); return rc; }
    M4D_CHECK_LAUNCH("attention");
    return 0;
}

extern "C" int m4d_attention(m4d_dtype dt, const void* q, int64_t q_bs, int64_t q_ls, const m4d_kv_segs* kv, void* out,
                             int64_t o_bs, int64_t o_ls, int B, int64_t Lq, int heads, int head_dim, float scale,
                             int accumulate, m4d_stream stream) {
    return attention_impl(dt, q, q_bs, q_ls, kv, out, o_bs, o_ls, B, Lq, heads, head_dim, scale, accumulate, nullptr, stream);
}

extern "C" int m4d_attention_lse(m4d_dtype dt, const void* q, int64_t q_bs, int64_t q_ls, const m4d_kv_segs* kv, void* out,
                                 int64_t o_bs, int64_t o_ls, int B, int64_t Lq, int heads, int head_dim, float scale,
                                 int accumulate, float* lse, m4d_stream stream) {
    M4D_CHECK_ARG(lse, "attention_lse: null lse");
    return attention_impl(dt, q, q_bs, q_ls, kv, out, o_bs, o_ls, B, Lq, heads, head_dim, scale, accumulate, lse, stream);
}
