// gemm_bt256w_kernel: 256x256 tile, FOUR waves (2 x 2, wave tile 128(m) x 128(n)), ONE wave per SIMD, K-tile 64.
//
// Why a second production structure next to gemm_bt256p_kernel (gemm_phased.h): at the board's power limit the phased kernel
// gains from fewer bytes per MFMA, not from schedule changes.  With 128 x 128 per wave every LDS fragment
// is used by FOUR MFMAs instead of two / four (8 fragment reads per 16 MFMAs = 0.5 reads per MFMA instead of 0.75), and a wave
// that owns its SIMD has 512 registers: the 128 x 128 fp32 accumulators (256 registers) live in AGPRs, four fragment sets
// (128 VGPRs) in VGPRs.  There is no partner wave to fill bubbles, so the wave overlaps its own work: MFMAs are asynchronous
// (16 passes each), the LDS reads / DMA issues / waits / the barrier of a phase sit in the issue shadow of the MFMAs.
//
// Staging units (8 KiB = 64 rows x 128 B).  Wave (wm, wn) reads activation units A0 = rows wm*128 + [0,64), A1 = +64 and weight
// units B0 = rows wn*128 + [0,64), B1 = +64.  A K-tile is four 16 KiB GROUPS, each the same-named unit of both wave rows / columns
// ({A0 of wm 0, A0 of wm 1}, ...); a group is staged by 4 DMA instructions per lane (16 per K-tile) and read in exactly ONE phase.
// Phase plan of K-tile k (FA0 / FA1 / FBx / FBy = fragment sets of 2 blocks x 4 k-steps; FBx / FBy swap roles every K-tile):
//   Q1: MFMA A0 x B0 (16)    read B1(k)   -> FBy       stage the group read 7 phases later
//   Q2: MFMA A0 x B1         read A1(k)   -> FA1
//   Q3: MFMA A1 x B1         read A0(k+1) -> FA0
//   Q4: MFMA A1 x B0         read B0(k+1) -> FBy' (= this tile's B1 set, dead after Q3)
// Two K-tile buffers (128 KiB).  Every phase ends  lgkmcnt(0) . vmcnt(24) . s_barrier : the group read in the NEXT phase was
// staged 7 phases ago (six younger groups x 4 instructions may stay in flight), and the group just read is re-staged right after
// the barrier (WAR).  The barrier itself is issued after the first MFMA of the next phase, so its latency and the waves' skew hide
// behind that MFMA's 16 passes.
#pragma once
#include "gemm_common.h"

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // (native vector type: inline-asm register operands)
constexpr int BM2 = 256, BN2 = 256;
extern __shared__ __attribute__((aligned(16))) char dyn_smem[];

constexpr int W_GRP = 16384, W_BUF = 4 * W_GRP;      // group order inside a K-tile buffer: A0, A1, B0, B1
constexpr int W_A0 = 0, W_A1 = W_GRP, W_B0 = 2 * W_GRP, W_B1 = 3 * W_GRP;


// Epilogue of one 64(m) x 64(n) block of a wave's sub-tile, lean form: the fp32 accumulators go to a wave-private 16 KiB LDS region
// exactly as they are (accumulator layout: lane = output ROW m, 4 consecutive n per register quad), and are read back ROW-WISE:
// lane -> (row it*8 + (lane >> 3), columns (lane & 7)*8 + [0, 8)).  Bias, activation, rounding and the gated-residual update run
// in that second layout, where a lane's columns are the same in every iteration (bias: one 8-element load per block, no
// per-element branch, no dependent load in front of a store) and 8 lanes cover one contiguous 128-byte (bf16) / 256-byte (fp32)
// row segment.  EPI is a template parameter: the arithmetic is the shared epilogue's (gemm_common.h: epilogue_half_lds), in the
// same order, so the results are bit-identical.
// INNER (gated residual only): the tile is not shifted into a neighbour (every row / column of it is this tile's to write) — the residual
// loads and stores carry no lane predicates.  Predicated, each of them sits in an exec-masked basic block, hipcc cannot count what is in
// flight across them and drains the queue (s_waitcnt vmcnt(0)) in front of every block's first update: the residual requests and stores
// of the previous block's last iterations are waited for four times per tile and wave.  Straight-line it waits for the two loads it needs.
template <typename T, int EPI, bool INNER = false>
M4D_DEV void epilogue_block64(const GemmArgs& p, char* wl, const f32x16& a00, const f32x16& a01, const f32x16& a10, const f32x16& a11,
                              int64_t m_base, int64_t n_base, int64_t m_lo, int64_t n_lo, int lane,
                              f32x4 (&r0)[8], f32x4 (&r1)[8], bool first, bool has_next, int64_t m_next, int64_t n_next) {
    // a[ni][mi2]: a00 = (ni 0, mi2 0), a01 = (ni 0, mi2 1), a10 = (ni 1, mi2 0), a11 = (ni 1, mi2 1)
    const int li = lane & 31, hi = lane >> 5;
    const int c8 = lane & 7;
    constexpr bool F32OUT = EPI == M4D_EPI_RESID_GATE || EPI == M4D_EPI_STORE_F32;
    // the lane's two 4-column chunks of a row: bf16 outputs take 8 consecutive columns (one 16-byte store); fp32 outputs take chunk c8
    // and chunk 8 + c8, so that the eight lanes of a row cover 128 CONTIGUOUS bytes per load / store instruction instead of 16 of every 32
    constexpr int CA_MUL = F32OUT ? 1 : 2, CB_OFF = F32OUT ? 8 : 1;
    const int cA = c8 * CA_MUL, cB = cA + CB_OFF;
    const int64_t nb = n_base + cA * 4, nb2 = n_base + cB * 4;
    // Gated residual: written as "v += load(dst); store(dst, v)" per iteration, every load sat behind the previous iteration's store to the same
    // array (may alias: hipcc keeps the order) and a block paid eight dependent HBM / L2 round trips — ~30 us of epilogue per 256 x 256 tile
    // against 7 us for the bf16 store (the K = 5120 gated-residual GEMMs ran 23 % behind the q / k / v projections of the same size).
    // Now a rolling window of eight iterations' residual values (r0 / r1, owned by the caller): the tile's first block requests all eight
    // before its accumulators go through LDS, and every iteration re-fills its slot with the same iteration of the NEXT block right after its
    // own store — a load has a whole block (~1.5 us) to arrive.
    // INNER addressing: scalar base of (block, iteration) + two per-lane 32-bit offsets for the whole block (row lane >> 3, the lane's two
    // chunks) — per-iteration 64-bit row pointers cost the 512-register kernel spills
    auto uni64 = [](int64_t v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi2 = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
        return (int64_t)(((unsigned long long)hi2 << 32) | lo);
    };
    const unsigned ldc4 = (unsigned)p.ldc * 4u;
    const unsigned voffA = (unsigned)(lane >> 3) * ldc4 + (unsigned)cA * 16u, voffB = (unsigned)(lane >> 3) * ldc4 + (unsigned)cB * 16u;
    auto fetch = [&](int it, int64_t mb, int64_t nbase) {
        if constexpr (INNER) {
            const char* bi = (const char*)p.out + (uni64(mb) * p.ldc + uni64(nbase)) * 4 + (size_t)it * 8 * ldc4;
            r0[it] = *reinterpret_cast<const f32x4*>(bi + voffA);
            r1[it] = *reinterpret_cast<const f32x4*>(bi + voffB);
            return;
        }
        const int64_t m = mb + it * 8 + (lane >> 3);
        const float* src = (const float*)p.out + m * p.ldc + nbase;
        if constexpr (INNER) { r0[it] = load4(src + cA * 4); r1[it] = load4(src + cB * 4); }
        else {
            r0[it] = (m >= m_lo && nbase + cA * 4 >= n_lo) ? load4(src + cA * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            r1[it] = (m >= m_lo && nbase + cB * 4 >= n_lo) ? load4(src + cB * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = {1.f, 1.f, 1.f, 1.f};
    bool one_gate = false;
    if constexpr (EPI == M4D_EPI_RESID_GATE) {
        if (first) {
#pragma unroll
            for (int it = 0; it < 8; ++it) fetch(it, m_base, n_base);
        }
        one_gate = p.gate && m_base / p.rows_per_sample == (m_base + 63) / p.rows_per_sample;      // (wave-uniform) the block lies in one sample
        if constexpr (INNER) one_gate = __builtin_amdgcn_readfirstlane((int)one_gate) != 0;        // ... and the compiler is told so
        if (one_gate) {
            const float* grow = p.gate + (m_base / p.rows_per_sample) * p.gate_stride;
            g0 = load4(grow + nb); g1 = load4(grow + nb2);
        }
    }
#pragma unroll
    for (int mi2 = 0; mi2 < 2; ++mi2) {
        const int r = mi2 * 32 + li;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const f32x16& acc = ni == 0 ? (mi2 == 0 ? a00 : a01) : (mi2 == 0 ? a10 : a11);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int ch = (ni * 32 + rq * 8 + hi * 4) >> 2;               // 16 chunks of 4 floats per 256-byte row
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[rq * 4 + e];
                *reinterpret_cast<f32x4*>(wl + r * 256 + ((ch ^ (r & 15)) << 4)) = v;
            }
        }
    }
    // wave-private region: program order + the compiler's lgkmcnt wait order the reads after the writes
    const T* bias = (const T*)p.bias;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if (bias && !p.bias_on_m) { b0 = load4(bias + nb); b1 = load4(bias + nb2); }
    // bias along m (the V^T projection): the eight rows' values up front — loaded inside the iterations each one queued behind the previous
    // iteration's store (same may-alias order as the residual above): that GEMM ran 41 % behind the q / k projections of the same FLOPs
    float bm8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias && p.bias_on_m) {
#pragma unroll
        for (int it = 0; it < 8; ++it) bm8[it] = (float)bias[m_base + it * 8 + (lane >> 3)];
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + (lane >> 3);
        const int64_t m = m_base + r;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(wl + r * 256 + ((cA ^ (r & 15)) << 4));
        f32x4 v1 = *reinterpret_cast<const f32x4*>(wl + r * 256 + ((cB ^ (r & 15)) << 4));
        if (bias) {
            if (p.bias_on_m) { const float bm = bm8[it]; v0 += bm; v1 += bm; }
            else { v0 += b0; v1 += b1; }
        }
        if constexpr (EPI == M4D_EPI_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = gelu_tanh_f(v0[e]); v1[e] = gelu_tanh_f(v1[e]); }
        }
        if constexpr (F32OUT) {
            if (EPI != M4D_EPI_STORE_F32 || !p.nb1) {      // (batched split-K partial sums keep their float32 accumulators)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = round_through<T>(v0[e]); v1[e] = round_through<T>(v1[e]); }
            }
            float* drow = (float*)p.out + m * p.ldc;
            if constexpr (EPI == M4D_EPI_RESID_GATE) {
                if (p.gate) {
                    if (one_gate) { v0 = v0 * g0; v1 = v1 * g1; }
                    else {
                        const float* grow = p.gate + (m / p.rows_per_sample) * p.gate_stride;
                        v0 = v0 * load4(grow + nb); v1 = v1 * load4(grow + nb2);
                    }
                }
                if constexpr (INNER) {
                    char* bi = (char*)p.out + (uni64(m_base) * p.ldc + uni64(n_base)) * 4 + (size_t)it * 8 * ldc4;
                    v0 += r0[it]; *reinterpret_cast<f32x4*>(bi + voffA) = v0;
                    v1 += r1[it]; *reinterpret_cast<f32x4*>(bi + voffB) = v1;
                } else {
                    if (m >= m_lo && nb >= n_lo) { v0 += r0[it]; store4(drow + nb, v0); }
                    if (m >= m_lo && nb2 >= n_lo) { v1 += r1[it]; store4(drow + nb2, v1); }
                }
                if (has_next) fetch(it, m_next, n_next);
            } else {
                if (m >= m_lo && nb >= n_lo) store4(drow + nb, v0);
                if (m >= m_lo && nb2 >= n_lo) store4(drow + nb2, v1);
            }
        } else {
            union { uint4 u; bf16x4 h[2]; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o.h[0][e] = (bf16_t)v0[e]; o.h[1][e] = (bf16_t)v1[e]; }
            // a chunk straddling n_lo rewrites the neighbour's identical values (same K order): benign
            if (m >= m_lo && nb + 8 > n_lo) *reinterpret_cast<uint4*>((T*)p.out + m * p.ldc + nb) = o.u;
        }
    }
}

// Persistent kernel: one 32(m) x 64(n) block (8 KiB of wave-private LDS next to the two live K-tile buffers), bf16 outputs, the bias
// of the block's columns already in registers.  Same arithmetic and order as epilogue_block64.  Stores go out as
// `global_store_dwordx4 voff, data, sbase` with ONE per-lane 32-bit offset for the whole kernel (row (lane >> 3), 8 columns (lane & 7))
// and a scalar base per (block, iteration): written as plain pointer arithmetic hipcc hoists the 32 per-lane 64-bit addresses of a
// tile's stores out of the tile loop and spills inside the K loop.
template <typename T, int EPI>
M4D_DEV void epilogue_block32(const GemmArgs& p, char* wl, const f32x16& a0, const f32x16& a1, int64_t m_base, int64_t n_base,
                              int lane, unsigned voff, bool has_bias, const u32x4& bias8, bool bias_m, const f32x4& bm) {
    static_assert(EPI == M4D_EPI_STORE || EPI == M4D_EPI_GELU_TANH, "bf16 outputs only");
    f32x4 b0, b1;
    {
        const bf16_t* bb = reinterpret_cast<const bf16_t*>(&bias8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { b0[e] = (float)bb[e]; b1[e] = (float)bb[4 + e]; }
    }
    const int li = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const f32x16& acc = ni == 0 ? a0 : a1;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int ch = (ni * 32 + rq * 8 + hi * 4) >> 2;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[rq * 4 + e];
            *reinterpret_cast<f32x4*>(wl + li * 256 + ((ch ^ (li & 15)) << 4)) = v;
        }
    }
    const int c8 = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3);
        f32x4 v0 = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((2 * c8) ^ (r & 15)) << 4));
        f32x4 v1 = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((2 * c8 + 1) ^ (r & 15)) << 4));
        if (bias_m) { v0 += bm[it]; v1 += bm[it]; }      // bias along m (the V^T projection): this iteration's row, wave-uniform branch
        else if (has_bias) { v0 += b0; v1 += b1; }
        if constexpr (EPI == M4D_EPI_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = gelu_tanh_f(v0[e]); v1[e] = gelu_tanh_f(v1[e]); }
        }
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ob[e] = (bf16_t)v0[e]; ob[4 + e] = (bf16_t)v1[e]; }
        const u32x4 ov = __builtin_bit_cast(u32x4, ob);
        const char* sb = uniform_ptr((const char*)p.out + ((m_base + it * 8) * p.ldc + n_base) * 2);
        // NO edge mask: a tile shifted inwards rewrites rows / columns of its neighbour with the neighbour's own values (same K order,
        // same instruction sequence per element: bit-identical), so every tile issues exactly 4 stores per block — the phases
        // after the epilogue count them
        // (s_nop 1: a store of more than 8 bytes reads its data registers up to two wait states after issue and the hardware does not
        // interlock; hipcc's hazard recognizer, which pads its own stores, does not look inside an asm statement — without the
        // nops it re-used a data register for an LDS address in the very next instruction and the address went out as data)
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(ov), "s"(sb) : "memory");
    }
}

// ABL: compile-time timing ablations (tool builds only; results wrong when != 0): 1 no DMA, 2 no fragment reads, 4 no barriers, 8 no MFMA,
// 64 = correct kernel + per-workgroup timestamps written over the first output row of every tile (tools/gemm_timeline.py)
// PERSIST: the workgroup walks tiles blockIdx.x, + gridDim.x, ... of the XCD-aware order and the DMA stream runs on from one tile into the
// next: a staging group whose K-tile counter reaches K/64 wraps to K-tile 0 of the workgroup's NEXT tile (the tile origin lives in the
// per-group VGPR offset, the scalar row bases are tile independent), so the next tile's first two K-tiles are in LDS and its first
// fragments in registers when the epilogue of the current one ends — no prologue, no launch gap.  The epilogue runs beside the live
// ring in 32 KiB of extra LDS (8 KiB per wave) and only ISSUES its stores (every tile issues all 32: edge tiles rewrite their
// neighbour's identical values instead of masking).
// Requirements (host-checked): K/64 even and >= 4, both operands below 4 GiB, bf16 output (bias along n or along m).
template <int ABL, int EPI, bool PERSIST = false>
__global__ __launch_bounds__(256, 1) void gemm_bt256w_kernel(GemmArgs p) {
    typedef bf16_t T;
    static_assert(!PERSIST || ((EPI == M4D_EPI_STORE || EPI == M4D_EPI_GELU_TANH) && ABL == 0), "persistent form: bf16 epilogues, no ablations");
    unsigned long long ts[6];
    if constexpr (ABL & 64) { ts[0] = __builtin_readcyclecounter(); ts[5] = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (!PERSIST && EPI == M4D_EPI_STORE_F32) {
        // batched launch (m4d_gemm_bt_taps: the K-slices of a conv weight gradient): blockIdx.y moves both operands along K and the
        // float32 output to its own [M, ldc] slab
        if (p.nb1) {
            p.A = (const char*)p.A + (int64_t)blockIdx.y * p.a_bs1 * 2;
            p.W = (const char*)p.W + (int64_t)blockIdx.y * p.w_bs1 * 2;
            p.out = (float*)p.out + (int64_t)blockIdx.y * p.M * p.ldc;
        }
    }
    int tm, tn;
    int bid = blockIdx.x;
    tile_coords(p, tm, tn, bid);
    int64_t m_lo = (int64_t)tm * BM2, n_lo = (int64_t)tn * BN2;
    int64_t m0 = min(m_lo, p.M - BM2), n0 = min(n_lo, p.N - BN2);     // edge tiles shifted inwards (see gemm_phased.h)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // scalar: everything derived from (tile, wave) stays in SGPRs
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS char*)dyn_smem;

    // ---- DMA: instruction i of a group covers group rows i*32 + wave*8 + (lane >> 3); group row u = unit (u >> 6), row (u & 63)
    // of the unit = operand row (u >> 6)*128 + half*64 + (u & 63).  The per-lane part of the source offset is the same for all
    // instructions of an operand (one VGPR each); everything else is wave-uniform and goes into the scalar base.
    const int lrow = lane >> 3, pc = lane & 7;
    unsigned oa, ow;
    {
        const int u = wave * 8 + lrow;
        const int lc = pc ^ ((u >> 1) & 7);
        oa = (unsigned)((u * p.lda + lc * 8) * 2);
        ow = (unsigned)((u * p.ldw + lc * 8) * 2);
    }
    const int nk = (int)(p.K / 64);
    // scalar row bases: group instruction i of half h reads operand rows (i >> 1)*128 + h*64 + (i & 1)*32 + (lane's row); the
    // K position of a staging is carried by the per-group VGPR offset (advanced by one K-tile = 128 B after every use), so a DMA
    // is two instructions: s_add m0 + global_load_lds
    const char* ra[8];
    const char* rw[8];
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) {
        const int row = (jx >> 2) * 128 + ((jx >> 1) & 1) * 64 + (jx & 1) * 32;     // jx = unit*4 + half*2 + (i & 1)
        int64_t tap = 0;
        if constexpr (!PERSIST && EPI == M4D_EPI_STORE_F32) {
            // stacked taps: rows [t * tap_rows, (t + 1) * tap_rows) are the SAME tap_rows rows of A, shifted along K by tap (dt, dh) = (t / tap_kh,
            // t % tap_kh); tap_rows is a multiple of 32, so a 32-row staging block never straddles two taps
            if (p.tap_rows) {
                const int64_t r = m0 + row;
                const int t_ = (int)(r / p.tap_rows);
                const int dt_ = t_ / p.tap_kh, dh_ = t_ - dt_ * p.tap_kh;
                tap = (dt_ * p.tap_s1 + dh_ * p.tap_s2 - (int64_t)t_ * p.tap_rows * p.lda) * 2;
            }
        }
        ra[jx] = uniform_ptr((const char*)p.A + ((PERSIST ? 0 : m0) + row) * p.lda * 2 + tap);
        rw[jx] = uniform_ptr((const char*)p.W + ((PERSIST ? 0 : n0) + row) * p.ldw * 2);
    }
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
    if constexpr (PERSIST) { oa += (unsigned)(m0 * p.lda * 2); ow += (unsigned)(n0 * p.ldw * 2); }      // (operands below 4 GiB)
    unsigned va0 = oa, va1 = oa, vb0 = ow, vb1 = ow;        // + K offset of the NEXT staging of the group
    int ka0 = 0, ka1 = 0, kb0 = 0, kb1 = 0;                  // K-tile index of that staging (scalar; clamps the advance at nk - 1)
    // PERSIST: what a group's offset moves by when it wraps from the last K-tile of this tile to K-tile 0 of the next one
    unsigned d_a = 0, d_w = 0;
    const int nwg_all = p.tiles_m * p.tiles_n;
    auto next_deltas = [&]() {
        const int nb_ = bid + (int)gridDim.x;
        int64_t m0n = m0, n0n = n0;
        if (nb_ < nwg_all) {
            int tm2, tn2;
            tile_coords(p, tm2, tn2, nb_);
            m0n = min((int64_t)tm2 * BM2, p.M - BM2); n0n = min((int64_t)tn2 * BN2, p.N - BN2);
        }
        d_a = (unsigned)((m0n - m0) * p.lda * 2 - (int64_t)(nk - 1) * 128);
        d_w = (unsigned)((n0n - n0) * p.ldw * 2 - (int64_t)(nk - 1) * 128);
    };
    if constexpr (PERSIST) next_deltas();
#define W_GLDS(VOFF, SRC, DSTOFF)                                                                            \
    if constexpr (!(ABL & 1)) asm volatile("s_add_u32 m0, %0, %3\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsw), "v"(VOFF), "s"(SRC), "i"(DSTOFF) : "memory", "m0", "scc")
    // instruction I (0..3) of the group (operand rows R, half H, group offset G) for the K-tile buffer BUFB (0 / W_BUF)
#define W_STG_I(R, VOFF, H, G, BUFB, I) W_GLDS(VOFF, R[((I) >> 1) * 4 + (H) * 2 + ((I) & 1)], (BUFB) + (G) + (I) * 4096)
#define W_ADV(VOFF, KC, DLT)                                                                                 \
    do {                                                                                                     \
        KC += 1;                                                                                             \
        if constexpr (PERSIST) { const bool wrap_ = KC == nk; VOFF += wrap_ ? DLT : 128u; KC = wrap_ ? 0 : KC; }  \
        else VOFF += (KC < nk) ? 128u : 0u;                                                                  \
    } while (0)
    f32x16 acc[4][4];   // [ni][mi]: rows n, column m = lane (W is the MFMA's A operand, see gemm.hip)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-lane fragment addresses (block 0 of the wave's unit; block 1 = +4096) in K-tile buffer 0 and 1
    // (ONE address set: Q1 / Q2 read the current K-tile buffer, Q3 / Q4 the next one, so the set moves by +-W_BUF after Q2 and stays
    // there for the next K-tile's Q1 / Q2: eight v_add per K-tile in MFMA shadows instead of eight more live registers)
    unsigned am0[4], an0[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned x = li * ROWB + (((kk * 2 + hi) ^ ((li >> 1) & 7)) << 4);
        am0[kk] = lds_base + wm * 8192 + x;
        an0[kk] = lds_base + wn * 8192 + x;
    }
    bf16x8 fa0[2][4], fa1[2][4], fbx[2][4], fby[2][4];
#define W_DSR(dst, addr, OFF) if constexpr (!(ABL & 2)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define W_PIN() __builtin_amdgcn_sched_barrier(0)
#define W_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define W_VM_(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define W_VM(N) W_VM_(N)
    // read one unit (2 blocks x 4 k-steps) from group offset G of the buffer addressed by AD
#define W_READ(F, AD, G)                                                                                   \
    do {                                                                                                   \
        W_DSR(F[0][0], AD[0], G); W_DSR(F[1][0], AD[0], G + 4096);                                          \
        W_DSR(F[0][1], AD[1], G); W_DSR(F[1][1], AD[1], G + 4096);                                          \
        W_DSR(F[0][2], AD[2], G); W_DSR(F[1][2], AD[2], G + 4096);                                          \
        W_DSR(F[0][3], AD[3], G); W_DSR(F[1][3], AD[3], G + 4096);                                          \
    } while (0)
    // (PERSIST: the MFMA builtin is a pure operation to hipcc, which in the larger function sinks whole phases of them past the asm
    // statements that make up the schedule; an empty asm that "modifies" the accumulator block right after the MFMA ties it to its slot)
#define W_MMA(FN, FM, NB, MB, KK, NI, MI)                                                                  \
    do {                                                                                                   \
        if constexpr (!(ABL & 8)) mma32(FN[NI][KK], FM[MI][KK], acc[(NB) + (NI)][(MB) + (MI)]);             \
        if constexpr (PERSIST) asm volatile("" : "+a"(acc[(NB) + (NI)][(MB) + (MI)]));                      \
    } while (0)
    // One phase: 16 MFMAs (FN x FM into the accumulator quadrant NB, MB); ONE other instruction in the issue shadow of each MFMA
    // (an MFMA holds the pipe for 8 passes = 32 cycles): after the first MFMA the wait for the group read in this phase and the
    // barrier, then the 8 fragment reads of the phase, then the 4 DMA instructions of the group that was read in the last phase.
#define W_PHASE(VMN, FN, FM, NB, MB, RF, RAD, RG, SR, SV, SK, SH, SG, SB, SD)                                   \
    do {                                                                                                   \
        W_LGKM0(); W_PIN();                                                                                \
        W_MMA(FN, FM, NB, MB, 0, 0, 0); W_PIN();                                                           \
        W_VM(VMN);                                                                                         \
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                               \
        W_PIN();                                                                                           \
        W_MMA(FN, FM, NB, MB, 0, 0, 1); W_PIN(); W_DSR(RF[0][0], RAD[0], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 0, 1, 0); W_PIN(); W_DSR(RF[1][0], RAD[0], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 0, 1, 1); W_PIN(); W_DSR(RF[0][1], RAD[1], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 1, 0, 0); W_PIN(); W_DSR(RF[1][1], RAD[1], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 1, 0, 1); W_PIN(); W_DSR(RF[0][2], RAD[2], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 1, 1, 0); W_PIN(); W_DSR(RF[1][2], RAD[2], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 1, 1, 1); W_PIN(); W_DSR(RF[0][3], RAD[3], RG); W_PIN();                       \
        W_MMA(FN, FM, NB, MB, 2, 0, 0); W_PIN(); W_DSR(RF[1][3], RAD[3], RG + 4096); W_PIN();                \
        W_MMA(FN, FM, NB, MB, 2, 0, 1); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 0); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 2, 1, 0); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 1); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 2, 1, 1); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 2); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 3, 0, 0); W_PIN(); W_STG_I(SR, SV, SH, SG, SB, 3); W_PIN();                    \
        W_MMA(FN, FM, NB, MB, 3, 0, 1); W_PIN(); W_ADV(SV, SK, SD); W_PIN();                                     \
        W_MMA(FN, FM, NB, MB, 3, 1, 0); W_MMA(FN, FM, NB, MB, 3, 1, 1); W_PIN();                            \
    } while (0)
    // K-tile KT in buffer (AMC, ANC) = byte offset BC, next K-tile's buffer (AMN, ANN) = BN; FX holds B0(KT) on entry, FY receives
    // B1(KT) and then B0(KT + 1).  Staging issued in phase h = the group read in phase h + 7, into the slot of the group read in
    // phase h - 1: Q1 of tile k is phase 4k -> B0(k + 2) (read in phase 4(k + 2) - 1); Q2 -> B1(k + 2); Q3 -> A1(k + 2);
    // Q4 -> A0(k + 3) (read in phase 4(k + 3) - 2, next buffer).
#define W_KTILE(V1, V2, FX, FY, BC, BN)                                                                    \
    do {                                                                                                   \
        W_PHASE(V1, FX, fa0, 0, 0, FY, an0, W_B1, rw, vb0, kb0, 0, W_B0, BC, d_w);                          \
        W_PHASE(V1, FY, fa0, 2, 0, fa1, am0, W_A1, rw, vb1, kb1, 1, W_B1, BC, d_w);                         \
        {                                                                                                  \
            const unsigned dl_ = (BN) > (BC) ? (unsigned)W_BUF : (unsigned)-W_BUF;                         \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { am0[kk] += dl_; an0[kk] += dl_; }           \
        }                                                                                                  \
        W_PHASE(V1, FY, fa1, 2, 2, fa0, am0, W_A0, ra, va1, ka1, 1, W_A1, BC, d_a);                         \
        W_PHASE(V2, FX, fa1, 0, 2, FY, an0, W_B0, ra, va0, ka0, 0, W_A0, BN, d_a);                          \
    } while (0)
#define W_STAGE(R, VOFF, KC, H, G, BUFB, DLT)                                                                   \
    do {                                                                                                   \
        W_STG_I(R, VOFF, H, G, BUFB, 0); W_STG_I(R, VOFF, H, G, BUFB, 1);                                   \
        W_STG_I(R, VOFF, H, G, BUFB, 2); W_STG_I(R, VOFF, H, G, BUFB, 3); W_ADV(VOFF, KC, DLT);                  \
    } while (0)

    const T* bias = (const T*)p.bias;
    // PERSIST: the bias of the wave's 128 columns travels like the operands — ONE LDS-DMA instruction (4 bytes per lane) at the start of
    // a tile into the wave's epilogue block, which nothing touches before the epilogue — so it retires in order with the stagings under
    // the same counted waits.  (An ordinary load would make hipcc wait vmcnt(0) at its first use, i.e. drain the DMA queue in front of
    // every epilogue; an inline-asm load into registers is unsafe: the compiler may copy "its result" before the data has arrived.)
    // ALWAYS issued (without a bias: the first weights, unused) so that the waits do not depend on it.
    const unsigned lds_bias = __builtin_amdgcn_readfirstlane(lds_base + 2 * W_BUF + wave * 8192);
    auto request_bias = [&]() {
        // (bias along m: the 128 ROWS of the wave instead of its 128 columns)
        const char* src = uniform_ptr(bias ? (const char*)(p.bias_on_m ? bias + m0 + wm * 128 : bias + n0 + wn * 128) : (const char*)p.W);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" :: "s"(lds_bias), "v"(lane * 4), "s"(src) : "memory", "m0");
    };
    if constexpr (PERSIST) request_bias();          // (first tile: older than every DMA of the prologue, retired by the first counted wait)
    // ---- prologue: both K-tile buffers in flight in consumption order, then A0(2) ----
    W_STAGE(ra, va0, ka0, 0, W_A0, 0, d_a); W_STAGE(rw, vb0, kb0, 0, W_B0, 0, d_w); W_STAGE(rw, vb1, kb1, 1, W_B1, 0, d_w); W_STAGE(ra, va1, ka1, 1, W_A1, 0, d_a);
    W_STAGE(ra, va0, ka0, 0, W_A0, W_BUF, d_a); W_STAGE(rw, vb0, kb0, 0, W_B0, W_BUF, d_w); W_STAGE(rw, vb1, kb1, 1, W_B1, W_BUF, d_w); W_STAGE(ra, va1, ka1, 1, W_A1, W_BUF, d_a);
    W_VM(24);                                  // A0(0), B0(0) of this wave have landed
    __builtin_amdgcn_s_barrier();
    W_READ(fa0, am0, W_A0);
    W_READ(fbx, an0, W_B0);
    W_LGKM0();
    __builtin_amdgcn_s_barrier();              // everybody has read A0(0): its slot may be re-staged
    W_STAGE(ra, va0, ka0, 0, W_A0, 0, d_a);    // A0(2)
    // entering phase 0 the wait is vmcnt(24): B1(0) is followed by A1(0), A0(1), B0(1), B1(1), A1(1), A0(2) = 24 instructions
    if constexpr (ABL & 64) ts[1] = __builtin_readcyclecounter();
    if constexpr (PERSIST) {
        char* wl8 = dyn_smem + 2 * W_BUF + wave * 8192;
        // first tile: nothing but DMAs in flight.  Later tiles: the 32 stores of the last epilogue and the bias request of this tile sit
        // between the DMAs.  vmcnt is ONE counter for loads and stores, and only loads are retired in order among themselves (stores may
        // be acknowledged before older loads: a wait that counted them — vmcnt(58) — raced), so a wait may only count the YOUNGER LOADS:
        // 24 DMA instructions + the bias request for the seven phases whose group was staged before the epilogue.  Stores still
        // pending make that wait stricter than needed: phase j needs all but the last 24 - 4j stores acknowledged, which they are
        // unless the fabric is more than a microsecond behind.
#define W_KLOOP(KT0)                                                                                       \
        for (int kt = KT0; kt < nk; kt += 2) {                                                             \
            W_KTILE(24, 24, fbx, fby, 0, W_BUF);                                                           \
            W_KTILE(24, 24, fby, fbx, W_BUF, 0);                                                           \
        }
        // XCD-wide tile rounds (p.sync): the 32 workgroups that share an XCD's L2 (blockIdx & 7) walk their tiles in rounds — round r is
        // an 8 x 4 block of tiles that needs 12 A / W panels between them — but only workgroups whose K loops stay within a few
        // K-tiles of each other find a neighbour's lines in the 4 MiB L2.  Every workgroup ARRIVES (one L2 atomic) when it enters the
        // epilogue of a tile and, before the first phase of its next tile, wave 0 polls until all workgroups that had a tile in the
        // finished round have arrived: drift stays below one epilogue.  A hint only — the poll gives up after ~20 us, nothing depends
        // on it for correctness, and the other waves simply meet wave 0 at the first phase's barrier.
        const int xcd_ = (int)blockIdx.x & 7;
        const int q_x = (nwg_all >> 3) + (xcd_ < (nwg_all & 7) ? 1 : 0);
        unsigned* const ctr = p.sync ? p.sync + xcd_ * 32 : nullptr;
        int round_ = 0;
        bool poll_ok = true;      // a poll that ran out (counters not visible: another launch shares them, or the workgroups of this
                                  // blockIdx class do not share an L2 after all) switches the polling off for the rest of the launch
        W_KLOOP(0)
        for (;;) {
            // (agent scope: other workgroups read the counter; on gfx950 the same L2 atomic as the workgroup-scope form)
            if (ctr && t == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // epilogue of this tile; the ring already holds K-tiles 0 and 1 of the next one, fa0 / fbx its first fragments.
            // The lane id is laundered through an empty asm so that hipcc recomputes the epilogue's per-lane addresses per tile
            // (a few dozen VALU instructions) instead of hoisting them out of the tile loop and spilling across the K loop.
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
#ifdef M4D_ABLATIONS
            if (p.abl & 512) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (p.abl & 1024) { asm volatile("s_sleep 127\n\ts_sleep 127\n\ts_sleep 127\n\ts_sleep 127" ::: "memory"); }
            if (p.abl & 2048) __builtin_amdgcn_s_barrier();
            if (p.abl & 128) {      // debug: known accumulators (tile-local n, or m with bit 256) instead of the products
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[a][b][r] = (p.abl & 256) ? (float)(wm * 128 + b * 32 + (lane_e & 31))
                                                         : (float)(wn * 128 + a * 32 + (r >> 2) * 8 + (lane_e >> 5) * 4 + (r & 3));
            }
#endif
            const unsigned evoff = (unsigned)(((lane_e >> 3) * p.ldc + (lane_e & 7) * 8) * 2);     // stores: row (lane >> 3), columns (lane & 7) * 8
            u32x4 bz[2];            // this lane's 2 x 8 bias values, out of the block before the first accumulators go in
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) bz[nh] = *reinterpret_cast<const u32x4*>(wl8 + (nh * 64 + (lane_e & 7) * 8) * 2);
            const bool bias_m = __builtin_amdgcn_readfirstlane((int)(bias != nullptr && p.bias_on_m)) != 0;
            f32x4 bmr[4];            // bias along m: the lane's row of every (block mi, iteration it): rows mi * 32 + it * 8 + (lane >> 3) of the wave's 128
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    bmr[mi][it] = bias_m ? (float)reinterpret_cast<const T*>(wl8)[mi * 32 + it * 8 + (lane_e >> 3)] : 0.f;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                {
                    epilogue_block32<T, EPI>(p, wl8, acc[nh * 2][mi], acc[nh * 2 + 1][mi], m0 + wm * 128 + mi * 32, n0 + wn * 128 + nh * 64,
                                             lane_e, evoff, bias != nullptr, bz[nh], bias_m, bmr[mi]);
                    __builtin_amdgcn_sched_barrier(0);       // one block at a time: interleaved blocks cost registers the K loop then spills
                }
            bid += (int)gridDim.x;
            if (bid >= nwg_all) break;
            tile_coords(p, tm, tn, bid);
            m_lo = (int64_t)tm * BM2; n_lo = (int64_t)tn * BN2;
            m0 = min(m_lo, p.M - BM2); n0 = min(n_lo, p.N - BN2);
            next_deltas();
            request_bias();
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            if (ctr) {
                ++round_;
                const unsigned target = (unsigned)min(q_x, 32 * round_);
                if (wave == 0 && poll_ok) {
                    poll_ok = false;
                    for (int k = 0; k < 64; ++k) {
                        unsigned seen;
                        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(ctr) : "memory");
                        if (seen >= target) { poll_ok = true; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
            }
            W_KTILE(25, 25, fbx, fby, 0, W_BUF);
            W_KTILE(25, 24, fby, fbx, W_BUF, 0);
            W_KLOOP(2)
        }
#undef W_KLOOP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead stagings of the last tile must not outlive the workgroup's LDS
        return;
    }
    {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            W_KTILE(24, 24, fbx, fby, 0, W_BUF);
            W_KTILE(24, 24, fby, fbx, W_BUF, 0);
        }
        if (kt < nk) W_KTILE(24, 24, fbx, fby, 0, W_BUF);
    }
#undef W_KTILE
#undef W_PHASE
#undef W_MMA
#undef W_READ
    W_LGKM0();
    if constexpr (ABL & 64) ts[2] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant tail stagings must land before the LDS is re-used
    __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 64) ts[3] = __builtin_readcyclecounter();
    char* wl = dyn_smem + wave * 16384;
    // (the host sends bf16 outputs whose rows are not 16-byte aligned, and the erf-GELU / SiLU epilogues of a few small GEMMs, to
    // gemm_bt256p_kernel)
    f32x4 res0[8], res1[8];      // rolling residual window of the gated-residual epilogue (unused by the others)
    auto epilogue = [&](auto INNER_T) {
        constexpr bool INNER = decltype(INNER_T)::value;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                const int nxt = nh * 2 + mh + 1;        // next block in this order: (nh', mh') = (nxt >> 1, nxt & 1)
                epilogue_block64<T, EPI, INNER>(p, wl, acc[nh * 2][mh * 2], acc[nh * 2][mh * 2 + 1], acc[nh * 2 + 1][mh * 2], acc[nh * 2 + 1][mh * 2 + 1],
                                                m0 + wm * 128 + mh * 64, n0 + wn * 128 + nh * 64, m_lo, n_lo, lane, res0, res1, nxt == 1, nxt < 4,
                                                m0 + wm * 128 + (nxt & 1) * 64, n0 + wn * 128 + (nxt >> 1) * 64);
            }
    };
    if constexpr (EPI == M4D_EPI_RESID_GATE) {
        if (m0 >= m_lo && n0 >= n_lo) epilogue(std::true_type{});        // (workgroup-uniform) the tile is not shifted into a neighbour
        else epilogue(std::false_type{});
    } else epilogue(std::false_type{});
    if constexpr (ABL & 64) {       // timestamps (shader cycles) + 100 MHz wall clock + hardware id into the tile's first output row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[4] = __builtin_readcyclecounter();
        if (t == 0) {
            unsigned long long* d = (unsigned long long*)((T*)p.out + m0 * p.ldc + n0);
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[0] = ts[0]; d[1] = ts[1]; d[2] = ts[2]; d[3] = ts[3]; d[4] = ts[4]; d[5] = ts[5];
            d[6] = __builtin_amdgcn_s_memrealtime(); d[7] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#undef W_STAGE
#undef W_STG_I
#undef W_GLDS
#undef W_ADV
#undef W_DSR
#undef W_PIN
#undef W_LGKM0
#undef W_VM
}

// launcher body shared by the per-epilogue translation units (gemm_wide_*.hip: one instantiation each, compiled in parallel)
template <int EPI>
int launch_gemm_wide(const GemmArgs& p, unsigned nwg, hipStream_t st, unsigned nby = 1) {
#define W_LAUNCH(A)                                                                                                    \
    do {                                                                                                               \
        static PerDeviceOnce configured;                                                                                \
        if (configured.pending()) {                                                                                             \
            if (hipFuncSetAttribute((const void*)gemm_bt256w_kernel<A, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W_BUF) != hipSuccess) \
                return -3;                                                                                             \
            configured.mark();                                                                                         \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_bt256w_kernel<A, EPI>), dim3(nwg, nby), dim3(256), 2 * W_BUF, st, p);                  \
    } while (0)
#ifdef M4D_ABLATIONS
    if constexpr (EPI == M4D_EPI_STORE) {
        if (p.abl & 64) {                       // timeline stamps, alone or on top of a timing ablation (clock = power proxy)
            switch (p.abl & 15) {
                case 1: W_LAUNCH(65); return 0;
                case 2: W_LAUNCH(66); return 0;
                case 3: W_LAUNCH(67); return 0;
                case 7: W_LAUNCH(71); return 0;
                default: W_LAUNCH(64); return 0;
            }
        }
        switch (p.abl & 15) {                  // (bit 16 = every workgroup on tile (0, 0): handled by tile_coords at run time)
            case 1: W_LAUNCH(1); return 0;
            case 2: W_LAUNCH(2); return 0;
            case 4: W_LAUNCH(4); return 0;
            case 8: W_LAUNCH(8); return 0;
            case 3: W_LAUNCH(3); return 0;
            case 7: W_LAUNCH(7); return 0;
            case 10: W_LAUNCH(10); return 0;
            default: break;
        }
    }
#endif
    W_LAUNCH(0);
#undef W_LAUNCH
    return 0;
}

// persistent form (bf16 epilogues): one workgroup per CU walks its tiles; 160 KiB of LDS (two K-tile buffers + 4 x 8 KiB epilogue blocks)
template <int EPI>
int launch_gemm_wide_persistent(const GemmArgs& p, unsigned nwg, unsigned ncu, hipStream_t st) {
    constexpr int LDS = 2 * W_BUF + 4 * 8192;
    static PerDeviceOnce configured;
    if (configured.pending()) {
        if (hipFuncSetAttribute((const void*)gemm_bt256w_kernel<0, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        configured.mark();
    }
    const unsigned grid = nwg < ncu ? nwg : ncu;
    hipLaunchKernelGGL((gemm_bt256w_kernel<0, EPI, true>), dim3(grid), dim3(256), LDS, st, p);
    return 0;
}

}  // namespace
