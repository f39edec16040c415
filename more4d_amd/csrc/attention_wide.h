// attn128w_kernel: flash-attention forward, bf16, head_dim 128, long key ranges — 64 queries per WAVE.
//
// What bounds attn128p_kernel (32 queries per wave): every MFMA takes a fresh 1 KiB K / V^T fragment from LDS, so a
// 64-key tile costs 8 waves x 32 KiB of LDS reads + 32 KiB of DMA writes = 2250 cycles of the 128 B/clk LDS port
// against 2048 cycles of matrix pipe per SIMD (measured: the MFMA stream alone runs at 57 % of peak).  Here a wave
// owns TWO 32-query halves (A, B): each fragment read feeds two MFMAs, LDS traffic per tile halves (4 waves x 32 KiB +
// 32 KiB = 1250 cycles) and the matrix pipe becomes the bound.  The price is 1 wave per SIMD (~430 VGPRs), so the
// softmax can no longer hide under ANOTHER wave's MFMAs; it is software-pipelined inside the wave instead:
//   iteration i:   MFMA stream  =  O += V^T(i-1) P^T(i-1)   then   S(i+1) = K(i+1) Q^T        (64 MFMAs, 32 fragments)
//                  VALU stream  =  softmax of S(i) -> P(i)   (both halves), cut into 32 chunks, one behind each
//                                  fragment step, so the VALU work issues in the shadow of the two MFMAs of that step.
// O is rescaled by alpha(i) = exp2(m(i-1) - m(i)) at the start of iteration i+1 (after PV(i-1), before PV(i)), only when
// some row's maximum moved.  K / V^T tiles: four 32 KiB LDS stages by global->LDS DMA; tile i+2 is requested at the
// start of iteration i into the stage tile i-2 left and must have landed before the single barrier that ends the
// iteration.  The ragged last tile is processed first, in plain order.  Single K/V segment.
#pragma once
#include <type_traits>

namespace wide {

constexpr int STAGE = 32768, VOFF = 16384, KVB = 64;

template <int OFF> M4D_DEV void dsr128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> M4D_DEV void lgkm_le() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Per-wave state.  Index [h] = query half (0: rows 0-31, 1: rows 32-63 of the wave's 64).
// O^T accumulators and Q fragments are NOT C++ values: they live in hand-assigned AGPRs for the whole kernel
//   O[h][d]  (f32x16)  = a[(h*4 + d)*16 .. +15]        (a0 .. a127)
//   Q[h][kk] (bf16x8)  = a[128 + (h*8 + kk)*4 .. +3]   (a128 .. a191)
// and every asm statement that touches them names all 192 as clobbered, so the compiler never parks anything there.
#define M4D_ACLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"
struct State {
    f32x16 s[2][2][2];     // [buffer][h][sub]: S of the tile being soft-maxed / S of the next tile being accumulated
    bf16x8 pf[2][4];       // [h][c]: P fragments (single buffer: PV(i-1) consumes them in steps 0..15, softmax(i) re-packs in steps >= 16)
    bf16x8 ring[8];
    unsigned ka[8], va[4];
    float m_run[2], l_run[2], alpha[2], mx[2];
    f32x2 ps2[2];
};

// fragment F of an iteration's stream: F < 16: V^T fragment (c = F/4 key group, d = F%4 head-dim block) of the PV tile;
// F >= 16: K fragment (kk = (F-16)/2, sub = (F-16)%2) of the QK tile
template <int F> M4D_DEV void frag_read(State& w) {
    if constexpr (F < 16) dsr128<(F & 3) * 4096>(w.ring[F & 7], w.va[F >> 2]);
    else dsr128<((F - 16) & 1) * 8192>(w.ring[F & 7], w.ka[(F - 16) >> 1]);
}
// MFMAs are issued from inline asm: left to itself the allocator kept S in AGPRs (VALU cannot read them: ~350
// v_accvgpr copies per tile) and spilled Q.  Software hazards: every accumulator is re-used >= 4 MFMAs later, S is read by
// VALU only in the next iteration, P is written by VALU >= 16 fragment steps before its next MFMA use.
template <int OB> M4D_DEV void mfma_pv(const bf16x8& frag, const bf16x8& pfrag) {        // O[OB] (AGPR) += frag x P
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" :: "v"(frag), "v"(pfrag), "n"(OB * 16), "n"(OB * 16 + 15) : M4D_ACLOB);
}
template <int QB> M4D_DEV void mfma_qk(f32x16& acc, const bf16x8& frag) {                // S (VGPR) += frag x Q[QB] (AGPR)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(frag), "n"(128 + QB * 4), "n"(128 + QB * 4 + 3) : M4D_ACLOB);
}
template <int QB> M4D_DEV void mfma_qk0(f32x16& acc, const bf16x8& frag) {               // S (VGPR)  = frag x Q[QB]
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=v"(acc) : "v"(frag), "n"(128 + QB * 4), "n"(128 + QB * 4 + 3) : M4D_ACLOB);
}
template <int R> M4D_DEV void acc_write(float v) { asm volatile("v_accvgpr_write_b32 a[%1], %0" :: "v"(v), "n"(R) : M4D_ACLOB); }
template <int R> M4D_DEV float acc_read() { float v; asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R) : M4D_ACLOB); return v; }
template <int R, int N> M4D_DEV void acc_fill(float v) { if constexpr (R < N) { acc_write<R>(v); acc_fill<R + 1, N>(v); } }
template <int R, int N> M4D_DEV void acc_scale(float a) {                                // a[R..N) *= a
    if constexpr (R < N) { acc_write<R>(acc_read<R>() * a); acc_scale<R + 1, N>(a); }
}
template <int F, int SB, int HALF> M4D_DEV void frag_mma(State& w) {   // SB: S buffer written by QK; HALF: query half
    if constexpr (F < 16) {
        mfma_pv<HALF * 4 + (F & 3)>(w.ring[F & 7], w.pf[HALF][F >> 2]);
    } else {
        constexpr int kk = (F - 16) >> 1, sub = (F - 16) & 1;
        if constexpr (kk == 0) mfma_qk0<HALF * 8>(w.s[SB][HALF][sub], w.ring[F & 7]);
        else mfma_qk<HALF * 8 + kk>(w.s[SB][HALF][sub], w.ring[F & 7]);
    }
}

// ---- softmax of S[SB^1... the buffer NOT being written] in 32 chunks ----
// element e = sub*16 + r of half h.  Work list per half: 4 max chunks (8 elements each), finalize, 16 exp units (2
// elements each), 4 pack groups (8 elements -> one P fragment).
template <int H, int Q, int SR> M4D_DEV void sm_max(State& w) {          // Q = 0..3: elements [8Q, 8Q+8)
    const f32x16& v = w.s[SR][H][Q >> 1];
    constexpr int r0 = (Q & 1) * 8;
    float mx = Q == 0 ? v[0] : w.mx[H];
    if constexpr (Q == 0) {
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(v[1]), "v"(v[2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(v[3]), "v"(v[4]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(v[5]), "v"(v[6]));
        mx = fmaxf(mx, v[7]);
    } else {
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(v[r0]), "v"(v[r0 + 1]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(v[r0 + 2]), "v"(v[r0 + 3]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(v[r0 + 4]), "v"(v[r0 + 5]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(v[r0 + 6]), "v"(v[r0 + 7]));
    }
    w.mx[H] = mx;
}
template <int H> M4D_DEV void sm_final(State& w, float sc) {
    float mx = w.mx[H];
    const unsigned u = __float_as_uint(mx);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    // lazy reference maximum: m only moves when the tile maximum exceeds it by more than 2^8 (probabilities then stay
    // <= 256, harmless in fp32 / bf16 range; the final O / l and the LSE are independent of the reference) -> the 128-register
    // O rescale almost never runs after the first tile
    const float cand = mx * sc;
    const float m_new = cand > w.m_run[H] + 8.f ? cand : w.m_run[H];
    w.alpha[H] = __builtin_amdgcn_exp2f(w.m_run[H] - m_new);     // exactly 1 when m did not move; 0 on the first tile
    w.m_run[H] = m_new;
    w.l_run[H] *= w.alpha[H];
    w.ps2[H] = f32x2{0.f, 0.f};
}
// The softmax of one tile as a flat list of 74 micro-ops, small enough to issue in the 28-cycle shadow of one MFMA:
//   0..3  max of half A (8 elements each)     4  finalize A (cross-half max, reference maximum, alpha)
//   5..8  max of half B                        9  finalize B
//   10 + 32*H + e : element e (= sub*16 + r) of half H:  even r: pk_fma of the pair (r, r+1), exp2 of r
//                                                         odd r:  exp2 of r, row-sum += pair; every 8th element also
//                                                                 packs the finished 8 elements into a P fragment
constexpr int NUOP = 74;
template <int U, int SR> M4D_DEV void sm_uop(State& w, float sc) {
    if constexpr (U < 4) sm_max<0, U, SR>(w);
    else if constexpr (U == 4) sm_final<0>(w, sc);
    else if constexpr (U < 9) sm_max<1, U - 5, SR>(w);
    else if constexpr (U == 9) sm_final<1>(w, sc);
    else if constexpr (U < NUOP) {
        constexpr int H = (U - 10) >> 5, e = (U - 10) & 31, sub = e >> 4, r = e & 15;
        f32x16& v = w.s[SR][H][sub];
        // single-issue VALU only: a packed fp32 op beside the MFMAs costs ~+22-26 cycles more than its two scalar halves
        // (MI355X_MICROARCH.md price list) — with one wave per SIMD that is what made MFMA time and VALU time ADD.  Everything
        // is volatile asm in a fixed order because hipcc's hazard recogniser does not see through inline asm: a v_exp_f32
        // result is consumed no closer than two instructions behind it (gfx940+ trans-use hazard needs one).
        if constexpr ((r & 1) == 0) {
            const float nm = -w.m_run[H];
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[r]) : "v"(v[r]), "s"(sc), "v"(nm));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[r + 1]) : "v"(v[r + 1]), "s"(sc), "v"(nm));
            asm volatile("v_exp_f32 %0, %1" : "=v"(v[r]) : "v"(v[r]));
        } else {
            asm volatile("v_exp_f32 %0, %1" : "=v"(v[r]) : "v"(v[r]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(w.ps2[H][0]) : "v"(w.ps2[H][0]), "v"(v[r - 1]));
            asm volatile("s_nop 0");
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(w.ps2[H][1]) : "v"(w.ps2[H][1]), "v"(v[r]));
            if constexpr ((e & 7) == 7) {
                constexpr int C = e >> 3;
                w.pf[H][C] = pack8<bf16_t>(w.s[SR][H][C >> 1], (C & 1) * 8);
                if constexpr (C == 3) w.l_run[H] += w.ps2[H][0] + w.ps2[H][1];
            }
        }
    }
}
template <int U, int UE, int SR> M4D_DEV void sm_uops(State& w, float sc) {      // micro-ops [U, UE)
    if constexpr (U < UE) { sm_uop<U, SR>(w, sc); sm_uops<U + 1, UE, SR>(w, sc); }
}

// One iteration's interleaved stream over fragments [F0, F1): per fragment  wait, MFMA(half A), VALU slot, MFMA(half B),
// next fragment read, VALU slot.  The 74 softmax micro-ops are spread evenly over the 2*(F1-F0) slots.
// SB = S buffer written by QK (the softmax reads SB^1).
template <int F, int F0, int F1, int SB, int PB, bool HAS_SM, int ABL = 0> M4D_DEV void stream(State& w, float sc) {
    if constexpr (F < F1) {
        constexpr int left = F1 - 1 - F;                    // fragments after this one
        constexpr int NSLOT = 2 * (F1 - F0), t0 = 2 * (F - F0);
        lgkm_le<(left < 7) ? left : 7>();
        if constexpr (!(ABL & 2)) frag_mma<F, SB, 0>(w);
        if constexpr (HAS_SM && !(ABL & 1)) sm_uops<(t0 * NUOP) / NSLOT, ((t0 + 1) * NUOP) / NSLOT, SB ^ 1>(w, sc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 2)) frag_mma<F, SB, 1>(w);
        if constexpr (F + 8 < F1) frag_read<F + 8>(w);
        if constexpr (HAS_SM && !(ABL & 1)) sm_uops<((t0 + 1) * NUOP) / NSLOT, ((t0 + 2) * NUOP) / NSLOT, SB ^ 1>(w, sc);
        __builtin_amdgcn_sched_barrier(0);
        stream<F + 1, F0, F1, SB, PB, HAS_SM, ABL>(w, sc);
    }
}
template <int F, int F1> M4D_DEV void prefetch(State& w) {
    if constexpr (F < F1) { frag_read<F>(w); prefetch<F + 1, F1>(w); }
}
// softmax without an MFMA stream (peeled ragged tile, first tile): all 32 chunks back to back
template <int F, int SR, int PW> M4D_DEV void softmax_all(State& w, float sc) { sm_uops<0, NUOP, SR>(w, sc); }


M4D_DEV void rescale_o(State& w) {     // O *= alpha (of the last softmax), skipped when no row's maximum moved
    if (__any(w.alpha[0] != 1.f || w.alpha[1] != 1.f)) {
        acc_scale<0, 64>(w.alpha[0]);
        acc_scale<64, 128>(w.alpha[1]);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // accvgpr writes -> MFMA SrcC
    }
}
M4D_DEV void step_va(State& w, int next_stage) {
    const unsigned dv = next_stage ? (unsigned)STAGE : (unsigned)(-3 * STAGE);
#pragma unroll
    for (int c = 0; c < 4; ++c) w.va[c] += dv;
}
M4D_DEV void step_ka(State& w, int next_stage) {
    const unsigned dk = next_stage ? (unsigned)STAGE : (unsigned)(-3 * STAGE);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) w.ka[kk] += dk;
}
// normalised O of one (head-dim block, row quad) of half OB/4 -> global
template <int OB, int DRQ> M4D_DEV void store_o(bf16_t* op, float inv, bool valid, int accumulate) {
    if constexpr (DRQ < 16) {
        constexpr int d = DRQ >> 2, rq = DRQ & 3, R = (OB + d) * 16 + rq * 4;
        f32x4 v = {acc_read<R>() * inv, acc_read<R + 1>() * inv, acc_read<R + 2>() * inv, acc_read<R + 3>() * inv};
        if (valid) {
            bf16_t* dst = op + d * 32 + rq * 8;
            if (accumulate) {
                f32x4 prev = load4(dst);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round_through<bf16_t>(v[e]) + prev[e];
            }
            store4(dst, v);
        }
        store_o<OB, DRQ + 1>(op, inv, valid, accumulate);
    }
}

}  // namespace wide

template <int ABL>
__global__ __launch_bounds__(256, 1) void attn128w_kernel(AttnArgs p) {
    using namespace wide;
    typedef bf16_t T;
    constexpr int D = 128, QB = 256;
    extern __shared__ __attribute__((aligned(16))) char wsmem[];   // 4 * STAGE

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

    State w;
    int64_t qrow[2];
    bool qvalid[2];
    acc_fill<0, 128>(0.f);                                               // O = 0
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        qrow[hh] = (int64_t)qt * QB + wave * 64 + hh * 32 + li;
        qvalid[hh] = qrow[hh] < p.Lq;
        w.m_run[hh] = -INFINITY; w.l_run[hh] = 0.f; w.alpha[hh] = 1.f; w.mx[hh] = 0.f;
        w.ps2[hh] = f32x2{0.f, 0.f};
    }
    {   // Q fragments -> a128.. (a bf16x8 is four dwords)
        auto load_q = [&](int hh, int kk) -> uint4 {
            const T* qp = (const T*)p.q + b * p.q_bs + qrow[hh] * p.q_ls + (int64_t)h * D + hi * 8 + kk * 16;
            return qvalid[hh] ? *reinterpret_cast<const uint4*>(qp) : make_uint4(0, 0, 0, 0);
        };
#define M4D_QW(HH, KK)                                                                                                \
        do {                                                                                                         \
            const uint4 q4 = load_q(HH, KK);                                                                         \
            acc_write<128 + ((HH) * 8 + (KK)) * 4 + 0>(__uint_as_float(q4.x));                                       \
            acc_write<128 + ((HH) * 8 + (KK)) * 4 + 1>(__uint_as_float(q4.y));                                       \
            acc_write<128 + ((HH) * 8 + (KK)) * 4 + 2>(__uint_as_float(q4.z));                                       \
            acc_write<128 + ((HH) * 8 + (KK)) * 4 + 3>(__uint_as_float(q4.w));                                       \
        } while (0)
        M4D_QW(0, 0); M4D_QW(0, 1); M4D_QW(0, 2); M4D_QW(0, 3); M4D_QW(0, 4); M4D_QW(0, 5); M4D_QW(0, 6); M4D_QW(0, 7);
        M4D_QW(1, 0); M4D_QW(1, 1); M4D_QW(1, 2); M4D_QW(1, 3); M4D_QW(1, 4); M4D_QW(1, 5); M4D_QW(1, 6); M4D_QW(1, 7);
#undef M4D_QW
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)wsmem;
    {
        const int kr = perm23(li);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) w.ka[kk] = lds0 + kr * 256 + (((kk * 2 + hi) ^ (kr & 15)) << 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) w.va[c] = lds0 + VOFF + li * 128 + (((c * 2 + hi) ^ ((li >> 1) & 7)) << 4);
    }

    // K/V may arrive as several segments (one per rank of the T-sharded loop) that share their strides (host-checked):
    // the full 64-key tiles of all segments form one pipelined tile list, each segment's ragged tail is peeled first
    const int64_t kls = p.kv.k_ls[0], vls = p.kv.vt_ls[0];
    int NT = 0;
    for (int sg = 0; sg < p.kv.nseg; ++sg) NT += p.kv.len[sg] > 0 ? (int)(p.kv.len[sg] / KVB) : 0;
    auto seg_k = [&](int sg) { return (const T*)p.kv.k[sg] + b * p.kv.k_bs[sg] + (int64_t)h * D; };
    auto seg_v = [&](int sg) { return (const T*)p.kv.vt[sg] + b * p.kv.vt_bs[sg] + (int64_t)h * D * p.kv.vt_ls[sg]; };
    int dseg = 0;                               // DMA iterator over the tile list (wave-uniform)
    int64_t dk0 = 0;
    while (dseg < p.kv.nseg && p.kv.len[dseg] < KVB) ++dseg;
    // base pointers / length of the iterator's CURRENT segment live in scalars and are re-read only when it moves on
    // (indexing the kernel-argument arrays per tile costs scalar memory loads on the critical path of every iteration)
    const T* dkb = dseg < p.kv.nseg ? seg_k(dseg) : nullptr;
    const T* dvb = dseg < p.kv.nseg ? seg_v(dseg) : nullptr;
    int64_t dlen = dseg < p.kv.nseg ? p.kv.len[dseg] : 0;

    // DMA: 4 waves x 4 instructions per operand; lane -> (row in instruction block, physical 16-byte chunk)
    const int k_r = lane >> 4, k_lc0 = lane & 15, v_r = lane >> 3, v_pc = lane & 7;
    unsigned offk[4], offv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int blk = wave * 4 + i;
        const int krow = blk * 4 + k_r, vrow = blk * 8 + v_r;
        offk[i] = (unsigned)((krow * kls + (k_lc0 ^ (krow & 15)) * 8) * 2);
        offv[i] = (unsigned)((vrow * vls + (v_pc ^ ((vrow >> 1) & 7)) * 8) * 2);
    }
    auto uniform_ptr = [](const char* q) {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi2 << 32) | lo);
    };
    auto dma_tile = [&](int stage, int64_t k0) {
        const char* kp = uniform_ptr((const char*)(dkb + dk0 * kls));      // the NEXT tile of the list (k0 unused)
        const char* vp = uniform_ptr((const char*)(dvb + dk0));
        dk0 += KVB;
        if (dk0 + KVB > dlen) {
            dk0 = 0;
            ++dseg;
            while (dseg < p.kv.nseg && p.kv.len[dseg] < KVB) ++dseg;
            if (dseg < p.kv.nseg) { dkb = seg_k(dseg); dvb = seg_v(dseg); dlen = p.kv.len[dseg]; }
        }
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + wave * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + i * 1024), "v"(offk[i]), "s"(kp) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + VOFF + i * 1024), "v"(offv[i]), "s"(vp) : "memory", "m0");
        }
    };
    // ---- peeled ragged tail: stage 0 through registers (zero filled), plain order ----
    for (int sg = 0; sg < p.kv.nseg; ++sg) {
        const int64_t len = p.kv.len[sg];
        if (len <= 0) continue;
        const int64_t tail0 = (len / KVB) * KVB;
        if (tail0 >= len) continue;
        const T* kbase = seg_k(sg);
        const T* vbase = seg_v(sg);
        char* base = wsmem;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + 256 * i;
            {
                const int row = c >> 4, ch = c & 15;
                const int64_t key = tail0 + row;
                const uint4 v = key < len ? *reinterpret_cast<const uint4*>(kbase + key * kls + ch * 8) : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(base + swz_off<256>(row, ch)) = v;
            }
            {
                const int row = c >> 3, ch = c & 7;
                const int64_t key = tail0 + ch * 8;
                const T* src = vbase + row * vls + key;
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
        __syncthreads();
        prefetch<16, 24>(w);
        stream<16, 16, 32, 0, 0, false>(w, p.sc);                        // S[0] = K Q^T
        const int k_lim = (int)(len - tail0);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7) >= k_lim) w.s[0][hh][sub][r] = -INFINITY;
        softmax_all<0, 0, 0>(w, p.sc);                                   // -> P[0]
        rescale_o(w);
        prefetch<0, 8>(w);
        stream<0, 0, 16, 0, 0, false>(w, p.sc);                          // O += V^T P[0]
        __syncthreads();
    }

    // ---- pipelined part: NT >= 4 full tiles (the host only selects this kernel for long key ranges) ----
    {
        // prologue: tiles 0, 1 requested and landed; S[0] = S(0)
        dma_tile(0, 0);
        dma_tile(1, KVB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prefetch<16, 24>(w);
        stream<16, 16, 32, 0, 0, false>(w, p.sc);                        // S buffer 0 <- tile 0
        step_ka(w, 1);                                                      // K side -> tile 1 (stage 1)
        // Iteration i (buffers by parity: S(i) in S[i&1], P(i) -> P[i&1], QK(i+1) -> S[(i+1)&1], PV(i-1) <- P[(i-1)&1]):
        //   i = 0:            softmax(0) || QK(1)
        //   0 < i <= NT-1:    softmax(i) || PV(i-1), QK(i+1)      (for i = NT-1 the QK runs on a stale stage; S unused)
        //   then:             PV(NT-1)
#define M4D_W_END_ITER()                                                                                              \
        do {                                                                                                         \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* the tile requested in this iteration landed */     \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            __builtin_amdgcn_s_barrier();                                                                            \
        } while (0)
#define M4D_W_ITER(I, SBPB)                                                                                           \
        do {                                                                                                         \
            if ((I) + 2 < NT) dma_tile(((I) + 2) & 3, (int64_t)((I) + 2) * KVB);                                     \
            rescale_o(w);                                                                                               \
            prefetch<0, 8>(w);                                                                                       \
            stream<0, 0, 32, SBPB, SBPB, true, ABL>(w, p.sc);                                                        \
            step_va(w, (I) & 3); step_ka(w, ((I) + 2) & 3);                                                                \
            M4D_W_END_ITER();                                                                                        \
        } while (0)
        // i = 0
        dma_tile(2, 2 * KVB);
        prefetch<16, 24>(w);
        stream<16, 16, 32, 1, 1, true>(w, p.sc);                         // QK(1) -> S[1]; softmax S[0] -> P[0]
        step_ka(w, 2);
        M4D_W_END_ITER();
        int i = 1;
        for (; i + 1 < NT; i += 2) {                                     // two iterations per trip: buffer roles are constants
            M4D_W_ITER(i, 0);                                            // odd i:  PV <- P[0], QK -> S[0], softmax S[1] -> P[1]
            M4D_W_ITER(i + 1, 1);                                        // even:   PV <- P[1], QK -> S[1], softmax S[0] -> P[0]
        }
        if (i < NT) M4D_W_ITER(i, 0);                                    // NT even: one odd iteration left
#undef M4D_W_ITER
#undef M4D_W_END_ITER
        // final PV(NT-1) <- P[(NT-1)&1]
        rescale_o(w);
        prefetch<0, 8>(w);
        if ((NT - 1) & 1) stream<0, 0, 16, 0, 1, false>(w, p.sc);
        else stream<0, 0, 16, 0, 0, false>(w, p.sc);
    }

    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");       // last MFMAs -> accvgpr reads
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const float l_tot = w.l_run[hh] + __shfl_xor(w.l_run[hh], 32, 64);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        if (p.lse && qvalid[hh] && hi == 0) p.lse[((int64_t)b * p.heads + h) * p.Lq + qrow[hh]] = w.m_run[hh] + log2f(l_tot);
        T* op = (T*)p.out + b * p.o_bs + qrow[hh] * p.o_ls + (int64_t)h * D + hi * 4;
        if (hh == 0) store_o<0, 0>(op, inv, qvalid[0], p.accumulate);
        else store_o<4, 0>(op, inv, qvalid[1], p.accumulate);
    }
}
#undef M4D_ACLOB
