// conv_halo_kernel — the production 3x3(x3) convolution of the VAE / adaptors (bf16, stride 1, zero padding 1 in H and W, "valid"
// in T over the caller's [tail + chunk] staging buffer).  Included by conv.hip.
//
// Why a second formulation: as an implicit GEMM (conv_cl256_kernel) every K-tile re-fetches its 256 x 64 activation tile through
// the global->LDS DMA path although each input pixel feeds 27 taps — 85 FLOP per DMA byte against the 128 of the 256^2 GEMM tile,
// with ~90 VALU instructions of gather bookkeeping per 16 MFMAs: 17 % of the MFMA peak over a whole encode / decode
// (profiles/r02a_vae_kernel_stats.csv).  Here a workgroup owns a TH x TW patch of output pixels of one frame and all taps:
//   * per 16-channel chunk the patch's HALO (KT frames x (TH+2) x (TW+2) pixels x 16 channels) is DMA'd into LDS ONCE and the
//     27 taps read it at shifted addresses: every MFMA B-fragment (32 pixels x 16 channels) is one ds_read_b128 at
//     lane_base[dw] + (dt, dh) offset — no per-tap address arithmetic, no validity masks (out-of-image halo pixels were
//     DMA'd from a zero page), ~200 FLOP per DMA byte;
//   * the weight tile streams through a double buffer one (dt, dh) row of 3 taps at a time, one barrier per 12 MFMAs per wave;
//   * TWO workgroups share a CU (72-80 KiB of LDS and <= 128 VGPRs each): ablations of the first version (one workgroup per CU,
//     double-buffered halo, tools/abl_conv.sh) showed its phases running back to back — MFMA stream 27 %, fragment reads 16 %,
//     DMA waits 16 %, prologue + uncoalesced epilogue 45 % of the kernel — so the latency of one workgroup's halo load, barrier
//     or store tail is covered by the other workgroup's MFMAs instead of by deeper buffering inside one workgroup;
//   * the epilogue goes through LDS (the halo buffer is free by then): 8-byte stores of 32 different pixels per instruction
//     became 64 x NT-byte row segments;
//   * EXACT output-channel tiles: the 8 waves sit along the pixel axis (32 pixels each) and every wave owns all NT 32-channel
//     column tiles of the workgroup (NT = 3 for 96 / 192 / 384 output channels = 1 / 2 / 4 tiles, NT = 4 for 128): no MFMA runs on padding
//     channels (the 4 x 2 wave layout with a fixed 128-channel tile wasted 25 % of the MFMAs at 96, 192 and 384 channels),
//     and one pixel fragment feeds NT MFMAs.  PMC (tools/prof_kernel.sh): the kernel
//     sits at the POWER limit at ~1.5 GHz (back-to-back launches slow down 1064 -> 1412 us within three launches), so what counts
//     is energy: fewer MFMAs and less fabric traffic, not only fewer stalls.
// LDS layout: halo pixel = 32 B (two 16-byte chunks), row pitch a multiple of 16 pixels; chunk c of pixel px lives at slot
// c ^ ((px >> 3) & 1): the 16 lanes a ds_read_b128 services per cycle (lanes {0-3,12-15,20-27} of 32 consecutive pixels, or 2 x 16)
// then cover all sixteen 16-byte bank slots.  Weight rows (one output channel, 16 input channels of one tap) use the same rule.
// The DMA writes LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE address.
#pragma once

namespace halo {
constexpr int CK = 16;                 // input channels per chunk = one MFMA K step
constexpr int PXB = CK * 2;            // bytes per halo pixel / per weight row of one tap
constexpr int KW = 3;

// LDS geometry of one instantiation (shared by the kernel and its launcher)
template <int KT, int KH, int TH, int TW, int NT, int MT, int SD = 1>
struct Cfg {
    static constexpr int NWAVE = MT == 3 ? 4 : 8 / MT;                                  // every wave owns MT 32-pixel tiles x NT 32-channel tiles
    static constexpr int NB = NT * 32;                                    // output channels per workgroup
    static constexpr int WTAP = NB * PXB;                                 // bytes of one tap of the weight tile
    // SD = 2 (the Resample stride-2 conv, ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2), wan_vae.py:96-100): the patch needs 2 TH + 1 input
    // rows of 2 TW + 1 pixels; a halo row is stored de-interleaved, [PE even pixels | PE odd pixels], so that the 32 output pixels of a
    // fragment still read 32 CONSECUTIVE halo pixels for every tap (dw = 0: even j, dw = 1: odd j, dw = 2: even j + 1)
    static constexpr int PE = (TW + 1 + 7) / 8 * 8;
    static constexpr int HH = SD == 2 ? 2 * TH + 1 : TH + KH - 1;
    static constexpr int PITCH = SD == 2 ? 2 * PE : (TW + KW - 1 + 7) / 8 * 8;     // (a multiple of 8 pixels = 256 B keeps the bank pattern of a row)
    // (the pitch padding behind the last row is never read: trimming it is what lets the 12 x 32 x 3-frame halo + THREE weight buffers
    //  fit one half of the CU's LDS, i.e. weight groups requested two steps ahead instead of one)
    static constexpr int NPIX = KT * HH * PITCH - (SD == 1 ? PITCH - (TW + KW - 1) : 0);
    static constexpr int HINSTR = ((NPIX * 2 + 63) / 64 + NWAVE - 1) / NWAVE * NWAVE;    // 1 KiB wave-instructions per halo, a multiple of the waves
    static constexpr int HALO_BYTES = HINSTR * 1024;
    static constexpr int WG_BYTES = KW * WTAP;                            // one (dt, dh) group of 3 taps
    static constexpr int NWB = HALO_BYTES + 3 * WG_BYTES <= 80 * 1024 ? 3 : 2;      // weight ring (two workgroups share the CU's 160 KiB)
    static constexpr int EROW = NT * 64;                                  // bytes of one pixel's NB channels in the epilogue staging block
    static constexpr int MAIN_BYTES = HALO_BYTES + NWB * WG_BYTES, EPI_BYTES = NWAVE * 32 * EROW;
    static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
};
}  // namespace halo

template <int KT, int KH, int TH, int TW, int NT, int MT, int SD = 1>
__global__ __launch_bounds__((MT == 3 ? 256 : 512 / MT), (MT == 1 ? 4 : 2)) void conv_halo_kernel(ConvArgs p) {      // two workgroups per CU: 128 (MT = 1) / 256 (MT = 2) VGPRs
#if defined(__HIP_DEVICE_COMPILE__)     // (the buffer-resource builtins exist only in the device pass; the host pass needs just the stub)
    using namespace halo;
    typedef bf16_t T;
    using C = Cfg<KT, KH, TH, TW, NT, MT, SD>;
    unsigned long long ts[4] = {0, 0, 0, 0}, rt0 = 0;
    if (M4D_ABL(p) & 64) { ts[0] = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    static_assert(SD == 1 || (SD == 2 && KT == 1), "stride 2: the 2-D down-sampling conv");
    constexpr int NWAVE = C::NWAVE, NB = C::NB, WTAP = C::WTAP, HH = C::HH, PITCH = C::PITCH, NPIX = C::NPIX, HINSTR = C::HINSTR;
    constexpr int HALO_BYTES = C::HALO_BYTES, WG_BYTES = C::WG_BYTES, NWB = C::NWB, EROW = C::EROW;
    constexpr int HPW = HINSTR / NWAVE;
    constexpr int WINSTR = WG_BYTES / 1024;                        // 9 / 12
    constexpr int WPW = (WINSTR + NWAVE - 1) / NWAVE;              // weight DMA instructions per wave (upper bound)
    constexpr int NG = KT * KH;
    static_assert(NG >= 2, "the halo reload assumes a weight group follows it");
    constexpr int ROWS_PER_MT = 32 / TW > 0 ? 32 / TW : 1;         // image rows covered by one 32-pixel MFMA tile (TW = 32: 1, TW = 16: 2)
    constexpr int ESW = NT % 2 == 0 ? 7 : 3;                       // epilogue swizzle mask: the XOR must stay inside the pixel's NT*4 chunks
    static_assert(TH * TW == NWAVE * MT * 32 && (TW == 32 || TW == 16), "256 (MT = 3: 384) output pixels per workgroup");
    static_assert(NT >= 1 && NT <= 4, "two workgroups per CU: <= 80 KiB of LDS (launch_halo sizes it for the epilogue blocks too)");

    const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tn = bid % p.tiles_n;
    int tm = bid / p.tiles_n;
    const int tw = tm % tiles_w; tm /= tiles_w;
    const int th = tm % tiles_h;
    const int to = tm / tiles_h;
    const int h0 = th * TH, w0 = tw * TW;
    const int n0 = tn * NB;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, hi = lane >> 5;

    // ---- per-lane DMA sources: 32-bit byte offsets into raw buffer descriptors of x / w.  `buffer_load_dwordx4 ... lds` writes ZEROS
    // for an offset past the end of the buffer (probed: tools/probes/buffer_lds_oob.hip), so halo pixels outside the image are
    // just offset -1: no zero page, no select, no 64-bit per-lane addresses ----
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), (short)0,
        (int)(p.xplane ? (int64_t)(p.Cin / 16 - 1) * p.xplane * 2 + (int64_t)p.Tin * p.Hin * p.Win * 32 : (int64_t)p.Tin * p.Hin * p.Win * p.xs * 2),
        0x00027000);                                       // (physical extent, < 2 GiB)
    // weights: the plain [Cout][taps][Cin] order (every 16-byte piece of a DMA request from another 2 K-byte row: 32 cache lines per
    // request) or, when the caller passes them, the TILED order of m4d_conv_pack_weights — [Cout / 32][Cin / 16][taps] units of 1 KiB,
    // each the LDS image (32 rows x 32 bytes, swizzled) of one tap's 32-channel tile, so that a request is one contiguous KiB
    const bool tiled = p.wt != nullptr;
    const int ntaps = KT * KH * KW, nchunk_w = p.Cin / CK;
    const __amdgpu_buffer_rsrc_t rw = tiled
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), (short)0, (int)(((p.Cout + 31) / 32) * nchunk_w * ntaps * 1024), 0x00027000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), (short)0, (int)(p.Cout * p.K * 2), 0x00027000);
    const int plane_bytes = (int)(p.xplane * 2);            // planar-16 input: [Cin/16][rows >= Tin*Hin*Win][16], p.xplane elements between planes
    int hoff[HPW];
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        const int q = (i * NWAVE + wave) * 64 + lane;
        const int px = q >> 1, physc = q & 1;
        const int fdt = px / (HH * PITCH), r = px % (HH * PITCH);
        const int hh = r / PITCH, ww = r % PITCH;
        const int c = physc ^ ((ww >> 3) & 1);
        // logical coordinates: frame ti of Tin << tsplit (frame f = channels (f & 1) * Cin + [0, Cin) of physical frame f >> 1,
        // wan_vae.py:138-141), pixel (hi_, wi) of the nearest-exact 2x up-sampled map when ups (:61-67)
        int ti = to + fdt, hi_ = h0 - 1 + hh, wi = w0 - 1 + ww;
        bool ok = px < NPIX && ww < TW + KW - 1 && ti < (p.Tin << p.tsplit) && hi_ >= 0 && hi_ < (p.Hin << p.ups) && wi >= 0 &&
                  wi < (p.Win << p.ups);
        if constexpr (SD == 2) {        // slot ww of the de-interleaved row = input column 2 j (+ 1 in the odd half); no padding on top / left
            const int part = ww / C::PE, j = ww % C::PE;
            hi_ = 2 * h0 + hh; wi = 2 * w0 + 2 * j + part;
            ok = px < NPIX && j < TW + 1 - part && hi_ < p.Hin && wi < p.Win;
        }
        const int64_t pix = ((int64_t)(ti >> p.tsplit) * p.Hin + (hi_ >> p.ups)) * p.Win + (wi >> p.ups);
        hoff[i] = ok ? (int)(pix * (p.xplane ? 32 : p.xs * 2)) + (p.tsplit ? (ti & 1) * p.Cin * 2 : 0) + c * 16 : -1;
    }
    int woff[WPW];          // (weights of one layer are far below 2 GiB)
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int q = (i * NWAVE + wave) * 64 + lane;                   // 16-byte slot inside the group: tap, row, physical chunk
        const int tig = q / (NB * 2), n = (q % (NB * 2)) >> 1, physc = q & 1;
        const int c = physc ^ ((n >> 3) & 1);
        const int64_t row = min(n0 + n, p.Cout - 1);
        woff[i] = (int)((row * p.K + tig * p.Cin + c * 8) * 2);
        // (tiled: the request is the unit of (row block tn * NT + n / 32, chunk 0, tap tig) — a block past the last one reads zeros)
        if (tiled) woff[i] = (((tn * NT + (n >> 5)) * nchunk_w) * ntaps + tig) * 1024 + lane * 16;
    }
    auto issue_halo = [&](int ck0) {
        char* dst = conv_dyn_smem;
        const int soff = p.xplane ? (ck0 >> 4) * plane_bytes : ck0 * 2;
#pragma unroll
        for (int i = 0; i < HPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (LDS_AS void*)(dst + (i * NWAVE + wave) * 1024), 16, hoff[i], soff, 0, 0);
    };
    auto issue_w = [&](int buf, int ck0, int g) {
        if (M4D_ABL(p) & 4) return;
        char* dst = conv_dyn_smem + HALO_BYTES + buf * WG_BYTES;
        const int koff = tiled ? ((ck0 >> 4) * ntaps + g * KW) * 1024 : (g * KW * p.Cin + ck0) * 2;
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            if (i * NWAVE + wave < WINSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (LDS_AS void*)(dst + (i * NWAVE + wave) * 1024), 16, woff[i], koff, 0, 0);
    };

    // ---- per-lane fragment addresses ----
    // MFMA "B" operand (activations): lane (li, hi) = pixel li of this wave's 32-pixel tile, 8 channels of chunk hi
    unsigned abase[MT][KW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int mt = wave * MT + mi;
        const int row = TW == 32 ? mt : mt * ROWS_PER_MT + li / TW;
        const int col = TW == 32 ? li : li % TW;
#pragma unroll
        for (int dw = 0; dw < KW; ++dw) {
            const int cc = SD == 2 ? (dw == 1 ? C::PE + col : col + (dw >> 1)) : col + dw;
            abase[mi][dw] = (unsigned)((row * SD * PITCH + cc) * PXB + ((hi ^ ((cc >> 3) & 1)) << 4));
        }
    }
    // MFMA "A" operand (weights): row n = ni*32 + li of the tap's [NB x 16] tile; further column tiles are +32 rows = +1024 bytes
    const unsigned wfrag = (unsigned)(li * PXB + ((hi ^ ((li >> 3) & 1)) << 4));

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS char*)conv_dyn_smem;
    const int nchunk = p.Cin / CK;
    const int total = nchunk * NG;
    // weight groups stream through a ring of NWB buffers, requested NWB-1 steps ahead (one step = MT*NT*3 MFMAs per wave is shorter
    // than an L2 round trip); jc/jg/jb = chunk, group and buffer of the next group to request
    int jc = 0, jg = 0, jb = 0;
    const bool wfull = WINSTR % NWAVE == 0 || wave < WINSTR % NWAVE;       // this wave issues WPW (else WPW - 1) pieces per group
    auto next_w = [&]() {
        if (jc < nchunk) issue_w(jb, jc * CK, jg);
        jb = jb + 1 == NWB ? 0 : jb + 1;
        if (++jg == NG) { jg = 0; ++jc; }
    };
    // wait until at most `n` of my weight groups are still in flight (everything older - halo pieces included - has landed)
#define HL_WAIT_W(n)                                                                                             \
    do {                                                                                                         \
        if (wfull) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) * WPW) : "memory");                              \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) * (WPW - 1)) : "memory");                              \
    } while (0)
    next_w();
    issue_halo(0);
#pragma unroll
    for (int d = 1; d < NWB - 1; ++d) next_w();
    int s = 0, rb = 0;
    if (M4D_ABL(p) & 64) ts[1] = __builtin_readcyclecounter();
#pragma unroll 1
    for (int ci = 0; ci < nchunk; ++ci) {
#pragma unroll 1
        for (int g = 0; g < NG; ++g, ++s) {       // (kept rolled: unrolled, hipcc hoists every fragment address and spills)
            // my share of weight group s (and of the first halo): group s+1 may stay in flight (NWB = 3)
            if (NWB == 2 || s + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else HL_WAIT_W(1);
            if (!(M4D_ABL(p) & 16)) __builtin_amdgcn_s_barrier();
            if (g == 0 && ci > 0) {
                // every wave is done with the previous chunk's halo: overwrite it (the co-resident workgroup computes meanwhile)
                if (!(M4D_ABL(p) & 8)) issue_halo(ci * CK);
                const bool more = jc < nchunk;
                next_w();
                // the halo pieces are older than that weight group: leave only it in flight
                if (more) HL_WAIT_W(1);
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(M4D_ABL(p) & 16)) __builtin_amdgcn_s_barrier();
            } else {
                next_w();
            }
            const int fdt = g / KH, dh = g % KH;
            const unsigned hb = lds_base + (unsigned)((fdt * HH + dh) * PITCH * PXB);
            const unsigned wb = lds_base + HALO_BYTES + rb * WG_BYTES + wfrag;
            rb = rb + 1 == NWB ? 0 : rb + 1;
            // MT pixel fragments + NT weight fragments feed MT x NT MFMAs (LDS reads per MFMA: 1.33 at 1 x 3, 0.83 at 2 x 3, 0.75 at
            // 2 x 4 - at one read per MFMA the LDS array is as busy as the MFMA pipe); the fragments of tap dw+1 are requested before
            // the MFMAs of tap dw (counted lgkmcnt): the LDS latency is exposed once per step, not once per tap
            bf16x8 fa[2][MT], fw[2][NT];
#define HL_LD(buf, dw)                                                                                          \
            do {                                                                                                \
                _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) {                                             \
                    const unsigned a_ = hb + abase[mi][dw];                                                     \
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[buf][mi]) : "v"(a_));                          \
                }                                                                                               \
                _Pragma("unroll") for (int ni = 0; ni < NT; ++ni) {                                             \
                    const unsigned w_ = wb + (dw) * WTAP + ni * 1024;                                           \
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fw[buf][ni]) : "v"(w_));                          \
                }                                                                                               \
            } while (0)
            if (!(M4D_ABL(p) & 2)) HL_LD(0, 0);
#pragma unroll
            for (int dw = 0; dw < KW; ++dw) {
                if (!(M4D_ABL(p) & 2)) {
                    if (dw == 0) { HL_LD(1, 1); }
                    else if (dw == 1) { HL_LD(0, 2); }
                    if (dw < KW - 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT + MT) : "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        if (!(M4D_ABL(p) & 1)) mma32(fw[dw & 1][ni], fa[dw & 1][mi], acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef HL_LD
        }
    }

#include "conv_halo_epi.inc"
#endif
}
