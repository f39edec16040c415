// conv_halo64_kernel — the 3 x 3 x 3 convolution of the VAE's 96-channel tiles as ONE WAVE PER SIMD (included by conv.hip behind
// conv_halo.h).  Same problem, same LDS-halo idea and same epilogue as conv_halo_kernel<3, 3, 12, 32, 3, 3> (conv_halo.h), which it
// replaces where it applies; the main loop is a generated, hand-placed instruction stream (tools/gen_conv_halo64.py, where the design
// and the reasoning are documented): 2 waves per workgroup (a 10 x 32 patch), every wave 5 pixel tiles x 3 channel tiles = 15
// accumulators in 240 AGPRs (8 fragment reads per 15 MFMAs instead of 6 per 9), halo slabs (one frame of one 16-channel chunk) and
// weight groups through rings of three, no LDS wait inside the MFMA stream.  Two workgroups share a CU = one wave per SIMD.
// Replaces: CausalConv3d (MoRe4D/models/wan_vae.py:21-40) of the residual blocks (:190-224) at 96 / 192 / 384 output channels.
// Restrictions (host-checked, everything else stays on conv_halo_kernel): kt = kh = kw = 3, stride 1, no fused up-sampling / time
// split, Cout % 96 == 0, bf16, no GroupNorm statistics.
#pragma once

#ifndef M4D_CV64_INC
#define M4D_CV64_INC "conv_halo64_gen.inc"
#endif

#define M4D_CV64_VCLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define M4D_CV64_ACLOB "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define M4D_CV64_SCLOB "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101"

#ifndef M4D_CV64K1_INC
#define M4D_CV64K1_INC "conv_halo64k1_gen.inc"
#endif

// LDS geometry of one instantiation (KT = 3: MT = 5, NT = 3, the 3 x 3 x 3 conv on 10 x 32 patches of 96-channel tiles; KT = 1: MT = 4, NT = 4, the
// 3 x 3 conv on 8 x 32 patches of 128-channel tiles — the adaptors' convs, same patch as conv_halo_kernel<1, 3, 8, 32, 4, 2> so that the
// per-patch GroupNorm statistics land in the same blocks)
template <int KT, int MT, int NT>
struct Halo64Cfg {
    static constexpr int NWAVE = 2, TH = NWAVE * MT, TW = 32, PITCH = 40, HH = TH + 2;
    static constexpr int SLAB = MT == 5 ? 16384 : 14336;          // one frame of one chunk, a whole number of KiB pieces per wave
    static constexpr int WG_BYTES = 3 * NT * 1024, W_RING = 3 * WG_BYTES, LDS_BYTES = W_RING + 3 * SLAB;
    static_assert(HH * PITCH * 32 <= SLAB && NWAVE * 32 * NT * 64 <= LDS_BYTES && LDS_BYTES <= 80 * 1024, "slab / epilogue staging must fit half a CU");
    static_assert((KT == 3 && MT == 5 && NT == 3) || (KT == 1 && MT == 4 && NT == 4), "generated streams: tools/gen_conv_halo64.py --shape");
};
namespace halo64 { constexpr int TH = 10, LDS_BYTES = Halo64Cfg<3, 5, 3>::LDS_BYTES; }

template <int KT, int MT, int NT>
__global__ __launch_bounds__(128, 1) void conv_halo64_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace halo;
    typedef bf16_t T;
    using C64 = Halo64Cfg<KT, MT, NT>;
    constexpr int NWAVE = C64::NWAVE, TH = C64::TH, TW = C64::TW, PITCH = C64::PITCH, HH = C64::HH;
    constexpr int NB = NT * 32, EROW = NT * 64, ESW = NT % 2 == 0 ? 7 : 3, ROWS_PER_MT = 1, NTAPS = KT * 9;
    constexpr int NSP = C64::SLAB / 2048, NWP = (C64::WG_BYTES / 1024 + 1) / 2;       // DMA pieces per wave: slab / weight group
    unsigned long long ts[4] = {0, 0, 0, 0}, rt0 = 0;
    (void)ts; (void)rt0;
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
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)conv_dyn_smem;

    // raw buffer resources as in conv_halo_kernel (base, stride 0, extent in bytes, flags 0x00027000): a request past the end of the
    // buffer writes ZEROS — halo pixels outside the image are just offset -1.  Built word by word: the stream takes them in SGPRs.
    const int pixb = p.xplane ? 32 : (int)(p.xs * 2);
    const unsigned long long xb = (unsigned long long)p.x;
    const unsigned rx2 = (unsigned)(p.xplane ? (int64_t)(p.Cin / 16 - 1) * p.xplane * 2 + (int64_t)p.Tin * p.Hin * p.Win * 32
                                              : (int64_t)p.Tin * p.Hin * p.Win * p.xs * 2);
#ifdef M4D_CV64_GATHER
    const unsigned long long wbp = (unsigned long long)p.w;
    const unsigned rw2 = (unsigned)(p.Cout * p.K * 2);
#else
    const unsigned long long wbp = (unsigned long long)p.wt;
    const unsigned rw2 = (unsigned)((p.Cout / 32) * (p.Cin / CK) * NTAPS * 1024);
#endif

    // ---- lane table (LDS offset 0, 32 dwords per work item): AB[5][3], WF, HO[8], WO[5] ----
    {
        unsigned* tab = reinterpret_cast<unsigned*>(conv_dyn_smem) + t * 32;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int row = wave * MT + mi;
#pragma unroll
            for (int dw = 0; dw < KW; ++dw) {
                const int cc = li + dw;
                tab[mi * 3 + dw] = lds0 + C64::W_RING + (unsigned)((row * PITCH + cc) * PXB + ((hi ^ ((cc >> 3) & 1)) << 4));
            }
        }
        tab[15] = lds0 + (unsigned)(li * PXB + ((hi ^ ((li >> 3) & 1)) << 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) tab[16 + i] = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < NSP; ++i) {      // slab piece 2 i + wave: 64 x 16-byte slots; slot q = pixel q >> 1, physical chunk q & 1
            const int q = (i * NWAVE + wave) * 64 + lane;
            const int px = q >> 1, physc = q & 1;
            const int hh = px / PITCH, ww = px % PITCH;
            const int c = physc ^ ((ww >> 3) & 1);
            const int hi_ = h0 - 1 + hh, wi = w0 - 1 + ww;
            const bool ok = hh < HH && ww < TW + KW - 1 && hi_ >= 0 && hi_ < p.Hin && wi >= 0 && wi < p.Win;
            tab[16 + i] = ok ? (unsigned)(((int64_t)hi_ * p.Win + wi) * pixb) + (unsigned)(c * 16) : 0xffffffffu;
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i) {      // weight piece 2 i + wave of a (dt, dh) group (nine pieces: wave 1's fifth repeats piece 8)
            int piece = i * NWAVE + wave;
            if (piece > C64::WG_BYTES / 1024 - 1) piece = C64::WG_BYTES / 1024 - 1;
            const int q = piece * 64 + lane;
            const int tig = q / (NB * 2), n = (q % (NB * 2)) >> 1;
#ifdef M4D_CV64_GATHER          // (tool builds of the first version: plain [Cout][27][Cin] weights, conv_halo64 --gather stream)
            const int c = (q & 1) ^ ((n >> 3) & 1);
            const int64_t row = min(n0 + n, p.Cout - 1);
            tab[24 + i] = (unsigned)((row * p.K + tig * p.Cin + c * 8) * 2);
#else                           // tiled weights (m4d_conv_pack_weights): the piece is the KiB of (row block tn * 3 + n / 32, chunk 0, tap tig)
            tab[24 + i] = (unsigned)((((tn * NT + (n >> 5)) * (p.Cin / CK)) * NTAPS + tig) * 1024 + lane * 16);
#endif
        }
#pragma unroll
        for (int i = 24 + NWP; i < 32; ++i) tab[i] = 0;
    }
    __syncthreads();

    const unsigned frb = (unsigned)((int64_t)p.Hin * p.Win * pixb);
    const unsigned cho = (unsigned)to * frb;                                   // chunk 0, frame `to` (+ dt frames inside the stream)
    const unsigned chb = p.xplane ? (unsigned)(p.xplane * 2) : 32u;            // next 16-channel chunk: next plane / next 32 bytes of a pixel
    const unsigned gs = (unsigned)(KW * p.Cin * 2);                            // next (dt, dh) group of three taps
    const unsigned nch = (unsigned)(p.Cin / CK);
    f32x16 acc[MT][NT];
#define M4D_U(x) __builtin_amdgcn_readfirstlane((unsigned)(x))
    if constexpr (KT == 3) {
        asm volatile(
#include M4D_CV64_INC
            : "={a[0:15]}"(acc[0][0]), "={a[16:31]}"(acc[0][1]), "={a[32:47]}"(acc[0][2]), "={a[48:63]}"(acc[1][0]), "={a[64:79]}"(acc[1][1]), "={a[80:95]}"(acc[1][2]), "={a[96:111]}"(acc[2][0]), "={a[112:127]}"(acc[2][1]), "={a[128:143]}"(acc[2][2]), "={a[144:159]}"(acc[3][0]), "={a[160:175]}"(acc[3][1]), "={a[176:191]}"(acc[3][2]), "={a[192:207]}"(acc[4][0]), "={a[208:223]}"(acc[4][1]), "={a[224:239]}"(acc[4][2])
          : [tid] "v"(threadIdx.x), [rx0] "s"(M4D_U(xb)), [rx1] "s"(M4D_U((xb >> 32) & 0xffff)), [rx2] "s"(M4D_U(rx2)), [rx3] "s"(M4D_U(0x00027000u)),
            [rw0] "s"(M4D_U(wbp)), [rw1] "s"(M4D_U((wbp >> 32) & 0xffff)), [rw2] "s"(M4D_U(rw2)), [rw3] "s"(M4D_U(0x00027000u)),
            [cho] "s"(M4D_U(cho)), [frb] "s"(M4D_U(frb)), [chb] "s"(M4D_U(chb)), [gs] "s"(M4D_U(gs)), [nch] "s"(M4D_U(nch)), [lds0] "s"(M4D_U(lds0))
            : "memory", "vcc", "scc", "m0", M4D_CV64_SCLOB, M4D_CV64_VCLOB, M4D_CV64_ACLOB);
    } else {
        asm volatile(
#include M4D_CV64K1_INC
            : "={a[0:15]}"(acc[0][0]), "={a[16:31]}"(acc[0][1]), "={a[32:47]}"(acc[0][2]), "={a[48:63]}"(acc[0][3]), "={a[64:79]}"(acc[1][0]), "={a[80:95]}"(acc[1][1]), "={a[96:111]}"(acc[1][2]), "={a[112:127]}"(acc[1][3]), "={a[128:143]}"(acc[2][0]), "={a[144:159]}"(acc[2][1]), "={a[160:175]}"(acc[2][2]), "={a[176:191]}"(acc[2][3]), "={a[192:207]}"(acc[3][0]), "={a[208:223]}"(acc[3][1]), "={a[224:239]}"(acc[3][2]), "={a[240:255]}"(acc[3][3])
          : [tid] "v"(threadIdx.x), [rx0] "s"(M4D_U(xb)), [rx1] "s"(M4D_U((xb >> 32) & 0xffff)), [rx2] "s"(M4D_U(rx2)), [rx3] "s"(M4D_U(0x00027000u)),
            [rw0] "s"(M4D_U(wbp)), [rw1] "s"(M4D_U((wbp >> 32) & 0xffff)), [rw2] "s"(M4D_U(rw2)), [rw3] "s"(M4D_U(0x00027000u)),
            [cho] "s"(M4D_U(cho)), [frb] "s"(M4D_U(frb)), [chb] "s"(M4D_U(chb)), [gs] "s"(M4D_U(gs)), [nch] "s"(M4D_U(nch)), [lds0] "s"(M4D_U(lds0))
            : "memory", "vcc", "scc", "m0", M4D_CV64_SCLOB, M4D_CV64_VCLOB);
    }
#undef M4D_U

#include "conv_halo_epi.inc"
#endif
}
