/* more4d_hip.h — C ABI of libmore4d_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * 4D-STraG denoising hot path of MoRe4D (Wan2.1-DiT forward, Motion-Sensitive 3D-VAE, Euler/CFG loop).
 *
 * The reference has no FFI for this path: its boundary is the Python operator surface of
 * MoRe4D/models/wan_transformer4d.py, wan_vae.py, trajectory_module.py and the loop in
 * MoRe4D/pipeline/pipeline_wan_fun_control.py.  Each entry point below names the
 * reference lines whose arithmetic it replaces.  Host code (more4d_amd/_lib.py) binds these with
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch); the library never allocates,
 *     frees or keeps them past the enqueued work;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); work is enqueued,
 *     never synchronised;
 *   - return 0 on success, <0 on error (-1 invalid argument, -2 unsupported shape, -3 launch
 *     failure); m4d_last_error() returns a thread-local message; nothing throws across the ABI;
 *   - dtype enums select the activation/weight element type `T`: M4D_BF16 is the production path
 *     (bf16 operands, fp32 accumulation, casts placed where the reference's autocast puts them),
 *     M4D_F32 is the parity path (exact-fp32 MFMA, checked against the CPU oracle to 1e-3).
 */
#ifndef MORE4D_HIP_H
#define MORE4D_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { M4D_F32 = 0, M4D_BF16 = 1 } m4d_dtype;
typedef void* m4d_stream;

int m4d_version(void);
/* sha256 (hex) of the csrc/ + include/ files the library was compiled from; more4d_amd/_lib.py refuses to load a
 * library whose hash differs from the tree's, more4d_amd/build.py rebuilds on a mismatch (no reference counterpart). */
const char* m4d_source_hash(void);
const char* m4d_last_error(void);

/* ---- diagnostics: which kernel structure the calls of this process dispatched to ----
 * Every launcher counts its launches per kernel class (host side, process wide).  Tests use the counters to assert that a
 * module-level parity case against the reference's outputs really ran the PRODUCTION tile paths (the reference has no counterpart:
 * its kernels are chosen by cuDNN / flash-attn). */
typedef enum {
    M4D_KC_GEMM_PHASED = 0,        /* gemm_bt256p_kernel: 8 waves, two phased groups */
    M4D_KC_GEMM_WIDE = 1,          /* gemm_bt256w_kernel: 4 waves x 128x128, AGPR accumulators */
    M4D_KC_GEMM_GENERIC = 2,       /* 128x128 register-staged (small / ragged / fp32) and the A/B structures */
    M4D_KC_ATTN_PHASED = 3,        /* attn128p_kernel */
    M4D_KC_ATTN_OTHER = 4,
    M4D_KC_CONV_HALO_MT3_12X32 = 5,/* conv_halo_kernel, three pixel tiles per wave, 12 x 32 patches */
    M4D_KC_CONV_HALO_MT3_24X16 = 6,/* ... 24 x 16 patches */
    M4D_KC_CONV_HALO = 7,          /* conv_halo_kernel, two pixel tiles per wave (8 x 32 / 16 x 16 patches) */
    M4D_KC_CONV_GENERIC = 8,       /* conv_cl256_kernel / conv_cl_kernel */
    M4D_KC_CONV_FUSED_NORM = 9,    /* a conv launch that wrote the next layer's RMS_norm(+SiLU) from its epilogue */
    M4D_KC_CONV_FUSED_NORM_RESID = 10, /* ... of a residual block's conv2 (shortcut added, norm of the NEXT block / head) */
    M4D_KC_CONV_GNSTATS = 11,      /* a conv launch that emitted GroupNorm partial statistics from its epilogue */
    M4D_KC_ATTN_BWD128 = 12,       /* one pass of the production attention backward (attn_bwd128_kernel: dQ / dK / dV, bf16, head_dim 128) */
    M4D_KC_ATTN_BWD_GENERIC = 13,  /* one pass of the generic two-pass backward (fp32 / other head dims) */
    M4D_KC_ATTN_XP = 14,           /* attn128x_kernel: persistent pipeline over (query tile, key tile) pairs for short key lists (cross-attention) */
    M4D_KC_ATTN_Q64 = 15,          /* attn128q_kernel: one wave per SIMD, 4 x 64 query rows, generated instruction stream (long self-attention) */
    M4D_KC_ATTN_BWD64 = 16,        /* attn_bwd_*64_kernel: one pass of the attention backward as one wave per SIMD, generated instruction stream */
    M4D_KC_CONV_HALO64 = 17,       /* conv_halo64_kernel: the 3x3x3 conv of 96-channel tiles as one wave per SIMD, generated main loop */
    M4D_KC_COUNT = 18
} m4d_kernel_class;
/* launches of `kernel_class` since process start (or the last reset); reset != 0 clears that counter after reading it;
 * kernel_class < 0 with reset != 0 clears all counters and returns 0. */
int64_t m4d_launch_count(int kernel_class, int reset);

/* ---- GEMM epilogues (m4d_gemm_bt) ---- */
typedef enum {
    M4D_EPI_STORE = 0,      /* out[T]   = acc + bias                                   nn.Linear */
    M4D_EPI_GELU_TANH = 1,  /* out[T]   = gelu_tanh(acc + bias)          wan_transformer4d.py:620-622, 900-902 */
    M4D_EPI_GELU_ERF = 2,   /* out[T]   = gelu_erf(acc + bias)           MLPProj :729-732 */
    M4D_EPI_SILU = 3,       /* out[T]   = silu(acc + bias)               time_embedding :904-906 */
    M4D_EPI_RESID_GATE = 4, /* resid[f32] += round_T(acc + bias) * gate[sample(m), n]   :669, :674, :684 */
    M4D_EPI_STORE_F32 = 5   /* out[f32] = round_T(acc + bias)            patch_embedding :1073, head :720 */
} m4d_epilogue;

/* C[M,N] = A[M,K] * W[N,K]^T (+ bias) with a fused epilogue.  Replaces every nn.Linear and the
 * kernel==stride convs (patch_embedding :898-899, ref_conv :946) of the DiT, the 1x1(x1) convs of the
 * VAE's attention block (its 3x3(x3) convs are m4d_conv_cl).
 *   A: T [M,K] row stride lda; W: T [N,K] row stride ldw (nn.Linear weight layout);
 *   bias: T [N] (or T [M] when bias_on_m != 0), may be NULL;
 *   out: T [M,N] (STORE/GELU/SILU) or float [M,N] (STORE_F32), row stride ldc;
 *   resid: float [M,N] row stride ldc, updated in place (RESID_GATE); gate: float, element
 *   [ (m / rows_per_sample) * gate_stride + n ], NULL => 1.
 * Requirements: K*sizeof(T) % 16 == 0, N % 4 == 0, lda/ldw*sizeof(T) % 16 == 0, ldc % 4 == 0.
 * Side effects besides the launch: large bf16 STORE / GELU_TANH problems (more 256 x 256 tiles than CUs) run as a persistent kernel whose
 * workgroups pace each other per XCD through eight arrival counters in a 1 KiB device buffer the library allocates on its first such
 * call outside a stream capture and clears with a hipMemsetAsync on `stream` in front of every such launch;
 * the counters are a performance hint only (two such GEMMs running concurrently on different streams lose the hint, not the result).
 * M4D_GEMM_SYNC=0 / M4D_GEMM_PERSIST=0 in the environment switch the hint / the persistent form off. */
int m4d_gemm_bt(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                int bias_on_m, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                const float* gate, int64_t gate_stride, int64_t rows_per_sample, m4d_stream stream);
/* m4d_gemm_bt with a caller-owned workspace (bytes from m4d_gemm_bt_workspace_bytes; NULL or too small = the plain single launch).
 * When the 256 x 256 tile grid leaves a partial last round on the 256 CUs (M = 43 680, N = 5120: 3 420 tiles = 13.36 rounds), the
 * tiles of that round are split along K so that the whole chip works on them; a small kernel sums the float32 slabs in a fixed
 * order (deterministic) and applies the epilogue.  Same results up to the summation order of the split tiles.  Enabled only with
 * M4D_GEMM_TAIL=1 (measured: no gain at the power limit, see csrc/gemm.hip); otherwise identical to m4d_gemm_bt. */
int64_t m4d_gemm_bt_workspace_bytes(m4d_dtype dt, int64_t M, int64_t N, int64_t K);
int m4d_gemm_bt_ws(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, int bias_on_m, void* out,
                   int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, const float* gate, int64_t gate_stride,
                   int64_t rows_per_sample, void* ws, int64_t ws_bytes, m4d_stream stream);

/* Weight pre-shuffle for the production GEMM.  m4d_pack_frag re-lays a row-major bf16 matrix W[rows, K] (an nn.Linear
 * weight) out ONCE into MFMA fragment order: out[rb = row/32][kb = k/16][lane][8], lane (li, hi) holding
 * W[rb*32+li][kb*16+hi*8 .. +8); rows are zero-padded to 32 (out has m4d_pack_frag_elems(rows, K) elements).
 * m4d_gemm_bt_packed is m4d_gemm_bt with that operand packed: packed_side 0 = W (the N side, ordinary Linear),
 * 1 = A (the M side: V^T = W_v x^T); the other operand stays row-major.  The packed fragments stream straight into
 * VGPRs, only the activation tile uses the global->LDS DMA path (the measured limiter of the unpacked kernel).
 * bf16 only, K % 64 == 0. */
int64_t m4d_pack_frag_elems(int64_t rows, int64_t K);
int m4d_pack_frag(m4d_dtype dt, const void* W, int64_t ld, void* out, int64_t rows, int64_t K, m4d_stream stream);
int m4d_gemm_bt_packed(m4d_dtype dt, const void* A, int64_t lda, const void* W, int64_t ldw, int packed_side,
                       const void* bias, int bias_on_m, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K,
                       int epilogue, const float* gate, int64_t gate_stride, int64_t rows_per_sample, m4d_stream stream);

/* LayerNorm (no affine | affine) * (1 + scale) + shift, optional spatial guidance, cast to T.
 * Replaces WanLayerNorm + modulation (:662, :677, :720), norm3 (:611-613, :674), the LayerNorms of
 * MLPProj (:729-732) and SpatialGuidanceModule.forward (:757-783).
 *   x: x_dt [rows, C] contiguous; out: out_dt [rows, C];
 *   shift/scale: float, element [(row / rows_per_sample) * mod_stride + c], both NULL => no modulation;
 *   ln_w/ln_b: float [C] or NULL;
 *   guidance (all NULL/0 to disable): g_ss float [B, g_period, 2C] = (scale | shift) per spatial
 *   position, g_gate float [C]; token l of a sample uses position l % g_period when l < g_len and
 *   zero scale/shift otherwise (the reference zero-pads, :772-776).
 * Requirements: C % 4 == 0, C <= 8192. */
int m4d_ln_modulate(m4d_dtype x_dt, const void* x, m4d_dtype out_dt, void* out, int64_t rows, int C,
                    int64_t rows_per_sample, const float* shift, const float* scale, int64_t mod_stride,
                    const float* ln_w, const float* ln_b, float eps, const float* g_ss,
                    const float* g_gate, int64_t g_period, int64_t g_len, m4d_stream stream);

/* m4d_ln_modulate with the guidance table indexed independently of the modulation: g_rows = rows per GUIDANCE sample (0: rows_per_sample).
 * Per-token timesteps (wan_transformer4d.py:655-657: one modulation vector per row, rows_per_sample = 1) together with spatial
 * guidance (:757-783: table row = token position inside its sample) need both. */
int m4d_ln_modulate_g(m4d_dtype x_dt, const void* x, m4d_dtype out_dt, void* out, int64_t rows, int C,
                      int64_t rows_per_sample, const float* shift, const float* scale, int64_t mod_stride,
                      const float* ln_w, const float* ln_b, float eps, const float* g_ss,
                      const float* g_gate, int64_t g_period, int64_t g_len, int64_t g_rows, m4d_stream stream);

/* In-place WanRMSNorm over the full channel dim (:386-394) followed by 3-axis RoPE on adjacent pairs
 * (:340-369) for up to two tensors (q and k) in one launch.
 *   x0/x1: T [rows, C] row stride ld (x1 may be NULL); w0/w1: float [C], both NULL => RoPE only (qk_norm=False, :431-432);
 *   cos/sin: float [table_rows, head_dim/2], per-token tables built by the host for the (f,h,w) grid
 *   (NULL => no RoPE, cross-attention q/k); token l = row % rows_per_sample gets table row
 *   pos_offset + l when l < rope_len, rows past rope_len are normalised but not rotated (:365). */
int m4d_rmsnorm_rope(m4d_dtype dt, void* x0, void* x1, int64_t ld, const float* w0, const float* w1,
                     int64_t rows, int C, int head_dim, float eps, const float* cos_t, const float* sin_t,
                     int64_t rows_per_sample, int64_t rope_len, int64_t pos_offset, m4d_stream stream);

/* Non-causal softmax(q k^T * scale) v, flash style (never materialises the score matrix).
 * Replaces attention()/flash_attention() (:66-236) as used by WanSelfAttention (:455-461) and
 * WanI2VCrossAttention (:533-552).  K/V may arrive as up to M4D_MAX_KV_SEGS segments (the T-sharded
 * denoise loop hands over one segment per rank after the RCCL all-gather, no concat copy).
 *   q:   T, element (b, l, h, d) at q + b*q_bs + l*q_ls + h*head_dim + d;  out likewise (o_bs, o_ls);
 *   seg s: k_s element (b, j, h, d) at k[s] + b*k_bs[s] + j*k_ls[s] + h*head_dim + d;
 *          V is supplied TRANSPOSED: vt_s element (b, h, d, j) at vt[s] + b*vt_bs[s] + (h*head_dim + d)*vt_ls[s] + j
 *          (the projection GEMM writes V^T directly: m4d_gemm_bt(A = W_v, W = x));
 *          len[s] keys are valid.
 *   accumulate != 0: out = round_T(round_T(o) + out)   (x + img_x, :552).
 * head_dim in {32, 64, 128}; all strides in elements, multiples of 8. */
#define M4D_MAX_KV_SEGS 8
typedef struct {
    const void* k[M4D_MAX_KV_SEGS];
    const void* vt[M4D_MAX_KV_SEGS];
    int64_t k_bs[M4D_MAX_KV_SEGS], k_ls[M4D_MAX_KV_SEGS];
    int64_t vt_bs[M4D_MAX_KV_SEGS], vt_ls[M4D_MAX_KV_SEGS];
    int64_t len[M4D_MAX_KV_SEGS];
    int32_t nseg;
    int32_t new_softmax;   /* bit s set: segment s starts a NEW softmax whose (normalised) output is added to that of the segments
                              before it — WanI2VCrossAttention's text + image branches (wan_transformer4d.py:533-552) in one launch,
                              sharing the query tile; bf16 / head_dim 128 only, not with lse.  0 = one softmax over all segments. */
} m4d_kv_segs;

int m4d_attention(m4d_dtype dt, const void* q, int64_t q_bs, int64_t q_ls, const m4d_kv_segs* kv, void* out,
                  int64_t o_bs, int64_t o_ls, int B, int64_t Lq, int heads, int head_dim, float scale,
                  int accumulate, m4d_stream stream);

/* Patch gather for the kernel==stride convs: out[b, (f,h,w), (c,pt,ph,pw)] = src[b, c, f*pt.., h*ph.., w*pw..]
 * with channels taken from src0 (c0 channels) then src1 (c1 channels; NULL/0 allowed) — the
 * torch.cat([x, y], dim=0 per sample) + Conv3d of :1069-1073, and ref_conv :1087 with F = 1. */
int m4d_patchify(m4d_dtype src_dt, const void* src0, int c0, const void* src1, int c1, m4d_dtype out_dt,
                 void* out, int B, int F, int H, int W, int pt, int ph, int pw, m4d_stream stream);

/* unpatchify (:1343-1366): tok float [B, tok_bs rows.., (pt,ph,pw,c)] starting at row tok_row0 of each
 * sample -> out out_dt [B, c, F*pt, H*ph, W*pw]. */
int m4d_unpatchify(const float* tok, int64_t tok_bs, int64_t tok_row0, m4d_dtype out_dt, void* out, int B,
                   int c, int F, int H, int W, int pt, int ph, int pw, m4d_stream stream);

/* Classifier-free guidance + Euler step in one pass (pipeline_wan_fun_control.py:820-825,
 * fm_solvers.py:415-483 order-1):  x <- x + dsigma * (v_u + g * (v_c - v_u)),  fp32 state.
 *   v: v_dt [2, n] (uncond half first); round_dt: the model dtype the reference casts the sample back
 *   to after the step (fm_solvers.py:789) — M4D_F32 keeps fp32. */
int m4d_cfg_euler(float* x, m4d_dtype v_dt, const void* v, int64_t n, float guidance, float dsigma,
                  m4d_dtype round_dt, m4d_stream stream);

/* Elementwise helpers: out = act(x) with dtype conversion (act: 0 copy/cast, 1 silu, 2 gelu_tanh, 3 gelu_erf). */
int m4d_unary(m4d_dtype in_dt, const void* x, m4d_dtype out_dt, void* out, int64_t n, int act,
              m4d_stream stream);

/* ------------------------------------------------------------------ Motion-Sensitive 3D-VAE (channels-last) */

/* Causal 3-D / 2-D convolution as an implicit GEMM on channels-last activations.  Replaces CausalConv3d
 * (wan_vae.py:21-40), the Conv2d of Resample (:81-100, incl. nn.Upsample nearest-exact x2 and ZeroPad2d((0,1,0,1))),
 * the 1x1 convs (:203, :238-239, :509-510) and the adaptors' Conv2d (trajectory_module.py:73-100, 142, 170, 216, 254).
 *   x:   T [Tin, Hin, Win, *] with `x_pixel_stride` elements between pixels; channels [0, Cin) are used
 *        (tsplit: logical frame f reads physical frame f>>1, channels (f&1)*Cin + [0, Cin) — upsample3d, :138-141);
 *   w:   T [Cout, kt*kh*kw*Cin], K ordered (dt, dh, dw, c)  (the host repacks nn.Conv weights once);
 *   out[(to,ho,wo), co] = bias[co] + sum x[to*st+dt-pad_t, ho*sh+dh-pad_h, wo*sw+dw-pad_w, c] * w[co, (dt,dh,dw,c)]
 *        (+ resid[(to,ho,wo), co]); taps outside [0,Tin<<tsplit) x [0,Hin<<ups) x [0,Win<<ups) read zero;
 *   ups: the H/W indices address a nearest-exact 2x up-sampled view of x (:61-67).
 * The caller provides causality by keeping the conv's 2-frame tail in front of the chunk (pad_t = 0) or by
 * pad_t = 2 for a cold start.  Requirements: Cin*sizeof(T) % 16 == 0, Cout % 4 == 0. */
int m4d_conv_cl(m4d_dtype dt, const void* x, int64_t x_pixel_stride, const void* w, const void* bias,
                const void* resid, int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin,
                int Cout, int kt, int kh, int kw, int st, int sh, int sw, int pad_t, int pad_h, int pad_w, int To,
                int Ho, int Wo, int ups, int tsplit, m4d_stream stream);

/* The same convolution for the stride-1 3x3(x3) layers (kh = kw = 3, pad (0,1,1), To = Tin - kt + 1, bf16) with x in PLANAR-16
 * layout: [Cin/16][rows][16], `x_plane_stride` elements between channel planes, row = (t*Hin + h)*Win + w.  That is the layout
 * m4d_rmsnorm_silu_cl_planar writes into a causal conv's [tail + chunk] staging buffer (wan_vae.py:199-224: norm -> SiLU -> conv):
 * a halo pixel's 16-channel piece sits next to its row neighbours', so the kernel's halo DMA uses every byte of the lines it
 * fetches (channels-last: 32 bytes of each Cin*2-byte pixel per pass).  Input extent < 2 GiB. */
int m4d_conv_cl_planar(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* bias, const void* resid,
                       int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout, int kt, int To,
                       m4d_stream stream);
/* ... with the NEXT layer's RMS_norm (+SiLU) fused into the epilogue (a ResidualBlock's conv -> norm -> SiLU -> conv, wan_vae.py:199-224):
 * norm_out (planar-16, rows [0, To*Hin*Win)) = rmsnorm_silu(conv result (+resid)) with `norm_gamma` float [Cout]; `out` may be NULL
 * when nobody reads the un-normalised result.  Cout in {32, 64, 96, 128} (one workgroup owns all channels of a pixel); results are
 * bit-identical to m4d_conv_cl_planar followed by m4d_rmsnorm_silu_cl_planar. */
int m4d_conv_cl_planar_norm(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* bias, const void* resid,
                            int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout, int kt, int To,
                            const float* norm_gamma, void* norm_out, int64_t norm_out_plane_stride, int silu, m4d_stream stream);

/* Tiled weights for the LDS-halo convolution kernels (3x3 / 3x3x3 taps, bf16, Cin % 16 == 0).  Those kernels stage one tap of a
 * 32-output-channel x 16-input-channel weight tile per 1 KiB DMA request; from the plain [Cout, taps*Cin] order the 64 16-byte pieces of a
 * request come from 32 different rows (32 cache lines per request).  m4d_conv_pack_weights writes the same values as
 * [ceil(Cout/32)][Cin/16][taps] units of 1 KiB, each the image the kernels keep in LDS (row r of the unit = output channel 32*block + r,
 * clamped to Cout-1; its two 16-byte halves = input channels [8c, 8c+8) of the chunk with c = half ^ ((r >> 3) & 1)), so that a request
 * is one contiguous KiB.  The host does this once per weight, next to the (dt, dh, dw, c) repack m4d_conv_cl already asks for
 * (reference weights: CausalConv3d / Conv2d of wan_vae.py:21-40, 81-100); m4d_conv_tiled_weight_bytes = size of the copy (0: not tileable).
 * m4d_conv_cl_tw / m4d_conv_cl_planar_tw are m4d_conv_cl / m4d_conv_cl_planar{,_norm,_gnstats} taking BOTH orders: `w_tiled` may be NULL,
 * kernels without a tiled form read `w`; results are bit-identical either way.  In m4d_conv_cl_planar_tw, norm_out != NULL selects the
 * fused next-layer norm (m4d_conv_cl_planar_norm), gn_partial != NULL the GroupNorm statistics (m4d_conv_cl_planar_gnstats). */
int64_t m4d_conv_tiled_weight_bytes(int Cin, int Cout, int taps);
int m4d_conv_pack_weights(m4d_dtype dt, const void* w, void* w_tiled, int Cin, int Cout, int taps, m4d_stream stream);
int m4d_conv_cl_tw(m4d_dtype dt, const void* x, int64_t x_pixel_stride, const void* w, const void* w_tiled, const void* bias,
                   const void* resid, int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin,
                   int Cout, int kt, int kh, int kw, int st, int sh, int sw, int pad_t, int pad_h, int pad_w, int To,
                   int Ho, int Wo, int ups, int tsplit, m4d_stream stream);
int m4d_conv_cl_planar_tw(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* w_tiled, const void* bias,
                          const void* resid, int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout,
                          int kt, int To, const float* norm_gamma, void* norm_out, int64_t norm_out_plane_stride, int silu,
                          float* gn_partial, m4d_stream stream);

/* RMS_norm over channels (F.normalize * sqrt(C) * gamma, wan_vae.py:43-58) fused with the following SiLU
 * (:199-201, :319, :424).  x/out: T [P, C] with row strides; gamma float [C]. */
int m4d_rmsnorm_silu_cl(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, void* out, int64_t out_ld,
                        int64_t P, int C, int silu, m4d_stream stream);
/* ... with the result in planar-16 layout: out[(c / 16) * out_plane_stride + row * 16 + c % 16], rows [0, P) (bf16, C % 16 == 0). */
int m4d_rmsnorm_silu_cl_planar(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, void* out, int64_t out_plane_stride,
                               int64_t P, int C, int silu, m4d_stream stream);

/* GroupNorm(G groups, eps) + affine (+ x*sigmoid(x)) per frame on [F, HW, C] (trajectory_module.py:54-60): two
 * deterministic passes; `partial` is a float workspace of m4d_groupnorm_cl_workspace(F, HW, G) elements. */
int64_t m4d_groupnorm_cl_workspace(int F, int64_t HW, int G);
int m4d_groupnorm_cl(m4d_dtype dt, const void* x, void* out, float* partial, int64_t partial_floats, const float* weight,
                     const float* bias, int F, int64_t HW, int C, int G, float eps, int silu, m4d_stream stream);
/* ... with the result in planar-16 layout for m4d_conv_cl_planar (the adaptors' norm -> swish -> conv, trajectory_module.py:54-71): the F
 * frames go into groups of `frames_per_group`, group g = [C/16][frames_per_group*HW][16] at out + g*out_group_stride with
 * out_plane_stride elements between channel planes (a group must stay below the 2 GiB the conv addresses).  bf16, C % 16 == 0. */
int m4d_groupnorm_cl_planar(m4d_dtype dt, const void* x, void* out, float* partial, int64_t partial_floats, const float* weight,
                            const float* bias, int F, int64_t HW, int C, int G, float eps, int silu, int frames_per_group,
                            int64_t out_plane_stride, int64_t out_group_stride, m4d_stream stream);
/* The adaptors' conv -> GroupNorm(32) -> swish -> conv chain without the statistics pass: m4d_conv_cl_planar_gnstats is
 * m4d_conv_cl_planar (Cout = 128) that also writes, per output patch, the (sum, sum of squares) of its stored result for the 32 groups
 * of 4 channels: gn_partial float [To][blocks][32][2], blocks = m4d_conv_cl_planar_gnstats_blocks(Hin, Win); a frame-group call
 * passes gn_partial + first_frame * blocks * 64.  m4d_groupnorm_cl_planar_apply then reduces the `partial_blocks` rows of every
 * frame in a fixed order into stat float [F][G][2] = (mean, rstd) and applies norm + affine (+ swish) like m4d_groupnorm_cl_planar. */
int m4d_conv_cl_planar_gnstats_blocks(int Hin, int Win);
int m4d_conv_cl_planar_gnstats(m4d_dtype dt, const void* x, int64_t x_plane_stride, const void* w, const void* bias, const void* resid,
                               int64_t resid_ld, void* out, int64_t out_ld, int Tin, int Hin, int Win, int Cin, int Cout, int kt, int To,
                               float* gn_partial, m4d_stream stream);
int m4d_groupnorm_cl_planar_apply(m4d_dtype dt, const void* x, void* out, const float* partial, int partial_blocks, float* stat,
                                  const float* weight, const float* bias, int F, int64_t HW, int C, int G, float eps, int silu,
                                  int frames_per_group, int64_t out_plane_stride, int64_t out_group_stride, m4d_stream stream);

/* out[r, c] = softmax_c(x[r, c] * scale) for c < C, 0 for C <= c < Cpad — the score matrix of the VAE's single-head
 * mid-block attention (wan_vae.py:256-260; head dim 384 > the flash kernel's 128). */
int m4d_softmax_rows(m4d_dtype in_dt, const void* x, int64_t ldx, m4d_dtype out_dt, void* out, int64_t ldo, int64_t rows,
                     int C, int Cpad, float scale, m4d_stream stream);

/* Layout boundaries of the VAE path.  ncthw_to_cl: dst[(t,h,w), c] = (src[c,t,h,w]*scale+shift)*ch_scale[c]+ch_shift[c]
 * for c < C, zero for C <= c < Cp.  cl_to_ncthw: the inverse with act: 0 none, 1 clamp(-1,1) (wan_vae.py:827),
 * 2 sigmoid(v + aux[c,t,h,w]) (trajectory_module.py:193). */
int m4d_ncthw_to_cl(m4d_dtype src_dt, const void* src, m4d_dtype dst_dt, void* dst, int64_t dst_pixel_stride, int C, int Cp,
                    int T, int H, int W, float scale, float shift, const float* ch_scale, const float* ch_shift,
                    m4d_stream stream);
int m4d_cl_to_ncthw(m4d_dtype src_dt, const void* src, int64_t src_pixel_stride, m4d_dtype dst_dt, void* dst, int C, int T,
                    int H, int W, float scale, float shift, const float* ch_scale, const float* ch_shift, int act,
                    const void* aux, m4d_stream stream);

/* out[b, i] = a[b, i] + bias[i]  (float32; a: [B, n], bias: [n]) — `(self.modulation + e)` of
 * WanAttentionBlock (:659) and Head (:718). */
int m4d_add_bcast(const float* a, const float* bias, float* out, int64_t B, int64_t n, m4d_stream stream);

/* Merge two m4d_attention_lse results of the SAME queries over DISJOINT key sets (T-sharded denoising: the local K/V shard is
 * attended while the peers' shards are still on the xGMI links, more4d_amd/dist): in place,
 *   o_a <- 2^(lse_a - lse) o_a + 2^(lse_b - lse) o_b,   lse_a <- lse = log2(2^lse_a + 2^lse_b)
 * o_* T [B, L, heads*head_dim] (batch / row strides in elements), lse_* float [B, heads, L] in the log2 domain; a side
 * without any valid key (lse = -inf) gets weight 0.  Equals one softmax over the union of the keys (reference attention(),
 * wan_transformer4d.py:66-236, over the all-gathered K/V :1187-1198). */
int m4d_attn_merge(m4d_dtype dt, void* o_a, int64_t oa_bs, int64_t oa_ls, float* lse_a, const void* o_b, int64_t ob_bs,
                   int64_t ob_ls, const float* lse_b, int B, int64_t L, int heads, int head_dim, m4d_stream stream);

/* TeaCache step skipping (cache_utils.py:19-74, wan_transformer4d.py:1201-1270), all float32:
 * axpby: out = a*x + b*y (residual re-use x + r, residual capture x_out - x_in);
 * rel_l1: out2 = { sum|cur - prev|, sum|prev| } over the modulated timestep embedding [B,6,C]. */
int m4d_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, m4d_stream stream);
/* out = a0*x0 + a1*x1 + a2*x2 + a3*x3 (float32; x1..x3 may be NULL; out may alias an input): the DPM-Solver++ multistep
 * updates of orders 2 and 3 (fm_solvers.py:486-677) and x0 = x - sigma v (:385-388) as linear combinations. */
int m4d_lincomb(const float* x0, float a0, const float* x1, float a1, const float* x2, float a2, const float* x3, float a3,
                float* out, int64_t n, m4d_stream stream);
int m4d_rel_l1(const float* prev, const float* cur, float* out2, int64_t n, m4d_stream stream);

/* Bilinear resize, align_corners=False, of a channels-last map [B,Hi,Wi,C] -> [B,Ho,Wo,C]: the OmniMAE feature map
 * of the Motion Perception Module resized to the latent token grid (wan_transformer4d.py:1152). */
int m4d_bilinear_cl(m4d_dtype dt, const void* x, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C, m4d_stream stream);

/* ------------------------------------------------------------------ stage-1 geometry (scripts/inference/infer.py)
 * The prologue / epilogue either side of the sampler, all float32 unless a dtype is given.
 * minmax: out[g] = {min, max} of each of n_groups contiguous groups of group_len floats (the depth range of :826 and the
 *   per-axis extent of the first frame's point cloud, :209-212).
 * backproject (:179-195 after the bilinear resize, which is m4d_bilinear_cl with C = 1): coords [3,H,W] = K^-1 (u, v, 1) * depth
 *   on the linspace(0,1) pixel grid, K = [[fx,0,.5],[0,fy,.5],[0,0,1]] (pass 1/fx, 1/fy); zclean [H,W] = z clamped to [0, 1e4]
 *   with nan / < 1e-5 replaced by 1 (:823-825).
 * depth_control (:826-828): out [3, hw] (dtype out_dt) = 2 (zclean - min) / (max - min + 1e-8) - 1, three identical channels.
 * flow_recover: rel [B,3,F,hw] (dtype in_dt) = decoded displacements, frame0 float [B,3,hw] = first-frame coordinates;
 *   out float [B,3,F,hw]: frame 0 = frame0 (:870; mode | 2 keeps the recovered frame 0 instead, what :198-219 itself returns),
 *   frames f > 0 = (rel + frame0/diff) * diff with diff = max over the three
 *   axes of (max - min) of frame0 (0 -> 1), minmax float [B*3, 2] from m4d_minmax (fminf / fmaxf: NaN inputs are skipped where torch.min / max would propagate them) (mode 0, inverse_flow_norm_transform_no_diff
 *   :198-219), or rel + frame0 (mode 1, --normalize_track_z :857-861; minmax may be NULL). */
int m4d_minmax(const float* x, int64_t n_groups, int64_t group_len, float* out, m4d_stream stream);
int m4d_backproject(const float* depth, int H, int W, float inv_fx, float inv_fy, float* coords, float* zclean, m4d_stream stream);
int m4d_depth_control(m4d_dtype out_dt, const float* zclean, const float* minmax, void* out, int64_t hw, m4d_stream stream);
int m4d_flow_recover(m4d_dtype in_dt, const void* rel, const float* frame0, const float* minmax, float* out, int B, int F,
                     int64_t hw, int mode, m4d_stream stream);

/* ------------------------------------------------------------------ training step (train_wan.py:1891-2015)
 * Backward halves of the DiT kernels above and the fused optimizer update.  The reference gets these from torch
 * autograd (accelerator.backward, :1988) and torch.optim.AdamW (:1136-1142, :2014); GEMM-shaped gradients
 * (dgrad = dy W, wgrad = dy^T x) are m4d_gemm_bt calls on m4d_transpose'd operands. */

/* m4d_attention that also writes lse[b, h, l] = log2(sum_j exp2(s_lj * scale * log2 e)) (float [B, heads, Lq]),
 * the only forward state the backward needs besides q, k, v, out. */
int m4d_attention_lse(m4d_dtype dt, const void* q, int64_t q_bs, int64_t q_ls, const m4d_kv_segs* kv, void* out,
                      int64_t o_bs, int64_t o_ls, int B, int64_t Lq, int heads, int head_dim, float scale,
                      int accumulate, float* lse, m4d_stream stream);

/* Flash-attention backward of softmax(q k^T scale) v (the autograd of flash_attn / SDPA, :138-169, :202-233).
 * Row-major operands: element (b, l, h, d) at ptr + b*bs + l*ls + h*head_dim + d.  Transposed operands (qt, kt, dot =
 * q^T, k^T, dO^T): element (b, h, d, l) at ptr + b*bs + (h*head_dim + d)*ls + l (what m4d_transpose of the [B*L, C]
 * matrix produces with bs = L, ls = B*L).  Only the first Lk of the Lk_rows key rows are real keys; dk / dv rows in
 * [Lk, Lk_rows) are written as zero.  delta: float workspace [B, heads, Lq]; lse from m4d_attention_lse.
 * qt / kt / dot may be NULL for bf16 with head_dim 128: those passes take the transposed fragments out of the row-major
 * tiles (ds_read_b64_tr_b16); the generic kernels (float, other head dims) require them. */
typedef struct {
    const void *q, *k, *v, *o, *d_o, *qt, *kt, *dot;
    const float* lse;
    float* delta;
    void *dq, *dk, *dv;
    int64_t q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, o_bs, o_ls, do_bs, do_ls;
    int64_t qt_bs, qt_ls, kt_bs, kt_ls, dot_bs, dot_ls;
    int64_t dq_bs, dq_ls, dk_bs, dk_ls, dv_bs, dv_ls;
    int64_t Lq, Lk, Lk_rows;
    int32_t B, heads, head_dim, accumulate_dq, accumulate_dkv;
    float scale;
    float* ws;          /* optional float workspace of ws_elems >= B*Lk_rows*heads*head_dim elements: lets the dk / dv */
    int64_t ws_elems;   /* passes split a long query loop over more workgroups when there are few keys (cross-attention) */
} m4d_attn_bwd_args;
int m4d_attention_bwd(m4d_dtype dt, const m4d_attn_bwd_args* args, m4d_stream stream);

/* out[c, r] = in[r, c] for a T matrix [R, C] (leading dims in elements, multiples of 4). */
int m4d_transpose(m4d_dtype dt, const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C,
                  m4d_stream stream);

/* out[g, c] += sum_{r in group g} a[r, c] (* b[r, c] when b != NULL); float out [ceil(R/rows_per_group), C] must be
 * initialised by the caller (atomic accumulation): bias gradients (sum of dy), gate gradients (sum of dx * y). */
int m4d_colsum(m4d_dtype a_dt, const void* a, int64_t lda, m4d_dtype b_dt, const void* b, int64_t ldb, float* out,
               int64_t R, int64_t C, int64_t rows_per_group, m4d_stream stream);

/* out T [R, C] = in float [R, C] * gate[r / rows_per_sample, c] (gate NULL => cast only): the gradient of the gated
 * residual `x + y * e[2]` (:669, :684) with respect to y. */
int m4d_scale_cast(const float* in, const float* gate, int64_t gate_stride, int64_t rows_per_sample, m4d_dtype out_dt,
                   void* out, int64_t R, int64_t C, m4d_stream stream);

/* out float [R, C] = x float [R, C] + y T [R, C] * gate[r / rows_per_sample, c] (gate NULL => 1): the gated residual
 * of :669 / :684 as a stand-alone op (the training recompute keeps y for the gate gradient). */
int m4d_resid_gate(const float* x, m4d_dtype y_dt, const void* y, const float* gate, int64_t gate_stride,
                   int64_t rows_per_sample, float* out, int64_t R, int64_t C, m4d_stream stream);

/* out = a + b (T, n % 4 == 0): sums of gradient branches. */
int m4d_add(m4d_dtype dt, const void* a, const void* b, void* out, int64_t n, m4d_stream stream);

/* dy *= act'(pre) in place (act: 1 silu, 2 gelu_tanh, 3 gelu_erf: the m4d_unary / GEMM-epilogue activations; 4 sigmoid, 5 sigmoid
 * with `pre` = its OUTPUT (the encoder adaptor's sigmoid(h + x), trajectory_module.py:194), 6 clamp(pre, -1, 1) (wan_vae.py:817)). */
int m4d_act_bwd(m4d_dtype dt, void* dy, const void* pre, int64_t n, int act, m4d_stream stream);

/* Backward of m4d_ln_modulate without guidance: y = LN(x) * m + s with m = 1 + scale[sample] | ln_w | 1.
 *   dx[r, :] += the LayerNorm input gradient (dx is the float residual-stream gradient, accumulated in place);
 *   dshift[sample*red_stride + c] += sum_r dy, dscale[...] += sum_r dy * xhat (both NULL to skip; red_stride = 0
 *   reduces over all samples: the affine weight / bias gradients of norm3, :674). */
int m4d_ln_modulate_bwd(const float* x, m4d_dtype dy_dt, const void* dy, float* dx, int B, int64_t rows_per_sample, int C,
                        const float* scale, int64_t mod_stride, const float* ln_w, float eps, float* dshift,
                        float* dscale, int64_t red_stride, m4d_stream stream);

/* Backward of the spatial-guidance tail of m4d_ln_modulate (SpatialGuidanceModule.forward, wan_transformer4d.py:757-783:
 * z = u*(1 + S*gate) + H*gate with u = LN(x)*(1+scale)+shift and (S | H) = g_ss[sample, l % g_period] for l < g_len).
 *   dz T [B*rows_per_sample, C] holds dL/dz on entry and dL/du on return (rows l >= g_len are untouched: z = u there),
 *   ready for m4d_ln_modulate_bwd;
 *   ab float [B, g_period, 2C] = (sum_f dz*u | sum_f dz) over the rows f*g_period + pos < g_len of each position
 *   (fully written).  dS = A*gate, dH = Bm*gate, dgate = sum_{b,pos} (A*S + Bm*H). */
int m4d_guidance_bwd(const float* x, m4d_dtype dz_dt, void* dz, int B, int64_t rows_per_sample, int C, const float* shift,
                     const float* scale, int64_t mod_stride, float eps, const float* g_ss, const float* g_gate,
                     int64_t g_period, int64_t g_len, float* ab, m4d_stream stream);

/* m4d_guidance_bwd with mod_rows = rows that share one (shift, scale) vector (0: rows_per_sample; 1: per-token modulation, :655-657). */
int m4d_guidance_bwd_m(const float* x, m4d_dtype dz_dt, void* dz, int B, int64_t rows_per_sample, int C, const float* shift,
                       const float* scale, int64_t mod_stride, int64_t mod_rows, float eps, const float* g_ss, const float* g_gate,
                       int64_t g_period, int64_t g_len, float* ab, m4d_stream stream);

/* Backward of m4d_rmsnorm_rope, in place on the gradient: dy0/dy1 T [rows, C] (row stride ld_dy) hold dL/d(output) on
 * entry and dL/d(input) on return; x0/x1 are the PRE-norm inputs (row stride ld_x); dw0/dw1 float [C] accumulate the
 * WanRMSNorm weight gradients (atomic, caller initialises). */
int m4d_rmsnorm_rope_bwd(m4d_dtype dt, void* dy0, void* dy1, int64_t ld_dy, const void* x0, const void* x1, int64_t ld_x,
                         const float* w0, const float* w1, float* dw0, float* dw1, int64_t rows, int C, int head_dim,
                         float eps, const float* cos_t, const float* sin_t, int64_t rows_per_sample, int64_t rope_len,
                         int64_t pos_offset, m4d_stream stream);

/* *out += sum x^2 (float accumulation; the global gradient norm of :1991-1993 / clip_grad_norm_ :2009). */
int m4d_sumsq(m4d_dtype dt, const void* x, int64_t n, float* out, m4d_stream stream);

/* One torch.optim.AdamW step (decoupled weight decay, bias correction from `step` >= 1) on a flat parameter, with the
 * clip coefficient fused: g = grad * (*grad_scale) (NULL => 1).  State dtype: float32 or the parameter dtype
 * (the reference keeps bf16 state next to bf16 parameters, :1089, :1136-1142). */
int m4d_adamw(m4d_dtype dt, void* param, const void* grad, m4d_dtype state_dt, void* exp_avg, void* exp_avg_sq, int64_t n,
              float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale,
              m4d_stream stream);

/* ------------------------------------------------------------------ VAE / adaptor training (train_vae.py:434-495)
 * Backward halves of the Motion-Sensitive-VAE kernels: the reference gets them from torch autograd through `encode_full` /
 * `decode_full` (wan_vae.py:549-676: per-chunk checkpoint, streaming cache detached between chunks) and the adaptors
 * (trajectory_module.py:101-279).  Convolution DATA gradients reuse m4d_conv_cl with flipped taps; the WEIGHT gradient is
 *   pad_transpose (dy and x -> pixel-major panels in a zero-padded frame geometry, every tap = a column offset)
 *   -> gemm_bt_batched (split-K partial products, K-slices x taps as the batch) -> wgrad_reduce.
 * pad_transpose: src = T frames of H x W pixels, C channels, pixel stride `pixel_stride`; out [nshift*C, cols] (row stride cols):
 *   out[s*C + c][q] = P[q + s][c], P = the frames zero-padded to [T, Hp, Wp] with the image at (pad_top, pad_left), flattened;
 *   zero beyond the last frame.
 * gemm_bt_batched: float32 out[i2][i1] [M, N] = A_(i1,i2) [M,K] . W_(i1,i2) [N,K]^T (unrounded accumulators), A_(i1,i2) =
 *   A + i1*a_bs1 + i2*a_bs2 elements with row stride lda, W likewise; i1 < nb1, i2 < nb2.
 * wgrad_reduce: dw[co][dt][dh][dw][ci] += sum_s part[dh][s][co][dw*cip + ci]; part float32 [kh, S, Mp, kw*cip], dw float32
 *   [cop, kt, kh, kw, cip] (the packed weight layout of m4d_conv_cl).
 * gemm_bt_taps (bf16; the same weight gradient on the production 256 x 256 kernel, for Cout %% 32 == 0 and kw*cip >= 256): float32
 *   out[s] [M, N] = A_s [M,K] . W_s [N,K]^T for K-slice s < nb1 (A_s = A + s*a_bs1, W_s = W + s*w_bs1 elements), where the
 *   M = taps * tap_rows rows of A are STACKED TAPS: rows [t*tap_rows, (t+1)*tap_rows) are rows [0, tap_rows) of the matrix at A read
 *   tap_s1*(t / tap_kh) + tap_s2*(t %% tap_kh) elements further along K — dy is the shifted operand, all kt*kh taps of a layer are
 *   one launch.  wgrad_reduce_taps: dw[co][dt][dh][dw][ci] += sum_s part[s][dt*kh + dh][co][dw*cip + ci].
 * rmsnorm_silu_cl_bwd: backward of m4d_rmsnorm_silu_cl (wan_vae.py:43-58 + SiLU): dx T [P, C], dgamma float32 [C] (+=).
 * softmax_rows_bwd: dS = scale * P * (dP - rowsum(P dP)) on [rows, Cpad] (columns >= C written as 0): mid-block attention :244-266.
 * upsample2x_cl: nearest-exact 2x of channels-last frames [t,h,w,c] -> [t',2h,2w,c] (tsplit: input pixels hold 2c channels and
 *   frame 2i / 2i+1 of the result reads the first / second half, :138-141); backward != 0 runs the transpose (sum of each 2x2
 *   block, frames back into channel halves).
 * groupnorm_cl_bwd: backward of m4d_groupnorm_cl (GroupNorm(32, eps 1e-6) + swish, trajectory_module.py:32-60): dx T [F,HW,C],
 *   dweight / dbias float32 [C] (+=); ws from m4d_groupnorm_cl_bwd_workspace floats. */
int m4d_pad_transpose(m4d_dtype dt, const void* src, int64_t pixel_stride, int C, int T, int H, int W, int Hp, int Wp, int pad_top,
                      int pad_left, int nshift, void* out, int64_t cols, m4d_stream stream);
int m4d_gemm_bt_batched(m4d_dtype dt, const void* A, int64_t lda, int64_t a_bs1, int64_t a_bs2, const void* W, int64_t ldw,
                        int64_t w_bs1, int64_t w_bs2, float* out, int64_t M, int64_t N, int64_t K, int nb1, int nb2, m4d_stream stream);
int m4d_wgrad_reduce(const float* part, float* dw, int S, int Mp, int cop, int kt, int kh, int kw, int cip, int dt, m4d_stream stream);
int m4d_gemm_bt_taps(m4d_dtype dt, const void* A, int64_t lda, int64_t a_bs1, const void* W, int64_t ldw, int64_t w_bs1, float* out,
                     int64_t M, int64_t N, int64_t K, int nb1, int tap_rows, int tap_kh, int64_t tap_s1, int64_t tap_s2, m4d_stream stream);
int m4d_wgrad_reduce_taps(const float* part, float* dw, int S, int cop, int kt, int kh, int kw, int cip, m4d_stream stream);
int m4d_rmsnorm_silu_cl_bwd(m4d_dtype dt, const void* x, int64_t x_ld, const float* gamma, const void* dy, int64_t dy_ld, void* dx,
                            int64_t dx_ld, float* dgamma, int64_t P, int C, int silu, m4d_stream stream);
int m4d_softmax_rows_bwd(m4d_dtype dt, const void* p, const float* dp, void* out, int64_t rows, int C, int Cpad, float scale,
                         m4d_stream stream);
int m4d_upsample2x_cl(m4d_dtype dt, const void* in, void* out, int t, int h, int w, int c, int tsplit, int backward, m4d_stream stream);
int64_t m4d_groupnorm_cl_bwd_workspace(int F, int64_t HW, int G);
int m4d_groupnorm_cl_bwd(m4d_dtype dt, const void* x, const float* weight, const float* bias, const void* dy, void* dx, float* dweight,
                         float* dbias, float* ws, int64_t ws_floats, int F, int64_t HW, int C, int G, float eps, int silu,
                         m4d_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MORE4D_HIP_H */
