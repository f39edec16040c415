// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of more4d_amd.
// Fragment convention used by every MFMA kernel here (32x32 tiles, K=16 per step):
//   lane l = (i = l & 31, hi = l >> 5) holds 8 consecutive K-elements k0 + hi*8 + [0..8)
//   of row i  ->  one v_mfma_f32_32x32x16_bf16 (T = bf16) or eight v_mfma_f32_32x32x2_f32
//   (T = float; slot (hi, j) of A pairs with slot (hi, j) of B, so the same register image works).
//   C/D: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * hi, r in [0,16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define M4D_DEV __device__ __forceinline__

template <typename T> struct Frag8;
template <> struct Frag8<bf16_t> { typedef bf16x8 type; };
template <> struct Frag8<float> { typedef f32x8 type; };

// D += A(32 x 16) * B(16 x 32)
M4D_DEV void mma32(const bf16x8& a, const bf16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
M4D_DEV void mma32(const f32x8& a, const f32x8& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
}

// value rounded through the compute dtype (mimics the reference's autocast casts)
template <typename T> M4D_DEV float round_through(float x);
template <> M4D_DEV float round_through<float>(float x) { return x; }
template <> M4D_DEV float round_through<bf16_t>(float x) { return (float)(bf16_t)x; }

template <typename T> M4D_DEV float to_f32(T x) { return (float)x; }

// 4-wide load/store helpers (8 B for bf16, 16 B for float)
M4D_DEV f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
M4D_DEV f32x4 load4(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r;
    r[0] = (float)v[0]; r[1] = (float)v[1]; r[2] = (float)v[2]; r[3] = (float)v[3];
    return r;
}
M4D_DEV void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
M4D_DEV void store4(bf16_t* p, const f32x4& v) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

M4D_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
M4D_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3): one v_exp_f32 and one v_rcp_f32 instead of libm's tanhf
// (~30 instructions with range branches, 5 % of the ffn_up GEMM in its epilogue).  |error| of the two hardware ops: ~1e-7 relative, far
// inside bf16 rounding and the fp32-mode parity budget; x -> -inf: exp -> +inf, rcp -> 0, result -0 like the tanh form.
M4D_DEV float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    const float e = __builtin_amdgcn_exp2f(u * -2.8853900817779268f);      // exp(-2u) = 2^(-2u log2 e)
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
M4D_DEV float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
// (v_rcp_f32 instead of the IEEE division sequence, ~10 instructions: eight SiLUs per read-back iteration sit in the latency-bound fused-norm
// epilogue of the VAE conv; 1 ulp, far inside the bf16 rounding that follows — every SiLU of the library goes through this one function)
M4D_DEV float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// root_c / max(|x|, 1e-12) of the VAE's RMS_norm (F.normalize): v_sqrt_f32 + v_rcp_f32 (1 ulp each) instead of the correctly rounded
// sqrtf and IEEE division sequences (~25 instructions per pixel in the fused-norm conv epilogue); ONE function for the fused epilogue and
// the separate kernel, whose results must agree bit for bit
M4D_DEV float rms_scale_f(float root_c, float ss) { return root_c * __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(ss), 1e-12f)); }

// ---- A/B switches and timing ablations ----
// Environment switches are read ONCE per process (M4D_ENV_ONCE).  Timing ablations (M4D_*_ABL: kernels that skip work,
// run faster and return WRONG results) exist only in tool builds: `python -m more4d_amd.build --ablations` compiles with
// -DM4D_ABLATIONS into lib/libmore4d_hip_abl.so; in the shipping library M4D_ABL() is the constant 0, the branches fold
// away and the environment variables are never read.
#include <stdlib.h>
// "Done once" state that HIP keeps PER DEVICE (hipFuncSetAttribute: the dynamic-LDS opt-in of a kernel): a process may drive several GPUs,
// so the launchers' first-use blocks are keyed by the current device, not by a process-wide bool.  Relaxed atomics: host
// threads may launch concurrently (a flag seen late only repeats the idempotent hipFuncSetAttribute); devices beyond the table repeat it
// on every launch (correct, slower).
#include <atomic>
struct PerDeviceOnce {
    static constexpr int N = 64;
    std::atomic<bool> done[N] = {};
    static int dev() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return d; }
    bool pending() const { const int d = dev(); return d < 0 || d >= N || !done[d].load(std::memory_order_relaxed); }
    void mark() { const int d = dev(); if (d >= 0 && d < N) done[d].store(true, std::memory_order_relaxed); }
};

#define M4D_ENV_ONCE(var, name, dflt)                      \
    static int var = -0x7fffffff;                          \
    if (var == -0x7fffffff) {                              \
        const char* v__ = getenv(name);                    \
        var = v__ ? atoi(v__) : (dflt);                    \
    }
#ifdef M4D_ABLATIONS
#define M4D_ABL(p) ((p).abl)
#else
#define M4D_ABL(p) 0
#endif

// ---- host-side error plumbing (api.cpp owns the storage) ----
extern "C" __attribute__((visibility("hidden"))) void m4d_set_error(const char* fmt, ...);   // library-internal: not part of the ABI
extern "C" __attribute__((visibility("hidden"))) void m4d_count_launch(int kernel_class);       // m4d_kernel_class (more4d_hip.h)
#define M4D_CHECK_ARG(cond, ...)                   \
    do {                                           \
        if (!(cond)) {                             \
            m4d_set_error(__VA_ARGS__);            \
            return -1;                             \
        }                                          \
    } while (0)
#define M4D_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            m4d_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -3;                                                      \
        }                                                                   \
    } while (0)
