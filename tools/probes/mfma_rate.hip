// Probe: issue rate of v_mfma_f32_32x32x16_bf16 from ONE wave under the conditions the attention kernel's MFMA phase runs in.
// A wave issues N back-to-back MFMAs (4 rotating accumulators unless CH says otherwise) and times them with s_memtime.
//   WPS   waves per SIMD resident (1: 256 threads, 2: 512 threads; the second wave of a SIMD parks at an s_barrier or spins on VALU)
//   ACC   0 = accumulators in arch VGPRs, 1 = AGPRs
//   GAP   instructions between MFMAs: 0 none, 1 = s_waitcnt lgkmcnt(7) (satisfied), 2 = s_waitcnt + ds_read_b128
//   OTHER what the other wave of the SIMD does: 0 = waits at the barrier, 1 = v_exp_f32 loop, 2 = v_fma_f32 loop
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/mfma_rate.bin && tools/probes/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WPS, int ACC, int GAP, int OTHER, int CH>
__global__ __launch_bounds__(256 * WPS, 1) void k(unsigned long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x16 c[4];
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) c[d][r] = 0.f;
    bf16x8 frag;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
    float x = 0.5f + lane * 0.001f;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (GAP >= 1) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                if (ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[j % CH]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j % CH]) : "v"(a), "v"(b));
                if (GAP >= 2) asm volatile("ds_read_b128 %0, %1" : "=v"(frag) : "v"(addr) : "memory");
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        if (WPS == 2) __builtin_amdgcn_s_barrier();
    } else {
        if (OTHER == 0) __builtin_amdgcn_s_barrier();
        else {
            for (int it = 0; it < iters * 40; ++it) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (OTHER == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                    else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
                }
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    float s = x;
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += c[d][r];
    if (GAP >= 2) s += (float)frag[0];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

// The other direction: how fast does the VALU wave of a SIMD run while its neighbour streams MFMAs?  MF: 0 = neighbour parked at the
// barrier, 1 = MFMAs with accumulators in arch VGPRs, 2 = in AGPRs.  OP: 0 = v_fma_f32 (4 independent chains), 1 = v_exp_f32.
template <int MF, int OP>
__global__ __launch_bounds__(512, 1) void kv(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x16 c[4];
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) c[d][r] = 0.f;
    float x0 = 0.5f + lane * 0.001f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (MF != 0) {
            for (int it = 0; it < iters * 3; ++it) {
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    if (MF == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[j % 4]) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j % 4]) : "v"(a), "v"(b));
                }
            }
        }
        __builtin_amdgcn_s_barrier();
    } else {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (OP == 0) {
                    asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x0)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x1));
                    asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x2)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x3));
                } else {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x2)); asm volatile("v_exp_f32 %0, %0" : "+v"(x3));
                }
            }
        }
        asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float s = x0 + x1 + x2 + x3;
    for (int d = 0; d < 4; ++d) for (int r = 0; r < 16; ++r) s += c[d][r];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0 && wave == 4) out[blockIdx.x] = t1 - t0;
}
template <int MF, int OP> void runv(const char* name, unsigned long long* d, float* sink) {
    const int iters = 50, nb = 256;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((kv<MF, OP>), dim3(nb), dim3(512), 0, 0, d, sink, iters); hipDeviceSynchronize(); }
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < nb; ++i) sum += (double)h[i];
    printf("%-64s %6.2f cycles per VALU instruction\n", name, sum / nb / (iters * 256.0));
}

template <int WPS, int ACC, int GAP, int OTHER, int CH> void run(const char* name, unsigned long long* d, float* sink) {
    const int iters = 50, nb = 256;
    hipLaunchKernelGGL((k<WPS, ACC, GAP, OTHER, CH>), dim3(nb), dim3(256 * WPS), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<WPS, ACC, GAP, OTHER, CH>), dim3(nb), dim3(256 * WPS), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < nb; ++i) sum += (double)h[i];
    printf("%-64s %6.1f cycles per MFMA\n", name, sum / nb / (iters * 32.0));
}

int main() {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 64);
    run<1, 1, 0, 0, 4>("1 wave/SIMD, AGPR acc, back to back", d, sink);
    run<1, 0, 0, 0, 4>("1 wave/SIMD, VGPR acc, back to back", d, sink);
    run<1, 0, 0, 0, 2>("1 wave/SIMD, VGPR acc, 2 chains", d, sink);
    run<1, 0, 0, 0, 1>("1 wave/SIMD, VGPR acc, 1 chain", d, sink);
    run<1, 1, 0, 0, 1>("1 wave/SIMD, AGPR acc, 1 chain", d, sink);
    run<1, 0, 1, 0, 4>("1 wave/SIMD, VGPR acc, s_waitcnt between", d, sink);
    run<1, 0, 2, 0, 4>("1 wave/SIMD, VGPR acc, s_waitcnt + ds_read between", d, sink);
    run<1, 1, 2, 0, 4>("1 wave/SIMD, AGPR acc, s_waitcnt + ds_read between", d, sink);
    run<2, 0, 0, 0, 4>("2 waves/SIMD (other at barrier), VGPR acc, back to back", d, sink);
    run<2, 1, 0, 0, 4>("2 waves/SIMD (other at barrier), AGPR acc, back to back", d, sink);
    run<2, 0, 2, 0, 4>("2 waves/SIMD (other at barrier), VGPR acc, waitcnt + ds_read", d, sink);
    run<2, 0, 2, 1, 4>("2 waves/SIMD (other: v_exp loop), VGPR acc, waitcnt + ds_read", d, sink);
    run<2, 0, 2, 2, 4>("2 waves/SIMD (other: v_fma loop), VGPR acc, waitcnt + ds_read", d, sink);
    run<2, 1, 2, 1, 4>("2 waves/SIMD (other: v_exp loop), AGPR acc, waitcnt + ds_read", d, sink);
    run<2, 1, 2, 2, 4>("2 waves/SIMD (other: v_fma loop), AGPR acc, waitcnt + ds_read", d, sink);
    runv<0, 0>("v_fma_f32 wave, neighbour parked", d, sink);
    runv<1, 0>("v_fma_f32 wave, neighbour: MFMAs with VGPR accumulators", d, sink);
    runv<2, 0>("v_fma_f32 wave, neighbour: MFMAs with AGPR accumulators", d, sink);
    runv<0, 1>("v_exp_f32 wave, neighbour parked", d, sink);
    runv<1, 1>("v_exp_f32 wave, neighbour: MFMAs with VGPR accumulators", d, sink);
    runv<2, 1>("v_exp_f32 wave, neighbour: MFMAs with AGPR accumulators", d, sink);
    return 0;
}
