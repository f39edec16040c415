// Probe: throughput of the fp32 atomic-add traffic a single-pass flash-attention backward would generate for dQ on an MI355X.
// One workgroup = (head, block of 128 keys), 4 waves; for every 64-query tile each wave adds a [64 q x 32 d] fp32 block into
// dq[head][q][d] (32 wave-instructions, each 2 rows x 128 B).  Heads are pinned per XCD (block b runs on XCD b % 8), so the
// updates of one head stay in one XCD's L2 — compared across memory scopes: SCOPE 0 = workgroup (plain L2 atomic), 1 = agent
// (device scope), 2 = plain load+add+store (no atomicity, upper bound of the RMW traffic itself).
// hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_dq.hip -o tools/probes/atomic_dq.bin && tools/probes/atomic_dq.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SCOPE>
__global__ __launch_bounds__(256) void k(float* dq, int L, int heads, int nkb, int spin) {
    const int HB = heads;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int h = xcd * (HB >> 3) + idx / nkb;
    if (h >= heads) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hi = lane >> 5;
    float* base = dq + (size_t)h * L * 128 + wave * 32 + li;
    float acc = 1.0f + blockIdx.x * 1e-6f;
    for (int t = 0; t * 64 + 64 <= L; ++t) {
        // stand-in for the 80 MFMAs of a tile: `spin` dependent FMAs per lane
        for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int q = t * 64 + (r >> 2) * 8 + hi * 4 + (r & 3);
            float* p = base + (size_t)q * 128;
            if (SCOPE == 0) __hip_atomic_fetch_add(p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (SCOPE == 1) __hip_atomic_fetch_add(p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *p += acc;
        }
    }
}

int main(int argc, char** argv) {
    const int L = 21840, heads = 40, nkb = argc > 1 ? atoi(argv[1]) : 171;
    float* dq;
    const size_t n = (size_t)heads * L * 128;
    hipMalloc(&dq, n * 4);
    hipMemset(dq, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)heads * nkb * (L / 64) * 4 * 32 * 64 * 4.0;
    for (int spin : {0, 600}) {
        for (int scope = 0; scope < 3; ++scope) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                dim3 g(heads * nkb), b(256);
                if (scope == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, dq, L, heads, nkb, spin);
                else if (scope == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, dq, L, heads, nkb, spin);
                else hipLaunchKernelGGL(k<2>, g, b, 0, 0, dq, L, heads, nkb, spin);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("spin %4d scope %d (%s): %8.2f ms  %6.2f TB/s of fp32 adds  (%.1f GB)\n", spin, scope,
                                scope == 0 ? "workgroup" : scope == 1 ? "agent" : "plain rmw", ms, bytes / ms / 1e9, bytes / 1e9);
            }
        }
    }
    float probe[4];
    hipMemcpy(probe, dq, 16, hipMemcpyDeviceToHost);
    printf("dq[0] = %g (expect ~%d adds x 4 launches per scope mix)\n", probe[0], nkb);
    return 0;
}
