// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS for lanes whose offset is past num_records (raw buffer, stride 0)?
// hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_oob.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const char* x, int nbytes, const int* offs, float* out) {
    for (int i = threadIdx.x; i < 64 * 4; i += 64) ((float*)smem)[i] = -7.f;     // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, nbytes, 0x00027000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, offs[threadIdx.x], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
    const int n = 64 * 4;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    std::vector<int> offs(64);
    for (int l = 0; l < 64; ++l) offs[l] = (l % 3 == 1) ? 0x7fffff00 : (l % 3 == 2 ? -16 : l * 16);   // OOB high, "negative", in range
    float *dx, *dout; int* doff;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&doff, 64 * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(doff, offs.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, (const char*)dx, n * 4, doff, dout);
    std::vector<float> o(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    for (int l = 0; l < 9; ++l) printf("lane %d off %d -> %g %g %g %g\n", l, offs[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    return 0;
}
