#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
// LDS filled with u16 index values (element e at byte 2e). Each lane reads at byte address addr[lane]; out[lane*4+j] = result element j.
__global__ void probe(const unsigned* addr, unsigned short* out) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds + addr[threadIdx.x];
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned h_addr[64]; unsigned short h_out[256];
    unsigned *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 2; ++pat) {
        // pattern 0: lane l reads 8 bytes at l*8 (lane-linear).  pattern 1: lane l reads row (l&15) of a 64-byte-pitch matrix, chunk (l>>4): addr = (l&15)*64 + (l>>4)*8
        for (int l = 0; l < 64; ++l) h_addr[l] = pat == 0 ? l * 8 : (l & 15) * 64 + (l >> 4) * 8;
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4u -> %5u %5u %5u %5u\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
    }
    return 0;
}
