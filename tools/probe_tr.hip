// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds its own element index; every lane reads at byte address lane*8.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned addr = (unsigned)(size_t)(lds) + threadIdx.x * 8;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main()
{
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
