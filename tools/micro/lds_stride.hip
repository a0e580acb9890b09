// Cycles per ds_read_b32 wave-instruction against the lane -> word mapping (one workgroup of 256 threads per CU, 64 dependent-free reads
// per thread): consecutive words, strides of 2 / 4 / 8 words, and the two sample patterns of pyramid_fused (lane pairs 4 or 8 words apart),
// plain and with the row skewed by one word per 32 (x + (x >> 5)).     hipcc --offload-arch=gfx950 -O3 tools/micro/lds_stride.hip -o tools/micro/lds_stride
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(unsigned long long* out, int mode) {
    __shared__ float g[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) g[i] = i;
    __syncthreads();
    const int l = threadIdx.x & 63;
    int w;
    switch (mode) {
        case 0: w = l; break;
        case 1: w = 2 * l; break;
        case 2: w = 4 * l; break;
        case 3: w = 8 * l; break;
        case 4: w = 4 * (l >> 1) + (l & 1); break;          // level 2 of pyramid_fused
        case 5: w = 8 * (l >> 1) + (l & 1); break;          // level 3
        case 6: w = 4 * (l >> 1) + (l & 1); w += w >> 5; break;
        default: w = 8 * (l >> 1) + (l & 1); w += w >> 5; break;
    }
    const float* p = g + w;
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int t = 0; t < 64; ++t) acc += p[t];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345.f) out[100] = 1;
    if (l == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 1024);
    const char* names[] = {"consecutive words", "stride 2", "stride 4", "stride 8", "pairs 4 apart (level 2)", "pairs 8 apart (level 3)",
                           "pairs 4 apart, row skewed by x >> 5", "pairs 8 apart, row skewed by x >> 5"};
    printf("cycles per ds_read_b32 wave-instruction, 4 waves of one workgroup reading together (64 reads each)\n");
    for (int m = 0; m < 8; ++m) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, m);
        unsigned long long h[4];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("  %-40s %6.1f\n", names[m], (double)(h[0] + h[1] + h[2] + h[3]) / 4 / 64 / 4 * 4);
    }
    return 0;
}
