// Microbenchmark: what does an LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave) cost a SIMD that is issuing
// bf16 MFMAs, by waves per SIMD?   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_mfma.hip -o /tmp/dma_mfma && /tmp/dma_mfma
// Every workgroup: W waves per SIMD (blockDim = 256 * W), each wave loops ITER times over [D DMA pieces, M MFMAs on 8
// independent accumulators]; the DMA source is a small L2-resident buffer; LDS destination per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int D, int M>
__global__ void k(const char* src, float* out, unsigned long long* cyc, int iters) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 1 << 20, 0x00020000);
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)(float)(lane - j); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 2048 + (d & 1) * 1024), 16,
                                                     ((it * D + d) & 255) * 1024 + lane * 16, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        if (D > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D > 0 ? D : 0) : "memory");   // keep one iteration of pieces in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
#endif
}

template <int D, int M>
void run(int W, const char* src, float* out, unsigned long long* cyc) {
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL((k<D, M>), dim3(blocks), dim3(256 * W), 2048 * 4 * W, 0, src, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<D, M>), dim3(blocks), dim3(256 * W), 2048 * 4 * W, 0, src, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // whole-kernel time (every wave of the workgroup finished), in cycles at a nominal 2.1 GHz, per iteration
    const double avg = (double)ms * 1e-3 * 2.1e9 / iters;
    // MFMA issue floor per SIMD per iteration: W waves x M MFMAs x 32 cycles
    printf("waves/SIMD %d  DMA %d  MFMA %2d per iteration: %7.0f cycles@2.1GHz per iteration (MFMA floor %5d) -> %5.1f cycles per DMA piece beyond the floor\n", W, D, M,
           avg, W * M * 32, D ? (avg - W * M * 32) / (D * W) : 0.0);
}

int main() {
    char* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20);
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    for (int W : {1, 2, 4}) {
        run<0, 24>(W, src, out, cyc);
        run<3, 24>(W, src, out, cyc);
        run<6, 24>(W, src, out, cyc);
        run<6, 48>(W, src, out, cyc);
        run<9, 48>(W, src, out, cyc);
    }
    return 0;
}
