// Microbenchmark: what the bf16 matrix pipes sustain in WALL-CLOCK terms, by duty cycle and operand data.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
// 256 workgroups x 8 waves (2 per SIMD, as gemm_x6), each wave loops over [48 v_mfma_f32_32x32x16_bf16 on 8 independent
// accumulators, then IDLE s_nop-free s_sleep units].  Reported: executed PFLOP/s from HIP events, and the implied clock
// (cycles from s_memtime / time).  Operands: zeros, small integers, or random bf16 bit patterns of normal magnitude.
// The question it answers: is a busier matrix pipe paid back in full, or does the chip lower its clock as the duty rises?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int IDLE>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ ops, float* out, unsigned long long* cyc, int iters) {
#if __HIP_DEVICE_COMPILE__
    const int lane = threadIdx.x & 63;
    floatx16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = __builtin_bit_cast(bf16x8, ops[(threadIdx.x * 6 + j) & 4095]);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(bf16x8, ops[(threadIdx.x * 6 + 4 + j) & 4095]);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 3], b[(m >> 2) & 1], acc[m & 7], 0, 0, 0);
        if (IDLE > 0) __builtin_amdgcn_s_sleep(IDLE);   // 64 cycles per unit
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    (void)lane;
#endif
}

template <int IDLE>
static void run(const char* what, const u32x4* d_ops, float* d_out, unsigned long long* d_cyc, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<IDLE>, dim3(256), dim3(512), 0, 0, d_ops, d_out, d_cyc, iters / 8);   // warm-up
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<IDLE>, dim3(256), dim3(512), 0, 0, d_ops, d_out, d_cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(256);
    (void)hipMemcpy(c.data(), d_cyc, sizeof(c[0]) * 256, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v / 256;
    const double flops = 256.0 * 8 * iters * 48.0 * 2 * 32 * 32 * 16;
    // duty of a SIMD's matrix pipe by construction: two waves, each [48 MFMAs = 1536 cycles of the pipe, then 64 * IDLE cycles asleep]:
    // period = max(3072, 1536 + 64 * IDLE), of which 3072 are MFMA; clock = MFMA cycles / (duty * time)
    const double period = 1536.0 + 64.0 * IDLE > 3072.0 ? 1536.0 + 64.0 * IDLE : 3072.0;
    const double duty = 3072.0 / period;
    const double mfma_cycles = (double)iters * 3072.0;
    printf("%-7s sleep %3d: %7.2f ms  %6.3f PFLOP/s executed   matrix-pipe duty %.2f by construction -> clock %.2f GHz   (s_memtime ticks %.0f)\n", what,
           IDLE, ms, flops / (ms * 1e-3) / 1e15, duty, mfma_cycles / duty / (ms * 1e-3) / 1e9, mean);
}

int main() {
    std::vector<unsigned> h(4096 * 4);
    u32x4* d_ops;
    float* d_out;
    unsigned long long* d_cyc;
    (void)hipMalloc(&d_ops, h.size() * 4);
    (void)hipMalloc(&d_out, 256 * 512 * 4);
    (void)hipMalloc(&d_cyc, 256 * 8);
    const int iters = 40000;
    for (int mode = 0; mode < 3; ++mode) {
        srand(1);
        for (auto& v : h) {
            if (mode == 0) v = 0;
            else if (mode == 1) v = 0x3f803f80u;   // 1.0, 1.0
            else {                                 // random sign / mantissa, exponents within 2^-4 .. 2^3
                unsigned lo = ((rand() & 1) << 15) | ((123 + rand() % 8) << 7) | (rand() & 127);
                unsigned hi = ((rand() & 1) << 15) | ((123 + rand() % 8) << 7) | (rand() & 127);
                v = (hi << 16) | lo;
            }
        }
        (void)hipMemcpy(d_ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const char* what = mode == 0 ? "zeros" : mode == 1 ? "ones" : "random";
        run<0>(what, d_ops, d_out, d_cyc, iters);
        run<8>(what, d_ops, d_out, d_cyc, iters);
        run<16>(what, d_ops, d_out, d_cyc, iters);
        run<32>(what, d_ops, d_out, d_cyc, iters);
        run<64>(what, d_ops, d_out, d_cyc, iters);
        run<96>(what, d_ops, d_out, d_cyc, iters);
    }
    return 0;
}
