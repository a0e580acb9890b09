// Issue cost of the vector instructions the Farneback kernels are made of (cycles per wave-instruction on one SIMD, 1 / 2 / 4 waves per
// SIMD), measured with s_memtime around 8 x 64 instructions (a loop of 8 with 64 unrolled: the loop branch is 1 in 67).   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP, int CH>
__global__ __launch_bounds__(1024) void rate(unsigned long long* out, float seed) {
    float f[16];
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { f[i] = seed + i + threadIdx.x; d[i] = seed * 3 + i + threadIdx.x; }
    const float c = seed * 0.5f;
    const double cd = seed * 0.25;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#define ONE(i)                                                                                                   \
    if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));                                     \
    if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(c));                                 \
    if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));                                    \
    if (OP == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));                                \
    if (OP == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));                                    \
    if (OP == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));                                  \
    if (OP == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));                                  \
    if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(c));                            \
    if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));                             \
    if (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));                                                  \
    if (OP == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(c));                                        \
    if (OP == 11) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(f[i]));
        if (CH == 1) { ONE(0) ONE(0) ONE(0) ONE(0) ONE(0) ONE(0) ONE(0) ONE(0) }
        if (CH == 2) { ONE(0) ONE(1) ONE(0) ONE(1) ONE(0) ONE(1) ONE(0) ONE(1) }
        if (CH == 4) { ONE(0) ONE(1) ONE(2) ONE(3) ONE(0) ONE(1) ONE(2) ONE(3) }
        if (CH == 8) { REP8(ONE) }
        if (CH == 16) { REP8(ONE) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15) }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += f[i] + (float)d[i];
    if (acc == 12345.678f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int OP, int CH = 8>
static void run(const char* name, unsigned long long* dbuf) {
    printf("%-16s", name);
    for (int waves : {4, 8, 16}) {   // per CU: 1, 2, 4 per SIMD; 32 = two 16-wave workgroups per CU (8 per SIMD)
        const int per_wg = waves > 16 ? 16 : waves;
        hipLaunchKernelGGL((rate<OP, CH>), dim3(256 * (waves / per_wg)), dim3(per_wg * 64), 0, 0, dbuf, 1.5f);
        unsigned long long h[16];
        hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < per_wg; ++i) s += (double)h[i];
        // ticks of SIMD time per wave-instruction = elapsed / (512 instructions x waves per SIMD)
        printf("  %d/SIMD: %6.2f", waves / 4, s / per_wg / (CH == 16 ? 1024.0 : 512.0) / (waves / 4));
    }
    printf("\n");
}

__global__ void spin(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long c0 = __builtin_readcyclecounter();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    out[0] = __builtin_amdgcn_s_memtime() - t0;
    out[1] = __builtin_readcyclecounter() - c0;
}

int main() {
    unsigned long long* dbuf;
    hipMalloc(&dbuf, 8192 * 8);
    {   // what a tick is: s_memtime against the events' clock
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, dbuf, 1000ull);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, dbuf, 100000000ull);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
        printf("s_memtime: %llu ticks in %.3f ms = %.1f MHz; readcyclecounter %llu = %.1f MHz\n", h[0], ms, h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
    }
    printf("cycles (s_memtime ticks) of one SIMD per wave-instruction, independent instructions\n");
    run<0>("v_add_f32", dbuf);
    run<1>("v_fma_f32", dbuf);
    run<8>("v_pk_fma_f32", dbuf);
    run<2>("v_add_f64", dbuf);
    run<3>("v_fma_f64", dbuf);
    run<4>("v_mul_f64", dbuf);
    run<5>("v_cvt_f64_f32", dbuf);
    run<6>("v_cvt_f32_f64", dbuf);
    run<7>("v_cndmask_b32", dbuf);
    run<9>("v_rcp_f32", dbuf);
    run<10>("v_mov_b32", dbuf);
    run<11>("v_cvt_f32_i32", dbuf);
    printf("independent chains per wave (the same register is rewritten every CH instructions):\n");
    run<1, 1>("v_fma_f32 x1", dbuf);
    run<1, 2>("v_fma_f32 x2", dbuf);
    run<1, 4>("v_fma_f32 x4", dbuf);
    run<1, 8>("v_fma_f32 x8", dbuf);
    run<1, 16>("v_fma_f32 x16", dbuf);
    run<3, 1>("v_fma_f64 x1", dbuf);
    run<3, 2>("v_fma_f64 x2", dbuf);
    run<3, 4>("v_fma_f64 x4", dbuf);
    run<3, 16>("v_fma_f64 x16", dbuf);
    return 0;
}
