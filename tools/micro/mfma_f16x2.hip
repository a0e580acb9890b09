// Microbenchmark (round 5, stage A of the two-plane fp16 arithmetic): an fp32 operand held as TWO fp16 numbers x * s = hi + lo
// (s a power of two that puts the row into the fp16 range; hi, lo round-to-nearest: 22 significant bits) and all FOUR partial
// products on v_mfma_f32_16x16x32_f16 in TWO instructions per 16-deep K chunk - the 32-deep K of the instruction takes both planes:
//        A[hi|lo] B[lo|lo] = ah bl + al bl        A[hi|lo] B[hi|hi] = ah bh + al bh                  (smallest first)
// against the six bf16 products of gemm_x6 in three instructions (tools/micro/mfma_shape.hip, V = 2).
//   part 1  numerics: 256 x 256 outputs at K = 32 / 768 / 3072 on random data, error against the exact (double) product of the
//           fp32 operands, for  (a) bf16x6 paired  (b) fp16 two planes, four products  (c) fp16, three products (al bl dropped)
//           (d) the fp32 FMA chain (host fmaf)
//   part 2  does the f16 MFMA keep fp16 subnormal operands?  (a lo plane of small values lives there)
//   part 3  rate: same output tile per wave (128 x 64) and LDS-resident operands re-read by ds_read_b128 every step, random data:
//           bf16x6 paired (96 MFMAs + 32 reads per step) against fp16x2 (64 MFMAs + 16 reads per step)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f16x2.hip -o /tmp/mfma_f16x2 && /tmp/mfma_f16x2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- part 1 / 2: one wave per 16 x 16 outputs, fragments straight from global memory ------------------------------------
// planes: [rows][K/16 chunks][P planes][16] 16-bit values (P = 3 bf16 or 2 fp16); A rows 0..255, B rows 256..511
template <int MODE>
__global__ __launch_bounds__(64) void numerics(const unsigned short* __restrict__ pl, float* __restrict__ out, int K) {
    const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
    const int tm = blockIdx.x >> 4, tn = blockIdx.x & 15;
    constexpr int P = MODE == 0 ? 3 : 2;
    const int chunks = K >> 4;
    const unsigned short* arow = pl + ((size_t)(tm * 16 + r16) * chunks) * P * 16;
    const unsigned short* brow = pl + ((size_t)(256 + tn * 16 + r16) * chunks) * P * 16;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    const int half = (g & 1) * 8, second = g >> 1;
    for (int c = 0; c < chunks; ++c) {
        const unsigned short* a = arow + (size_t)c * P * 16;
        const unsigned short* b = brow + (size_t)c * P * 16;
#define LD(p_, plane_) (*reinterpret_cast<const u32x4*>((p_) + (plane_) * 16 + half))
        if (MODE == 0) {   // A[lo|hi] B[hi|lo], A[mid|hi] B[hi|mid], A[hi|mid] B[hi|mid]
            const bf16x8 a_lh = __builtin_bit_cast(bf16x8, LD(a, second ? 0 : 2)), a_mh = __builtin_bit_cast(bf16x8, LD(a, second ? 0 : 1)),
                         a_hm = __builtin_bit_cast(bf16x8, LD(a, second ? 1 : 0));
            const bf16x8 b_hl = __builtin_bit_cast(bf16x8, LD(b, second ? 2 : 0)), b_hm = __builtin_bit_cast(bf16x8, LD(b, second ? 1 : 0));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lh, b_hl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mh, b_hm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hm, b_hm, acc, 0, 0, 0);
        } else {
            const f16x8 a_hl = __builtin_bit_cast(f16x8, LD(a, second));
            const f16x8 b_ll = __builtin_bit_cast(f16x8, LD(b, 1)), b_hh = __builtin_bit_cast(f16x8, LD(b, 0));
            if (MODE == 1) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hl, b_ll, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hl, b_hh, acc, 0, 0, 0);
            } else {   // three products: ah bl, then ah bh + al bh
                const u32x4 z = {0u, 0u, 0u, 0u};
                const f16x8 b_l0 = __builtin_bit_cast(f16x8, second ? z : LD(b, 1));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hl, b_l0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hl, b_hh, acc, 0, 0, 0);
            }
        }
#undef LD
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(tm * 16 + 4 * g + r) * 256 + tn * 16 + r16] = acc[r];
}

// ---- part 3: the rate ---------------------------------------------------------------------------------------------------
constexpr int ROWS = 512, NCHUNK = 2;
// V = 0: bf16x6 paired on the linear 96-byte image (mfma_shape.hip V = 2).  V = 1: fp16x2, 64-byte rows [16 hi][16 lo]; unit u of row
// r sits in slot u ^ f((r >> 2) & 3), f = {0, 3, 2, 1}: the 16-lane groups of a ds_read_b128 then touch every bank once for all
// three forms (A[hi|lo]: lane (r, g) reads unit g; B[hi|hi]: unit g & 1; B[lo|lo]: unit 2 + (g & 1)).
template <int V>
__global__ __launch_bounds__(512, 2) void rate(const u32x4* __restrict__ img, unsigned long long* cyc, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CHUNK = V == 0 ? 96 : 64, STAGE = ROWS * CHUNK;
    for (int i = threadIdx.x; i < NCHUNK * STAGE / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = img[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int r16 = lane & 15, g = lane >> 4;
    if (V == 0) {
        const int second = g >> 1;
        const int rowoff = r16 * CHUNK + ((g & 1) << 4);
        const int a_hm = (wm * 128) * CHUNK + rowoff + (second ? 32 : 0), a_mh = (wm * 128) * CHUNK + rowoff + (second ? 0 : 32),
                  a_lh = (wm * 128) * CHUNK + rowoff + (second ? 0 : 64);
        const int b_hm = (256 + wn * 64) * CHUNK + rowoff + (second ? 32 : 0), b_hl = (256 + wn * 64) * CHUNK + rowoff + (second ? 64 : 0);
        for (int it = 0; it < iters; ++it) {
            const char* sp = smem + (it & (NCHUNK - 1)) * STAGE;
            bf16x8 bhm[4], bhl[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                bhm[y] = *reinterpret_cast<const bf16x8*>(sp + b_hm + y * 16 * CHUNK);
                bhl[y] = *reinterpret_cast<const bf16x8*>(sp + b_hl + y * 16 * CHUNK);
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const bf16x8 ahm = *reinterpret_cast<const bf16x8*>(sp + a_hm + x * 16 * CHUNK);
                const bf16x8 amh = *reinterpret_cast<const bf16x8*>(sp + a_mh + x * 16 * CHUNK);
                const bf16x8 alh = *reinterpret_cast<const bf16x8*>(sp + a_lh + x * 16 * CHUNK);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alh, bhl[y], acc[x][y], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amh, bhm[y], acc[x][y], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm, bhm[y], acc[x][y], 0, 0, 0);
            }
        }
    } else {
        const int f = (4 - ((r16 >> 2) & 3)) & 3;
        const int a_hl = (wm * 128 + r16) * CHUNK + ((g ^ f) << 4);
        const int b_hh = (256 + wn * 64 + r16) * CHUNK + (((g & 1) ^ f) << 4), b_ll = (256 + wn * 64 + r16) * CHUNK + (((2 + (g & 1)) ^ f) << 4);
        for (int it = 0; it < iters; ++it) {
            const char* sp = smem + (it & (NCHUNK - 1)) * STAGE;
            f16x8 bhh[4], bll[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                bhh[y] = *reinterpret_cast<const f16x8*>(sp + b_hh + y * 16 * CHUNK);
                bll[y] = *reinterpret_cast<const f16x8*>(sp + b_ll + y * 16 * CHUNK);
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const f16x8 ahl = *reinterpret_cast<const f16x8*>(sp + a_hl + x * 16 * CHUNK);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahl, bll[y], acc[x][y], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahl, bhh[y], acc[x][y], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static unsigned short bf16_rne(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short b) {
    unsigned u = (unsigned)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static unsigned short f16_rne(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}
static float f16_f(unsigned short b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}

template <int MODE>
static void check(const char* what, const std::vector<unsigned short>& planes, int K, double scale, const std::vector<double>& exact) {
    unsigned short* d;
    float* o;
    (void)hipMalloc(&d, planes.size() * 2);
    (void)hipMalloc(&o, 256 * 256 * 4);
    (void)hipMemcpy(d, planes.data(), planes.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(numerics<MODE>, dim3(256), dim3(64), 0, 0, d, o, K);
    std::vector<float> got(256 * 256);
    (void)hipMemcpy(got.data(), o, got.size() * 4, hipMemcpyDeviceToHost);
    double se = 0, sa = 0, mx = 0;
    for (size_t i = 0; i < got.size(); ++i) {
        const double e = std::fabs((double)got[i] * scale - exact[i]);
        se += e;
        sa += std::fabs(exact[i]);
        if (e > mx) mx = e;
    }
    printf("  %-44s K = %5d: mean |err| / mean |exact| = %.3e   max |err| = %.3e\n", what, K, se / sa, mx);
    (void)hipFree(d);
    (void)hipFree(o);
}

template <int V>
static void run(const char* what, const u32x4* d_img, unsigned long long* d_cyc, float* d_sink, int iters) {
    constexpr int CHUNK = V == 0 ? 96 : 64;
    const size_t lds = (size_t)NCHUNK * ROWS * CHUNK;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(rate<V>, dim3(256), dim3(512), lds, 0, d_img, d_cyc, d_sink, iters / 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<V>, dim3(256), dim3(512), lds, 0, d_img, d_cyc, d_sink, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(256);
    (void)hipMemcpy(c.data(), d_cyc, sizeof(c[0]) * 256, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v / 256;
    const double nmfma = V == 0 ? 96 : 64;
    const double alg = 256.0 * iters * 2.0 * 256 * 256 * 16;   // algorithmic fp32 FLOPs
    printf("  %-36s %8.2f ms  %6.1f TFLOP/s algorithmic  %6.3f PFLOP/s executed  %6.0f ticks per 16-deep step (%4.0f = the matrix pipe's own)\n", what,
           ms, alg / (ms * 1e-3) / 1e12, alg * (V == 0 ? 6.0 : 4.0) / (ms * 1e-3) / 1e15, mean / iters, nmfma * 16 * 2);
}

int main() {
    std::mt19937 rng(5);
    std::normal_distribution<float> nd(0.f, 1.f);
    printf("part 1: error against the exact product of the fp32 operands (A ~ N(0,1), B ~ N(0, 0.02))\n");
    for (int K : {32, 768, 3072}) {
        const int chunks = K / 16;
        std::vector<float> val((size_t)512 * K);
        for (int r = 0; r < 512; ++r)
            for (int k = 0; k < K; ++k) val[(size_t)r * K + k] = nd(rng) * (r < 256 ? 1.f : 0.02f);
        std::vector<double> exact(256 * 256);
        std::vector<float> chain(256 * 256);
        for (int m = 0; m < 256; ++m)
            for (int n = 0; n < 256; ++n) {
                double s = 0;
                float c = 0.f;
                for (int k = 0; k < K; ++k) {
                    s += (double)val[(size_t)m * K + k] * (double)val[(size_t)(256 + n) * K + k];
                    c = fmaf(val[(size_t)m * K + k], val[(size_t)(256 + n) * K + k], c);
                }
                exact[m * 256 + n] = s;
                chain[m * 256 + n] = c;
            }
        {
            double se = 0, sa = 0;
            for (int i = 0; i < 256 * 256; ++i) se += std::fabs((double)chain[i] - exact[i]), sa += std::fabs(exact[i]);
            printf("  %-44s K = %5d: mean |err| / mean |exact| = %.3e\n", "(d) fp32 FMA chain (host fmaf)", K, se / sa);
        }
        std::vector<unsigned short> p3((size_t)512 * chunks * 48), p2((size_t)512 * chunks * 32);
        // fp16 scales: one power of two per operand matrix here (the product kernel: per tensor / per weight row), max -> [2^14, 2^15)
        float amax = 0.f, bmax = 0.f;
        for (int r = 0; r < 512; ++r)
            for (int k = 0; k < K; ++k) (r < 256 ? amax : bmax) = std::fmax(r < 256 ? amax : bmax, std::fabs(val[(size_t)r * K + k]));
        const float sa = std::ldexp(1.f, 14 - (int)std::floor(std::log2(amax))), sb = std::ldexp(1.f, 14 - (int)std::floor(std::log2(bmax)));
        for (int r = 0; r < 512; ++r)
            for (int k = 0; k < K; ++k) {
                const float x = val[(size_t)r * K + k];
                const unsigned short hi = bf16_rne(x);
                const float r1 = x - bf16_f(hi);
                const unsigned short mid = bf16_rne(r1);
                const unsigned short lo = bf16_rne(r1 - bf16_f(mid));
                unsigned short* q = &p3[((size_t)r * chunks + (k >> 4)) * 48 + (k & 15)];
                q[0] = hi, q[16] = mid, q[32] = lo;
                const float v = x * (r < 256 ? sa : sb);
                const unsigned short h = f16_rne(v);
                const unsigned short l = f16_rne(v - f16_f(h));
                unsigned short* q2 = &p2[((size_t)r * chunks + (k >> 4)) * 32 + (k & 15)];
                q2[0] = h, q2[16] = l;
            }
        check<0>("(a) bf16x6, three paired instructions", p3, K, 1.0, exact);
        check<1>("(b) fp16 two planes, four products, two instr.", p2, K, 1.0 / ((double)sa * sb), exact);
        check<2>("(c) fp16 two planes, three products", p2, K, 1.0 / ((double)sa * sb), exact);
    }
    printf("part 2: fp16 subnormal operands\n");
    {
        // a[0][k=0] = hi plane 2^-20 (an fp16 subnormal), b[0][k=0] = hi 2^10: the product is 2^-10 if the MFMA keeps subnormals;
        // a[1][k=0] = lo plane 2^-24 (the smallest subnormal), b = 2^14: 2^-10 again
        const int K = 16, chunks = 1;
        std::vector<unsigned short> p2((size_t)512 * chunks * 32, 0);
        p2[(size_t)0 * 32 + 0] = f16_rne(std::ldexp(1.f, -20));
        p2[(size_t)1 * 32 + 16] = f16_rne(std::ldexp(1.f, -24));
        p2[(size_t)256 * 32 + 0] = f16_rne(1024.f);
        p2[(size_t)257 * 32 + 0] = f16_rne(16384.f);
        unsigned short* d;
        float* o;
        (void)hipMalloc(&d, p2.size() * 2);
        (void)hipMalloc(&o, 256 * 256 * 4);
        (void)hipMemcpy(d, p2.data(), p2.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(numerics<1>, dim3(256), dim3(64), 0, 0, d, o, K);
        std::vector<float> got(256 * 256);
        (void)hipMemcpy(got.data(), o, got.size() * 4, hipMemcpyDeviceToHost);
        printf("  2^-20 (hi plane) x 2^10 = %.6e (kept: %.6e)   2^-24 (lo plane) x 2^14 = %.6e (kept: %.6e)\n", got[0 * 256 + 0],
               std::ldexp(1.0, -10), got[1 * 256 + 1], std::ldexp(1.0, -10));
    }
    printf("part 3: rate (256 CUs x 8 waves, 128 x 64 per wave, operands re-read from LDS every step, random data)\n");
    {
        std::vector<unsigned short> i3((size_t)NCHUNK * ROWS * 48), i2((size_t)NCHUNK * ROWS * 32);
        for (int c = 0; c < NCHUNK; ++c)
            for (int r = 0; r < ROWS; ++r)
                for (int kk = 0; kk < 16; ++kk) {
                    const float x = nd(rng);
                    const unsigned short hi = bf16_rne(x);
                    const float r1 = x - bf16_f(hi);
                    const unsigned short mid = bf16_rne(r1);
                    const unsigned short lo = bf16_rne(r1 - bf16_f(mid));
                    unsigned short* rl = &i3[((size_t)c * ROWS + r) * 48];
                    rl[kk] = hi, rl[16 + kk] = mid, rl[32 + kk] = lo;
                    const float v = x * 2048.f;
                    const unsigned short h = f16_rne(v), l = f16_rne(v - f16_f(f16_rne(v)));
                    const int f = (4 - ((r >> 2) & 3)) & 3;
                    unsigned short* r2 = &i2[((size_t)c * ROWS + r) * 32];
                    r2[(((kk >> 3) ^ f) << 3) + (kk & 7)] = h;
                    r2[(((2 + (kk >> 3)) ^ f) << 3) + (kk & 7)] = l;
                }
        u32x4 *d3, *d2;
        unsigned long long* d_cyc;
        float* d_sink;
        (void)hipMalloc(&d3, i3.size() * 2);
        (void)hipMalloc(&d2, i2.size() * 2);
        (void)hipMalloc(&d_cyc, 256 * 8);
        (void)hipMalloc(&d_sink, 4);
        (void)hipMemcpy(d3, i3.data(), i3.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(d2, i2.data(), i2.size() * 2, hipMemcpyHostToDevice);
        const int iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            run<0>("bf16x6 (3 instr. per chunk)", d3, d_cyc, d_sink, iters);
            run<1>("fp16x2 (2 instr. per chunk)", d2, d_cyc, d_sink, iters);
        }
    }
    return 0;
}
