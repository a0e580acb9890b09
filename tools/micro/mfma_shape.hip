// Microbenchmark: the six-product split-plane arithmetic of gemm_x6 on the two bf16 MFMA shapes of gfx950, same output tile per wave
// (128 x 64), same LDS image (512 rows x 96 B per 16-deep K chunk, planes [16 hi][16 mid][16 lo], halves swapped in rows with bit 3
// set), fragments re-read from LDS every step by ds_read_b128, no global traffic in the loop, random data.
//   V = 0: v_mfma_f32_32x32x16_bf16, one product per instruction: (al bh) (am bm) (ah bl) (am bh) (ah bm) (ah bh)  -> 48 MFMAs, 18 reads / step
//   V = 1: v_mfma_f32_16x16x32_bf16, TWO products per instruction: the 32-deep K of the instruction is [16 k of plane p | 16 k of plane q]:
//          A[h|l] B[l|h] = ah bl + al bh,  A[h|m] B[m|h] = ah bm + am bh,  A[h|m] B[h|m] = ah bh + am bm       -> 96 MFMAs, 28 reads / step
// Reported: wall-clock time (HIP events), executed PFLOP/s, in-kernel cycles, and the error of both against the exact (double) product.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 512, CHUNK = 96, NCHUNK = 2, STAGE = ROWS * CHUNK;

template <int V>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ img, float* out, unsigned long long* cyc, int iters) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < NCHUNK * STAGE / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = img[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (V == 0) {
        floatx16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int r32 = lane & 31;
        const int frag = r32 * CHUNK + ((((lane >> 5) ^ (r32 >> 3)) & 1) << 4);
        const int a_off = (wm * 128) * CHUNK + frag, b_off = (256 + wn * 64) * CHUNK + frag;
        for (int it = 0; it < iters; ++it) {
            const char* sp = smem + (it & (NCHUNK - 1)) * STAGE;
            bf16x8 a[4][3], b[2][3];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[x][pl] = *reinterpret_cast<const bf16x8*>(sp + a_off + x * 32 * CHUNK + pl * 32);
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[y][pl] = *reinterpret_cast<const bf16x8*>(sp + b_off + y * 32 * CHUNK + pl * 32);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int pa = t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0;
                const int pb = t == 2 ? 2 : (t == 1 || t == 4) ? 1 : 0;
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][pa], b[y][pb], acc[x][y], 0, 0, 0);
            }
        }
        if (out && blockIdx.x == 0) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * 128 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int col = wn * 64 + y * 32 + (lane & 31);
                        out[row * 256 + col] = acc[x][y][r];
                    }
        }
    } else {
        floatx4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const int r16 = lane & 15, g = lane >> 4;
        // lane (r, g): k group g of the 32-deep instruction = half (g & 1) of the first (g < 2) or the second (g >= 2) plane of the pair
        const int half = (V == 2 ? (g & 1) : (((g & 1) ^ (r16 >> 3)) & 1)) << 4;   // V == 2: linear image (no half swap): conflict-free for these reads
        const int second = g >> 1;
        const int rowoff = r16 * CHUNK + half;
        const int a_hm = (wm * 128) * CHUNK + rowoff + (second ? 32 : 0);   // A [hi | mid]
        const int a_hl = (wm * 128) * CHUNK + rowoff + (second ? 64 : 0);   // A [hi | lo]
        const int b_hm = (256 + wn * 64) * CHUNK + rowoff + (second ? 32 : 0);   // B [hi | mid]
        const int b_mh = (256 + wn * 64) * CHUNK + rowoff + (second ? 0 : 32);   // B [mid | hi]
        const int b_lh = (256 + wn * 64) * CHUNK + rowoff + (second ? 0 : 64);   // B [lo | hi]
        for (int it = 0; it < iters; ++it) {
            const char* sp = smem + (it & (NCHUNK - 1)) * STAGE;
            bf16x8 bhm[4], bmh[4], blh[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                bhm[y] = *reinterpret_cast<const bf16x8*>(sp + b_hm + y * 16 * CHUNK);
                bmh[y] = *reinterpret_cast<const bf16x8*>(sp + b_mh + y * 16 * CHUNK);
                blh[y] = *reinterpret_cast<const bf16x8*>(sp + b_lh + y * 16 * CHUNK);
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const bf16x8 ahm = *reinterpret_cast<const bf16x8*>(sp + a_hm + x * 16 * CHUNK);
                const bf16x8 ahl = *reinterpret_cast<const bf16x8*>(sp + a_hl + x * 16 * CHUNK);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahl, blh[y], acc[x][y], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm, bmh[y], acc[x][y], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm, bhm[y], acc[x][y], 0, 0, 0);
            }
        }
        if (out && blockIdx.x == 0) {
#pragma unroll
            for (int x = 0; x < 8; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm * 128 + x * 16 + 4 * g + r;
                        const int col = wn * 64 + y * 16 + r16;
                        out[row * 256 + col] = acc[x][y][r];
                    }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
#endif
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

template <int V>
static double run(const char* what, const u32x4* d_img, float* d_out, unsigned long long* d_cyc, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t lds = NCHUNK * STAGE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), lds, 0, d_img, nullptr, d_cyc, iters / 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), lds, 0, d_img, nullptr, d_cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(256);
    (void)hipMemcpy(c.data(), d_cyc, sizeof(c[0]) * 256, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v / 256;
    const double flops = 256.0 * iters * 6.0 * 2 * 256 * 256 * 16;
    printf("%-34s %8.2f ms  %6.3f PFLOP/s executed  %7.0f ticks per step (1536 = the matrix pipe's own)  clock %.2f GHz\n", what, ms,
           flops / (ms * 1e-3) / 1e15, mean / iters, mean / (ms * 1e-3) / 1e9 * 1.0);
    return ms;
}

template <int V>
static void check(const char* what, const u32x4* d_img, float* d_out, unsigned long long* d_cyc, const std::vector<double>& exact, int steps) {
    const size_t lds = NCHUNK * STAGE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(512), lds, 0, d_img, d_out, d_cyc, steps);
    std::vector<float> got(256 * 256);
    (void)hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost);
    double se = 0, sa = 0, mx = 0;
    for (size_t i = 0; i < got.size(); ++i) {
        const double e = std::fabs((double)got[i] - exact[i]);
        se += e;
        sa += std::fabs(exact[i]);
        if (e > mx) mx = e;
    }
    printf("%-34s K = %5d: mean |err| / mean |exact| = %.3e   max |err| = %.3e\n", what, steps * 16, se / sa, mx);
}

int main() {
    // fp32 operands -> split planes on the host (round to nearest even at every step, as csrc/sp3.h)
    std::mt19937 rng(5);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> val((size_t)NCHUNK * ROWS * 16);
    for (auto& v : val) v = nd(rng);
    std::vector<unsigned short> img((size_t)NCHUNK * ROWS * 48), img_lin((size_t)NCHUNK * ROWS * 48);
    std::vector<float> kept(val.size());   // hi + mid + lo (== val: the split is exact)
    for (int c = 0; c < NCHUNK; ++c)
        for (int r = 0; r < ROWS; ++r)
            for (int kk = 0; kk < 16; ++kk) {
                const float x = val[((size_t)c * ROWS + r) * 16 + kk];
                const unsigned short hi = bf16_rne(x);
                const float r1 = x - bf16_f(hi);
                const unsigned short mid = bf16_rne(r1);
                const float r2 = r1 - bf16_f(mid);
                const unsigned short lo = bf16_rne(r2);
                kept[((size_t)c * ROWS + r) * 16 + kk] = bf16_f(hi) + bf16_f(mid) + bf16_f(lo);
                const int h = kk >> 3, j = kk & 7;
                const int ph = h ^ ((r >> 3) & 1);   // rows with bit 3 set hold the halves swapped
                unsigned short* row = &img[((size_t)c * ROWS + r) * 48];
                row[0 * 16 + ph * 8 + j] = hi;
                row[1 * 16 + ph * 8 + j] = mid;
                row[2 * 16 + ph * 8 + j] = lo;
                unsigned short* rl = &img_lin[((size_t)c * ROWS + r) * 48];
                rl[0 * 16 + h * 8 + j] = hi;
                rl[1 * 16 + h * 8 + j] = mid;
                rl[2 * 16 + h * 8 + j] = lo;
            }
    u32x4 *d_img, *d_lin;
    float* d_out;
    unsigned long long* d_cyc;
    (void)hipMalloc(&d_img, img.size() * 2);
    (void)hipMalloc(&d_out, 256 * 256 * 4);
    (void)hipMalloc(&d_cyc, 256 * 8);
    (void)hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&d_lin, img.size() * 2);
    (void)hipMemcpy(d_lin, img_lin.data(), img.size() * 2, hipMemcpyHostToDevice);
    // numerics: `steps` K steps cycling over the NCHUNK chunks = (steps / NCHUNK) x the 32-deep product
    for (int steps : {2, 48, 192}) {
        std::vector<double> exact(256 * 256, 0.0);
        for (int m = 0; m < 256; ++m)
            for (int n = 0; n < 256; ++n) {
                double s = 0;
                for (int c = 0; c < NCHUNK; ++c)
                    for (int kk = 0; kk < 16; ++kk)
                        s += (double)val[((size_t)c * ROWS + m) * 16 + kk] * (double)val[((size_t)c * ROWS + 256 + n) * 16 + kk];
                exact[m * 256 + n] = s * (steps / NCHUNK);
            }
        check<0>("32x32x16, six products", d_img, d_out, d_cyc, exact, steps);
        check<1>("16x16x32, three paired products", d_img, d_out, d_cyc, exact, steps);
        check<2>("16x16x32 paired, linear image", d_lin, d_out, d_cyc, exact, steps);
    }
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        run<0>("32x32x16, six products", d_img, d_out, d_cyc, iters);
        run<1>("16x16x32, three paired products", d_img, d_out, d_cyc, iters);
        run<2>("16x16x32 paired, linear image", d_lin, d_out, d_cyc, iters);
    }
    return 0;
}
