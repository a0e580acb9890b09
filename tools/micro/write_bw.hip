// Streaming rates of the part for stores, loads and copies of buffers far larger than the Infinity Cache (16-byte accesses, consecutive lanes
// on consecutive addresses), and for the store pattern of poly_expansion (blocks that walk down rows, ~5 KB per row at a 77 KB row stride).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/write_bw.hip -o tools/micro/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_write(f32x4* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (f32x4){v, v, v, v};
}
__global__ __launch_bounds__(256) void k_read(const f32x4* p, size_t n, float* out) {
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += p[i];
    if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* a, f32x4* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// read r16 + write w16 16-byte units per thread and row: bands of 256 threads walking down `rows` rows of an image `pitch16` units wide
__global__ __launch_bounds__(256) void k_rows(const f32x4* in, f32x4* out, int rows, int in_pitch16, int out_pitch16, int w16, int band_stride16 = 0, int n16 = 0) {
    const int band = blockIdx.x, seg = blockIdx.y;
    const size_t img = blockIdx.z;
    const f32x4* src = in + img * (size_t)in_pitch16 * rows * gridDim.y + (size_t)seg * rows * in_pitch16 + band * 64;
    f32x4* dst = out + img * (size_t)out_pitch16 * rows * gridDim.y + (size_t)seg * rows * out_pitch16 + band * (band_stride16 ? band_stride16 : 64 * w16);
    const int nq = n16 ? n16 : 64 * w16;
    for (int y = 0; y < rows; ++y) {
        f32x4 v = {0, 0, 0, 0};
        if (threadIdx.x < 64) v = src[(size_t)y * in_pitch16 + threadIdx.x];
        v.x += __shfl(v.y, (threadIdx.x + 1) & 63);
        for (int q = threadIdx.x; q < nq; q += 256) dst[(size_t)y * out_pitch16 + q] = v;
    }
}

// poly_expansion's skeleton with its parts switchable: every thread reads one float per row 5 rows ahead (ring), a barrier per row,
// MATH dependent double FMAs per thread and row, 1230 floats per band row written as 16-byte stores by the first 308 threads
template <bool LOADS, bool BARRIER, int MATH, bool STORES>
__global__ __launch_bounds__(256) void k_poly(const float* in, f32x4* out, int rows, int w, int segs) {
    const int band = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x;
    const size_t img = blockIdx.z;
    const float* col = in + (img * segs + seg) * (size_t)rows * w + band * 246 + tid;
    f32x4* dst = out + ((img * segs + seg) * (size_t)rows * w + band * 246) * 5 / 4;
    float ring[6];
    for (int i = 0; i < 6; ++i) ring[i] = LOADS ? col[(size_t)i * w] : (float)i;
    double acc = tid;
    for (int y = 0; y < rows; y += 6) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float cur = ring[j];
            if (LOADS) ring[j] = col[(size_t)(y + j + 6 < rows ? y + j + 6 : rows - 1) * w];
            if (BARRIER) __syncthreads();
            double a = cur;
#pragma unroll
            for (int m = 0; m < MATH; ++m) a = __builtin_fma(a, 1.0000001, acc);
            acc = a;
            if (STORES) {
                const float v = (float)a;
                for (int q = tid; q < 308; q += 256) dst[(size_t)(y + j) * w * 5 / 4 + q] = (f32x4){v, v, v, v};
            }
        }
    }
    if (acc == 12345.678) out[0] = (f32x4){1, 1, 1, 1};
}

int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    f32x4 *a, *b;
    float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, double moved, auto&& launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms / 5, moved / (ms / 5 * 1e-3) / 1e12);
    };
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d x 256 threads, 4 GiB buffers\n", grid);
        time("  stores only", (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n, 1.f); });
        time("  loads only", (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, o); });
        time("  copy (bytes read + written)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    }
    // the poly_expansion shape at 2160p: 16 bands x 34 segments of 64 rows x 16 images; 1 KB read + 5 KB written per band row
    {
        const int W16in = 16 * 64, W16out = 16 * 64 * 5, rows = 64, segs = 34, imgs = 16;
        const double moved = (double)imgs * segs * rows * (W16in + W16out) * 16.0;
        time("rows: 1 KB in + 5 KB out per band row", moved, [&] { hipLaunchKernelGGL(k_rows, dim3(16, segs, imgs), dim3(256), 0, 0, a, b, rows, W16in, W16out, 5); });
        const double moved1 = (double)imgs * segs * rows * (W16in + W16in) * 16.0;
        time("rows: 1 KB in + 1 KB out per band row", moved1, [&] { hipLaunchKernelGGL(k_rows, dim3(16, segs, imgs), dim3(256), 0, 0, a, b, rows, W16in, W16in, 1); });
    }
    // the same bytes written as wider chunks per block row (the pitch stays one 2160p coefficient row, 76.8 KB): does the rate depend on the chunk?
    for (int bands : {16, 8, 4, 2, 1}) {
        const int pitch16 = 4800, w16 = 4800 / bands / 64 * 64 / 64, rows = 60, segs = 36, imgs = 16;   // w16 units per thread-of-64 and row
        const double moved = (double)imgs * segs * rows * bands * (64.0 * w16 + 64) * 16.0;
        char name[96];
        snprintf(name, sizeof(name), "rows: %d bands, %.1f KB out per band row", bands, 64.0 * w16 * 16 / 1024);
        time(name, moved, [&] { hipLaunchKernelGGL(k_rows, dim3(bands, segs, imgs), dim3(256), 0, 0, a, b, rows, 1024, pitch16, w16); });
    }
    // where a band's chunk starts: poly_expansion's 246-pixel bands are 4920 bytes wide (start 56 b mod 128); 224 pixels = 4480 bytes = 35 lines
    for (int n16 : {308, 320, 280}) {
        const int pitch16 = 4800, rows = 60, segs = 36, imgs = 16, bands = 4800 / n16;
        const double moved = (double)imgs * segs * rows * bands * (n16 + 64.0) * 16.0;
        char name[96];
        snprintf(name, sizeof(name), "rows: bands of %d x 16 B (start %d mod 128)", n16, n16 * 16 % 128);
        time(name, moved, [&] { hipLaunchKernelGGL(k_rows, dim3(bands, segs, imgs), dim3(256), 0, 0, a, b, rows, 1024, pitch16, 5, n16, n16); });
    }
    {   // 2160p level 0: 16 bands x 36 segments of 60 rows x 16 images
        const int w = 3840, rows = 60, segs = 36, imgs = 16;
        const double rd = (double)imgs * segs * rows * 16 * 256 * 4, wr = (double)imgs * segs * rows * 16 * 308 * 16;
        printf("poly_expansion skeleton (2160p, 16 images): bytes = loads + stores actually issued\n");
#define RUN(L, B, M, S, name) time(name, (L ? rd : 0) + (S ? wr : 0), [&] { hipLaunchKernelGGL((k_poly<L, B, M, S>), dim3(16, segs, imgs), dim3(256), 0, 0, (const float*)a, b, rows, w, segs); })
        RUN(false, false, 0, true, "  stores only");
        RUN(true, false, 0, true, "  loads + stores");
        RUN(true, true, 0, true, "  loads + barrier + stores");
        RUN(true, true, 40, true, "  loads + barrier + 40 dfma + stores");
        RUN(true, true, 150, true, "  loads + barrier + 150 dfma + stores");
        RUN(true, true, 150, false, "  loads + barrier + 150 dfma");
        RUN(false, true, 150, true, "  barrier + 150 dfma + stores");
        RUN(false, false, 150, true, "  150 dfma + stores");
    }
    return 0;
}
