// Farneback dense optical flow + the reference's flow visualisation on gfx950.
//
// Reference call sites (xinyiW915/ReLaX-VQA):
//   src/main_fragment_layerstack.py:313-315  cv2.calcOpticalFlowFarneback(gray(orig), gray(next), None, 0.5, 3, 15, 3, 5, 1.2, 0)
//   src/main_fragment_layerstack.py:162-175  flow_to_rgb (cartToPolar, NORM_MINMAX, hue = ang*180/pi/2, HSV -> BGR)
// The algorithm is OpenCV's (opencv-python 4.9, modules/video/src/optflowgf.cpp), restated - not ported from source,
// which is not available here - and pinned by tolerance to the reference's example flow PNGs (oracle/flow_ref.py).
// All stages are streaming kernels, batched over the pairs of a clip (one launch per stage and pyramid level for ALL pairs, so the
// coarse levels still fill the chip):
//   pyramid_fused: the uint8 frames -> gray (fixed-point BGR2GRAY) -> the four level inputs (Gaussian blur of the full-resolution frame,
//   3 / 3 / 9 / 19 taps, reflect-101, then the linear resize) in ONE pass over the frame's bytes   [sides not multiples of 8: flow_gray +
//   gauss3_v4 / gauss_pass / gauss_h_sampled / gauss_v_sampled_resize / resize_linear_f32 per level]
//   per level, coarsest first: poly_expansion (both passes of the polynomial expansion) -> 3 x flow_iteration (matrix update + 15x15 box
//   blur in double + 2x2 solve in one kernel; the first one takes the x2 linear upsampling of the coarser level's flow on the fly, the
//   last one of the finest level the magnitude range)   [flow_fused = 0: update_matrices_k + box_solve_fused, the matrix plane through HBM]
//   flow_visualise: fastAtan2, normalisation, 8-bit HSV -> BGR (float formula, truncated).
// LAB_NOTES.md section 3.3 has the measurements and what was tried and not kept.
#include <cfloat>
#include <cmath>

#include <atomic>

#include "relax_internal.h"

// This file is compiled with -ffp-contract=off (Makefile): a * b + c is a multiply and an add, as in the oracle and in OpenCV's scalar code,
// in EVERY kernel - left to itself the compiler fuses such expressions differently from one kernel to the next (also through the
// __fmul_rn / __fadd_rn wrappers, which are plain operators in this toolchain), and results would depend on which kernel evaluated
// them.  Where a fused multiply-add is wanted it is written: fmaf / fma / __fmaf_rn / __fma_rn.
namespace relax {

// Workgroup barrier for data that goes through LDS only: waits for this wave's LDS operations, not for its global loads and
// stores (__syncthreads() also waits vmcnt(0): in a row-walking kernel that drains the rows requested ahead and the stores of the
// row just written at every row).
#define RELAX_LDS_BARRIER()                                   \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_s_barrier();                         \
    } while (0)

constexpr int POLY_N = 5;
constexpr int WINSIZE = 15;
constexpr int ITERS = 3;
constexpr int MAX_GAUSS = 32;
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: 16-byte loads / stores of the 4-pixel kernels

struct PolyConsts {
    float g[POLY_N + 1], xg[POLY_N + 1], xxg[POLY_N + 1];
    double ig11, ig03, ig33, ig55;
};
struct GaussKernel {
    float k[MAX_GAUSS];
    int ksize;
};

__device__ inline int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);   // (only reached for n smaller than the kernel radius)
}
__device__ inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }
// cv::resize's source coordinate of destination index d, (d + 0.5) * scale - 0.5 in double, and the flow magnitude: one definition each
// with named roundings - a contraction into an FMA in one kernel and not in another would make results depend on the kernel
__device__ __forceinline__ float resize_src_coord(int d, double scale) { return (float)__dsub_rn(__dmul_rn(d + 0.5, scale), 0.5); }
__device__ __forceinline__ float flow_magnitude(float x, float y) { return sqrtf(__fmaf_rn(y, y, __fmul_rn(x, x))); }

// uint8 BGR frame pair -> float gray [P][2][H][W]; (B*1868 + G*9617 + R*4899 + 2^13) >> 14
__global__ __launch_bounds__(256) void flow_gray(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ next,
                                                 int64_t pair_stride, int HW, float* __restrict__ gray) {
    const int pix = blockIdx.x * 256 + threadIdx.x;      // grid (HW / 256, 1, P * 2): no 64-bit div / mod per pixel
    if (pix >= HW) return;
    const int img = blockIdx.z;                           // pair * 2 + which
    const uint8_t* src = ((img & 1) ? next : orig) + (int64_t)(img >> 1) * pair_stride + (int64_t)pix * 3;
    gray[(int64_t)img * HW + pix] = (float)(((int)src[0] * 1868 + (int)src[1] * 9617 + (int)src[2] * 4899 + (1 << 13)) >> 14);
}

// 4 pixels per thread: 12 bytes in as three dwords, one 16-byte store (frames whose pixel count and pointers are multiples of 4)
__global__ __launch_bounds__(256) void flow_gray_v4(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ next,
                                                    int64_t pair_stride, int HW, float* __restrict__ gray) {
    const int pix = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (pix >= HW) return;
    const int img = blockIdx.z;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(((img & 1) ? next : orig) + (int64_t)(img >> 1) * pair_stride + (int64_t)pix * 3);
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];   // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
#define RELAX_GRAY(b_, g_, r_) (float)(((int)(b_) * 1868 + (int)(g_) * 9617 + (int)(r_) * 4899 + (1 << 13)) >> 14)
    const f32x4 o = {RELAX_GRAY(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255), RELAX_GRAY(w0 >> 24, w1 & 255, (w1 >> 8) & 255),
                     RELAX_GRAY((w1 >> 16) & 255, w1 >> 24, w2 & 255), RELAX_GRAY((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24)};
#undef RELAX_GRAY
    *reinterpret_cast<f32x4*>(gray + (int64_t)img * HW + pix) = o;
}

// The taps arrive by value in the kernel argument; indexing that with a loop counter costs one dependent scalar load per tap,
// so every block first copies them into LDS (and the common 3-tap blur of the two finest levels is unrolled from registers).
__device__ inline void stage_taps(const GaussKernel& gk, float* sk) {
    if (threadIdx.x < MAX_GAUSS) sk[threadIdx.x] = gk.k[threadIdx.x];
    __syncthreads();
}

template <bool VERTICAL>
__global__ __launch_bounds__(256) void gauss_pass(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                  GaussKernel gk) {
    __shared__ float sk[MAX_GAUSS];
    stage_taps(gk, sk);
    const int x = blockIdx.x * 256 + threadIdx.x;         // grid (W / 256, H, images)
    if (x >= W) return;
    const int y = blockIdx.y;
    const float* img = src + (int64_t)blockIdx.z * ((int64_t)W * H);
    const int r = gk.ksize / 2;
    float acc = 0.f;
    for (int t = 0; t < gk.ksize; ++t) {
        const float v = VERTICAL ? img[(int64_t)reflect101(y + t - r, H) * W + x] : img[(int64_t)y * W + reflect101(x + t - r, W)];
        acc = fmaf(sk[t], v, acc);
    }
    dst[(int64_t)blockIdx.z * ((int64_t)W * H) + (int64_t)y * W + x] = acc;
}

// 3-tap blur (pyramid levels 0 and 1), both passes in one kernel, 4 pixels per thread: reads the 3 rows once with 16-byte
// loads, same products and order as gauss_pass<false> then gauss_pass<true> (horizontal first, rounded to float, then vertical).
__device__ __forceinline__ f32x4 fma4(float k, f32x4 v, f32x4 a) {
    return (f32x4){fmaf(k, v.x, a.x), fmaf(k, v.y, a.y), fmaf(k, v.z, a.z), fmaf(k, v.w, a.w)};
}
__global__ __launch_bounds__(256) void gauss3_v4(const float* __restrict__ src, float* __restrict__ dst, int H, int W, float k0,
                                                 float k1, float k2) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;  // grid (W / 1024, H, images)
    if (x0 >= W) return;
    const int y = blockIdx.y;
    const float* img = src + (int64_t)blockIdx.z * ((int64_t)W * H);
    f32x4 hrow[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float* row = img + (int64_t)reflect101(y + t - 1, H) * W;
        const f32x4 c = *reinterpret_cast<const f32x4*>(row + x0);
        const float l = row[reflect101(x0 - 1, W)], rr = row[reflect101(x0 + 4, W)];
        // acc = 0 + k0*v(-1); acc += k1*v(0); acc += k2*v(+1)
        f32x4 a = (f32x4){l, c.x, c.y, c.z} * k0;
        a = fma4(k1, c, a);
        a = fma4(k2, (f32x4){c.y, c.z, c.w, rr}, a);
        hrow[t] = a;
    }
    f32x4 o = hrow[0] * k0;
    o = fma4(k1, hrow[1], o);
    o = fma4(k2, hrow[2], o);
    *reinterpret_cast<f32x4*>(dst + (int64_t)blockIdx.z * ((int64_t)W * H) + (int64_t)y * W + x0) = o;
}

// a * (1 - t) + b * t with every operation rounded on its own (no fused multiply-add: what cv::resize and the oracle compute); every
// bilinear step of the pyramid goes through this one function, so the stand-alone resize, the sampled blur + resize and the copy inside
// update_matrices_k<true> agree bit for bit
__device__ inline float lerp_rn(float a, float b, float t) { return __fadd_rn(__fmul_rn(a, __fsub_rn(1.f, t)), __fmul_rn(b, t)); }

// Coarse pyramid levels (scale 1/4, 1/8): cv::resize only samples the blurred frame at 2 columns x 2 rows per output
// pixel, so the blur is evaluated at those samples only - same taps, same order, same interpolation arithmetic as
// gauss_pass + resize_linear_f32 (bit-identical), a fraction of the work.
__device__ inline void linear_tap(int d, double scale, int n_in, int* s0, int* s1, float* f) {
    float fx = resize_src_coord(d, scale);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= n_in - 1) { fx = 0.f; sx = n_in - 1; }
    *s0 = sx;
    *s1 = sx + 1 < n_in ? sx + 1 : n_in - 1;
    *f = fx;
}

// horizontal blur at the two sample columns of every output column: src [B][H][W] -> tmp [B][H][w][2].  A block first copies
// the stretch of the source row its 256 outputs tap (reflected at the borders while copying) into LDS with coalesced loads;
// the 9 / 19 taps per output then come from LDS instead of strided global loads.
constexpr int GH_SEG = 1280;   // floats of source row per block: 128 output columns x scale <= 8, plus the taps
__global__ __launch_bounds__(256) void gauss_h_sampled(const float* __restrict__ src, float* __restrict__ tmp, int H, int W,
                                                       int w, double scale_x, GaussKernel gk) {
    __shared__ float sk[MAX_GAUSS];
    __shared__ float seg[GH_SEG];
    const int e0 = blockIdx.x * 256;                      // grid (2w / 256, H, images): e = dx * 2 + j
    const int64_t row = (int64_t)blockIdx.z * H + blockIdx.y;            // b*H + y
    const float* r_ = src + row * W;
    const int r = gk.ksize / 2;
    int a0, a1, b0, b1;
    float f;
    linear_tap(e0 >> 1, scale_x, W, &a0, &a1, &f);
    const int dx_last = (e0 + 255) >> 1 < w ? (e0 + 255) >> 1 : w - 1;
    linear_tap(dx_last, scale_x, W, &b0, &b1, &f);
    const int xlo = a0 - r;
    const int n = b1 + r - xlo + 1;                       // <= GH_SEG (checked by the launcher)
    if (threadIdx.x < MAX_GAUSS) sk[threadIdx.x] = gk.k[threadIdx.x];
    for (int i = threadIdx.x; i < n; i += 256) seg[i] = r_[reflect101(xlo + i, W)];
    __syncthreads();
    const int e = e0 + threadIdx.x;
    if (e >= 2 * w) return;
    int s0, s1;
    linear_tap(e >> 1, scale_x, W, &s0, &s1, &f);
    const int x = (e & 1) ? s1 : s0;
    float acc = 0.f;
    for (int t = 0; t < gk.ksize; ++t) acc = fmaf(sk[t], seg[x + t - r - xlo], acc);
    tmp[row * (2 * w) + e] = acc;
}

// vertical blur at the two sample rows + the bilinear combine: tmp [B][H][w][2] -> dst [B][h][w]
__global__ __launch_bounds__(256) void gauss_v_sampled_resize(const float* __restrict__ tmp, float* __restrict__ dst, int H,
                                                              int W, int h, int w, double scale_y, double scale_x,
                                                              GaussKernel gk) {
    __shared__ float sk[MAX_GAUSS];
    stage_taps(gk, sk);
    const int dx = blockIdx.x * 256 + threadIdx.x;       // grid (w / 256, h, images)
    if (dx >= w) return;
    const int dy = blockIdx.y;
    const int64_t b = blockIdx.z;
    int sy0, sy1, sx0, sx1;
    float fy, fx;
    linear_tap(dy, scale_y, H, &sy0, &sy1, &fy);
    linear_tap(dx, scale_x, W, &sx0, &sx1, &fx);
    const float* img = tmp + b * ((int64_t)H * w * 2) + dx * 2;
    const int r = gk.ksize / 2;
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;   // blurred (row sy0|sy1, col sx0|sx1)
    for (int t = 0; t < gk.ksize; ++t) {
        const float k = sk[t];
        const float* p0 = img + (int64_t)reflect101(sy0 + t - r, H) * w * 2;
        const float* p1 = img + (int64_t)reflect101(sy1 + t - r, H) * w * 2;
        a00 = fmaf(k, p0[0], a00);
        a01 = fmaf(k, p0[1], a01);
        a10 = fmaf(k, p1[0], a10);
        a11 = fmaf(k, p1[1], a11);
    }
    dst[(b * h + dy) * w + dx] = lerp_rn(lerp_rn(a00, a01, fx), lerp_rn(a10, a11, fx), fy);
}

// cv::resize INTER_LINEAR on float [B][H][W][C] -> [B][h][w][C], result scaled by mul (flow upsampling: 1/pyr_scale)
template <int C>
__global__ __launch_bounds__(256) void resize_linear_f32(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                         int h, int w, double scale_y, double scale_x, float mul) {
    const int e = blockIdx.x * 256 + threadIdx.x;        // grid (w * C / 256, h, images): e = dx * C + c
    if (e >= w * C) return;
    const int c = e % C;
    const int dx = e / C;
    const int dy = blockIdx.y;
    const int64_t b = blockIdx.z;
    float fx = resize_src_coord(dx, scale_x);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= W - 1) { fx = 0.f; sx = W - 1; }
    float fy = resize_src_coord(dy, scale_y);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    if (sy < 0) { fy = 0.f; sy = 0; }
    if (sy >= H - 1) { fy = 0.f; sy = H - 1; }
    const int sx1 = sx + 1 < W ? sx + 1 : W - 1;
    const int sy1 = sy + 1 < H ? sy + 1 : H - 1;
    const float* im = src + b * ((int64_t)H * W * C);
    const float r0 = lerp_rn(im[((int64_t)sy * W + sx) * C + c], im[((int64_t)sy * W + sx1) * C + c], fx);
    const float r1 = lerp_rn(im[((int64_t)sy1 * W + sx) * C + c], im[((int64_t)sy1 * W + sx1) * C + c], fx);
    dst[((b * h + dy) * w) * C + e] = __fmul_rn(lerp_rn(r0, r1, fy), mul);
}

// ---- the whole pyramid of one frame in ONE pass over its bytes -------------------------------------------------------------
// OpenCV's Farneback builds every pyramid level from the full-resolution frame: GaussianBlur (3 / 3 / 9 / 19 taps for the
// scales 1, 1/2, 1/4, 1/8) then cv::resize.  As separate kernels (flow_gray, gauss3_v4 x 2, resize_linear_f32, gauss_h_sampled x 2,
// gauss_v_sampled_resize x 2) that is 42 bytes of HBM traffic per pixel and frame; here a block walks down a band of 256 columns
// once, reading the uint8 frame (3 bytes per pixel) and writing the four level inputs (4 + 1 + 1/4 + 1/16 floats per pixel):
//   per input row: 68 threads turn 3 dwords each into 4 gray values (the row lives in LDS, reflect-101 halo of 8 columns);
//   every thread forms the two horizontal 3-tap blurs of its column (levels 0 / 1) and keeps the last rows in registers: the
//   vertical 3-tap of level 0 gives I0 row y - 1, that of level 1 + the 2 x 2 bilinear combine (lane pairs) I1 row (y - 1) / 2;
//   threads 0-127 / 128-191 form the horizontal 9- / 19-tap blurs at the two sample columns of every level-2 / level-3 output
//   column into LDS rings of 16 / 32 rows, and when the last row of an output row's window has arrived, its vertical taps at
//   the two sample rows and the bilinear combine.
// Exact scales only (H and W multiples of 8: the sample positions are then 2 d + {0, 1}, 4 d + {1, 2}, 8 d + {3, 4} with weights
// 1/2; other sizes take the separate kernels), the same products in the same order as those kernels: bit-identical level inputs.
// One barrier per row: the gray row is double-buffered, and the vertical part of levels 2 / 3 runs one row late.
struct PyramidTaps {
    float k0[3], k1[3], k2[9], k3[19];
};
constexpr int PYR_HALO = 8;
constexpr int PYR_G = 256 + 2 * PYR_HALO;     // gray values per row and band
// the value of the next lane of the wave (lane 63 keeps its own: only even lanes use the result) as a DPP wave shift - a vector
// instruction, where __shfl_down goes through the LDS crossbar (ds_bpermute_b32) of a kernel that is short of LDS cycles
__device__ __forceinline__ float next_lane(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__global__ __launch_bounds__(256) void pyramid_fused(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ next, int64_t pair_stride,
                                                     int H, int W, float* __restrict__ I0, float* __restrict__ I1, float* __restrict__ I2,
                                                     float* __restrict__ I3, int seg, const PyramidTaps tp) {
    __shared__ __attribute__((aligned(16))) float g[2][PYR_G];
    __shared__ float ring2[16][128];
    __shared__ float ring3[32][64];
    const int x = threadIdx.x;
    const int c0 = blockIdx.x * 256;
    const int c = c0 + x;
    const int y0 = blockIdx.y * seg;                       // seg is a multiple of 8
    const int y1 = y0 + seg < H ? y0 + seg : H;
    const int img = blockIdx.z;                            // pair * 2 + which
    const uint8_t* src = ((img & 1) ? next : orig) + (int64_t)(img >> 1) * pair_stride;
    const int64_t HW = (int64_t)H * W;
    float* o0 = I0 + (int64_t)img * HW;
    float* o1 = I1 + (int64_t)img * (HW / 4);
    float* o2 = I2 + (int64_t)img * (HW / 16);
    float* o3 = I3 + (int64_t)img * (HW / 64);
    // stage A role: thread q < 68 owns the 4 gray values of columns c0 - 8 + 4 q .. + 3
    const int cq = c0 - PYR_HALO + 4 * x;
    const bool loader = x < PYR_G / 4;
    const bool fast = loader && cq >= 0 && cq + 3 < W;     // whole group inside the row: three aligned dwords
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    float slow[4] = {0.f, 0.f, 0.f, 0.f};
#define PYR_GRAY(b_, g_, r_) (float)(((int)(b_) * 1868 + (int)(g_) * 9617 + (int)(r_) * 4899 + (1 << 13)) >> 14)
    auto request_row = [&](int y) {
        if (!loader) return;
        const uint8_t* row = src + (int64_t)reflect101(y, H) * W * 3;
        if (fast) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(row + (int64_t)cq * 3);
            w0 = p[0]; w1 = p[1]; w2 = p[2];
        } else {                                           // image edge: reflected columns, byte loads
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint8_t* q = row + (int64_t)reflect101(cq + i, W) * 3;
                slow[i] = PYR_GRAY(q[0], q[1], q[2]);
            }
        }
    };
    float h0m1 = 0.f, h0m2 = 0.f, h1m1 = 0.f, h1m2 = 0.f, b1prev = 0.f;
    request_row(y0 - 6);
    for (int y = y0 - 6; y <= y1 + 6; ++y) {
        const int par = y & 1;
        if (loader && y <= y1 + 5) {                       // A: the gray values of row y, then the request for row y + 1
            f32x4 gv;                                      // one 16-byte store (four scalar stores at a 16-byte lane stride conflict 4 ways)
            if (fast) {
                gv.x = PYR_GRAY(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255);     // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
                gv.y = PYR_GRAY(w0 >> 24, w1 & 255, (w1 >> 8) & 255);
                gv.z = PYR_GRAY((w1 >> 16) & 255, w1 >> 24, w2 & 255);
                gv.w = PYR_GRAY((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24);
            } else {
                gv = (f32x4){slow[0], slow[1], slow[2], slow[3]};
            }
            *reinterpret_cast<f32x4*>(&g[par][4 * x]) = gv;
            if (y < y1 + 5) request_row(y + 1);
        }
        RELAX_LDS_BARRIER();
        const float* gr = g[par];
        const int gi = x + PYR_HALO;
        if (y <= y1 + 5) {
            // B: horizontal passes of row y
            float h0c = gr[gi - 1] * tp.k0[0];
            h0c = fmaf(tp.k0[1], gr[gi], h0c);
            h0c = fmaf(tp.k0[2], gr[gi + 1], h0c);
            float h1c = gr[gi - 1] * tp.k1[0];
            h1c = fmaf(tp.k1[1], gr[gi], h1c);
            h1c = fmaf(tp.k1[2], gr[gi + 1], h1c);
            // levels 0 / 1: output row r = y - 1 (rows r - 1, r, r + 1 of the horizontal blurs)
            const int r = y - 1;
            float v0 = h0m2 * tp.k0[0];
            v0 = fmaf(tp.k0[1], h0m1, v0);
            v0 = fmaf(tp.k0[2], h0c, v0);
            float b1 = h1m2 * tp.k1[0];
            b1 = fmaf(tp.k1[1], h1m1, b1);
            b1 = fmaf(tp.k1[2], h1c, b1);
            const bool in_rows = r >= y0 && r < y1;
            if (in_rows && c < W) o0[(int64_t)r * W + c] = v0;
            // level 1: rows 2 d, 2 d + 1 and columns 2 d, 2 d + 1 of the blurred frame, weights 1/2 (all lanes take part in the exchange)
            const float a01 = next_lane(b1prev), a11 = next_lane(b1);
            if (in_rows && (r & 1) && !(x & 1) && c < W)
                o1[(int64_t)(r >> 1) * (W / 2) + (c >> 1)] = lerp_rn(lerp_rn(b1prev, a01, 0.5f), lerp_rn(b1, a11, 0.5f), 0.5f);
            h0m2 = h0m1; h0m1 = h0c;
            h1m2 = h1m1; h1m1 = h1c;
            b1prev = b1;
            if (x < 128) {                                 // level 2: the 9-tap blur at columns 4 d + 1, 4 d + 2
                const int gs = PYR_HALO + 4 * (x >> 1) + 1 + (x & 1);
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(tp.k2[t], gr[gs + t - 4], acc);
                ring2[y & 15][x] = acc;
            } else if (x < 192) {                          // level 3: the 19-tap blur at columns 8 d + 3, 8 d + 4
                const int e = x - 128;
                const int gs = PYR_HALO + 8 * (e >> 1) + 3 + (e & 1);
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 19; ++t) acc = fmaf(tp.k3[t], gr[gs + t - 9], acc);
                ring3[y & 31][e] = acc;
            }
        }
        // levels 2 / 3, one row late (the rows up to y - 1 are in the rings since the barrier above)
        const int yy = y - 1;
        if (x < 128) {
            if (((yy - 6) & 3) == 0) {                     // uniform: the window of output row d = (yy - 6) / 4 is complete
                const int d = (yy - 6) >> 2;
                if (d >= (y0 >> 2) && d < (y1 >> 2)) {
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        a0 = fmaf(tp.k2[t], ring2[(4 * d + 1 + t - 4) & 15][x], a0);
                        a1 = fmaf(tp.k2[t], ring2[(4 * d + 2 + t - 4) & 15][x], a1);
                    }
                    const float a01 = next_lane(a0), a11 = next_lane(a1);
                    const int dx = (c0 >> 2) + (x >> 1);
                    if (!(x & 1) && dx < (W >> 2)) o2[(int64_t)d * (W >> 2) + dx] = lerp_rn(lerp_rn(a0, a01, 0.5f), lerp_rn(a1, a11, 0.5f), 0.5f);
                }
            }
        } else if (x < 192) {
            if (((yy - 13) & 7) == 0) {
                const int d = (yy - 13) >> 3;
                if (d >= (y0 >> 3) && d < (y1 >> 3)) {
                    const int e = x - 128;
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int t = 0; t < 19; ++t) {
                        a0 = fmaf(tp.k3[t], ring3[(8 * d + 3 + t - 9) & 31][e], a0);
                        a1 = fmaf(tp.k3[t], ring3[(8 * d + 4 + t - 9) & 31][e], a1);
                    }
                    const float a01 = next_lane(a0), a11 = next_lane(a1);
                    const int dx = (c0 >> 3) + (e >> 1);
                    if (!(e & 1) && dx < (W >> 3)) o3[(int64_t)d * (W >> 3) + dx] = lerp_rn(lerp_rn(a0, a01, 0.5f), lerp_rn(a1, a11, 0.5f), 0.5f);
                }
            }
        }
    }
#undef PYR_GRAY
}

// FarnebackPolyExp, both passes in one kernel: I [B][h][w] -> R, per pair [2][h][w][4] + [2][h][w] (c0..c3 of a pixel as one 16-byte
// unit, c4 in a plane of its own: two sequential streams per frame, every access a whole aligned unit.  History: five planes [5][h][w] -
// coalesced dword rows, sixteen streams per block - took 0.73 ms per 2160p pair, the five coefficients side by side [h][w][5] 0.68).  A block owns a band of 246 output columns (256 threads = 246 + the 5-column halo on each side, replicated border)
// and a segment of rows, and walks down the rows:
//   * vertical 11-tap filters (float, as OpenCV), one COLUMN per thread: the 11 rows of the window live in a register ring of 16
//     slots (static indices: the row loop is unrolled over the 16), so every input value is loaded once per segment - with one row
//     per block each thread re-read its 11 rows and the kernel ran at the L2 rate on 84 bytes per pixel; the other 5 slots hold the
//     rows requested ahead;
//   * the three vertical results go through LDS (double-buffered: one barrier per row), and the horizontal 11-tap part (double
//     accumulators, as OpenCV) reads its neighbours there.
// The intermediate [h][w][3] plane never goes to HBM: 4 + 20 bytes per pixel.
constexpr int POLY_OUT = 256 - 2 * POLY_N;   // 246 (224-pixel bands, whose rows start on 128-byte lines, measured the same: profiles/r04_write_bw.txt has the pattern alone)
constexpr int POLY_RING = 2 * POLY_N + 1;    // 11
constexpr int POLY_SLOTS = 16;               // the window + 5 rows in flight
__global__ __launch_bounds__(256) void poly_expansion(const float* __restrict__ I, float* __restrict__ R, int h, int w, int seg,
                                                      PolyConsts pc) {
    __shared__ f32x4 lt[2][256];                           // (t0, t1, t2, -) of a column side by side: one 16-byte read per neighbour
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const int x = blockIdx.x * POLY_OUT - POLY_N + tid;     // grid (w / 246, h / seg, images)
    const int xc = clampi(x, 0, w - 1);
    const int y0 = blockIdx.y * seg;
    const int y1 = y0 + seg < h ? y0 + seg : h;
    const float* col = I + (int64_t)blockIdx.z * ((int64_t)w * h) + xc;
    const bool writer = tid >= POLY_N && tid < POLY_N + POLY_OUT && x < w;
    // at the row y = y0 + s: slot (s + i) % 16 holds row y - 5 + i (clamped): i = 0 .. 10 the window, i = 11 .. 15 rows already
    // requested (with one row requested per step a thread had ONE 4-byte load in flight and the kernel waited on the read latency:
    // 3.3 TB/s; five rows ahead the reads are no longer what it waits for)
    float ring[POLY_SLOTS];
#pragma unroll
    for (int i = 0; i < POLY_SLOTS; ++i) ring[i] = col[(int64_t)clampi(y0 - POLY_N + i, 0, h - 1) * w];
    // Output layout (a pair = two consecutive images z): [frame][h][w][4] holds c0..c3 of a pixel as one aligned 16-byte unit, then
    // [frame][h][w] holds c4 - 40 bytes per pixel and pair as before, but a wave's stores (here) and loads (matrix_request) are whole
    // consecutive 16-byte / 4-byte units.  With the five coefficients side by side (20 bytes per pixel: 16- + 4-byte accesses at a
    // 20-byte lane stride) the texture-address FIFO of the iteration kernels was full 3.1 x as often as a memory instruction was
    // active (profiles/r04_flow_ta.txt): their producers waited for the load unit, not for HBM.
    const int64_t hw = (int64_t)h * w;
    float* const Rp = R + (int64_t)(blockIdx.z >> 1) * hw * 10;
    float* const R4 = Rp + (int64_t)(blockIdx.z & 1) * hw * 4;
    float* const R1 = Rp + hw * 8 + (int64_t)(blockIdx.z & 1) * hw;
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // (odd h * w: the second pair starts 8 bytes off a 16-byte line)
    int par = 0;
    for (int yb = y0; yb < y1; yb += POLY_SLOTS) {
#pragma unroll
        for (int j = 0; j < POLY_SLOTS; ++j) {
            const int y = yb + j;
            if (y < y1) {                                   // uniform over the block
                const float entering = col[(int64_t)clampi(y + POLY_N + 1 + (POLY_SLOTS - POLY_RING), 0, h - 1) * w];   // row y + 11
                // pairs of independent fp32 operations as 2-vectors (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per
                // instruction, the same roundings): 143 instead of 172 vector instructions per pixel
                f32x2 t02 = {ring[(j + POLY_N) % POLY_SLOTS] * pc.g[0], 0.f};   // (t0, t2)
                float t1 = 0.f;
#pragma unroll
                for (int k = 1; k <= POLY_N; ++k) {
                    const float up = ring[(j + POLY_N - k) % POLY_SLOTS];
                    const float dn = ring[(j + POLY_N + k) % POLY_SLOTS];
                    const float p = up + dn;
                    t02 = __builtin_elementwise_fma((f32x2){pc.g[k], pc.xxg[k]}, (f32x2){p, p}, t02);
                    t1 = fmaf(pc.xg[k], dn - up, t1);
                }
                ring[j] = entering;                         // row y + 11 takes the slot of row y - 5
                const float t0 = t02.x, t2 = t02.y;
                lt[par][tid] = (f32x4){t0, t1, t2, 0.f};
                RELAX_LDS_BARRIER();
                if (writer) {
                    double b1 = t0 * pc.g[0], b2 = 0, b3 = t1 * pc.g[0], b4 = 0, b5 = t2 * pc.g[0], b6 = 0;
#pragma unroll
                    for (int k = 1; k <= POLY_N; ++k) {
                        const f32x4 P = lt[par][tid + k], M = lt[par][tid - k];
                        const f32x2 p01 = {P.x, P.y}, m01 = {M.x, M.y};
                        const f32x2 s01 = p01 + m01;                               // (p0 + m0, p1 + m1)
                        const f32x2 d01 = (p01 - m01) * (f32x2){pc.xg[k], pc.xg[k]};   // ((p0 - m0) xg, (p1 - m1) xg)
                        const float s2 = P.z + M.z;
                        const double tg = s01.x;
                        b1 = fma(tg, (double)pc.g[k], b1);
                        b4 = fma(tg, (double)pc.xxg[k], b4);
                        b2 += d01.x;
                        b3 += s01.y * pc.g[k];
                        b6 += d01.y;
                        b5 += s2 * pc.g[k];
                    }
                    const int64_t px = (int64_t)y * w + x;
                    const float o1 = (float)(b2 * pc.ig11);
                    const float o0 = (float)(b3 * pc.ig11);
                    const float o3 = (float)(b1 * pc.ig03 + b4 * pc.ig33);
                    const float o2 = (float)(b1 * pc.ig03 + b5 * pc.ig33);
                    *reinterpret_cast<f32x4u*>(R4 + px * 4) = (f32x4u){o0, o1, o2, o3};
                    R1[px] = (float)(b6 * pc.ig55);
                }
                par ^= 1;   // the next row writes the other buffer: its barrier orders these reads before the buffer's reuse
            }
        }
    }
}

// FarnebackUpdateMatrices for one pixel, in two halves so that a kernel can keep the loads of one pixel in flight while it computes
// another: `matrix_request` loads what the entries need (R0 = the pixel's 5 expansion coefficients in frame 0, the 2 x 2 x 5
// coefficients of frame 1 around the displaced position), `matrix_compute` forms the 5 entries of the pixel's G matrix / h vector.
struct MatrixPix {
    float r0[5];          // frame 0 at (x, y)
    float top[10];        // frame 1 at (x1, y1), (x1 + 1, y1): 5 coefficients each
    float bot[10];        // frame 1 at (x1, y1 + 1), (x1 + 1, y1 + 1)
    float fx, fy;         // fractional part of the displaced position
    float dx, dy;         // the flow at the pixel
    bool inb;             // displaced position inside the image (else the frame-1 terms are dropped)
};
// Loads go through a buffer resource over the pair's coefficients ([2][h][w][4] + [2][h][w], 40 bytes per pixel): one 32-bit byte offset
// per pixel in a VGPR, the frame / plane offset in an SGPR - no 64-bit address arithmetic per load (17 VALU instructions per pixel with
// flat pointers); a pixel arrives as a 16- and a 4-byte load, a pixel pair of the displaced frame as 16 + 16 + 8 bytes.
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ float2 buf_f32x2(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_f32x4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ void matrix_request(MatrixPix& p, __amdgpu_buffer_rsrc_t rs, int hw_px, int x, int y, float dx, float dy, int h,
                                               int w) {
    // a pair's coefficients: [frame][h][w][4] (c0..c3, 16 bytes per pixel) followed by [frame][h][w] (c4) - see poly_expansion
    const int px0 = y * w + x;
    {
        const f32x4 a = buf_f32x4(rs, px0 * 16, 0);
        p.r0[0] = a.x; p.r0[1] = a.y; p.r0[2] = a.z; p.r0[3] = a.w;
        p.r0[4] = buf_f32(rs, px0 * 4, hw_px * 32);
    }
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= x1;
    fy -= y1;
    p.fx = fx; p.fy = fy; p.dx = dx; p.dy = dy;
    p.inb = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
    // outside the image the values are not used: the loads go to a clamped (valid) address instead of sitting in a divergent branch
    const int p1 = clampi(y1, 0, h - 2) * w + clampi(x1, 0, w - 2), p2 = p1 + w;
    {
        const int f4 = hw_px * 16, f1 = hw_px * 36;          // frame 1: its 4-vectors, its fifth coefficients
        const f32x4 a = buf_f32x4(rs, p1 * 16, f4), b = buf_f32x4(rs, p1 * 16 + 16, f4);
        const float2 c2 = buf_f32x2(rs, p1 * 4, f1);
        p.top[0] = a.x; p.top[1] = a.y; p.top[2] = a.z; p.top[3] = a.w; p.top[4] = c2.x;
        p.top[5] = b.x; p.top[6] = b.y; p.top[7] = b.z; p.top[8] = b.w; p.top[9] = c2.y;
        const f32x4 d = buf_f32x4(rs, p2 * 16, f4), e = buf_f32x4(rs, p2 * 16 + 16, f4);
        const float2 f2 = buf_f32x2(rs, p2 * 4, f1);
        p.bot[0] = d.x; p.bot[1] = d.y; p.bot[2] = d.z; p.bot[3] = d.w; p.bot[4] = f2.x;
        p.bot[5] = e.x; p.bot[6] = e.y; p.bot[7] = e.z; p.bot[8] = e.w; p.bot[9] = f2.y;
    }
}
// Every product / sum below names its rounding (__fmul_rn / __fadd_rn / __fmaf_rn): left to itself the compiler contracts these
// expressions into FMAs differently from one kernel to the next (packed FMAs in one, separate multiplies and adds in another), and
// the flow would depend on which kernel evaluated it.  The bilinear sums are FMA chains, smallest index first.
__device__ __forceinline__ float dot4_rn(float a0, float b0, float a1, float b1, float a2, float b2, float a3, float b3) {
    return __fmaf_rn(a3, b3, __fmaf_rn(a2, b2, __fmaf_rn(a1, b1, __fmul_rn(a0, b0))));
}
__device__ __forceinline__ void matrix_compute(const MatrixPix& p, int x, int y, int h, int w, float out[5]) {
    const float fx = p.fx, fy = p.fy, dx = p.dx, dy = p.dy;
    float r2, r3, r4, r5, r6;
    {   // the in-bounds case is always evaluated (the loads went to clamped addresses anyway) and the other one selected: no divergent
        // branch in the producers' instruction stream (-0.6 % of the stage against the branch; same values)
        const float gx = __fsub_rn(1.f, fx), gy = __fsub_rn(1.f, fy);
        const float a00 = __fmul_rn(gx, gy), a01 = __fmul_rn(fx, gy), a10 = __fmul_rn(gx, fy), a11 = __fmul_rn(fx, fy);
        r2 = dot4_rn(a00, p.top[0], a01, p.top[5], a10, p.bot[0], a11, p.bot[5]);
        r3 = dot4_rn(a00, p.top[1], a01, p.top[6], a10, p.bot[1], a11, p.bot[6]);
        r4 = dot4_rn(a00, p.top[2], a01, p.top[7], a10, p.bot[2], a11, p.bot[7]);
        r5 = dot4_rn(a00, p.top[3], a01, p.top[8], a10, p.bot[3], a11, p.bot[8]);
        r6 = dot4_rn(a00, p.top[4], a01, p.top[9], a10, p.bot[4], a11, p.bot[9]);
        r4 = __fmul_rn(__fadd_rn(p.r0[2], r4), 0.5f);
        r5 = __fmul_rn(__fadd_rn(p.r0[3], r5), 0.5f);
        r6 = __fmul_rn(__fadd_rn(p.r0[4], r6), 0.25f);
        r2 = p.inb ? r2 : 0.f;
        r3 = p.inb ? r3 : 0.f;
        r4 = p.inb ? r4 : p.r0[2];
        r5 = p.inb ? r5 : p.r0[3];
        r6 = p.inb ? r6 : __fmul_rn(p.r0[4], 0.5f);
    }
    r2 = __fmul_rn(__fsub_rn(p.r0[0], r2), 0.5f);
    r3 = __fmul_rn(__fsub_rn(p.r0[1], r3), 0.5f);
    r2 = __fadd_rn(r2, __fmaf_rn(r6, dx, __fmul_rn(r4, dy)));
    r3 = __fadd_rn(r3, __fmaf_rn(r5, dx, __fmul_rn(r6, dy)));
    const int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        // OpenCV's border table {0.14, 0.14, 0.4472, 0.4472, 0.4472} by distance to the edge, as selects (an indexed constant array is a
        // memory load, and its wait would drain the load pipeline of the fused iteration kernel)
#define RELAX_BORDER(d_) ((d_) < 2 ? 0.14f : 0.4472f)
        const float sc = __fmul_rn(__fmul_rn(x < BORDER ? RELAX_BORDER(x) : 1.f, x >= w - BORDER ? RELAX_BORDER(w - x - 1) : 1.f),
                                   __fmul_rn(y < BORDER ? RELAX_BORDER(y) : 1.f, y >= h - BORDER ? RELAX_BORDER(h - y - 1) : 1.f));
#undef RELAX_BORDER
        r2 = __fmul_rn(r2, sc); r3 = __fmul_rn(r3, sc); r4 = __fmul_rn(r4, sc); r5 = __fmul_rn(r5, sc); r6 = __fmul_rn(r6, sc);
    }
    out[0] = __fmaf_rn(r6, r6, __fmul_rn(r4, r4));
    out[1] = __fmul_rn(__fadd_rn(r4, r5), r6);
    out[2] = __fmaf_rn(r6, r6, __fmul_rn(r5, r5));
    out[3] = __fmaf_rn(r6, r3, __fmul_rn(r4, r2));
    out[4] = __fmaf_rn(r5, r3, __fmul_rn(r6, r2));
}
// The 2x2 solve of FarnebackUpdateFlow from the five box sums A (unscaled: the 1 / 225 of the box filter is folded into the
// regulariser, 1e-3 * 225^2 = 50.625, so no sum is multiplied first): flow = (A0 A4 - A1 A3, A2 A3 - A1 A4) / (A0 A2 - A1^2 + 50.625),
// in double with named FMAs (one form for every kernel that solves); the reciprocal is v_rcp_f64 refined by two Newton steps
// (the full division sequence costs twice as much and the quotient is rounded to float anyway).
__device__ __forceinline__ float2 solve_flow(double A0, double A1, double A2, double A3, double A4) {
    const double det = __fma_rn(A0, A2, __fma_rn(-A1, A1, 50.625));     // >= 50.625 up to rounding: the sums form a Gram matrix
    double r = __builtin_amdgcn_rcp(det);
    r = __fma_rn(__fma_rn(-det, r, 1.0), r, r);
    r = __fma_rn(__fma_rn(-det, r, 1.0), r, r);
    const double nx = __fma_rn(A0, A4, -__dmul_rn(A1, A3));
    const double ny = __fma_rn(A2, A3, -__dmul_rn(A1, A4));
    return make_float2((float)__dmul_rn(nx, r), (float)__dmul_rn(ny, r));
}
__device__ inline void matrix_entries(__amdgpu_buffer_rsrc_t rs, int hw_px, int x, int y, float dx, float dy, int h, int w, float out[5]) {
    MatrixPix p;
    matrix_request(p, rs, hw_px, x, y, dx, dy, h, w);
    matrix_compute(p, x, y, h, w, out);
}

// The pyramid step as the first matrix update of a level takes it: the flow at pixel (x, y) of this level = the coarser level's
// flow resized by cv::resize INTER_LINEAR and doubled - the operations of resize_linear_f32<2>, bit for bit, in two halves like
// the matrix entries (request the four coarse values / interpolate).
struct FlowUp {
    const float* src;     // coarser flow [P][H][W][2]
    int H, W;
    double scale_y, scale_x;
};
struct FlowUpPix {
    float2 a00, a01, a10, a11;
    float fx, fy;
};
__device__ __forceinline__ void flow_up_request(FlowUpPix& u, const FlowUp& up, const float* __restrict__ im, int x, int y) {
    float fx = resize_src_coord(x, up.scale_x);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= up.W - 1) { fx = 0.f; sx = up.W - 1; }
    float fy = resize_src_coord(y, up.scale_y);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    if (sy < 0) { fy = 0.f; sy = 0; }
    if (sy >= up.H - 1) { fy = 0.f; sy = up.H - 1; }
    const int sx1 = sx + 1 < up.W ? sx + 1 : up.W - 1;
    const int sy1 = sy + 1 < up.H ? sy + 1 : up.H - 1;
    u.a00 = *reinterpret_cast<const float2*>(im + ((int64_t)sy * up.W + sx) * 2);
    u.a01 = *reinterpret_cast<const float2*>(im + ((int64_t)sy * up.W + sx1) * 2);
    u.a10 = *reinterpret_cast<const float2*>(im + ((int64_t)sy1 * up.W + sx) * 2);
    u.a11 = *reinterpret_cast<const float2*>(im + ((int64_t)sy1 * up.W + sx1) * 2);
    u.fx = fx;
    u.fy = fy;
}
__device__ __forceinline__ float2 flow_up_value(const FlowUpPix& u) {
    return make_float2(__fmul_rn(lerp_rn(lerp_rn(u.a00.x, u.a01.x, u.fx), lerp_rn(u.a10.x, u.a11.x, u.fx), u.fy), 2.0f),
                       __fmul_rn(lerp_rn(lerp_rn(u.a00.y, u.a01.y, u.fx), lerp_rn(u.a10.y, u.a11.y, u.fx), u.fy), 2.0f));
}

// R [P][2][h][w][5], flow [P][h][w][2] -> M [P][5][h][w] (planar: the box filters stream it coalesced).
// One row of 256 columns per block.  The bilinear gather of row y reads rows y1, y1 + 1 of R1 and row y + 1 reads y1 + 1, y1 + 2:
// the shared row should come from L2, but workgroups are dealt round-robin to the 8 XCDs, so with a plain (x, y) grid the
// neighbouring rows ran on other XCDs and every R1 row was fetched twice from HBM (fetched / written bytes 3.45 against the
// algorithmic 2.4).  The block index is therefore remapped: XCD x (= linear block id mod 8) owns the band of rows
// [x * rpb, (x + 1) * rpb) and walks it row-major: fetched / written 2.4, 7.85 -> 6.0 GB per launch at 2160p; the time moved
// less (1115 -> 1084 us: 5.5 TB/s of useful bytes either way).  Eight rows per block in a loop instead: 1.7 % slower.
// UP: the flow of this level is the coarser level's flow resized by cv::resize INTER_LINEAR and doubled (the pyramid step).  The
// first matrix update of a level is its only reader (box_solve_fused writes the level's own flow), so the resized plane is never
// stored: the update computes the two values where it needs them - the operations of resize_linear_f32<2>, bit for bit (one kernel
// and 16 bytes per pixel fewer per level).
template <bool UP>
__global__ __launch_bounds__(256) void update_matrices_k(const float* __restrict__ R, const float* __restrict__ flow,
                                                         float* __restrict__ M, int h, int w, int rpb, const FlowUp up) {
    const int L = blockIdx.x + gridDim.x * blockIdx.y;    // grid (w / 256, 8 * rpb, pairs), rpb = rows per band = ceil(h / 8)
    const int band = L & 7, k = L >> 3;
    const int y = band * rpb + k / (int)gridDim.x;
    const int x = (k % (int)gridDim.x) * 256 + threadIdx.x;
    if (x >= w || y >= h || k / (int)gridDim.x >= rpb) return;
    const int64_t pair = blockIdx.z;
    const int64_t hw = (int64_t)w * h;
    const int64_t i = pair * hw + (int64_t)y * w + x;
    float e[5];
    float fx0, fy0;
    if constexpr (UP) {
        FlowUpPix u;
        flow_up_request(u, up, up.src + pair * ((int64_t)up.H * up.W * 2), x, y);
        const float2 f = flow_up_value(u);
        fx0 = f.x;
        fy0 = f.y;
    } else {
        fx0 = flow[i * 2];
        fy0 = flow[i * 2 + 1];
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R + (pair * 2) * hw * 5), 0, (int)(hw * 40), 0x00020000);
    matrix_entries(rs, (int)hw, x, y, fx0, fy0, h, w, e);
    float* o = M + pair * 5 * hw + ((int64_t)y * w + x);
#pragma unroll
    for (int c = 0; c < 5; ++c) o[c * hw] = e[c];
}

// The 15x15 box blur and the 2x2 solve of one Farneback iteration in ONE pass over M (no intermediate plane of row sums).
// A block owns a band of 240 output columns (254 threads = 240 + the 7-column halo on each side) and a segment of rows, and
// walks down the rows three at a time:
//   * vertical part, one COLUMN per thread: the last 15 rows of the 5 planes live in a register ring (static indices: the
//     row loop is unrolled over the ring period), so each M element is loaded once per segment and the running sums (double,
//     as OpenCV's FarnebackUpdateFlow_Blur and the oracle keep them) cost one add and one subtract per row; the three rows that
//     enter the window next are requested a group ahead;
//   * the column sums of the 3 rows go through LDS in double; then 180 threads each take a strip of 4 adjacent outputs of one
//     row: 9 16-byte LDS reads per plane, the 15-column window slid in double (sums of 15 doubles that are sums of 15 floats
//     are exact, so sliding does not change them), the 2x2 solve.  (One output per lane needed 75 LDS reads and adds per
//     pixel and was LDS-bound: 597 us for 4 pairs of 2160p; appending FarnebackUpdateMatrices to the row loop puts a dependent
//     gather between two barriers and was slower than a separate update_matrices_k; so was computing the matrix entries in
//     the vertical part instead of reading M: 1.59 vs 1.30 ms per 2160p pair.)
// HBM traffic per pixel: ~24 bytes of M + 8 of flow instead of 68 with a separate vertical pass that stores its row sums.
constexpr int FUSE_ROWS = 3;                          // divides the ring period
constexpr int FUSE_OUT = 240;                         // output columns per block
constexpr int FUSE_STRIPS = FUSE_OUT / 4;             // 60 strips per row, 180 strip threads per group
static_assert(WINSIZE % FUSE_ROWS == 0 && FUSE_OUT + WINSIZE - 1 <= 256 && FUSE_ROWS * FUSE_STRIPS <= 256, "fused box geometry");
// MINMAX (the last iteration of the finest level, when the flow image is asked for): the per-pair minimum and maximum of the flow
// magnitude, which flow_to_rgb's normalisation needs, are taken from the values as they are written (mag_minmax's expression,
// same bits; its pass over the flow plane disappears).
template <bool MINMAX>
__global__ __launch_bounds__(256) void box_solve_fused(const float* __restrict__ M, float* __restrict__ flow, int h, int w,
                                                       int seg, unsigned* __restrict__ mm) {
    __shared__ __attribute__((aligned(16))) double lds[FUSE_ROWS][5][256];
    constexpr int m = WINSIZE / 2;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * FUSE_OUT;                     // first output column of the band
    const int xc = clampi(c0 - m + tid, 0, w - 1);            // this thread's column (clamped = replicated border)
    const bool vert = tid < FUSE_OUT + 2 * m;
    const int y0 = blockIdx.y * seg;
    const int y1 = y0 + seg < h ? y0 + seg : h;
    const int64_t hw = (int64_t)h * w;
    const int64_t pair = blockIdx.z;
    const float* col = M + pair * 5 * hw + xc;
    // strip role
    const int sr = tid / FUSE_STRIPS, sq = tid - sr * FUSE_STRIPS;
    const int x0 = c0 + 4 * sq;
    const bool strip = tid < FUSE_ROWS * FUSE_STRIPS && x0 < w;

    [[maybe_unused]] unsigned mag_lo = 0xffffffffu, mag_hi = 0u;
    float ring[5][WINSIZE];   // the 15 rows of the window; slot of row r = (r - y0 + 7) % 15
    double s[5];
    float nxt[FUSE_ROWS][5];  // rows entering the window at the rows of the next group
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        s[c] = 0;
#pragma unroll
        for (int k = 0; k < WINSIZE; ++k) {                   // rows y0-7 .. y0+7 in order
            ring[c][k] = vert ? col[c * hw + (int64_t)clampi(y0 - m + k, 0, h - 1) * w] : 0.f;
            s[c] += ring[c][k];
        }
#pragma unroll
        for (int r = 0; r < FUSE_ROWS; ++r) nxt[r][c] = vert ? col[c * hw + (int64_t)clampi(y0 + r + m, 0, h - 1) * w] : 0.f;
    }

    for (int yb = y0; yb < y1; yb += WINSIZE) {
#pragma unroll
        for (int g = 0; g < WINSIZE / FUSE_ROWS; ++g) {
            const int yg = yb + g * FUSE_ROWS;
            if (yg < y1) {                                    // uniform over the block
                float cur[FUSE_ROWS][5];
#pragma unroll
                for (int r = 0; r < FUSE_ROWS; ++r)
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        cur[r][c] = nxt[r][c];
                        nxt[r][c] = vert ? col[c * hw + (int64_t)clampi(yg + FUSE_ROWS + r + m, 0, h - 1) * w] : 0.f;
                    }
#pragma unroll
                for (int r = 0; r < FUSE_ROWS; ++r) {
                    const int k = g * FUSE_ROWS + r;          // row y = yb + k; the leaving and the entering row share slot (k - 1) mod 15
                    const int y = yb + k;
                    if (y > y0 && y < y1) {
                        const int slot = (k + WINSIZE - 1) % WINSIZE;
#pragma unroll
                        for (int c = 0; c < 5; ++c) {
                            s[c] += (double)cur[r][c] - (double)ring[c][slot];
                            ring[c][slot] = cur[r][c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 5; ++c) lds[r][c][tid] = s[c];
                }
                __syncthreads();
                const int y = yg + sr;
                if (strip && y < y1) {
                    double acc[5][4];
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        double v[18];   // column sums at band columns 4*sq .. 4*sq + 17 (output o uses o .. o + 14)
#pragma unroll
                        for (int j = 0; j < 9; ++j) {
                            const double2 t = *reinterpret_cast<const double2*>(&lds[sr][c][4 * sq + 2 * j]);
                            v[2 * j] = t.x;
                            v[2 * j + 1] = t.y;
                        }
                        double a = 0;
#pragma unroll
                        for (int j = 0; j < WINSIZE; ++j) a += v[j];
                        acc[c][0] = a;
#pragma unroll
                        for (int e = 1; e < 4; ++e) {
                            a += v[WINSIZE - 1 + e] - v[e - 1];
                            acc[c][e] = a;
                        }
                    }
                    float* dst = flow + (pair * hw + (int64_t)y * w + x0) * 2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (x0 + e < w) {
                            const float2 f = solve_flow(acc[0][e], acc[1][e], acc[2][e], acc[3][e], acc[4][e]);
                            const float fxo = f.x, fyo = f.y;
                            dst[2 * e] = fxo;
                            dst[2 * e + 1] = fyo;
                            if constexpr (MINMAX) {
                                const unsigned u = __float_as_uint(flow_magnitude(fxo, fyo));
                                mag_lo = u < mag_lo ? u : mag_lo;
                                mag_hi = u > mag_hi ? u : mag_hi;
                            }
                        }
                    }
                }
                __syncthreads();   // the next group overwrites the three rows
            }
        }
    }
    if constexpr (MINMAX) {   // wave minimum / maximum, one atomic pair per wave
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned a = __shfl_xor(mag_lo, o), b = __shfl_xor(mag_hi, o);
            mag_lo = a < mag_lo ? a : mag_lo;
            mag_hi = b > mag_hi ? b : mag_hi;
        }
        if ((tid & 63) == 0) {
            atomicMin(&mm[pair], mag_lo);                 // mins [0,P), maxs [P,2P)
            atomicMax(&mm[gridDim.z + pair], mag_hi);
        }
    }
}

// ---- one Farneback iteration in ONE kernel ----------------------------------------------------------------------------
// FarnebackUpdateMatrices + the 15x15 box blur + the 2x2 solve, without the matrix plane M ever going to HBM: per pixel and
// iteration the memory system sees R0 (20 B), R1 at the displaced position (20 B), the flow in (8 B) and out (8 B) = 56 bytes
// instead of 68 (update_matrices_k) + 20 + 8 (box_solve_fused) = 96.  update_matrices_k ran at the HBM rate and box_solve_fused
// waited on its loads (3.2 TB/s with 2 waves per SIMD and 15 loads per thread and group in flight); here the two halves run side
// by side on one CU as a producer / consumer pair of wave groups:
//   * waves 4-7, the PRODUCERS: thread = a column of the band (256 = 240 outputs + 7 halo columns each side, replicated
//     border).  Per step they form the matrix entries of the 3 rows that enter the box window next and put them into LDS.  The
//     loads are software-pipelined across steps: the flow of step t + 2 (or the four coarse-flow values the pyramid step
//     interpolates, UP) is requested while the R0 / R1 loads of step t + 1 - whose addresses need the flow of t + 1 - go out row by
//     row as the entries of step t are computed from the registers the previous step filled.  3 rows x 27 dwords per thread stay
//     in flight (83 KB per CU) across the barriers: raw s_barrier after `s_waitcnt lgkmcnt(0)`, never __syncthreads (its vmcnt(0)
//     would drain the pipeline).
//   * waves 0-3, the BOX waves: box_solve_fused's body.  Thread = the same column: the 15 window rows of the 5 planes in a
//     register ring, running column sums in double (one add and one subtract per row), taken from LDS instead of HBM; the
//     column sums of the 3 rows go to LDS in double, then 180 threads slide the 15-column window over 4-output strips and solve.
//   One barrier per step: the box waves consume the rows the producers wrote during the PREVIOUS step (M double-buffered) and run
//   the strips of the step before that (column sums double-buffered), so neither side ever waits for the other inside a step.
// The ring starts as zeros and the first 5 steps (15 rows) fill it - `s += row - 0` - so the loop has no special first window:
// entering rows y0 - 8 + 3t + r (the very first, y0 - 8, lies outside the window of y0 and enters as zeros).  Same additions in the
// same order as box_solve_fused, same matrix entries as update_matrices_k (shared source above): the flow is bit-identical.
// Phase-stamp hooks of the diagnostic build: empty in the product.  tools/abl/flow_stamps.hip defines them (and the reader
// relax_debug_flow_stamps) and #includes this file; nothing in this translation unit can record or print a stamp.
#ifndef IT_STAMP
#define IT_STAMP_DECL
#define IT_STAMP(i_)
#define IT_STAMP_WAIT_LOADS
#define IT_STAMP_FLUSH_PRODUCER
#define IT_STAMP_FLUSH_BOX
#endif
constexpr int IT_ROWS = 3;
constexpr int IT_OUT = FUSE_OUT;                 // 240 output columns per band
constexpr int IT_STRIPS = IT_OUT / 4;
constexpr int IT_FILL = WINSIZE / IT_ROWS;       // 5 steps fill the window
constexpr size_t IT_M_BYTES = sizeof(float) * 2 * IT_ROWS * 5 * 256;
constexpr size_t IT_LDS = IT_M_BYTES + sizeof(double) * 2 * IT_ROWS * 5 * 256;   // 30 KB + 60 KB
static_assert(WINSIZE % IT_ROWS == 0 && IT_OUT + WINSIZE - 1 <= 256 && IT_ROWS * IT_STRIPS <= 256, "fused iteration geometry");

template <bool UP, bool MINMAX>
__global__ __launch_bounds__(512) void flow_iteration(const float* __restrict__ R, const float* __restrict__ flow_in,
                                                      float* __restrict__ flow_out, int h, int w, int seg, const FlowUp up,
                                                      unsigned* __restrict__ mm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float MTile[IT_ROWS][5][256];
    typedef double STile[IT_ROWS][5][256];
    MTile* Mb = reinterpret_cast<MTile*>(smem);
    STile* Sb = reinterpret_cast<STile*>(smem + IT_M_BYTES);
    constexpr int m = WINSIZE / 2;
    const int tid = threadIdx.x & 255;
    const bool producer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;   // wave-uniform: a scalar branch
    const int c0 = blockIdx.x * IT_OUT;                       // first output column of the band
    const int xc = clampi(c0 - m + tid, 0, w - 1);            // this thread's column (clamped = replicated border)
    const int y0 = blockIdx.y * seg;
    const int y1 = y0 + seg < h ? y0 + seg : h;
    const int64_t hw = (int64_t)h * w;
    const int64_t pair = blockIdx.z;
    const int Q = IT_FILL + (y1 - y0 + IT_ROWS - 1) / IT_ROWS;   // steps of entering rows; steps t >= 5 complete output rows

    if (producer) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R + (pair * 2) * hw * 5), 0, (int)(hw * 40), 0x00020000);
        const int hw_px = (int)hw;
        const float* fin = UP ? up.src + pair * ((int64_t)up.H * up.W * 2) : flow_in + pair * hw * 2;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fin), 0, UP ? 0 : (int)(hw * 8), 0x00020000);
        // One step of operands in flight: while the entries of step t are computed row by row from the registers the previous step
        // filled, the operands of step t + 1 go out into the registers just consumed, and the flow of step t + 2 (the operand addresses of
        // a step need its flow a step earlier; two flow sets, even / odd steps).  The loop is unrolled over the two parities so that
        // every loaded value is consumed from the register it landed in (a copy at the loop head waits for the youngest load:
        // vmcnt(0)), and the body is straight-line code without a divergent branch (threads 254 / 255 compute two columns nobody
        // reads): the compiler's wait-count bookkeeping turns pessimistic at control-flow joins.  (Two steps of operands in flight
        // need 2 x 90 registers: over the 256 the two waves of a SIMD have each, it spilled.)
#define IT_ROW(t_, r_) clampi(y0 - (m + 1) + IT_ROWS * (t_) + (r_), 0, h - 1)
        [[maybe_unused]] FlowUpPix fu0[IT_ROWS], fu1[IT_ROWS];      // UP: the four coarse values per pixel
        [[maybe_unused]] float f0x[IT_ROWS], f0y[IT_ROWS], f1x[IT_ROWS], f1y[IT_ROWS];
        MatrixPix px[IT_ROWS];
#define IT_REQUEST_FLOW(t_, r_, FU_, FX_, FY_)                                                               \
    do {                                                                                                     \
        const int y_ = IT_ROW(t_, r_);                                                                       \
        if constexpr (UP) flow_up_request(FU_[r_], up, fin, xc, y_);                                         \
        else {                                                                                               \
            const float2 q_ = buf_f32x2(rsf, (y_ * w + xc) * 8, 0);                                          \
            FX_[r_] = q_.x;                                                                                  \
            FY_[r_] = q_.y;                                                                                  \
        }                                                                                                    \
    } while (0)
#define IT_REQUEST_OPERANDS(t_, r_, FU_, FX_, FY_)                                                           \
    do {                                                                                                     \
        float dx_, dy_;                                                                                      \
        if constexpr (UP) {                                                                                  \
            const float2 f_ = flow_up_value(FU_[r_]);                                                        \
            dx_ = f_.x; dy_ = f_.y;                                                                          \
        } else {                                                                                             \
            dx_ = FX_[r_]; dy_ = FY_[r_];                                                                    \
        }                                                                                                    \
        matrix_request(px[r_], rs, hw_px, xc, IT_ROW(t_, r_), dx_, dy_, h, w);                               \
    } while (0)
        // one step: FUC_ / FXC_ / FYC_ = the flow set of step t_ + 1 (consumed), FUN_ / .. = that of step t_ + 2 (requested)
#define IT_STEP(t_, FUC_, FXC_, FYC_, FUN_, FXN_, FYN_, OPS_, FLOW_)                                         \
    do {                                                                                                     \
        IT_STAMP_WAIT_LOADS;                                                                                 \
        if constexpr (UP) {   /* the coarse flow is small and mostly in L2: ONE set, requested at the top of the step that consumes it \
                                 (with two sets in flight this instantiation needed 267 registers and spilled) */                     \
            if (OPS_) { _Pragma("unroll") for (int r = 0; r < IT_ROWS; ++r) flow_up_request(fu0[r], up, fin, xc, IT_ROW((t_) + 1, r)); } \
        }                                                                                                    \
        _Pragma("unroll") for (int r = 0; r < IT_ROWS; ++r) {                                                \
            float e[5];                                                                                      \
            matrix_compute(px[r], xc, IT_ROW(t_, r), h, w, e);                                               \
            const bool skip = (t_) == 0 && r == 0;   /* row y0 - 8 is outside the window of y0 */            \
            _Pragma("unroll") for (int c = 0; c < 5; ++c) Mb[(t_) & 1][r][c][tid] = skip ? 0.f : e[c];      \
            if constexpr (UP) {                                                                              \
                if (OPS_) IT_REQUEST_OPERANDS((t_) + 1, r, fu0, FXC_, FYC_);                                 \
            } else {                                                                                         \
                if (OPS_) IT_REQUEST_OPERANDS((t_) + 1, r, FUC_, FXC_, FYC_);                                \
                if (FLOW_) IT_REQUEST_FLOW((t_) + 2, r, FUN_, FXN_, FYN_);                                   \
            }                                                                                                \
        }                                                                                                    \
        IT_STAMP(0);   /* entries + requests (includes the waits for the operands) */                         \
        RELAX_LDS_BARRIER();                                                                                 \
        IT_STAMP(1);   /* barrier */                                                                          \
    } while (0)
#define IT_STEP_EVEN(t_, OPS_, FLOW_) IT_STEP(t_, fu1, f1x, f1y, fu0, f0x, f0y, OPS_, FLOW_)
#define IT_STEP_ODD(t_, OPS_, FLOW_) IT_STEP(t_, fu0, f0x, f0y, fu1, f1x, f1y, OPS_, FLOW_)
#pragma unroll
        for (int r = 0; r < IT_ROWS; ++r) IT_REQUEST_FLOW(0, r, fu0, f0x, f0y);
        if constexpr (!UP) {
#pragma unroll
            for (int r = 0; r < IT_ROWS; ++r) IT_REQUEST_FLOW(1, r, fu1, f1x, f1y);
        }
#pragma unroll
        for (int r = 0; r < IT_ROWS; ++r) IT_REQUEST_OPERANDS(0, r, fu0, f0x, f0y);
        IT_STAMP_DECL;
        int t = 0;
        for (; t + 3 < Q; t += 2) {                            // steps t and t + 1 both request operands (t + 2 < Q) and flow (t + 3 < Q)
            IT_STEP_EVEN(t, true, true);
            IT_STEP_ODD(t + 1, true, true);
        }
        for (; t < Q; ++t) {                                   // the last one to three steps: requests by (uniform) condition
            const bool ops = t + 1 < Q, flo = t + 2 < Q;
            if (t & 1) IT_STEP_ODD(t, ops, flo);
            else IT_STEP_EVEN(t, ops, flo);
        }
        RELAX_LDS_BARRIER();                                   // the barrier of step Q (the box waves' last column sums)
        IT_STAMP_FLUSH_PRODUCER;
#undef IT_STEP_EVEN
#undef IT_STEP_ODD
#undef IT_STEP
#undef IT_REQUEST_OPERANDS
#undef IT_REQUEST_FLOW
#undef IT_ROW
        return;
    }

    // ---- box waves ----
    const int sr = tid / IT_STRIPS, sq = tid - sr * IT_STRIPS;
    const int x0 = c0 + 4 * sq;
    const bool strip = tid < IT_ROWS * IT_STRIPS && x0 < w;
    const bool vec = (w & 3) == 0;                            // 32-byte aligned strips: two 16-byte stores
    [[maybe_unused]] unsigned mag_lo = 0xffffffffu, mag_hi = 0u;
    float ring[5][WINSIZE];   // slot of the row that enters at step t as row r: (3 t + r) % 15
    double s[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        s[c] = 0;
#pragma unroll
        for (int k = 0; k < WINSIZE; ++k) ring[c][k] = 0.f;
    }
    IT_STAMP_DECL;
    for (int tb = 0; tb <= Q + 1; tb += IT_FILL) {
#pragma unroll
        for (int j = 0; j < IT_FILL; ++j) {
            const int t = tb + j;                             // this step: V(t - 1), then H(t - 2)
            if (t <= Q + 1) {                                 // uniform over the block
                IT_STAMP(5);
                if (t >= 1 && t <= Q) {
                    const int par = (t - 1) & 1;
#pragma unroll
                    for (int r = 0; r < IT_ROWS; ++r) {
                        const int slot = (IT_ROWS * (j + IT_FILL - 1) + r) % WINSIZE;   // (3 (t - 1) + r) % 15: tb is a multiple of 5
#pragma unroll
                        for (int c = 0; c < 5; ++c) {
                            const float cur = Mb[par][r][c][tid];
                            s[c] += (double)cur - (double)ring[c][slot];
                            ring[c][slot] = cur;
                            Sb[par][r][c][tid] = s[c];
                        }
                    }
                }
                IT_STAMP(2);   // column sums
                if (t >= IT_FILL + 2) {                       // (t <= Q + 1 holds)
                    const int q = t - 2, par = q & 1;
                    const int y = y0 + IT_ROWS * (q - IT_FILL) + sr;
                    if (strip && y < y1) {
                        double acc[5][4];
#pragma unroll
                        for (int c = 0; c < 5; ++c) {
                            double v[18];   // column sums at band columns 4*sq .. 4*sq + 17 (output o uses o .. o + 14)
#pragma unroll
                            for (int jj = 0; jj < 9; ++jj) {
                                const double2 tt = *reinterpret_cast<const double2*>(&Sb[par][sr][c][4 * sq + 2 * jj]);
                                v[2 * jj] = tt.x;
                                v[2 * jj + 1] = tt.y;
                            }
                            double a = 0;
#pragma unroll
                            for (int jj = 0; jj < WINSIZE; ++jj) a += v[jj];
                            acc[c][0] = a;
#pragma unroll
                            for (int e = 1; e < 4; ++e) {
                                a += v[WINSIZE - 1 + e] - v[e - 1];
                                acc[c][e] = a;
                            }
                        }
                        float* dst = flow_out + (pair * hw + (int64_t)y * w + x0) * 2;
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float2 f = solve_flow(acc[0][e], acc[1][e], acc[2][e], acc[3][e], acc[4][e]);
                            o[2 * e] = f.x;
                            o[2 * e + 1] = f.y;
                        }
                        if (vec) {
                            *reinterpret_cast<f32x4*>(dst) = (f32x4){o[0], o[1], o[2], o[3]};
                            *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){o[4], o[5], o[6], o[7]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (x0 + e < w) {
                                    dst[2 * e] = o[2 * e];
                                    dst[2 * e + 1] = o[2 * e + 1];
                                }
                        }
                        if constexpr (MINMAX) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (x0 + e < w) {
                                    const unsigned u = __float_as_uint(flow_magnitude(o[2 * e], o[2 * e + 1]));
                                    mag_lo = u < mag_lo ? u : mag_lo;
                                    mag_hi = u > mag_hi ? u : mag_hi;
                                }
                        }
                    }
                }
                IT_STAMP(3);   // strips: window slide + solve + stores
                if (t <= Q) RELAX_LDS_BARRIER();
                IT_STAMP(4);   // barrier
            }
        }
    }
    IT_STAMP_FLUSH_BOX;
    if constexpr (MINMAX) {   // wave minimum / maximum, one atomic pair per wave
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned a = __shfl_xor(mag_lo, o), b = __shfl_xor(mag_hi, o);
            mag_lo = a < mag_lo ? a : mag_lo;
            mag_hi = b > mag_hi ? b : mag_hi;
        }
        if ((tid & 63) == 0) {
            atomicMin(&mm[pair], mag_lo);                 // mins [0,P), maxs [P,2P)
            atomicMax(&mm[gridDim.z + pair], mag_hi);
        }
    }
}

// ---- flow_to_rgb ------------------------------------------------------------------------------------------------
__device__ inline float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// per-pair min / max of the flow magnitude (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void mag_minmax(const float* __restrict__ flow, int HW, unsigned* __restrict__ mm) {
    __shared__ unsigned smin[256], smax[256];
    const int pair = blockIdx.y;
    const float* f = flow + (int64_t)pair * HW * 2;
    const bool al8 = (reinterpret_cast<uintptr_t>(f) & 7) == 0;   // one 8-byte load per pixel when the pair's plane allows it
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float x, y;
        if (al8) {
            const float2 v = reinterpret_cast<const float2*>(f)[i];
            x = v.x; y = v.y;
        } else {
            x = f[i * 2]; y = f[i * 2 + 1];
        }
        const unsigned u = __float_as_uint(flow_magnitude(x, y));
        lo = u < lo ? u : lo;
        hi = u > hi ? u : hi;
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = smin[threadIdx.x + s] < smin[threadIdx.x] ? smin[threadIdx.x + s] : smin[threadIdx.x];
            smax[threadIdx.x] = smax[threadIdx.x + s] > smax[threadIdx.x] ? smax[threadIdx.x + s] : smax[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMin(&mm[pair], smin[0]);                 // mins [0,P), maxs [P,2P)
        atomicMax(&mm[gridDim.y + pair], smax[0]);
    }
}

__device__ inline void minmax_affine(double smin, double smax, float* scale, float* shift) {
    const double d = smax - smin;
    const double sc = 255.0 * (d > DBL_EPSILON ? 1.0 / d : 0.0);
    *scale = (float)sc;
    *shift = (float)(0.0 - smin * sc);
}

// the two NORM_MINMAX maps of flow_to_rgb for a pair whose magnitudes span [mn, mx]: mag = normalize(mag); V = trunc(normalize(mag)) -
// the 2nd min / max are the images of the 1st (monotone affine map).  Four double divisions: once per thread, not per pixel.
struct VisNorm {
    float s1, b1, s2, b2;
};
__device__ inline VisNorm visualise_norm(float mn, float mx) {
    VisNorm n;
    minmax_affine(mn, mx, &n.s1, &n.b1);
    minmax_affine((double)(mn * n.s1 + n.b1), (double)(mx * n.s1 + n.b1), &n.s2, &n.b2);
    return n;
}
// one pixel of flow_to_rgb: (x, y) -> packed b | g << 8 | r << 16
__device__ inline unsigned visualise_pixel(float x, float y, const VisNorm& n) {
    float mag = flow_magnitude(x, y);
    const float ang = fast_atan2_deg(y, x) * (float)(M_PI / 180);
    mag = mag * n.s1 + n.b1;
    const float vf = mag * n.s2 + n.b2;
    const float hue = ang * 180.f / (float)M_PI / 2.f;
    const int Hh = (int)(uint8_t)(int)hue;              // numpy float32 -> uint8 assignment truncates
    const int Vv = (int)(uint8_t)(int)vf;
    // 8-bit HSV -> BGR, S = 255: OpenCV's float formula, converted by truncation (oracle/flow_ref.py)
    const float hh = (float)Hh * (6.0f / 180.0f);
    int sector = (int)floorf(hh);
    const float f = hh - (float)sector;
    sector %= 6;
    const float s = 255.f * (1.0f / 255.0f), v = (float)Vv * (1.0f / 255.0f);
    const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * f), t3 = v * (1.f - s * (1.f - f));
    // OpenCV's sector table {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0} (B, G, R) as selects: no indexed local arrays
    const float fb = sector <= 1 ? t1 : (sector == 2 ? t3 : (sector <= 4 ? t0 : t2));
    const float fg = sector == 0 ? t3 : (sector <= 2 ? t0 : (sector == 3 ? t2 : t1));
    const float fr = sector == 0 ? t0 : (sector == 1 ? t2 : (sector <= 3 ? t1 : (sector == 4 ? t3 : t0)));
    float q0 = floorf(fb * 255.f), q1 = floorf(fg * 255.f), q2 = floorf(fr * 255.f);
    q0 = q0 < 0.f ? 0.f : (q0 > 255.f ? 255.f : q0);
    q1 = q1 < 0.f ? 0.f : (q1 > 255.f ? 255.f : q1);
    q2 = q2 < 0.f ? 0.f : (q2 > 255.f ? 255.f : q2);
    return (unsigned)q0 | ((unsigned)q1 << 8) | ((unsigned)q2 << 16);
}

__global__ __launch_bounds__(256) void flow_visualise(const float* __restrict__ flow, const unsigned* __restrict__ mm, int P, int HW,
                                                      uint8_t* __restrict__ bgr) {
    const int pix = blockIdx.x * 256 + threadIdx.x;      // grid (HW / 256, 1, pairs)
    if (pix >= HW) return;
    const int pair = blockIdx.z;
    const int64_t i = (int64_t)pair * HW + pix;
    const unsigned c = visualise_pixel(flow[i * 2], flow[i * 2 + 1], visualise_norm(__uint_as_float(mm[pair]), __uint_as_float(mm[P + pair])));
    uint8_t* o = bgr + i * 3;
    o[0] = (uint8_t)c;
    o[1] = (uint8_t)(c >> 8);
    o[2] = (uint8_t)(c >> 16);
}

// 4 pixels per thread: two 16-byte loads, three dword stores (pixel counts and pointers that allow it)
__global__ __launch_bounds__(256) void flow_visualise_v4(const float* __restrict__ flow, const unsigned* __restrict__ mm, int P,
                                                         int HW, uint8_t* __restrict__ bgr) {
    const int pix = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (pix >= HW) return;
    const int pair = blockIdx.z;
    const int64_t i = (int64_t)pair * HW + pix;
    const f32x4 a = *reinterpret_cast<const f32x4*>(flow + i * 2), b = *reinterpret_cast<const f32x4*>(flow + i * 2 + 4);
    const VisNorm n = visualise_norm(__uint_as_float(mm[pair]), __uint_as_float(mm[P + pair]));
    const unsigned c0 = visualise_pixel(a.x, a.y, n), c1 = visualise_pixel(a.z, a.w, n);
    const unsigned c2 = visualise_pixel(b.x, b.y, n), c3 = visualise_pixel(b.z, b.w, n);
    uint32_t* o = reinterpret_cast<uint32_t*>(bgr + i * 3);
    o[0] = c0 | (c1 << 24);
    o[1] = (c1 >> 8) | (c2 << 16);
    o[2] = (c2 >> 16) | (c3 << 8);
}

// ---- host ---------------------------------------------------------------------------------------------------------
static void prepare_poly(PolyConsts* pc) {
    const int n = POLY_N;
    const double sigma = 1.2;
    float g[2 * POLY_N + 1];
    double s = 0.;
    for (int x = -n; x <= n; ++x) {
        g[x + n] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x + n];
    }
    s = 1. / s;
    float xg[2 * POLY_N + 1], xxg[2 * POLY_N + 1];
    for (int x = -n; x <= n; ++x) {
        g[x + n] = (float)(g[x + n] * s);
        xg[x + n] = (float)(x * g[x + n]);
        xxg[x + n] = (float)(x * x * g[x + n]);
    }
    double G00 = 0, G11 = 0, G33 = 0, G55 = 0;
    for (int y = -n; y <= n; ++y)
        for (int x = -n; x <= n; ++x) {
            const double w = (double)g[y + n] * g[x + n];
            G00 += w;
            G11 += w * x * x;
            G33 += w * x * x * x * x;
            G55 += w * x * x * y * y;
        }
    // invert the 6x6 moment matrix: rows/cols {0,3,4} couple ([G00 G11 G11; G11 G33 G55; G11 G55 G33]), 1, 2, 5 are diagonal
    const double a = G00, b = G11, c = G33, d = G55;
    const double det = a * (c * c - d * d) - 2.0 * b * b * (c - d);
    pc->ig11 = 1.0 / G11;
    pc->ig03 = (b * d - b * c) / det;          // cofactor (0,3) / det (symmetric)
    pc->ig33 = (a * c - b * b) / det;
    pc->ig55 = 1.0 / G55;
    for (int k = 0; k <= n; ++k) {
        pc->g[k] = g[k + n];
        pc->xg[k] = xg[k + n];
        pc->xxg[k] = xxg[k + n];
    }
}

static void make_gauss(int ksize, double sigma, GaussKernel* gk) {
    gk->ksize = ksize;
    if (ksize == 3 && sigma <= 0) {
        gk->k[0] = 0.25f; gk->k[1] = 0.5f; gk->k[2] = 0.25f;
        return;
    }
    const double s = sigma > 0 ? sigma : ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    double sum = 0;
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        gk->k[i] = (float)std::exp(-0.5 / (s * s) * x * x);
        sum += gk->k[i];
    }
    const float inv = (float)(1.0 / sum);
    for (int i = 0; i < ksize; ++i) gk->k[i] *= inv;
}

static inline unsigned nblocks(int64_t total) { return (unsigned)((total + 255) / 256); }
// the 4-pixel kernels use 16-byte loads / stores: row length a multiple of 4 and 16-byte aligned planes
static inline bool vec4_ok(int w, int64_t plane_elems, const void* a, const void* b) {
    return w % 4 == 0 && plane_elems % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

static int minmax_reset(relax_handle* h, int P, unsigned* mm, hipStream_t s) {
    RELAX_HIP_CHECK(h, hipMemsetAsync(mm, 0xff, sizeof(unsigned) * P, s));       // running minima
    RELAX_HIP_CHECK(h, hipMemsetAsync(mm + P, 0, sizeof(unsigned) * P, s));       // running maxima
    return RELAX_OK;
}

// have_minmax: mm already holds the magnitude range of every pair (box_solve_fused<true> wrote the flow)
static int visualise(relax_handle* h, const float* flow, int P, int HW, uint8_t* bgr, unsigned* mm, hipStream_t s, bool have_minmax = false) {
    if (!have_minmax) {
        RELAX_TRY(minmax_reset(h, P, mm, s));
        hipLaunchKernelGGL(mag_minmax, dim3(HW / 4096 > 64 ? (HW / 4096 < 1024 ? HW / 4096 : 1024) : 64, P), dim3(256), 0, s, flow, HW, mm);
    }
    if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(flow) & 15) == 0 && (reinterpret_cast<uintptr_t>(bgr) & 3) == 0)
        hipLaunchKernelGGL(flow_visualise_v4, dim3(nblocks(HW / 4), 1, P), dim3(256), 0, s, flow, mm, P, HW, bgr);
    else
        hipLaunchKernelGGL(flow_visualise, dim3(nblocks(HW), 1, P), dim3(256), 0, s, flow, mm, P, HW, bgr);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// Rows per block of the box-filter kernels.  Long segments keep the 15-row warm-up (and its re-read operands) small; shorter ones when
// the level is small, so that a clip's 32 pairs still give every CU work.  The choice depends on the level's geometry only - never on
// the number of pairs in the launch: the running column sums restart at a segment's first row, so the segmentation is part of what
// fixes the last bits of the flow, and a pair's flow must not depend on its batch.
static int flow_segment_rows(const relax_handle* h, int bands, int hh) {
    if (h->gemm.flow_seg_rows > 0) return (h->gemm.flow_seg_rows + 14) / 15 * 15;   // tests: force a segmentation (multiples of the ring period)
    for (const int cand : {270, 135, 90, 60, 45})
        if (bands * ((hh + cand - 1) / cand) >= 16) return cand;
    return 30;
}

template <bool UP, bool MINMAX>
static int launch_flow_iteration(relax_handle* h, const float* R, const float* flow_in, float* flow_out, int hh, int w, int P, const FlowUp& up,
                                 unsigned* mm, hipStream_t s) {
    // per (instantiation, device); atomic: two handles of one device may launch from two threads (the call is idempotent, so the worst
    // a lost race does is set the attribute twice)
    static std::atomic<bool> attr_set[kMaxDevices];
    if (!attr_set[h->device].load(std::memory_order_acquire)) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&flow_iteration<UP, MINMAX>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)IT_LDS));
        attr_set[h->device].store(true, std::memory_order_release);
    }
    const int bands = (w + IT_OUT - 1) / IT_OUT;
    const int seg = flow_segment_rows(h, bands, hh);
    hipLaunchKernelGGL((flow_iteration<UP, MINMAX>), dim3(bands, (hh + seg - 1) / seg, P), dim3(512), IT_LDS, s, R, flow_in, flow_out, hh, w,
                       seg, up, mm);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

static int flow_chunk(relax_handle* h, const uint8_t* orig, const uint8_t* next, int64_t pair_stride, int P, int H, int W,
                      float* flow_out, uint8_t* bgr_out, hipStream_t s) {
    const int64_t HW = (int64_t)H * W;
    // the matrix kernels address a pair's two coefficient images through ONE buffer resource with 32-bit byte offsets
    RELAX_REQUIRE(h, HW * 40 < (int64_t)0x7fffffff, "optical flow: frames of %d x %d exceed the 2 GB a buffer resource addresses (40 bytes per pixel)", W, H);
    // workspace carve (floats unless noted), all sized for level 0
    const size_t per_pair = (size_t)HW * (2 + 2 + 2 + 2 + 10 + 5 + 2 + 2) * sizeof(float) + 16;
    RELAX_TRY(ensure_buf(h, h->flow_ws, per_pair * P));
    float* gray = static_cast<float*>(h->flow_ws.p);
    float* tmp = gray + (size_t)P * 2 * HW;
    float* blur = tmp + (size_t)P * 2 * HW;
    float* I = blur + (size_t)P * 2 * HW;
    float* R = I + (size_t)P * 2 * HW;         // [P][2][h][w][5]
    float* M = R + (size_t)P * 10 * HW;        // [P][h][w][5]
    float* flowA = M + (size_t)P * 5 * HW;     // [P][h][w][2]
    float* flowB = flowA + (size_t)P * 2 * HW;
    unsigned* mm = reinterpret_cast<unsigned*>(flowB + (size_t)P * 2 * HW);

    PolyConsts pc;
    prepare_poly(&pc);
    // measurement (relax_profile_read kind 6): the whole stage, first launch to last; its algorithmic bytes are added up launch by launch
    double stage_bytes = 0;
    int stage_span;
    RELAX_TRY(prof_begin(h, s, 4, 0.0, &stage_span));
    int levels = 0;
    {
        double sc = 1.0;
        for (; levels < 3; ++levels) {
            sc *= 0.5;
            if (W * sc < 32 || H * sc < 32) break;
        }
    }
    // the four level inputs in one pass over the frames (pyramid_fused) when the scales are exact: blur -> level 0, I -> level 1,
    // tmp -> levels 2 and 3; otherwise the gray plane and the per-level blur / resize kernels below
    const bool pyr = h->gemm.flow_pyramid_fused && levels == 3 && H % 8 == 0 && W % 8 == 0 && pair_stride % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(orig) | reinterpret_cast<uintptr_t>(next)) & 3) == 0;
    float* lvl_in[4] = {blur, I, tmp, tmp + (size_t)P * 2 * (HW / 16)};
    if (pyr) {
        PyramidTaps tp;
        for (int k = 0; k <= 3; ++k) {
            const double scale = 1.0 / (1 << k), sigma = (1. / scale - 1) * 0.5;
            int smooth = (int)lrint(sigma * 5) | 1;
            if (smooth < 3) smooth = 3;
            GaussKernel gk;
            make_gauss(smooth, sigma, &gk);
            RELAX_REQUIRE(h, gk.ksize == (k < 2 ? 3 : (k == 2 ? 9 : 19)), "optical flow: unexpected smoothing kernel %d at level %d", gk.ksize, k);
            float* dst = k == 0 ? tp.k0 : (k == 1 ? tp.k1 : (k == 2 ? tp.k2 : tp.k3));
            for (int i = 0; i < gk.ksize; ++i) dst[i] = gk.k[i];
        }
        const int seg = 256;
        hipLaunchKernelGGL(pyramid_fused, dim3((W + 255) / 256, (H + seg - 1) / seg, P * 2), dim3(256), 0, s, orig, next, pair_stride, H, W,
                           lvl_in[0], lvl_in[1], lvl_in[2], lvl_in[3], seg, tp);
        stage_bytes += 2.0 * P * HW * (3 + 4 * (1 + 0.25 + 0.0625 + 0.015625));      // the frames once, the four level inputs once
    } else if (HW % 4 == 0 && pair_stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(orig) | reinterpret_cast<uintptr_t>(next)) & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(gray) & 15) == 0)
        hipLaunchKernelGGL(flow_gray_v4, dim3(nblocks(HW / 4), 1, P * 2), dim3(256), 0, s, orig, next, pair_stride, (int)HW, gray);
    else
        hipLaunchKernelGGL(flow_gray, dim3(nblocks(HW), 1, P * 2), dim3(256), 0, s, orig, next, pair_stride, (int)HW, gray);
    if (!pyr) stage_bytes += 2.0 * P * HW * (3 + 4);

    float* prev_flow = nullptr;
    int ph = 0, pw = 0;
    float* cur = flowA;
    for (int k = levels; k >= 0; --k) {
        double scale = 1.0;
        for (int i = 0; i < k; ++i) scale *= 0.5;
        const double sigma = (1. / scale - 1) * 0.5;
        int smooth = (int)lrint(sigma * 5) | 1;
        if (smooth < 3) smooth = 3;
        RELAX_REQUIRE(h, smooth <= MAX_GAUSS, "optical flow: smoothing kernel %d too large", smooth);
        RELAX_REQUIRE(h, k < 2 || 129.0 / scale + smooth + 4 <= GH_SEG, "optical flow: pyramid scale %g too coarse for the row stage", scale);
        const int w = (int)lrint(W * scale), hh = (int)lrint(H * scale);
        const int64_t hw = (int64_t)w * hh;
        if (!prev_flow) RELAX_HIP_CHECK(h, hipMemsetAsync(cur, 0, sizeof(float) * P * 2 * hw, s));
        // (finer levels: the first matrix update resizes the coarser flow where it reads it - no resized plane)
        GaussKernel gk;
        make_gauss(smooth, sigma, &gk);
        const float* Isrc;
        const dim3 g_full(nblocks(W), H, P * 2), g_lvl2(nblocks(w), hh, P * 2), g_um(nblocks(w), 8 * ((hh + 7) / 8), P);
        if (!pyr)   // per-level blur (+ resize): the gray frame read once, the level input written once (+ the full-size blur written and read at level 1)
            stage_bytes += 2.0 * P * (4.0 * HW + 4.0 * hw + (k == 1 ? 8.0 * HW : 0.0));
        if (pyr) {
            Isrc = lvl_in[k];
        } else if (k >= 2) {   // coarse levels: blur only where the resize samples
            hipLaunchKernelGGL(gauss_h_sampled, dim3(nblocks(2 * w), H, P * 2), dim3(256), 0, s, gray, tmp, H, W, w, (double)W / w, gk);
            hipLaunchKernelGGL(gauss_v_sampled_resize, g_lvl2, dim3(256), 0, s, tmp, I, H, W, hh, w, (double)H / hh,
                               (double)W / w, gk);
            Isrc = I;
        } else {
            if (gk.ksize == 3 && vec4_ok(W, HW, gray, blur)) {
                hipLaunchKernelGGL(gauss3_v4, dim3(nblocks((W + 3) / 4), H, P * 2), dim3(256), 0, s, gray, blur, H, W, gk.k[0], gk.k[1],
                                   gk.k[2]);
            } else {
                hipLaunchKernelGGL(gauss_pass<false>, g_full, dim3(256), 0, s, gray, tmp, H, W, gk);
                hipLaunchKernelGGL(gauss_pass<true>, g_full, dim3(256), 0, s, tmp, blur, H, W, gk);
            }
            Isrc = blur;
            if (w != W || hh != H) {
                hipLaunchKernelGGL(resize_linear_f32<1>, g_lvl2, dim3(256), 0, s, blur, I, H, W, hh, w, (double)H / hh,
                                   (double)W / w, 1.0f);
                Isrc = I;
            }
        }
        {
            int seg = 64;   // rows per block (16 to fill the ring before the first one); shorter segments when the level is small (enough blocks for 256 CUs)
            const int bands = (w + POLY_OUT - 1) / POLY_OUT;
            while (seg > 16 && (int64_t)bands * ((hh + seg - 1) / seg) * P * 2 < 2048) seg -= 16;
            hipLaunchKernelGGL(poly_expansion, dim3(bands, (hh + seg - 1) / seg, P * 2), dim3(256), 0, s, Isrc, R, hh, w, seg, pc);
            stage_bytes += 2.0 * P * hw * 24;      // 4 bytes in, 5 coefficients out per pixel and frame
        }
        if (h->gemm.flow_fused) {
            // one kernel per iteration (flow_iteration): M stays on the chip.  The iterations ping-pong between the two flow buffers:
            // a block reads flow rows its neighbours may already have rewritten otherwise.
            // measurement (relax_profile_read kind 5): algorithmic bytes per pixel = 2 x 5 floats of R at the pixel and at the displaced
            // position (counted once: neighbours share the lines) + the flow in and out = 56 bytes
            const double it_bytes = 56.0 * (double)hw * P;
            stage_bytes += ITERS * it_bytes;
            float* other = (cur == flowA) ? flowB : flowA;
            const float* in = cur;                // coarsest level: zeros
            float* out = other;
            const FlowUp up = prev_flow ? FlowUp{prev_flow, ph, pw, (double)ph / hh, (double)pw / w} : FlowUp{};
            if (prev_flow) out = cur;             // finer levels: the first iteration interpolates the coarser flow (in `other`)
            for (int it = 0; it < ITERS; ++it) {
                int span;
                RELAX_TRY(prof_begin(h, s, 3, it_bytes, &span));
                if (it == 0 && prev_flow) {
                    RELAX_TRY((launch_flow_iteration<true, false>(h, R, nullptr, out, hh, w, P, up, nullptr, s)));
                } else if (k == 0 && it == ITERS - 1 && bgr_out) {   // the final flow: its magnitude range is taken on the way out
                    RELAX_TRY(minmax_reset(h, P, mm, s));
                    RELAX_TRY((launch_flow_iteration<false, true>(h, R, in, out, hh, w, P, FlowUp{}, mm, s)));
                } else {
                    RELAX_TRY((launch_flow_iteration<false, false>(h, R, in, out, hh, w, P, FlowUp{}, nullptr, s)));
                }
                RELAX_TRY(prof_end(h, s, span));
                in = out;
                out = (out == flowA) ? flowB : flowA;
            }
            cur = const_cast<float*>(in);         // the level's result
        } else {
            // measurement (relax_profile_read kind 5): the dominant kernel of the stage, with its algorithmic bytes: per pixel
            // 2 x 5 floats of R at the pixel + 5 floats of R1 gathered at the displaced position (counted once: neighbours share
            // the lines) ... = 40 + 8 (flow) + 20 (M written) = 68 bytes
            const double um_bytes = 68.0 * (double)hw * P;
            stage_bytes += ITERS * (um_bytes + 28.0 * (double)hw * P);      // + box_solve_fused: M read, flow written
            int um_span;
            RELAX_TRY(prof_begin(h, s, 3, um_bytes, &um_span));
            if (prev_flow) {
                const FlowUp up{prev_flow, ph, pw, (double)ph / hh, (double)pw / w};
                hipLaunchKernelGGL(update_matrices_k<true>, g_um, dim3(256), 0, s, R, cur, M, hh, w, (hh + 7) / 8, up);
            } else {
                hipLaunchKernelGGL(update_matrices_k<false>, g_um, dim3(256), 0, s, R, cur, M, hh, w, (hh + 7) / 8, FlowUp{});
            }
            RELAX_TRY(prof_end(h, s, um_span));
            {
                const int bands = (w + FUSE_OUT - 1) / FUSE_OUT;
                const int seg = flow_segment_rows(h, bands, hh);   // multiples of the 15-row ring period; the same segments as flow_iteration
                const dim3 gf(bands, (hh + seg - 1) / seg, P);
                for (int it = 0; it < ITERS; ++it) {
                    if (k == 0 && it == ITERS - 1 && bgr_out) {   // the final flow: its magnitude range is taken on the way out
                        RELAX_TRY(minmax_reset(h, P, mm, s));
                        hipLaunchKernelGGL(box_solve_fused<true>, gf, dim3(256), 0, s, M, cur, hh, w, seg, mm);
                    } else {
                        hipLaunchKernelGGL(box_solve_fused<false>, gf, dim3(256), 0, s, M, cur, hh, w, seg, nullptr);
                    }
                    if (it < ITERS - 1) {
                        RELAX_TRY(prof_begin(h, s, 3, um_bytes, &um_span));
                        hipLaunchKernelGGL(update_matrices_k<false>, g_um, dim3(256), 0, s, R, cur, M, hh, w, (hh + 7) / 8, FlowUp{});
                        RELAX_TRY(prof_end(h, s, um_span));
                    }
                }
            }
        }
        prev_flow = cur;
        ph = hh;
        pw = w;
        cur = (cur == flowA) ? flowB : flowA;
    }
    RELAX_HIP_CHECK(h, hipGetLastError());
    if (flow_out)
        RELAX_HIP_CHECK(h, hipMemcpyAsync(flow_out, prev_flow, sizeof(float) * P * 2 * HW, hipMemcpyDeviceToDevice, s));
    if (bgr_out) {
        RELAX_TRY(visualise(h, prev_flow, P, (int)HW, bgr_out, mm, s, true));
        stage_bytes += 11.0 * P * HW;              // the flow read, 3 bytes per pixel written
    }
    prof_set_work(h, stage_span, stage_bytes);
    RELAX_TRY(prof_end(h, s, stage_span));
    return RELAX_OK;
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_optical_flow(relax_handle* h, const uint8_t* orig, const uint8_t* next, int64_t pair_stride, int T, int H, int W,
                       float* flow, uint8_t* flow_bgr, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, orig && next && T > 0 && H >= 16 && W >= 16, "relax_optical_flow: bad arguments");
    RELAX_REQUIRE(h, flow || flow_bgr, "relax_optical_flow: no output requested");
    RELAX_REQUIRE(h, pair_stride >= (int64_t)H * W * 3 || T == 1, "relax_optical_flow: pair stride smaller than a frame");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // bound the workspace: 108 B per pixel per pair; chunks of up to RELAX_FLOW_WS_GB (default 32) GB - 288 GB of HBM make
    // large chunks cheap, and more pairs per launch fill the chip better at the coarse pyramid levels
    const int64_t HW = (int64_t)H * W;
    int64_t ws_gb = 32;
    if (const char* e = getenv("RELAX_FLOW_WS_GB")) ws_gb = atoll(e) > 0 ? atoll(e) : ws_gb;
    int chunk = (int)((ws_gb << 30) / (HW * 112));
    if (h->gemm.flow_max_pairs > 0 && chunk > h->gemm.flow_max_pairs) chunk = h->gemm.flow_max_pairs;   // tests: force the chunk loop
    if (chunk < 1) chunk = 1;
    if (chunk > T) chunk = T;
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int P = T - t0 < chunk ? T - t0 : chunk;
        RELAX_TRY(flow_chunk(h, orig + t0 * pair_stride, next + t0 * pair_stride, pair_stride, P, H, W,
                             flow ? flow + (int64_t)t0 * HW * 2 : nullptr, flow_bgr ? flow_bgr + (int64_t)t0 * HW * 3 : nullptr, s));
    }
    return RELAX_OK;
}

int relax_flow_to_rgb(relax_handle* h, const float* flow, int T, int H, int W, uint8_t* flow_bgr, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, flow && flow_bgr && T > 0 && H > 0 && W > 0, "relax_flow_to_rgb: bad arguments");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    RELAX_TRY(ensure_buf(h, h->scratch, sizeof(unsigned) * 2 * (size_t)T));
    return visualise(h, flow, T, H * W, flow_bgr, static_cast<unsigned*>(h->scratch.p), static_cast<hipStream_t>(stream));
}

}  // extern "C"
