// Whole-frame front-ends: H x W -> 224 x 224 exactly as Pillow's 8-bit resample does it, for
// the two filters the reference uses:
//   BILINEAR (antialiased)  transforms.Resize((224,224)) on a PIL image   src/extractor/visualise_resnet.py:40-47
//   LANCZOS                 img.resize((224,224), Image.Resampling.LANCZOS) src/extractor/visualise_vit_layer.py:466-469
// Pillow (libImaging/Resample.c): separable, horizontal pass first; coefficients are computed in double, normalised,
// and quantised to 22-bit fixed point; every output is clip8((2^21 + sum(pixel * k)) >> 22); the horizontal result is
// stored as uint8 before the vertical pass.  The coefficient tables are built on the host with the same double
// arithmetic and cached per (size, filter); the kernels are exact integer arithmetic, so the result is bit-identical.
// HBM-bound: each frame is read once (both filters share the horizontal read of a row through LDS).
#include <cmath>

#include "relax_internal.h"

namespace relax {

constexpr int OUT = RELAX_TARGET;  // 224
constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int ROWS_PER_BLOCK = 4;

static double filt_bilinear(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}
static double sinc(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double filt_lanczos(double x) { return (-3.0 <= x && x < 3.0) ? sinc(x) * sinc(x / 3) : 0.0; }

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size)
static void build_table(int in_size, int filt, std::vector<int32_t>& bounds, std::vector<int32_t>& coeffs, int* ksize_out) {
    double (*f)(double) = filt == 0 ? filt_bilinear : filt_lanczos;
    const double support0 = filt == 0 ? 1.0 : 3.0;
    const double scale = (double)in_size / OUT;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = support0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign(OUT * 2, 0);
    coeffs.assign((size_t)OUT * ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> k(ksize);
    for (int xx = 0; xx < OUT; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = f((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x];
            coeffs[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[xx * 2] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    *ksize_out = ksize;
}

__device__ inline uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct ResizeArgs {
    const int32_t* bounds[2];  // per filter
    const int32_t* coeffs[2];
    int ksize[2];
    uint8_t* out[2];           // null = filter not requested
};

// Horizontal pass: a workgroup stages ROWS_PER_BLOCK input rows in LDS (16-byte coalesced loads when aligned) and
// produces the 224 x 3 outputs of both filters for each of them; the coefficient of a tap is loaded once and applied
// to all staged rows.
__global__ __launch_bounds__(256) void resize_horizontal(const uint8_t* __restrict__ frames, int64_t item_stride, int H,
                                                         int W, ResizeArgs a, bool aligned) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rows[];  // [ROWS_PER_BLOCK][row_pad]
    const int n = blockIdx.y;
    const int y0 = blockIdx.x * ROWS_PER_BLOCK;
    const int row_bytes = W * 3;
    const int row_pad = (row_bytes + 15) & ~15;
    const int nrows = H - y0 < ROWS_PER_BLOCK ? H - y0 : ROWS_PER_BLOCK;
    const uint8_t* src = frames + n * item_stride + (int64_t)y0 * row_bytes;
    if (aligned) {
        const int chunks = row_bytes / 16;
        for (int i = threadIdx.x; i < nrows * chunks; i += blockDim.x) {
            const int r = i / chunks, c = i % chunks;
            *reinterpret_cast<uint4*>(rows + r * row_pad + c * 16) =
                *reinterpret_cast<const uint4*>(src + (int64_t)r * row_bytes + c * 16);
        }
    } else {
        for (int i = threadIdx.x; i < nrows * row_bytes; i += blockDim.x) {
            const int r = i / row_bytes, c = i % row_bytes;
            rows[r * row_pad + c] = src[(int64_t)r * row_bytes + c];
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * OUT * 3; o += blockDim.x) {
        const int f = o / (OUT * 3);
        if (!a.out[f]) continue;
        const int rem = o - f * OUT * 3;
        const int xx = rem / 3, c = rem - xx * 3;
        const int xmin = a.bounds[f][xx * 2], cnt = a.bounds[f][xx * 2 + 1];
        const int32_t* kk = a.coeffs[f] + (size_t)xx * a.ksize[f];
        int acc[ROWS_PER_BLOCK];
#pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; ++r) acc[r] = 1 << (PRECISION_BITS - 1);
        const uint8_t* p = rows + xmin * 3 + c;
        for (int x = 0; x < cnt; ++x) {
            const int k = kk[x];
#pragma unroll
            for (int r = 0; r < ROWS_PER_BLOCK; ++r) acc[r] += (int)p[r * row_pad + x * 3] * k;
        }
        uint8_t* dst = a.out[f] + ((int64_t)n * H + y0) * (OUT * 3) + rem;
#pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; ++r)
            if (r < nrows) dst[(int64_t)r * (OUT * 3)] = clip8(acc[r]);
    }
}

// Vertical pass on the uint8 intermediate [N, H, 224, 3] -> [N, 224, 224, 3]; lanes walk adjacent bytes of a row.
__global__ __launch_bounds__(256) void resize_vertical(const uint8_t* __restrict__ tmp, int H, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ coeffs, int ksize,
                                                       uint8_t* __restrict__ out) {
    const int n = blockIdx.z, yy = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= OUT * 3) return;
    const int ymin = bounds[yy * 2], cnt = bounds[yy * 2 + 1];
    const int32_t* kk = coeffs + (size_t)yy * ksize;
    const uint8_t* p = tmp + ((int64_t)n * H + ymin) * (OUT * 3) + col;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) acc += (int)p[(int64_t)y * (OUT * 3)] * kk[y];
    out[((int64_t)n * OUT + yy) * (OUT * 3) + col] = clip8(acc);
}

static int get_table(relax_handle* h, int in_size, int filt, const ResizeTable** out) {
    for (const ResizeTable& t : h->resize_tables)
        if (t.in_size == in_size && t.filt == filt) {
            *out = &t;
            return RELAX_OK;
        }
    std::vector<int32_t> bounds, coeffs;
    ResizeTable t;
    t.in_size = in_size;
    t.filt = filt;
    build_table(in_size, filt, bounds, coeffs, &t.ksize);
    RELAX_HIP_CHECK(h, hipMalloc(reinterpret_cast<void**>(&t.bounds), bounds.size() * sizeof(int32_t)));
    RELAX_HIP_CHECK(h, hipMalloc(reinterpret_cast<void**>(&t.coeffs), coeffs.size() * sizeof(int32_t)));
    RELAX_HIP_CHECK(h, hipMemcpy(t.bounds, bounds.data(), bounds.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    RELAX_HIP_CHECK(h, hipMemcpy(t.coeffs, coeffs.data(), coeffs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->resize_tables.push_back(t);
    *out = &h->resize_tables.back();
    return RELAX_OK;
}

void free_resize(relax_handle* h) {
    for (ResizeTable& t : h->resize_tables) {
        (void)hipFree(t.bounds);
        (void)hipFree(t.coeffs);
    }
    h->resize_tables.clear();
    if (h->resize_ws.p) (void)hipFree(h->resize_ws.p);
    h->resize_ws = DevBuf();
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_resize_frames(relax_handle* h, const uint8_t* frames, int64_t item_stride, int N, int H, int W,
                        uint8_t* out_bilinear, uint8_t* out_lanczos, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, frames && N > 0 && H > 0 && W > 0, "relax_resize_frames: bad arguments");
    RELAX_REQUIRE(h, out_bilinear || out_lanczos, "relax_resize_frames: no output requested");
    RELAX_REQUIRE(h, item_stride >= (int64_t)H * W * 3 || N == 1, "relax_resize_frames: item stride smaller than a frame");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // tables first (h->resize_tables may reallocate while growing: fetch every pointer after the last insertion)
    const ResizeTable* t;
    for (int f = 0; f < 2; ++f) {
        RELAX_TRY(get_table(h, W, f, &t));
        RELAX_TRY(get_table(h, H, f, &t));
    }
    const size_t tmp_bytes = (size_t)N * H * OUT * 3;
    RELAX_TRY(ensure_buf(h, h->resize_ws, 2 * tmp_bytes));
    uint8_t* tmp[2] = {static_cast<uint8_t*>(h->resize_ws.p), static_cast<uint8_t*>(h->resize_ws.p) + tmp_bytes};
    uint8_t* outs[2] = {out_bilinear, out_lanczos};
    ResizeArgs a{};
    for (int f = 0; f < 2; ++f) {
        RELAX_TRY(get_table(h, W, f, &t));
        a.bounds[f] = t->bounds;
        a.coeffs[f] = t->coeffs;
        a.ksize[f] = t->ksize;
        a.out[f] = outs[f] ? tmp[f] : nullptr;
    }
    const int row_pad = (W * 3 + 15) & ~15;
    const bool aligned = (W * 3) % 16 == 0 && item_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(frames) & 15) == 0;
    const size_t lds = (size_t)ROWS_PER_BLOCK * row_pad;
    RELAX_REQUIRE(h, lds <= 64 * 1024, "relax_resize_frames: W=%d too wide for the LDS row stage", W);
    hipLaunchKernelGGL(resize_horizontal, dim3((H + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, N), dim3(256), lds, s, frames,
                       item_stride, H, W, a, aligned);
    for (int f = 0; f < 2; ++f) {
        if (!outs[f]) continue;
        RELAX_TRY(get_table(h, H, f, &t));
        hipLaunchKernelGGL(resize_vertical, dim3((OUT * 3 + 255) / 256, OUT, N), dim3(256), 0, s, tmp[f], H, t->bounds,
                           t->coeffs, t->ksize, outs[f]);
    }
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

}  // extern "C"
