// ResNet-50 conv1 (7x7, stride 2, pad 3, 3 -> 64 channels, no bias: the raw pre-BN tap) under "bf16x6", straight from the uint8
// fragments: ToTensor + Normalize, im2col, the split into bf16 planes and the contraction in one kernel.
//
// The generic implicit-GEMM kernel (gemm_x6.hip) wants 16-channel chunks per tap; with 3 input channels that would be 49 taps x 16 =
// 5.3x the MACs.  Here the K axis is (ky, kx, c) with the 7 x 3 = 21 values of one filter row padded to 32 (two 16-deep MFMA steps):
// K = 7 x 32 = 224 against 147 real - 9.1 bf16 MFMA passes per fp32 MAC instead of the 16 of the exact-fp32 kernel that ran this
// layer before (3.7 ms per 1024 fragments, 1.5 % of a config-3 step).  The 21 values of a (pixel, ky) are CONTIGUOUS in an RGB-interleaved
// image row, so an MFMA A fragment (8 consecutive k of one output pixel) is 8 consecutive floats of the input patch:
//   * a persistent workgroup (8 waves) keeps the 64 x 224 weights as split planes in LDS (85 KB) and walks 256-pixel tiles of the
//     112x112 output (49 per image: a tile never straddles two images);
//   * per tile it normalises the up to 13 input rows the tile's pixels tap into an fp32 patch in LDS (zero padded borders); the patch
//     is double-buffered: the next tile's rows are requested (as dwords, into registers) before the MFMA loop and written after it,
//   * every wave owns 32 output pixels x 64 channels: per K step it reads its 8 patch floats per lane, splits them into hi / mid / lo
//     (exact), reads the weight fragments and issues the six partial products (smallest first, as gemm_x6);
//   * the accumulators go straight to the NHWC fp32 output (a register of a 32x32 tile is 32 consecutive channels of one pixel),
//     and the 16-pixel group sums of the fused spatial mean (tap 0 of the layer stack) are formed in registers.
// Columns 21 .. 31 of a padded filter row read whatever follows in the patch (finite values) against ZERO weights.
// H2 ("f16x2", with "rn_h2_early"): the same kernel on two fp16 planes - the patch values x 2^13 (|normalised pixel| <= 2.64: a static
// scale) split in registers, the weights as planes with one power-of-two scale per filter, the three partial products al bh, ah bl, ah bh
// on v_mfma_f32_32x32x16_f16 with the two small ones in accumulators of their own (gemm_x6.hip, H2), half the matrix instructions.
#include "relax_internal.h"
#include "sp3.h"
#include "h2.h"

namespace relax {

typedef float c1_floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c1_f16x8 __attribute__((ext_vector_type(8)));
[[maybe_unused]] constexpr float kC1PatchScale = 8192.f;   // H2: 2.64 * 2^13 = 21.6 k < 65504
typedef float c1_f32x2 __attribute__((ext_vector_type(2)));

constexpr int C1_HW = 224, C1_OW = 112, C1_OPIX = C1_OW * C1_OW;   // 12544 = 49 * 256
constexpr int C1_TILE = 256;
constexpr int C1_KROW = 32;                         // 21 real (kx, c) values per filter row, padded
constexpr int C1_K = 7 * C1_KROW;                   // 224
constexpr int C1_CHUNKS = C1_K / 16;                // 14
constexpr int C1_WROW = C1_CHUNKS * kChunkBytes + 16;   // LDS bytes per weight row (+16: conflict-free ds_read_b128 across rows)
constexpr int C1_PCOLS = (C1_HW + 6) * 3;           // 690 floats per patch row (3 padding pixels on each side)
constexpr int C1_PROW = C1_PCOLS + 2;               // 692 (even: 8-byte aligned fragment reads)
constexpr int C1_PROWS = 13;                        // 2 * 3 + 7: a tile spans at most 4 output rows
constexpr int C1_W_BYTES = 64 * C1_WROW;
constexpr int C1_PATCH_FLOATS = C1_PROWS * C1_PROW + 64;   // slack: the padded k columns of the last row read past its end
constexpr size_t C1_LDS = C1_W_BYTES + 2 * sizeof(float) * C1_PATCH_FLOATS + 3 * 256 * sizeof(float);   // weights + two patch buffers + the normalisation table: 158.8 of 160 KiB

// fp32 weights [64][Kpad] of the exact-fp32 kernel, k = (ky*7 + kx)*4 + c  ->  [64][224], k = ky*32 + kx*3 + c (zeros elsewhere)
__global__ __launch_bounds__(256) void conv1_repack(const float* __restrict__ w, int kpad, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * C1_K) return;
    const int n = i / C1_K, k = i - n * C1_K;
    const int ky = k / C1_KROW, j = k - ky * C1_KROW;
    float v = 0.f;
    if (j < 21) v = w[(int64_t)n * kpad + (ky * 7 + j / 3) * 4 + j % 3];
    out[i] = v;
}

// one dword = 4 consecutive bytes k = 4d .. 4d+3 of a 672-byte BGR image row -> normalised RGB floats at their patch positions
// (k = 3 px + c, c: 0 = B, 1 = G, 2 = R  ->  patch column 9 + 3 px + (2 - c) = 11 + k - 2c).  lut[c][byte] = ((byte / 255) - mean_c) /
// std_c, computed once per workgroup with the operations of rn_preprocess (two fp32 divisions per value are ~25 instructions:
// looked up, not redone 20 times per thread and tile)
__device__ inline void conv1_put4(float* prow, int d, unsigned v, bool valid, const float* lut) {
    const int d3 = d % 3;                                  // 4d mod 3
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int c = (d3 + b) % 3;
        prow[11 + 4 * d + b - 2 * c] = valid ? lut[c * 256 + ((v >> (8 * b)) & 255u)] : 0.f;   // zeros above / below the image
    }
}

constexpr int C1_ROW_DW = C1_HW * 3 / 4;                   // 168 dwords per image row
[[maybe_unused]] constexpr int C1_LOADS = (C1_PROWS * C1_ROW_DW + 511) / 512;   // 5 dwords per thread and tile

template <bool H2>
__global__ __launch_bounds__(512) void conv1_x6(const uint8_t* __restrict__ frags, const char* __restrict__ w_sp3, float* __restrict__ out,
                                                float* __restrict__ gap, int n_tiles, const float* __restrict__ w_inv) {
#if __HIP_DEVICE_COMPILE__
    constexpr int CH = H2 ? kH2ChunkBytes : kChunkBytes;          // bytes of a 16-deep chunk of a weight row (two fp16 / three bf16 planes)
    constexpr int UPR = C1_CHUNKS * CH / 16;                      // 16-byte units per weight row
    constexpr int WROW = C1_CHUNKS * CH + 16;                     // (+16: conflict-free ds_read_b128 across rows, both forms)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    float* patch0 = reinterpret_cast<float*>(smem + C1_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;

    // weights: 64 rows x 84 16-byte units, rows padded to C1_WROW bytes; both patch buffers start as zeros (their 3-pixel side
    // borders and the slack are never written again)
    for (int u = tid; u < 64 * UPR; u += 512) {
        const int n = u / UPR, q = u - n * UPR;
        *reinterpret_cast<sp3_u32x4*>(wl + n * WROW + q * 16) = *reinterpret_cast<const sp3_u32x4*>(w_sp3 + ((int64_t)n * UPR + q) * 16);
    }
    for (int i = tid; i < 2 * C1_PATCH_FLOATS; i += 512) patch0[i] = 0.f;
    float* lut = patch0 + 2 * C1_PATCH_FLOATS;
    for (int i = tid; i < 3 * 256; i += 512) {
        const int c = i >> 8;
        const float u = (float)(i & 255) / 255.0f;                                                      // as rn_preprocess
        lut[i] = c == 2 ? (u - 0.485f) / 0.229f : c == 1 ? (u - 0.456f) / 0.224f : (u - 0.406f) / 0.225f;
    }
    __syncthreads();

    // the input rows of a tile as raw dwords in registers (requested before the MFMA loop of the tile before, written after it)
    unsigned raw[C1_LOADS];
    auto tile_geom = [&](int t, int& img, int& p0, int& y0, int& iy0, int& rows) {
        img = t / (C1_OPIX / C1_TILE);
        p0 = (t - img * (C1_OPIX / C1_TILE)) * C1_TILE;
        y0 = p0 / C1_OW;
        iy0 = 2 * y0 - 3;
        rows = 2 * ((p0 + C1_TILE - 1) / C1_OW - y0) + 7;
    };
    auto request = [&](int t) {
        int img, p0, y0, iy0, rows;
        tile_geom(t, img, p0, y0, iy0, rows);
        const unsigned* im = reinterpret_cast<const unsigned*>(frags + (int64_t)img * (C1_HW * C1_HW * 3));
#pragma unroll
        for (int i = 0; i < C1_LOADS; ++i) {
            const int e = tid + 512 * i;
            const int row = e / C1_ROW_DW, d = e - row * C1_ROW_DW;
            const int iy = iy0 + row;
            raw[i] = (row < rows && (unsigned)iy < (unsigned)C1_HW) ? im[iy * C1_ROW_DW + d] : 0u;
        }
    };
    auto deposit = [&](int t, float* patch) {
        int img, p0, y0, iy0, rows;
        tile_geom(t, img, p0, y0, iy0, rows);
#pragma unroll
        for (int i = 0; i < C1_LOADS; ++i) {
            const int e = tid + 512 * i;
            const int row = e / C1_ROW_DW, d = e - row * C1_ROW_DW;
            if (row < rows) conv1_put4(patch + row * C1_PROW, d, raw[i], (unsigned)(iy0 + row) < (unsigned)C1_HW, lut);
        }
    };

    int cur = 0;
    if ((int)blockIdx.x < n_tiles) {
        request(blockIdx.x);
        deposit(blockIdx.x, patch0);
    }
    __syncthreads();
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        int img, p0, y0, iy0, rows;
        tile_geom(t, img, p0, y0, iy0, rows);
        const float* patch = patch0 + cur * C1_PATCH_FLOATS;
        const int next = t + gridDim.x;
        if (next < n_tiles) request(next);                 // in flight under the MFMAs below

        const int p = p0 + wave * 32 + r;                  // this lane's output pixel (A rows of the wave's 32x64 block)
        const int y = p / C1_OW, x = p - y * C1_OW;
        const float* arow = patch + (2 * (y - y0)) * C1_PROW + 6 * x + 8 * half;
        c1_floatx16 acc[2], accs[2];   // (accs: H2's small products)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; accs[0][i] = 0.f; accs[1][i] = 0.f; }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int jc = 0; jc < 2; ++jc) {
                const float* a = arow + ky * C1_PROW + 16 * jc;
                const c1_f32x2 a0 = *reinterpret_cast<const c1_f32x2*>(a), a1 = *reinterpret_cast<const c1_f32x2*>(a + 2),
                               a2 = *reinterpret_cast<const c1_f32x2*>(a + 4), a3 = *reinterpret_cast<const c1_f32x2*>(a + 6);
                const int chunk = ky * 2 + jc;
                if constexpr (H2) {
                    h2_u32x4 ah, al;
                    split2_x8((h2_f32x4){a0.x, a0.y, a1.x, a1.y} * kC1PatchScale, (h2_f32x4){a2.x, a2.y, a3.x, a3.y} * kC1PatchScale, ah, al);
                    const c1_f16x8 Ah = __builtin_bit_cast(c1_f16x8, ah), Al = __builtin_bit_cast(c1_f16x8, al);
                    c1_f16x8 Bh[2], Bl[2];
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const char* wp = wl + (cb * 32 + r) * WROW + chunk * CH + half * 16;
                        Bh[cb] = *reinterpret_cast<const c1_f16x8*>(wp);
                        Bl[cb] = *reinterpret_cast<const c1_f16x8*>(wp + 32);
                    }
                    accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[0], accs[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[1], accs[1], 0, 0, 0);
                    accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[0], accs[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[1], accs[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[1], acc[1], 0, 0, 0);
                    continue;
                }
                sp3_u32x4 ah, am, al;
                split3_x8((sp3_f32x4){a0.x, a0.y, a1.x, a1.y}, (sp3_f32x4){a2.x, a2.y, a3.x, a3.y}, ah, am, al);
                const c1_bf16x8 Ah = __builtin_bit_cast(c1_bf16x8, ah), Am = __builtin_bit_cast(c1_bf16x8, am),
                                Al = __builtin_bit_cast(c1_bf16x8, al);
                c1_bf16x8 Bh[2], Bm[2], Bl[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const char* wp = wl + (cb * 32 + r) * WROW + chunk * CH + half * 16;
                    Bh[cb] = *reinterpret_cast<const c1_bf16x8*>(wp);
                    Bm[cb] = *reinterpret_cast<const c1_bf16x8*>(wp + 32);
                    Bl[cb] = *reinterpret_cast<const c1_bf16x8*>(wp + 64);
                }
                // the six partial products, smallest first; the two accumulators alternate (no MFMA waits on the one before it)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm[1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh[1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm[1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[1], acc[1], 0, 0, 0);
            }
        }
        // acc[cb][i]: channel cb*32 + r of pixel  32*wave + (i & 3) + 8 * (i >> 2) + 4 * half
        const int64_t pix0 = (int64_t)img * C1_OPIX + p0 + wave * 32;
        if constexpr (H2) {   // the scales are powers of two: exact
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[cb] = (acc[cb] + accs[cb]) * (w_inv[cb * 32 + r] * (1.f / kC1PatchScale));
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                out[(pix0 + (i & 3) + 8 * (i >> 2) + 4 * half) * 64 + cb * 32 + r] = acc[cb][i];
            if (gap) {   // sums over the two aligned groups of 16 pixels of this wave (pixels in a fixed order: batch-invariant)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s += acc[cb][8 * g + i];
                    s += __shfl_xor(s, 32);
                    if (half == 0) gap[((pix0 >> 4) + g) * 64 + cb * 32 + r] = s;
                }
            }
        }
        if (next < n_tiles) deposit(next, patch0 + (cur ^ 1) * C1_PATCH_FLOATS);   // the other buffer: nobody reads it during this tile
        __syncthreads();                                   // next patch complete, every wave done with this one
        cur ^= 1;
    }
#endif
}

// w_f32 [64][224] scratch -> w_sp3 [64][224 * 6 B]
int make_conv1_x6_weights(relax_handle* h, const float* w_packed, int kpad, void** w_sp3_out, std::vector<void*>& allocs) {
    float* tmp = nullptr;
    void* q = nullptr;
    if (hipMalloc(&tmp, sizeof(float) * 64 * C1_K) != hipSuccess || hipMalloc(&q, (size_t)64 * C1_K * 6) != hipSuccess) {
        if (tmp) (void)hipFree(tmp);
        set_error(h, "resnet50: hipMalloc of the conv1 split-plane weights failed");
        return RELAX_ERR_NOMEM;
    }
    allocs.push_back(q);
    hipLaunchKernelGGL(conv1_repack, dim3((64 * C1_K + 255) / 256), dim3(256), 0, nullptr, w_packed, kpad, tmp);
    int rc = launch_to_sp3(h, tmp, C1_K, q, 64, C1_K, nullptr);
    if (hipDeviceSynchronize() != hipSuccess && rc == RELAX_OK) {
        set_error(h, "resnet50: conv1 weight conversion failed");
        rc = RELAX_ERR_HIP;
    }
    (void)hipFree(tmp);
    *w_sp3_out = q;
    return rc;
}

// the same weights as two fp16 planes [64][224 * 4 B] with one power-of-two scale per filter (w_inv_out [64]: the inverse scales)
int make_conv1_h2_weights(relax_handle* h, const float* w_packed, int kpad, void** w_h2_out, float** w_inv_out, std::vector<void*>& allocs) {
    float *tmp = nullptr, *inv = nullptr;
    void* q = nullptr;
    if (hipMalloc(&tmp, sizeof(float) * 64 * C1_K) != hipSuccess || hipMalloc(&q, (size_t)64 * C1_K * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&inv), sizeof(float) * 64) != hipSuccess) {
        if (tmp) (void)hipFree(tmp);
        if (q) (void)hipFree(q);
        set_error(h, "resnet50: hipMalloc of the conv1 fp16-plane weights failed");
        return RELAX_ERR_NOMEM;
    }
    allocs.push_back(q);
    allocs.push_back(inv);
    hipLaunchKernelGGL(conv1_repack, dim3((64 * C1_K + 255) / 256), dim3(256), 0, nullptr, w_packed, kpad, tmp);
    int rc = launch_to_h2_rows(h, tmp, C1_K, q, 64, C1_K, inv, nullptr);
    if (hipDeviceSynchronize() != hipSuccess && rc == RELAX_OK) {
        set_error(h, "resnet50: conv1 weight conversion failed");
        rc = RELAX_ERR_HIP;
    }
    (void)hipFree(tmp);
    *w_h2_out = q;
    *w_inv_out = inv;
    return rc;
}

// frags uint8 [N,224,224,3] BGR -> out fp32 [N,112,112,64] (raw conv1), gap_groups [N*784][64] (16-pixel sums) or null.
// w_inv != null: `w` holds fp16 planes (make_conv1_h2_weights) and the kernel runs its f16x2 form
int launch_conv1_x6(relax_handle* h, const uint8_t* frags, const void* w_sp3, float* out, float* gap_groups, int N, hipStream_t s, const float* w_inv) {
    RELAX_REQUIRE(h, frags && w_sp3 && out && N > 0, "conv1_x6: bad arguments");
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_x6<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_LDS));
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_x6<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_LDS));
        attr_set[h->device] = true;
    }
    const int n_tiles = N * (C1_OPIX / C1_TILE);
    const double flops = 2.0 * N * (double)C1_OPIX * 64.0 * 147.0;
    const double bytes = (double)N * (C1_HW * C1_HW * 3 + (double)C1_OPIX * 64 * 4);
    int span;
    RELAX_TRY(prof_begin(h, s, w_inv ? 5 : 2, flops, &span, bytes));
    if (w_inv)
        hipLaunchKernelGGL(conv1_x6<true>, dim3(n_tiles < 256 ? n_tiles : 256), dim3(512), C1_LDS, s, frags, static_cast<const char*>(w_sp3), out,
                           gap_groups, n_tiles, w_inv);
    else
        hipLaunchKernelGGL(conv1_x6<false>, dim3(n_tiles < 256 ? n_tiles : 256), dim3(512), C1_LDS, s, frags, static_cast<const char*>(w_sp3), out,
                           gap_groups, n_tiles, static_cast<const float*>(nullptr));
    if (hipGetLastError() != hipSuccess) { prof_abort(h, span); set_error(h, "conv1_x6: launch failed"); return RELAX_ERR_HIP; }
    RELAX_TRY(prof_end(h, s, span));
    return RELAX_OK;
}

}  // namespace relax
