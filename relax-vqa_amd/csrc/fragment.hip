// Stage A of the ReLaX-VQA hot path on gfx950: fused residual + 16x16 patch score, exact top-n
// selection, fragment gather, 50/50 merge.  Integer/byte work, HBM-bound: every input byte is read
// once by the score kernel (16-byte coalesced loads, v_sad_u8), the residual frame is never
// materialised, and the gather touches only the 196 selected patches.
//
// Reference semantics (file:line in xinyiW915/ReLaX-VQA):
//   src/main_fragment_layerstack.py:302      cv2.absdiff
//   src/main_fragment_layerstack.py:177-189  get_patch_diff  (sum of bytes per 16x16x3 patch)
//   src/main_fragment_layerstack.py:191-210  extract_important_patches (top 196, raster re-tile)
//   src/main_fragment_layerstack.py:212-230  get_original_frame_patches
//   src/main_fragment_layerstack.py:242-245  merge_fragments (addWeighted .5/.5 -> round half even)
#include "relax_internal.h"

namespace relax {

constexpr int P = RELAX_PATCH;            // 16
constexpr int PATCH_ROW_BYTES = P * 3;    // 48 = 3 chunks of 16 B
constexpr int TILES_PER_ROW = RELAX_TARGET / P;  // 14
constexpr int FRAG_ROW_BYTES = RELAX_TARGET * 3; // 672

// ---- patch score ------------------------------------------------------------------------------
// One workgroup per (patch row py, item t): a strip of 16 image rows.  A lane owns one 16-byte
// chunk column c (patch c/3) for the 16 rows; lanes are consecutive chunks, so a wave reads 1 KiB
// of contiguous frame bytes per instruction and keeps 16 such loads in flight.
template <bool PAIR>
__global__ __launch_bounds__(512) void patch_score_aligned(const uint8_t* __restrict__ a_base,
                                                           const uint8_t* __restrict__ b_base,
                                                           int64_t item_stride, int W, int pw, int ph,
                                                           uint32_t* __restrict__ scores) {
    extern __shared__ uint32_t lds_sum[];  // [pw]
    const int py = blockIdx.x;
    const int t = blockIdx.y;
    const int row_bytes = W * 3;
    const int nchunks = pw * 3;
    for (int p = threadIdx.x; p < pw; p += blockDim.x) lds_sum[p] = 0;
    __syncthreads();
    const uint8_t* a = a_base + t * item_stride + (int64_t)py * P * row_bytes;
    const uint8_t* b = PAIR ? b_base + t * item_stride + (int64_t)py * P * row_bytes : nullptr;
    // a lane owns chunk column c for all 16 rows of the strip: 8 rows (16 x 16-byte loads) in flight at a time
    for (int c = threadIdx.x; c < nchunks; c += blockDim.x) {
        uint32_t s = 0;
#pragma unroll
        for (int r0 = 0; r0 < P; r0 += 8) {
            uint4 va[8], vb[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int64_t off = (int64_t)(r0 + r) * row_bytes + c * 16;
                va[r] = *reinterpret_cast<const uint4*>(a + off);
                if (PAIR) vb[r] = *reinterpret_cast<const uint4*>(b + off);
                else vb[r] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                s = __builtin_amdgcn_sad_u8(va[r].x, vb[r].x, s);
                s = __builtin_amdgcn_sad_u8(va[r].y, vb[r].y, s);
                s = __builtin_amdgcn_sad_u8(va[r].z, vb[r].z, s);
                s = __builtin_amdgcn_sad_u8(va[r].w, vb[r].w, s);
            }
        }
        atomicAdd(&lds_sum[c / 3], s);  // integer LDS atomic: order-independent, exact
    }
    __syncthreads();
    uint32_t* out = scores + ((int64_t)t * ph + py) * pw;
    for (int p = threadIdx.x; p < pw; p += blockDim.x) out[p] = lds_sum[p];
}

// Any width / alignment: byte loads.  Work item = (row r, patch p): 48 bytes.
template <bool PAIR>
__global__ __launch_bounds__(256) void patch_score_generic(const uint8_t* __restrict__ a_base,
                                                           const uint8_t* __restrict__ b_base,
                                                           int64_t item_stride, int W, int pw, int ph,
                                                           uint32_t* __restrict__ scores) {
    extern __shared__ uint32_t lds_sum[];
    const int py = blockIdx.x;
    const int t = blockIdx.y;
    const int row_bytes = W * 3;
    for (int p = threadIdx.x; p < pw; p += blockDim.x) lds_sum[p] = 0;
    __syncthreads();
    const uint8_t* a = a_base + t * item_stride + (int64_t)py * P * row_bytes;
    const uint8_t* b = PAIR ? b_base + t * item_stride + (int64_t)py * P * row_bytes : nullptr;
    for (int item = threadIdx.x; item < pw * P; item += blockDim.x) {
        const int r = item / pw, p = item % pw;
        const int64_t off = (int64_t)r * row_bytes + p * PATCH_ROW_BYTES;
        uint32_t s = 0;
        for (int i = 0; i < PATCH_ROW_BYTES; ++i) {
            const int x = a[off + i];
            const int y = PAIR ? (int)b[off + i] : 0;
            s += (uint32_t)(x > y ? x - y : y - x);
        }
        atomicAdd(&lds_sum[p], s);
    }
    __syncthreads();
    uint32_t* out = scores + ((int64_t)t * ph + py) * pw;
    for (int p = threadIdx.x; p < pw; p += blockDim.x) out[p] = lds_sum[p];
}

// ---- exact top-n selection ------------------------------------------------------------------------
// One workgroup per item.  Order = (score desc, flat index asc); output = the selected flat
// indices in ascending order (== the reference's re-sort by (y,x)).  Two-level 16-bit radix select
// for the n-th largest score, then an ordered compaction with a block-wide prefix sum.
constexpr int SEL_THREADS = 256;   // 4 waves per pair: the work is latency, not bandwidth (32 KB of scores per 1080p pair)
constexpr int SEL_BINS = 1024;

__device__ inline uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off);
        if (lane >= off) v += u;
    }
    return v;
}

// exclusive prefix sum over the 256 threads (wave scans by lane shuffles, then the 4 wave totals through LDS)
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t incl = wave_inclusive_scan(v);
    __syncthreads();                       // wave_tot may still be read from the previous scan
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
        const uint32_t t = wave_tot[w];
        before += w < wave ? t : 0u;
        all += t;
    }
    *total = all;
    return before + incl - v;
}

__global__ __launch_bounds__(SEL_THREADS) void select_topn(const uint32_t* __restrict__ scores, int npatch, int pw,
                                                           int top_n, int32_t* __restrict__ positions,
                                                           int32_t* __restrict__ counts) {
    __shared__ uint32_t hist[SEL_BINS];  // 10 bits per level x 2 levels = 20-bit scores (max 195840 < 2^18)
    __shared__ uint32_t wave_tot[SEL_THREADS / 64];
    __shared__ uint32_t sh_digit, sh_remaining;
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t* s = scores + (int64_t)t * npatch;
    int32_t* pos = positions + (int64_t)t * RELAX_TOP_N * 2;
    const int want = top_n < npatch ? top_n : npatch;

    // scores are < 2^18 (16*16*3*255 = 195840); two 10-bit levels cover 2^20
    uint32_t thr = 0;        // the want-th largest score
    uint32_t need_ties = 0;  // how many entries == thr are selected (lowest indices first)
    if (want == 0) {
        thr = 0xffffffffu;        // nothing is selected (top_n = 0): no score exceeds or equals this
        need_ties = 0;            // (the digit search below never finds a bin for remaining = 0 and would leave thr undefined)
    } else if (want == npatch) {
        thr = 0;
        need_ties = 0xffffffffu;  // everything is selected
    } else {
        uint32_t remaining = (uint32_t)want;
        uint32_t prefix = 0;
        for (int level = 1; level >= 0; --level) {
            const int shift = level * 10;
            for (int b = tid; b < SEL_BINS; b += SEL_THREADS) hist[b] = 0;
            __syncthreads();
            for (int i = tid; i < npatch; i += SEL_THREADS) {
                const uint32_t v = s[i];
                if (level == 1 || (v >> 10) == prefix) atomicAdd(&hist[(v >> shift) & 1023u], 1u);
            }
            __syncthreads();
            // the digit d with  count(bins > d) < remaining <= count(bins >= d):  thread k owns the 4 bins
            // 1023-4k .. 1020-4k (descending), a block scan gives the count of everything above its bins
            uint32_t h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = hist[1023 - 4 * tid - j];
            uint32_t tot;
            uint32_t above = block_exclusive_scan(h[0] + h[1] + h[2] + h[3], wave_tot, &tot);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (above < remaining && above + h[j] >= remaining) {   // exactly one (thread, j) satisfies this
                    sh_digit = (uint32_t)(1023 - 4 * tid - j);
                    sh_remaining = remaining - above;  // still to take from bin d (>= 1)
                }
                above += h[j];
            }
            __syncthreads();
            remaining = sh_remaining;
            if (level == 1) prefix = sh_digit;
            else thr = (prefix << 10) | sh_digit;
            __syncthreads();
        }
        need_ties = remaining;
    }

    // ordered compaction: each thread owns a contiguous run of flat indices
    const int per = (npatch + SEL_THREADS - 1) / SEL_THREADS;
    const int lo = tid * per < npatch ? tid * per : npatch;
    const int hi = lo + per < npatch ? lo + per : npatch;
    uint32_t n_gt = 0, n_eq = 0;
    for (int i = lo; i < hi; ++i) {
        const uint32_t v = s[i];
        n_gt += v > thr;
        n_eq += v == thr;
    }
    uint32_t total_eq, total_sel;
    const uint32_t eq_before = block_exclusive_scan(n_eq, wave_tot, &total_eq);
    uint32_t eq_take = 0;
    if (need_ties == 0xffffffffu) eq_take = n_eq;
    else if (eq_before < need_ties) eq_take = (need_ties - eq_before) < n_eq ? (need_ties - eq_before) : n_eq;
    const uint32_t out_before = block_exclusive_scan(n_gt + eq_take, wave_tot, &total_sel);
    uint32_t o = out_before, eq_left = eq_take;
    for (int i = lo; i < hi; ++i) {
        const uint32_t v = s[i];
        bool take = v > thr;
        if (!take && v == thr && eq_left > 0) { take = true; --eq_left; }
        if (take) {
            pos[2 * o] = i / pw;
            pos[2 * o + 1] = i % pw;
            ++o;
        }
    }
    for (int k = (int)total_sel + tid; k < RELAX_TOP_N; k += SEL_THREADS) {
        pos[2 * k] = -1;
        pos[2 * k + 1] = -1;
    }
    if (tid == 0) counts[t] = (int32_t)total_sel;
}

// ---- fragment gather --------------------------------------------------------------------------------
__device__ inline uint32_t absdiff_u8x4(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < 32; s += 8) {
        const int x = (a >> s) & 255, y = (b >> s) & 255;
        r |= (uint32_t)(x > y ? x - y : y - x) << s;
    }
    return r;
}

// MODE 0: copy patches of `a`; MODE 1: |b - a| patches.  One work item = one 16-byte chunk of a patch row.
template <int MODE, bool ALIGNED>
__global__ __launch_bounds__(256) void gather_fragment(const uint8_t* __restrict__ a_base,
                                                       const uint8_t* __restrict__ b_base, int64_t item_stride,
                                                       int H, int W, const int32_t* __restrict__ positions,
                                                       const int32_t* __restrict__ counts,
                                                       uint8_t* __restrict__ frag) {
    const int t = blockIdx.y;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;  // (k, r, c): 196*16*3
    if (item >= RELAX_TOP_N * P * 3) return;
    const int k = item / (P * 3);
    const int rem = item % (P * 3);
    const int r = rem / 3, c = rem % 3;
    uint8_t* dst = frag + (int64_t)t * RELAX_FRAG_BYTES +
                   ((int64_t)((k / TILES_PER_ROW) * P + r) * FRAG_ROW_BYTES) + (k % TILES_PER_ROW) * PATCH_ROW_BYTES +
                   c * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < counts[t]) {
        const int y = positions[((int64_t)t * RELAX_TOP_N + k) * 2];
        const int x = positions[((int64_t)t * RELAX_TOP_N + k) * 2 + 1];
        // caller-supplied positions (relax_gather_patches) are not trusted: an out-of-range patch yields a zero tile
        if ((unsigned)y >= (unsigned)(H / P) || (unsigned)x >= (unsigned)(W / P)) {
            *reinterpret_cast<uint4*>(dst) = v;
            return;
        }
        const int64_t off = t * item_stride + ((int64_t)(y * P + r) * W + (int64_t)x * P) * 3 + c * 16;
        if (ALIGNED) {
            v = *reinterpret_cast<const uint4*>(a_base + off);
            if (MODE == 1) {
                const uint4 w = *reinterpret_cast<const uint4*>(b_base + off);
                v.x = absdiff_u8x4(v.x, w.x);
                v.y = absdiff_u8x4(v.y, w.y);
                v.z = absdiff_u8x4(v.z, w.z);
                v.w = absdiff_u8x4(v.w, w.w);
            }
        } else {
            uint32_t words[4] = {0, 0, 0, 0};
            for (int i = 0; i < 16; ++i) {
                int x0 = a_base[off + i];
                if (MODE == 1) {
                    const int y0 = b_base[off + i];
                    x0 = x0 > y0 ? x0 - y0 : y0 - x0;
                }
                words[i >> 2] |= (uint32_t)x0 << ((i & 3) * 8);
            }
            v = make_uint4(words[0], words[1], words[2], words[3]);
        }
    }
    *reinterpret_cast<uint4*>(dst) = v;  // fragment rows are 672 B: every chunk is 16-B aligned
}

// ---- merge: round-half-to-even(0.5a + 0.5b) on uint8 ------------------------------------------------
__device__ inline uint32_t merge_u8x4(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < 32; s += 8) {
        const uint32_t sum = ((a >> s) & 255u) + ((b >> s) & 255u);
        const uint32_t half = sum >> 1;
        r |= (half + ((sum & 1u) & (half & 1u))) << s;
    }
    return r;
}

__global__ __launch_bounds__(256) void merge_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                    uint8_t* __restrict__ out, int64_t n, bool vec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n16 = n / 16;
        if (i < n16) {
            const uint4 x = reinterpret_cast<const uint4*>(a)[i];
            const uint4 y = reinterpret_cast<const uint4*>(b)[i];
            reinterpret_cast<uint4*>(out)[i] =
                make_uint4(merge_u8x4(x.x, y.x), merge_u8x4(x.y, y.y), merge_u8x4(x.z, y.z), merge_u8x4(x.w, y.w));
        }
        if (i < n - n16 * 16) {
            const int64_t j = n16 * 16 + i;
            out[j] = (uint8_t)merge_u8x4(a[j], b[j]);
        }
    } else if (i < n) {
        out[i] = (uint8_t)merge_u8x4(a[i], b[i]);
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// shared driver for relax_fragment_pairs / relax_fragment_image
static int fragment_common(relax_handle* h, bool pair, const uint8_t* a, const uint8_t* b, int64_t item_stride,
                           int T, int H, int W, int top_n, int32_t* positions, int32_t* counts, uint8_t* frag_a,
                           uint8_t* frag_diff, uint32_t* scores, hipStream_t s) {
    RELAX_REQUIRE(h, a && (!pair || b), "fragment: NULL frame pointer");
    RELAX_REQUIRE(h, T > 0 && H > 0 && W > 0, "fragment: bad shape T=%d H=%d W=%d", T, H, W);
    RELAX_REQUIRE(h, top_n >= 0 && top_n <= RELAX_TOP_N, "fragment: top_n=%d must be in [0,%d]", top_n, RELAX_TOP_N);
    RELAX_REQUIRE(h, positions && counts, "fragment: positions/counts must not be NULL");
    RELAX_REQUIRE(h, item_stride >= (int64_t)H * W * 3 || T == 1, "fragment: item stride smaller than a frame");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    const int ph = H / P, pw = W / P;
    const int npatch = ph * pw;
    if (npatch == 0) {  // frame smaller than one patch: empty selection, zero canvases
        RELAX_HIP_CHECK(h, hipMemsetAsync(counts, 0, sizeof(int32_t) * T, s));
        RELAX_HIP_CHECK(h, hipMemsetAsync(positions, 0xff, sizeof(int32_t) * 2 * RELAX_TOP_N * T, s));
        if (frag_a) RELAX_HIP_CHECK(h, hipMemsetAsync(frag_a, 0, (size_t)RELAX_FRAG_BYTES * T, s));
        if (frag_diff) RELAX_HIP_CHECK(h, hipMemsetAsync(frag_diff, 0, (size_t)RELAX_FRAG_BYTES * T, s));
        return RELAX_OK;
    }
    if (!scores) {
        RELAX_TRY(ensure_buf(h, h->scratch, sizeof(uint32_t) * (size_t)npatch * T));
        scores = static_cast<uint32_t*>(h->scratch.p);
    }
    const bool al = ((W * 3) % 16 == 0) && (item_stride % 16 == 0) && aligned16(a) && (!pair || aligned16(b));
    const size_t lds = sizeof(uint32_t) * pw;
    const double bytes = (pair ? 2.0 : 1.0) * ph * P * (double)pw * PATCH_ROW_BYTES * T + 4.0 * npatch * T;
    int span;
    RELAX_TRY(prof_begin(h, s, 1, bytes, &span));
    dim3 grid(ph, T);
    const int nchunks = pw * 3;
    const int passes = (nchunks + 511) / 512;
    const int ablock = (((nchunks + passes - 1) / passes + 63) / 64) * 64;  // 1080p: 360 chunks -> 384 lanes, 1 pass
    if (pair) {
        if (al) hipLaunchKernelGGL(patch_score_aligned<true>, grid, ablock, lds, s, a, b, item_stride, W, pw, ph, scores);
        else hipLaunchKernelGGL(patch_score_generic<true>, grid, 256, lds, s, a, b, item_stride, W, pw, ph, scores);
    } else {
        if (al) hipLaunchKernelGGL(patch_score_aligned<false>, grid, ablock, lds, s, a, b, item_stride, W, pw, ph, scores);
        else hipLaunchKernelGGL(patch_score_generic<false>, grid, 256, lds, s, a, b, item_stride, W, pw, ph, scores);
    }
    RELAX_TRY(prof_end(h, s, span));
    hipLaunchKernelGGL(select_topn, dim3(T), SEL_THREADS, 0, s, scores, npatch, pw, top_n, positions, counts);
    dim3 ggrid((RELAX_TOP_N * P * 3 + 255) / 256, T);
    if (frag_a) {
        if (al) hipLaunchKernelGGL((gather_fragment<0, true>), ggrid, 256, 0, s, a, b, item_stride, H, W, positions, counts, frag_a);
        else hipLaunchKernelGGL((gather_fragment<0, false>), ggrid, 256, 0, s, a, b, item_stride, H, W, positions, counts, frag_a);
    }
    if (frag_diff && pair) {
        if (al) hipLaunchKernelGGL((gather_fragment<1, true>), ggrid, 256, 0, s, a, b, item_stride, H, W, positions, counts, frag_diff);
        else hipLaunchKernelGGL((gather_fragment<1, false>), ggrid, 256, 0, s, a, b, item_stride, H, W, positions, counts, frag_diff);
    }
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_fragment_pairs(relax_handle* h, const uint8_t* orig, const uint8_t* next, int64_t pair_stride, int T,
                         int H, int W, int top_n, int32_t* positions, int32_t* counts, uint8_t* ori_frag,
                         uint8_t* diff_frag, uint32_t* scores, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    return fragment_common(h, true, orig, next, pair_stride, T, H, W, top_n, positions, counts, ori_frag, diff_frag,
                           scores, static_cast<hipStream_t>(stream));
}

int relax_fragment_image(relax_handle* h, const uint8_t* image, int64_t item_stride, int T, int H, int W, int top_n,
                         int32_t* positions, int32_t* counts, uint8_t* frag, uint32_t* scores, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    return fragment_common(h, false, image, nullptr, item_stride, T, H, W, top_n, positions, counts, frag, nullptr,
                           scores, static_cast<hipStream_t>(stream));
}

int relax_gather_patches(relax_handle* h, const uint8_t* image, int64_t item_stride, int T, int H, int W,
                         const int32_t* positions, const int32_t* counts, uint8_t* frag, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, image && positions && counts && frag, "relax_gather_patches: NULL pointer");
    RELAX_REQUIRE(h, T > 0 && H > 0 && W > 0, "relax_gather_patches: bad shape");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool al = ((W * 3) % 16 == 0) && (item_stride % 16 == 0) && aligned16(image);
    dim3 ggrid((RELAX_TOP_N * P * 3 + 255) / 256, T);
    if (al) hipLaunchKernelGGL((gather_fragment<0, true>), ggrid, 256, 0, s, image, image, item_stride, H, W, positions, counts, frag);
    else hipLaunchKernelGGL((gather_fragment<0, false>), ggrid, 256, 0, s, image, image, item_stride, H, W, positions, counts, frag);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int relax_merge_fragments(relax_handle* h, const uint8_t* a, const uint8_t* b, uint8_t* out, int64_t n_bytes,
                          relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, a && b && out && n_bytes >= 0, "relax_merge_fragments: bad arguments");
    if (n_bytes == 0) return RELAX_OK;
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    const bool vec = aligned16(a) && aligned16(b) && aligned16(out);
    const int64_t threads = vec ? (n_bytes / 16 > 16 ? n_bytes / 16 : 16) : n_bytes;
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)((threads + 255) / 256)), 256, 0, static_cast<hipStream_t>(stream), a,
                       b, out, n_bytes, vec);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

}  // extern "C"
