// The non-contraction kernels of the backbones: LayerNorm (wavefront reductions), fused
// single-tile attention for 197 tokens x 64-d heads (fp32 MFMA, softmax in registers),
// BN+ReLU+max-pool, deterministic global-average-pool, NHWC->NCHW tap export.
#include "relax_internal.h"
#include "sp3.h"
#include "h2.h"

namespace relax {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- LayerNorm: one 64-lane wave per row, row held in registers (dim <= 768) ---------------------------
// SP3: the normalised row leaves as split planes (bf16 hi + mid + lo, gemm_x6.hip) for the bf16x6 contraction that
// consumes it; the arithmetic before the store is the same, so the fp32 value that is split is the one the fp32 path stores.
// MODE 2: as two fp16 planes of value * h2_scale (csrc/h2.h; h2_scale = the static power of two of this tensor) for the f16x2 contraction.
template <int MODE>   // 0: fp32, 1: split planes, 2: two fp16 planes
__global__ __launch_bounds__(256) void layernorm_rows(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, void* __restrict__ yv, int rows,
                                                      int dim, float eps, float h2_scale) {
    constexpr bool SP3 = MODE == 1;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int nvec = dim >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * dim);
    float4 v[3];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = lane + 64 * j;
        v[j] = i < nvec ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = lane + 64 * j;
        if (i < nvec) {
            const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = lane + 64 * j;
        if (i < nvec) {
            const float4 gg = g4[i], bb = b4[i];
            float4 o;
            o.x = (v[j].x - mean) * rstd * gg.x + bb.x;
            o.y = (v[j].y - mean) * rstd * gg.y + bb.y;
            o.z = (v[j].z - mean) * rstd * gg.z + bb.z;
            o.w = (v[j].w - mean) * rstd * gg.w + bb.w;
            if (SP3) {
                // values k = 4i .. 4i+3 = 8 bytes of each plane of chunk k / 16.  Lanes 2j and 2j+1 hold the two halves of a
                // 16-byte unit: after one exchange the even lane stores the hi and lo units, the odd lane the mid unit (three
                // 16-byte stores per pair instead of six 8-byte ones: 8-byte stores run at 0.5-0.7 of the 16-byte rate)
                sp3_u32x2 hi, mid, lo;
                split3_x4((sp3_f32x4){o.x, o.y, o.z, o.w}, hi, mid, lo);
                const bool odd = lane & 1;
                // what the partner needs from this lane: even lane gives its mid, odd lane gives its hi and lo
                const unsigned s0 = __shfl_xor(odd ? hi.x : mid.x, 1), s1 = __shfl_xor(odd ? hi.y : mid.y, 1);
                const unsigned t0 = __shfl_xor(lo.x, 1), t1 = __shfl_xor(lo.y, 1);
                char* d = static_cast<char*>(yv) + (int64_t)row * dim * 6 + (i >> 2) * 96 + ((i >> 1) & 1) * 16;
                if (!odd) {
                    *reinterpret_cast<sp3_u32x4*>(d) = (sp3_u32x4){hi.x, hi.y, s0, s1};            // hi: own 4 values, partner's 4
                    *reinterpret_cast<sp3_u32x4*>(d + 64) = (sp3_u32x4){lo.x, lo.y, t0, t1};       // lo
                } else {
                    *reinterpret_cast<sp3_u32x4*>(d + 32) = (sp3_u32x4){s0, s1, mid.x, mid.y};     // mid: partner's 4 values, own 4
                }
            } else if (MODE == 2) {
                // values k = 4i .. 4i+3 = 8 bytes of each plane of chunk k / 16: lanes 2j and 2j+1 hold the two halves of a 16-byte unit;
                // after one exchange the even lane stores the hi unit, the odd lane the lo unit (16-byte stores, as above)
                h2_u32x2 hi, lo;
                split2_x4((h2_f32x4){o.x, o.y, o.z, o.w} * h2_scale, hi, lo);
                const bool odd = lane & 1;
                const unsigned s0 = __shfl_xor(odd ? hi.x : lo.x, 1), s1 = __shfl_xor(odd ? hi.y : lo.y, 1);
                char* d = static_cast<char*>(yv) + (int64_t)row * dim * 4 + (i >> 2) * kH2ChunkBytes + ((i >> 1) & 1) * 16;
                if (!odd) *reinterpret_cast<h2_u32x4*>(d) = (h2_u32x4){hi.x, hi.y, s0, s1};            // hi: own 4 values, partner's 4
                else *reinterpret_cast<h2_u32x4*>(d + 32) = (h2_u32x4){s0, s1, lo.x, lo.y};            // lo: partner's 4 values, own 4
            } else {
                reinterpret_cast<float4*>(static_cast<float*>(yv) + (int64_t)row * dim)[i] = o;
            }
        }
    }
}

int launch_layernorm(relax_handle* h, const float* x, const float* g, const float* b, float* y, int rows, int dim,
                     float eps, hipStream_t s) {
    RELAX_REQUIRE(h, dim % 4 == 0 && dim > 0 && dim <= 768, "layernorm: dim=%d must be a multiple of 4, <= 768", dim);
    RELAX_REQUIRE(h, rows > 0, "layernorm: rows=%d", rows);
    hipLaunchKernelGGL(layernorm_rows<0>, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, rows, dim, eps, 1.f);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_layernorm_sp3(relax_handle* h, const float* x, const float* g, const float* b, void* y_sp3, int rows, int dim,
                         float eps, hipStream_t s) {
    RELAX_REQUIRE(h, dim % 16 == 0 && dim > 0 && dim <= 768, "layernorm_sp3: dim=%d must be a multiple of 16, <= 768", dim);
    RELAX_REQUIRE(h, rows > 0, "layernorm: rows=%d", rows);
    hipLaunchKernelGGL(layernorm_rows<1>, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y_sp3, rows, dim, eps, 1.f);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_layernorm_h2(relax_handle* h, const float* x, const float* g, const float* b, void* y_h2, float scale, int rows, int dim,
                        float eps, hipStream_t s) {
    RELAX_REQUIRE(h, dim % 16 == 0 && dim > 0 && dim <= 768, "layernorm_h2: dim=%d must be a multiple of 16, <= 768", dim);
    RELAX_REQUIRE(h, rows > 0 && scale > 0.f, "layernorm_h2: rows=%d scale=%g", rows, (double)scale);
    hipLaunchKernelGGL(layernorm_rows<2>, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y_h2, rows, dim, eps, scale);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- attention: persistent workgroups walk the (image, head) items; 7 waves = 7 query tiles of 32 -----------------
// S^T = K Q^T is computed with the KEY on the MFMA row (register) axis and the QUERY on the lane, so each lane owns one
// query's scores: softmax is register-local plus one lane^32 exchange, and the probabilities already sit in the
// A-operand layout of the P V product (no LDS round trip).  K and V share ONE LDS buffer (61 KB): K is resident during
// the score phase while V streams into registers, V replaces K for the P V phase while the next item's K streams into
// registers - the global traffic (150 KB per item, ~6 us at the per-CU load rate) hides under the MFMA phases instead
// of preceding them.
constexpr int NTOK = 197;
constexpr int KTILES = 7;         // 224 = 7 * 32 padded keys
constexpr int KPAD = KTILES * 32;
constexpr int KV_LD = 68;         // 64 + 4: conflict-free ds_read_b128 of K rows
constexpr int ATT_THREADS = KTILES * 64;
constexpr size_t ATT_LDS = sizeof(float) * (KPAD * KV_LD + KTILES * 32);
constexpr int STAGE_ITERS = KPAD * 16 / ATT_THREADS;   // 8 float4 per thread per matrix
static_assert(KPAD * 16 % ATT_THREADS == 0, "K/V staging must divide evenly over the workgroup");

typedef float att_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(ATT_THREADS) void attention_197x64(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 int heads, int total_items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* KV = smem;
    float* rowsum = smem + KPAD * KV_LD;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int dim = heads * 64;
    const int ld = 3 * dim;
    const att_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int item = blockIdx.x;
    if (item >= total_items) return;   // workgroup-uniform

    att_f32x4 kv[STAGE_ITERS], qf[8];
#define ATT_LOAD_KV(item_, col_)                                                                                  \
    {                                                                                                             \
        const float* b_ = qkv + (int64_t)((item_) / heads) * NTOK * ld + ((item_) % heads) * 64 + (col_);        \
        _Pragma("unroll") for (int it = 0; it < STAGE_ITERS; ++it) {                                              \
            const int idx = it * ATT_THREADS + tid;                                                               \
            const int row = idx >> 4, c = idx & 15;                                                               \
            kv[it] = zero4;                                                                                       \
            if (row < NTOK) kv[it] = *reinterpret_cast<const att_f32x4*>(b_ + (int64_t)row * ld + c * 4);         \
        }                                                                                                         \
    }
#define ATT_STORE_KV()                                                                                            \
    _Pragma("unroll") for (int it = 0; it < STAGE_ITERS; ++it) {                                                  \
        const int idx = it * ATT_THREADS + tid;                                                                   \
        *reinterpret_cast<att_f32x4*>(KV + (idx >> 4) * KV_LD + (idx & 15) * 4) = kv[it];                          \
    }
#define ATT_LOAD_Q(item_)                                                                                         \
    {                                                                                                             \
        const int qrow_ = wave * 32 + li;                                                                         \
        const float* q_ = qkv + (int64_t)((item_) / heads) * NTOK * ld + ((item_) % heads) * 64 +                 \
                          (int64_t)(qrow_ < NTOK ? qrow_ : NTOK - 1) * ld + 4 * half;                              \
        _Pragma("unroll") for (int q8 = 0; q8 < 8; ++q8)                                                          \
            qf[q8] = *reinterpret_cast<const att_f32x4*>(q_ + 8 * q8) * (0.125f * 1.44269504088896341f);         \
    }

    ATT_LOAD_KV(item, dim);
    ATT_LOAD_Q(item);
    ATT_STORE_KV();
    __syncthreads();

    while (true) {
        ATT_LOAD_KV(item, 2 * dim);   // V of this item: in flight during the score phase
        floatx16 sacc[KTILES];
#pragma unroll
        for (int kt = 0; kt < KTILES; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
            const float* kp = KV + (kt * 32 + li) * KV_LD + 4 * half;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                const att_f32x4 kf = *reinterpret_cast<const att_f32x4*>(kp + 8 * q8);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[q8].x, sacc[kt], 0, 0, 0);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[q8].y, sacc[kt], 0, 0, 0);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[q8].z, sacc[kt], 0, 0, 0);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[q8].w, sacc[kt], 0, 0, 0);
            }
        }
        // sacc[kt][r] = score(query li, key kt*32 + (r&3) + 8*(r>>2) + 4*half)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= NTOK) sacc[kt][r] = -INFINITY;
                mx = fmaxf(mx, sacc[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // scores carry log2(e) (folded into the Q scale with head_dim^-0.5), so exp() is one v_exp_f32
                const float e = __builtin_amdgcn_exp2f(sacc[kt][r] - mx);
                sacc[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        if (half == 0) rowsum[wave * 32 + li] = sum;
        __syncthreads();          // every wave is done with K
        ATT_STORE_KV();           // V takes its place
        __syncthreads();
        const int next = item + gridDim.x;
        if (next < total_items) ATT_LOAD_KV(next, dim);   // K of the next item: in flight during the P V phase

        floatx16 oacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < KTILES; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v0 = KV[key * KV_LD + li];
                const float v1 = KV[key * KV_LD + 32 + li];
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[kt][r], v0, oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[kt][r], v1, oacc[1], 0, 0, 0);
            }
        // oacc[dt][r] = O(query wave*32 + (r&3)+8*(r>>2)+4*half, d = dt*32 + li)
        float* ob = out + (int64_t)(item / heads) * NTOK * dim + (item % heads) * 64 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qt = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int q = wave * 32 + qt;
            if (q < NTOK) {
                const float inv = 1.0f / rowsum[wave * 32 + qt];
                ob[(int64_t)q * dim] = oacc[0][r] * inv;
                ob[(int64_t)q * dim + 32] = oacc[1][r] * inv;
            }
        }
        if (next >= total_items) break;
        ATT_LOAD_Q(next);
        __syncthreads();          // every wave is done with V (and with rowsum)
        ATT_STORE_KV();           // next item's K
        __syncthreads();
        item = next;
    }
#undef ATT_LOAD_KV
#undef ATT_STORE_KV
#undef ATT_LOAD_Q
}

int launch_attention(relax_handle* h, const float* qkv, float* out, int Nimg, int heads, hipStream_t s) {
    RELAX_REQUIRE(h, Nimg > 0 && heads > 0, "attention: Nimg=%d heads=%d", Nimg, heads);
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_197x64),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_LDS));
        attr_set[h->device] = true;
    }
    const int total = Nimg * heads;
    // one persistent workgroup (7 waves, ~250 VGPRs) per CU
    const int grid = total < 256 ? total : 256;
    hipLaunchKernelGGL(attention_197x64, dim3(grid), dim3(ATT_THREADS), ATT_LDS, s, qkv, out, heads, total);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- relu(bn(x)) then 3x3 / stride 2 / pad 1 max-pool, NHWC, 4 channels per thread ---------------------------
template <bool SP3>   // SP3: the pooled map leaves as split planes (bf16 hi + mid + lo) for the bf16x6 convolutions of layer1
__global__ __launch_bounds__(256) void bn_relu_maxpool_nhwc(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, void* __restrict__ yv,
                                                            int Nimg, int H, int W, int C, unsigned* __restrict__ block_max) {
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const int64_t total = (int64_t)Nimg * Ho * Wo * C4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;   // (block_max: the launcher makes total a multiple of the block - no thread leaves before the barrier below)
    const int c4 = (int)(i % C4);
    int64_t pix = i / C4;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
    const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const float4 v = reinterpret_cast<const float4*>(x + (((int64_t)n * H + iy) * W + ix) * C)[c4];
            m.x = fmaxf(m.x, fmaxf(v.x * sc.x + sh.x, 0.f));
            m.y = fmaxf(m.y, fmaxf(v.y * sc.y + sh.y, 0.f));
            m.z = fmaxf(m.z, fmaxf(v.z * sc.z + sh.z, 0.f));
            m.w = fmaxf(m.w, fmaxf(v.w * sc.w + sh.w, 0.f));
        }
    }
    const int64_t opix = ((int64_t)n * Ho + oy) * Wo + ox;
    if (SP3) {
        store_sp3_x4(static_cast<char*>(yv) + opix * C * 6, 4 * c4, (sp3_f32x4){m.x, m.y, m.z, m.w});
    } else {
        reinterpret_cast<float4*>(static_cast<float*>(yv) + opix * C)[c4] = m;
    }
    if (block_max) {
        // the largest output of this block (outputs are >= 0: the bits order like the values), one plain store per block: the blocks of an
        // image are reduced by image_max_of_blocks below - no atomics, no order dependence
        __shared__ unsigned wmax[4];
        float mx = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = __float_as_uint(mx);
        __syncthreads();
        if (threadIdx.x == 0) block_max[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    }
}

// amax[n] = the largest of image n's `per_image` block maxima (bits of non-negative floats)
__global__ __launch_bounds__(64) void image_max_of_blocks(const unsigned* __restrict__ block_max, int per_image, unsigned* __restrict__ amax) {
    const unsigned* b = block_max + (int64_t)blockIdx.x * per_image;
    unsigned m = 0u;
    for (int i = threadIdx.x; i < per_image; i += 64) m = max(m, b[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (threadIdx.x == 0) amax[blockIdx.x] = m;
}

int launch_bn_relu_maxpool(relax_handle* h, const float* x, const float* scale, const float* shift, float* y,
                           int Nimg, int H, int W, int C, hipStream_t s) {
    RELAX_REQUIRE(h, Nimg > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "bn_relu_maxpool: bad shape");
    const int64_t total = (int64_t)Nimg * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(bn_relu_maxpool_nhwc<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, scale, shift, y,
                       Nimg, H, W, C, nullptr);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_bn_relu_maxpool_f32(relax_handle* h, const float* x, const float* scale, const float* shift, float* y, int Nimg, int H, int W, int C,
                               hipStream_t s, unsigned* amax_out, unsigned* block_ws) {
    RELAX_REQUIRE(h, Nimg > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "bn_relu_maxpool_f32: bad shape");
    const int64_t total = (int64_t)Nimg * (H / 2) * (W / 2) * (C / 4);
    const int64_t per_image = (int64_t)(H / 2) * (W / 2) * (C / 4);
    RELAX_REQUIRE(h, !amax_out || (block_ws && per_image % 256 == 0), "bn_relu_maxpool_f32: per-image maxima need whole blocks per image");
    hipLaunchKernelGGL(bn_relu_maxpool_nhwc<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, scale, shift, y, Nimg, H, W, C,
                       amax_out ? block_ws : nullptr);
    if (amax_out) hipLaunchKernelGGL(image_max_of_blocks, dim3(Nimg), dim3(64), 0, s, block_ws, (int)(per_image / 256), amax_out);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_bn_relu_maxpool_sp3(relax_handle* h, const float* x, const float* scale, const float* shift, void* y_sp3,
                               int Nimg, int H, int W, int C, hipStream_t s, unsigned* amax_out, unsigned* block_ws) {
    RELAX_REQUIRE(h, Nimg > 0 && H % 2 == 0 && W % 2 == 0 && C % 16 == 0, "bn_relu_maxpool_sp3: bad shape");
    const int64_t total = (int64_t)Nimg * (H / 2) * (W / 2) * (C / 4);
    const int64_t per_image = (int64_t)(H / 2) * (W / 2) * (C / 4);
    // amax_out [Nimg]: the per-image maximum of the outputs (block_ws: total / 256 words of scratch); a block must lie inside one image
    RELAX_REQUIRE(h, !amax_out || (block_ws && per_image % 256 == 0), "bn_relu_maxpool_sp3: per-image maxima need whole blocks per image");
    hipLaunchKernelGGL(bn_relu_maxpool_nhwc<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, scale, shift, y_sp3,
                       Nimg, H, W, C, amax_out ? block_ws : nullptr);
    if (amax_out) hipLaunchKernelGGL(image_max_of_blocks, dim3(Nimg), dim3(64), 0, s, block_ws, (int)(per_image / 256), amax_out);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- global average pool, deterministic two-stage ---------------------------------------------------------------
// stage 1: grid (C/64, Nimg, S); a workgroup owns 64 channels of one image and 1/S of the pixels;
//          16 lanes x float4 cover the channels (256 B coalesced), 16 lane-groups stride the pixels;
//          fixed-order LDS reduction -> partial[n][s][c].
// stage 2: sums the S partials in order and divides by HW.
__global__ __launch_bounds__(256) void gap_partial(const float* __restrict__ x, float* __restrict__ partial, int HW,
                                                   int C, int S) {
    __shared__ float4 red[256];
    const int cc = blockIdx.x, n = blockIdx.y, sp = blockIdx.z;
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int per = (HW + S - 1) / S;
    const int lo = sp * per;
    const int hi = lo + per < HW ? lo + per : HW;
    const float* xb = x + (int64_t)n * HW * C + cc * 64 + l16 * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = lo + grp; p < hi; p += 16) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * C);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < 16) {
        float4 t = red[threadIdx.x];
        for (int g = 1; g < 16; ++g) {
            const float4 u = red[g * 16 + threadIdx.x];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        reinterpret_cast<float4*>(partial + ((int64_t)n * S + sp) * C + cc * 64)[threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void gap_finish(const float* __restrict__ partial, float* __restrict__ out, int Nimg,
                                                  int HW, int C, int S, int64_t out_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Nimg * C) return;
    const int n = (int)(i / C), c = (int)(i % C);
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += partial[((int64_t)n * S + s) * C + c];
    out[(int64_t)n * out_stride + c] = t / (float)HW;
}

// stage 2 of the spatial mean fused into the bf16x6 epilogue: out[n, c] = (sum of the G group sums of image n, in order) / HW.
// The loads of 16 groups are in flight together, the additions keep the order (one load per addition left the kernel waiting on
// memory latency G times: 100 us for 100 MB of group sums).
__global__ __launch_bounds__(256) void gap_groups_finish(const float* __restrict__ groups, float* __restrict__ out, int Nimg,
                                                         int G, int HW, int C, int64_t out_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Nimg * C) return;
    const int n = (int)(i / C), c = (int)(i % C);
    const float* g = groups + (int64_t)n * G * C + c;
    float t = 0.f;
    int k = 0;
    for (; k + 16 <= G; k += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = g[(int64_t)(k + j) * C];
#pragma unroll
        for (int j = 0; j < 16; ++j) t += v[j];
    }
    for (; k < G; ++k) t += g[(int64_t)k * C];
    out[(int64_t)n * out_stride + c] = t / (float)HW;
}

// group = rows per group sum: 16, or 4 where HW is a multiple of 4 only (the 14x14 maps)
int launch_gap_groups_finish(relax_handle* h, const float* groups, float* out, int Nimg, int HW, int C, int64_t out_stride,
                             hipStream_t s) {
    RELAX_REQUIRE(h, HW % 4 == 0 && Nimg > 0 && C > 0, "gap_groups_finish: bad shape HW=%d C=%d", HW, C);
    const int group = HW % 16 == 0 ? 16 : 4;
    hipLaunchKernelGGL(gap_groups_finish, dim3((unsigned)(((int64_t)Nimg * C + 255) / 256)), dim3(256), 0, s, groups, out, Nimg,
                       HW / group, HW, C, out_stride);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

constexpr int GAP_MAX_SPLIT = 16;

int launch_gap_ws(relax_handle* h, const float* x, float* out, int Nimg, int HW, int C, int64_t out_stride,
                  float* partial_ws, hipStream_t s) {
    RELAX_REQUIRE(h, C % 64 == 0 && HW > 0 && Nimg > 0, "gap: bad shape HW=%d C=%d", HW, C);
    // the split depends on the layer geometry only (never on the batch), so a frame's features are bit-identical
    // whatever batch it travels in
    int S = HW / 196;
    if (S > GAP_MAX_SPLIT) S = GAP_MAX_SPLIT;
    if (S < 1) S = 1;
    hipLaunchKernelGGL(gap_partial, dim3(C / 64, Nimg, S), dim3(256), 0, s, x, partial_ws, HW, C, S);
    hipLaunchKernelGGL(gap_finish, dim3((unsigned)(((int64_t)Nimg * C + 255) / 256)), dim3(256), 0, s, partial_ws, out,
                       Nimg, HW, C, S, out_stride);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_gap(relax_handle* h, const float* x, float* out, int Nimg, int HW, int C, int64_t out_stride,
               hipStream_t s) {
    // stand-alone form: partial sums live in the stage-A scratch buffer
    RELAX_TRY(ensure_buf(h, h->scratch, sizeof(float) * (size_t)Nimg * GAP_MAX_SPLIT * C));
    return launch_gap_ws(h, x, out, Nimg, HW, C, out_stride, static_cast<float*>(h->scratch.p), s);
}

// ---- [N,HW,C] -> [N,C,HW] (tap export for the process_video_frame dict) -----------------------------------------
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int HW,
                                                           int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < C) tile[j][tx] = x[((int64_t)n * HW + p) * C + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (p < HW && c < C) y[((int64_t)n * C + c) * HW + p] = tile[tx][j];
    }
}

int launch_nhwc_to_nchw(relax_handle* h, const float* x, float* y, int Nimg, int HW, int C, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, Nimg), dim3(256), 0, s, x, y, HW, C);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- per-clip mean over frames (src/demo_test.py:171-175; src/data_processing/extract_npy2mat.py:121-126) --------------------
// dst[s, dst_col0 + c] = mean over the rows [row0 + off[s], row0 + off[s+1]) of src[:, c]: rows summed in order, then one
// division (numpy's mean over axis 0 does the same), so a clip's vector does not depend on which batch it travelled in
constexpr int kSegChunk = 64;
struct SegOffsets {   // passed by value: nothing to copy to the device, and a captured launch (HIP graph) keeps its own copy
    int32_t off[kSegChunk + 1];
};

__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ src, int64_t src_stride, int ncols, int row0,
                                                           const SegOffsets so, int seg0, float* __restrict__ dst,
                                                           int64_t dst_stride, int dst_col0) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    const int sgm = seg0 + blockIdx.y;
    const int lo = so.off[blockIdx.y], hi = so.off[blockIdx.y + 1];
    const float* p = src + (int64_t)(row0 + lo) * src_stride + c;
    float t = 0.f;
    for (int r = lo; r < hi; ++r, p += src_stride) t += *p;
    dst[(int64_t)sgm * dst_stride + dst_col0 + c] = t / (float)(hi - lo);
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_segment_mean(relax_handle* h, const float* src, int64_t src_stride, int ncols, int row0, const int32_t* seg_offsets,
                       int nseg, float* dst, int64_t dst_stride, int dst_col0, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, src && seg_offsets && dst && ncols > 0 && nseg > 0 && row0 >= 0, "relax_segment_mean: bad arguments");
    for (int i = 0; i < nseg; ++i)
        RELAX_REQUIRE(h, seg_offsets[i + 1] > seg_offsets[i] && seg_offsets[0] >= 0, "relax_segment_mean: segment %d is empty or out of order", i);
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    for (int s0 = 0; s0 < nseg; s0 += kSegChunk) {
        const int n = nseg - s0 < kSegChunk ? nseg - s0 : kSegChunk;
        SegOffsets so;
        for (int i = 0; i <= n; ++i) so.off[i] = seg_offsets[s0 + i];
        hipLaunchKernelGGL(segment_mean_kernel, dim3((ncols + 255) / 256, n), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                           src_stride, ncols, row0, so, s0, dst, dst_stride, dst_col0);
    }
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int relax_op_layernorm(relax_handle* h, const float* x, const float* gamma, const float* beta, float* y, int rows,
                       int dim, float eps, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, x && gamma && beta && y, "relax_op_layernorm: NULL operand");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    return launch_layernorm(h, x, gamma, beta, y, rows, dim, eps, static_cast<hipStream_t>(stream));
}

int relax_op_attention(relax_handle* h, const float* qkv, float* out, int Nimg, int heads, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, qkv && out, "relax_op_attention: NULL operand");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    if (h->gemm.precision == 3 && h->gemm.att_h2) return launch_attention_h2_op(h, qkv, out, Nimg, heads, static_cast<hipStream_t>(stream));
    if (h->gemm.precision >= 2) return launch_attention_x6(h, qkv, out, nullptr, Nimg, heads, static_cast<hipStream_t>(stream));
    return launch_attention(h, qkv, out, Nimg, heads, static_cast<hipStream_t>(stream));
}

int relax_op_bn_relu_maxpool(relax_handle* h, const float* x, const float* scale, const float* shift, float* y,
                             int Nimg, int H, int W, int C, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, x && scale && shift && y, "relax_op_bn_relu_maxpool: NULL operand");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    return launch_bn_relu_maxpool(h, x, scale, shift, y, Nimg, H, W, C, static_cast<hipStream_t>(stream));
}

int relax_op_gap(relax_handle* h, const float* x, float* out, int Nimg, int HW, int C, int64_t out_stride,
                 relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, x && out, "relax_op_gap: NULL operand");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    return launch_gap(h, x, out, Nimg, HW, C, out_stride, static_cast<hipStream_t>(stream));
}

}  // extern "C"
