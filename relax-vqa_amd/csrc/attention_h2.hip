// Attention of the DINO ViT (197 tokens, 64-d heads) under "f16x2": softmax(q k^T / 8) v   (src/extractor/visualise_vit_layer.py:93-106)
// with q, k, v taken as the two fp16 planes the qkv GEMM's epilogue writes (csrc/h2.h: x * s = hi + lo, one static power-of-two scale
// for the whole qkv tensor from the Cauchy-Schwarz bound behind LayerNorm) and both contractions as three partial products
// (lo hi, hi lo, hi hi) on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  Against attention_x6.hip (three bf16 planes, six products):
//   * half the matrix instructions;
//   * NO conversion passes: the K and V rows of an (image, head) item are 256 contiguous bytes of planes per token and go global -> LDS by
//     LDS-DMA straight into the images the MFMA fragments are read from (attention_x6 stages fp32 rows and splits / transposes them LDS -> LDS:
//     17 % of an item), both images resident (2 x 56 KB), two barriers per item instead of four;
//   * V stays [key][d]: the A operand of O^T = V^T P^T (8 consecutive KEYS of one d per lane) is read with ds_read_b64_tr_b16, the
//     transposing LDS read of gfx950 (two per fragment), so no transposed copy exists.
// Accuracy: the error of these sums is the fp32 accumulation's, not the 22-bit operands' - with the rest of the network exact, ViT-B tokens
// come out 5.9e-8 (regular weights) / 1.2e-6 (logits of +-20) from fp64 with this arithmetic against 2.0e-7 / 3.2e-6 for plain fp32
// attention (tools/attn_emul.py); the device tests hold the whole network to the same gates as before (tests/test_gpu_h2.py).
// Structure as attention_x6: one persistent workgroup of 7 waves per CU walks the items; wave w owns queries 32 w .. 32 w + 31;
// S^T[key, query] = K Q^T puts a query on a lane (softmax in registers + one lane^32 exchange) and leaves the probabilities in the
// B-operand layout of the second product.
// LDS images: [224 keys][256 B = 4 chunks of 16 d x (16 fp16 hi | 16 fp16 lo)]; the sixteen 16-byte units of a row are XOR-permuted by
// f(key & 15) (a bit permutation: b0 = k2, b1 = k0, b2 = k3, b3 = k1), applied on the SOURCE offset of the DMA: the 16 lanes of a
// ds_read_b128 group (keys {0-3, 12-15, 20-27} ...) and the 32 lanes of a transposed read (4 keys x 2 chunks x 4 column groups) then
// hit distinct banks.  Rows 197 .. 223 lie beyond the buffer resource and arrive as zeros with every item.
// Measured and not kept: issue priority for the first wave of each SIMD-sharing pair (waves w, w + 4), so that its softmax would run under
// the partner's MFMAs: 633 against 625 us per layer and 1024 images.  The 2^14 of the probability planes folded into the exponent: an
// argument near 14 has an ulp of 1e-6 and nearly uniform rows lose a digit (the device test caught it).
#include "relax_internal.h"
#include "h2.h"

namespace relax {

typedef float ah_floatx16 __attribute__((ext_vector_type(16)));
typedef float ah_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ah_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ah_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ah_u32x2 __attribute__((ext_vector_type(2)));
typedef short ah_i16x4 __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int AH_NTOK = 197;
constexpr int AH_KT = 7;                    // key tiles of 32 = query tiles = waves
constexpr int AH_KPAD = AH_KT * 32;         // 224
[[maybe_unused]] constexpr int AH_VC = (AH_NTOK + 15) / 16;  // 13 steps of 16 keys hold a real key
constexpr int AH_THREADS = AH_KT * 64;
constexpr int AH_ROW = 256;                 // bytes of one key row of an image
constexpr int AH_IMG = AH_KPAD * AH_ROW;    // 57344
constexpr int AH_LDS = 2 * AH_IMG;          // K image, V image
constexpr int AH_PPW = AH_IMG / 1024 / AH_KT;   // DMA pieces (4 rows) per wave and image: 8
static_assert(AH_PPW * AH_KT * 1024 == AH_IMG, "the pieces divide over the waves");

__device__ inline int ah_f(int k) {   // the unit permutation of row k (k & 15 matters)
    return ((k >> 2) & 1) | ((k & 1) << 1) | (((k >> 3) & 1) << 2) | (((k >> 1) & 1) << 3);
}
__device__ inline ah_f16x8 ah_frag(const ah_u32x4 v) { return __builtin_bit_cast(ah_f16x8, v); }

// dev_scalars (operator-level entry only): {alpha, out_mul} computed on the device from the tensor's measured maximum
__global__ __launch_bounds__(AH_THREADS) void attention_h2(const char* __restrict__ qkvp, char* __restrict__ out_h2, int heads, int total_items,
                                                           float alpha, float out_mul, const float* __restrict__ dev_scalars) {
#if __HIP_DEVICE_COMPILE__
    if (dev_scalars) {
        alpha = dev_scalars[0];
        out_mul = dev_scalars[1];
    }
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int dim = heads * 64;
    const int ldb = 3 * dim * 4;              // bytes of a token's qkv row of planes
    int item = blockIdx.x;
    if (item >= total_items) return;          // workgroup-uniform
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // K / V rows of an item -> image img_ (0 = K, 1 = V): 4 rows per DMA instruction, lane l fills position l & 15 of row 4 piece + (l >> 4)
    // with the source unit (l & 15) ^ f(row)
#define AH_DMA(item_, which_, img_)                                                                                  \
    {                                                                                                                \
        const char* b_ = qkvp + (int64_t)((item_) / heads) * AH_NTOK * ldb;                                          \
        __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b_), 0, AH_NTOK * ldb, 0x00020000); \
        const int col_ = ((which_) * dim + ((item_) % heads) * 64) * 4;                                              \
        _Pragma("unroll") for (int j = 0; j < AH_PPW; ++j) {                                                         \
            const int piece = wave_u * AH_PPW + j;                                                                   \
            const int row_ = piece * 4 + (lane >> 4);                                                                \
            const int src_ = (lane & 15) ^ ah_f(row_);                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(smem + (img_) * AH_IMG + piece * 1024), \
                                                     16, row_ * ldb + col_ + src_ * 16, 0, 0, 0);                    \
        }                                                                                                            \
    }
#define AH_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    // the planes of this lane's query: d = 16 s + 8 half .. + 7 for s = 0 .. 3, hi and lo (a clamped row for the padding queries)
    ah_u32x4 qh[4], ql[4];
#define AH_REQUEST_Q(item_)                                                                                          \
    {                                                                                                                \
        const int qrow_ = wave * 32 + li;                                                                            \
        const char* q_ = qkvp + ((int64_t)((item_) / heads) * AH_NTOK + (qrow_ < AH_NTOK ? qrow_ : AH_NTOK - 1)) * ldb + \
                         ((item_) % heads) * 256 + half * 16;                                                        \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                              \
            qh[s] = *reinterpret_cast<const ah_u32x4*>(q_ + s * 64);                                                 \
            ql[s] = *reinterpret_cast<const ah_u32x4*>(q_ + s * 64 + 32);                                            \
        }                                                                                                            \
    }

    // K fragment addresses: key kt * 32 + li, d step s, plane p: position ((4 s + 2 p + half) ^ f(li)) of the row = (((2 p + half) ^ f) << 4) ^ (s << 6)
    const int fk = ah_f(li);
    int kaddr[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) kaddr[s][p] = li * AH_ROW + ((((2 * p + half) ^ fk) << 4) ^ (s << 6));
    // V^T fragment addresses (transposed reads): lane = 16 g + 4 q + p; half = g >> 1, chunk = 2 dt + (g & 1); block row q = key
    // 16 c + 4 half + q (+ 8 for the second read of a fragment), columns 4 p .. 4 p + 3 of the chunk's 16 d of one plane
    int vaddr[2][2][2];   // [read j][dt][plane], key step c adds 4096 c
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ko = 4 * (g >> 1) + q + 8 * j;
            const int fv = ah_f(ko);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int u = 4 * (2 * dt + (g & 1)) + 2 * pl + (p >> 1);
                    vaddr[j][dt][pl] = AH_IMG + ko * AH_ROW + ((u ^ fv) << 4) + 8 * (p & 1);
                }
        }
    }

    AH_DMA(item, 1, 0);
    AH_REQUEST_Q(item);
    AH_WAIT_DMA();
    __syncthreads();

    while (true) {
        AH_DMA(item, 2, 1);   // the V rows: in flight during the score phase
        // ---- scores: S^T tile kt = K[kt] Q^T, 4 d-steps x 3 partial products (smallest first); the K fragments of group g + 1 are read under
        // the MFMAs of group g (a pinned software pipeline, as in attention_x6)
        ah_floatx16 sacc[AH_KT];
        ah_f16x8 kf[2][2];
#define AH_READ_K(buf_, g_)                                                                                          \
        {                                                                                                            \
            kf[buf_][0] = *reinterpret_cast<const ah_f16x8*>(smem + kaddr[(g_) & 3][0] + ((g_) >> 2) * (32 * AH_ROW)); \
            kf[buf_][1] = *reinterpret_cast<const ah_f16x8*>(smem + kaddr[(g_) & 3][1] + ((g_) >> 2) * (32 * AH_ROW)); \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }
        AH_READ_K(0, 0);
#pragma unroll
        for (int g = 0; g < 4 * AH_KT; ++g) {
            const int kt = g >> 2, s = g & 3, b = g & 1;
            if (g + 1 < 4 * AH_KT) AH_READ_K((g + 1) & 1, g + 1);
            if (s == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
            }
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[b][1], ah_frag(qh[s]), sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[b][0], ah_frag(ql[s]), sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[b][0], ah_frag(qh[s]), sacc[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef AH_READ_K
        // sacc[kt][r] * alpha = score(query li, key kt*32 + (r&3) + 8*(r>>2) + 4*half) * log2(e); only the last key tile holds padding keys.
        // e = exp2(score - max): one fma + one v_exp_f32 per score (alpha > 0, so the maximum is taken on the raw sums).  (The 2^14 of the
        // probability planes is NOT folded into the exponent: an argument near 14 has an ulp of 1e-6, and nearly uniform rows lost a digit.)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < AH_KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kt == AH_KT - 1 && kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= AH_NTOK) sacc[kt][r] = -INFINITY;
                mx = fmaxf(mx, sacc[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float shift = -mx * alpha;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < AH_KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], alpha, shift));
                sacc[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = out_mul / sum;
        AH_WAIT_DMA();
        __syncthreads();          // every wave is done with K, and the V rows have landed
        const int next = item + gridDim.x;
        if (next < total_items) AH_DMA(next, 1, 0);       // K rows of the next item: in flight during the output phase

        // ---- output: O^T[d, query] = V^T P^T over 13 steps of 16 keys; probabilities (<= 1) as planes of e * 2^14.  Registers 8*(c&1) .. +7 of
        // score tile c>>1 are keys 16c + 8*(j>>2) + 4*half + (j&3): the B fragment of step c; the A fragment = the same 8 keys of d = dt*32 + li
        ah_floatx16 oacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
        ah_u32x4 vf[2][2][2];   // [buffer][dt][plane]
        ah_u32x4 pp[2][2];      // [buffer][plane]
#define AH_TR(addr_, c_) __builtin_bit_cast(ah_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(                        \
                             (__attribute__((address_space(3))) ah_i16x4*)(smem + (addr_) + (c_) * (16 * AH_ROW))))
#define AH_READ_V(buf_, c_)                                                                                          \
        _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {          \
            const ah_u32x2 a_ = AH_TR(vaddr[0][dt][pl], c_), b_ = AH_TR(vaddr[1][dt][pl], c_);                       \
            vf[buf_][dt][pl] = (ah_u32x4){a_.x, a_.y, b_.x, b_.y};                                                   \
        }
#define AH_SPLIT_P(buf_, c_)                                                                                         \
        split2_x8((h2_f32x4){sacc[(c_) >> 1][8 * ((c_) & 1) + 0], sacc[(c_) >> 1][8 * ((c_) & 1) + 1],               \
                             sacc[(c_) >> 1][8 * ((c_) & 1) + 2], sacc[(c_) >> 1][8 * ((c_) & 1) + 3]} * 16384.f,    \
                  (h2_f32x4){sacc[(c_) >> 1][8 * ((c_) & 1) + 4], sacc[(c_) >> 1][8 * ((c_) & 1) + 5],               \
                             sacc[(c_) >> 1][8 * ((c_) & 1) + 6], sacc[(c_) >> 1][8 * ((c_) & 1) + 7]} * 16384.f,    \
                  pp[buf_][0], pp[buf_][1])
        AH_READ_V(0, 0);
        AH_SPLIT_P(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < AH_VC; ++c) {
            const int b = c & 1;
            if (c + 1 < AH_VC) {
                AH_READ_V(b ^ 1, c + 1);
                __builtin_amdgcn_sched_barrier(0);       // the reads go out first ...
                AH_SPLIT_P(b ^ 1, c + 1);                // ... the split's VALU interleaves with the MFMAs below
            }
            if (c == AH_KT && next < total_items) AH_REQUEST_Q(next);   // the next item's query planes (this item's are dead)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_frag(vf[b][dt][1]), ah_frag(pp[b][0]), oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_frag(vf[b][dt][0]), ah_frag(pp[b][1]), oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_frag(vf[b][dt][0]), ah_frag(pp[b][0]), oacc[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef AH_READ_V
#undef AH_SPLIT_P
#undef AH_TR
        // oacc[dt][r] = O(query li, d = dt*32 + (r&3) + 8*(r>>2) + 4*half) * (s 2^14): runs of 4 consecutive d; the lane^32 partner holds the
        // runs in between: after swapping two runs per tile each lane owns two units of 8 consecutive d (as attention_x6)
        // The planes leave through a buffer resource that covers the item's image exactly: the padding queries (rows >= 197) are out of its
        // range and their stores are dropped - no branch around the stores, so every wave issues all eight of them and the waits below can be
        // COUNTED: the eight youngest memory operations of the wave are these stores, everything older - the next item's K rows (DMA) and query
        // planes - has landed when vmcnt <= 8.  (With `if (q < 197)` around them the count had to be 0: every item ended by waiting for the
        // write acknowledgements of its own output.)
        const int q = wave * 32 + li;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_h2 + (int64_t)(item / heads) * AH_NTOK * ((int64_t)dim * 4), 0,
                                                                                AH_NTOK * dim * 4, 0x00020000);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            float own[16], got[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) own[r] = oacc[dt][r] * inv;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) got[4 * u + j] = __shfl_xor(half ? own[8 * u + j] : own[8 * u + 4 + j], 32);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ah_f32x4 lo4, hi4;
                if (half == 0) {
                    lo4 = (ah_f32x4){own[8 * u], own[8 * u + 1], own[8 * u + 2], own[8 * u + 3]};
                    hi4 = (ah_f32x4){got[4 * u], got[4 * u + 1], got[4 * u + 2], got[4 * u + 3]};
                } else {
                    lo4 = (ah_f32x4){got[4 * u], got[4 * u + 1], got[4 * u + 2], got[4 * u + 3]};
                    hi4 = (ah_f32x4){own[8 * u + 4], own[8 * u + 5], own[8 * u + 6], own[8 * u + 7]};
                }
                const int d0 = (item % heads) * 64 + dt * 32 + 16 * u + 8 * half;
                h2_u32x4 ph, pl;
                split2_x8(lo4, hi4, ph, pl);                                           // (the output scale is folded into inv)
                const int vo = q * (dim * 4) + (int)h2_offset(d0);
                __builtin_amdgcn_raw_buffer_store_b128(ph, rs_out, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(pl, rs_out, vo + 32, 0, 0);
            }
        }
        if (next >= total_items) break;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the next item's K rows and query planes have landed (the 8 stores above may still be in flight)
        __syncthreads();          // every wave is done with V, and the next K rows have landed
        item = next;
    }
#undef AH_DMA
#undef AH_WAIT_DMA
#undef AH_REQUEST_Q
#endif
}

// qkv_planes: fp16 planes [Nimg * 197][3 * dim * 4 B] of qkv * s_qkv; out_planes: [Nimg * 197][dim * 4 B] of the attention output * out_scale
int launch_attention_h2(relax_handle* h, const void* qkv_planes, float s_qkv, void* out_planes, float out_scale, int Nimg, int heads, hipStream_t s) {
    RELAX_REQUIRE(h, Nimg > 0 && heads > 0 && qkv_planes && out_planes, "attention_h2: Nimg=%d heads=%d", Nimg, heads);
    RELAX_REQUIRE(h, s_qkv > 0.f && s_qkv < 3.0e38f && out_scale > 0.f && out_scale < 3.0e38f, "attention_h2: bad scales");
    RELAX_REQUIRE(h, (int64_t)AH_NTOK * 3 * heads * 64 * 4 < 0x7ffffff0, "attention_h2: too many heads");
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_h2), hipFuncAttributeMaxDynamicSharedMemorySize, AH_LDS));
        attr_set[h->device] = true;
    }
    const int total = Nimg * heads;
    const int grid = total < 256 ? total : 256;   // one persistent workgroup (7 waves) per CU
    // log2(e) / 8 and the two operand scales folded into the logits; the probabilities' 2^14, V's scale and the output scale into 1 / sum
    const float alpha = (float)(0.125 * 1.44269504088896341 / ((double)s_qkv * (double)s_qkv));
    const float out_mul = (float)((double)out_scale / ((double)s_qkv * 16384.0));
    hipLaunchKernelGGL(attention_h2, dim3(grid), dim3(AH_THREADS), AH_LDS, s, static_cast<const char*>(qkv_planes), static_cast<char*>(out_planes),
                       heads, total, alpha, out_mul, static_cast<const float*>(nullptr));
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- operator-level entry (relax_op_attention under "f16x2"): fp32 qkv in, fp32 out.  The planes are made here with ONE scale for the whole
// tensor from its measured maximum (the model driver has a static bound instead), everything on the device: no host synchronisation
__global__ void ah_op_scalars(const unsigned* __restrict__ amax, float* __restrict__ tab) {
    const float s = h2_scale_for(__uint_as_float(amax[0]));   // qkv and - its rows being convex combinations of V rows - the output
    tab[0] = s;
    tab[1] = 1.f / s;
    tab[2] = (float)(0.125 * 1.44269504088896341 / ((double)s * (double)s));   // alpha
    tab[3] = 1.f / 16384.f;                                                     // out_mul = output scale / (s 2^14), the output scale being s
}

__global__ __launch_bounds__(256) void ah_from_h2(const char* __restrict__ y, float* __restrict__ x, int K, int64_t total8, const float* __restrict__ inv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int k8 = K >> 3;
    const int64_t row = i / k8;
    const int k = (int)(i - row * k8) * 8;
    const char* src = y + row * (int64_t)K * 4 + h2_offset(k);
    const h2_u32x4 hi = *reinterpret_cast<const h2_u32x4*>(src), lo = *reinterpret_cast<const h2_u32x4*>(src + 32);
    const float r = inv[0];
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned a = j == 0 ? hi.x : j == 1 ? hi.y : j == 2 ? hi.z : hi.w;
        const unsigned b = j == 0 ? lo.x : j == 1 ? lo.y : j == 2 ? lo.z : lo.w;
        o[2 * j] = ((float)__builtin_bit_cast(_Float16, (unsigned short)(a & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(b & 0xffffu))) * r;
        o[2 * j + 1] = ((float)__builtin_bit_cast(_Float16, (unsigned short)(a >> 16)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(b >> 16))) * r;
    }
    float* dst = x + row * K + k;
    *reinterpret_cast<ah_f32x4*>(dst) = (ah_f32x4){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<ah_f32x4*>(dst + 4) = (ah_f32x4){o[4], o[5], o[6], o[7]};
}

// (operator level, tests: ONE scale from the maximum of the whole qkv tensor of the call - not batch-invariant, unlike the engine's ViT path,
// which hands the kernel planes written with a static scale: csrc/h2.h)
int launch_attention_h2_op(relax_handle* h, const float* qkv, float* out, int Nimg, int heads, hipStream_t s) {
    RELAX_REQUIRE(h, Nimg > 0 && heads > 0 && qkv && out, "attention_h2 (operator): Nimg=%d heads=%d", Nimg, heads);
    const int dim = heads * 64;
    const int64_t rows = (int64_t)Nimg * AH_NTOK;
    const size_t q_bytes = ((size_t)rows * 3 * dim * 4 + 255) & ~(size_t)255, o_bytes = ((size_t)rows * dim * 4 + 255) & ~(size_t)255;
    RELAX_TRY(ensure_buf(h, h->sp3_ws, q_bytes + o_bytes + 512));
    char* Qp = static_cast<char*>(h->sp3_ws.p);
    char* Op = Qp + q_bytes;
    unsigned* amax = reinterpret_cast<unsigned*>(Op + o_bytes);
    float* tab = reinterpret_cast<float*>(Op + o_bytes + 256);
    RELAX_TRY(launch_image_absmax(h, qkv, rows * 3 * dim, 1, amax, s));
    hipLaunchKernelGGL(ah_op_scalars, dim3(1), dim3(1), 0, s, amax, tab);
    RELAX_TRY(launch_to_h2(h, qkv, 3 * dim, Qp, rows, 3 * dim, 1.f, tab, s, (int)rows));
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_h2), hipFuncAttributeMaxDynamicSharedMemorySize, AH_LDS));
        attr_set[h->device] = true;
    }
    const int total = Nimg * heads;
    hipLaunchKernelGGL(attention_h2, dim3(total < 256 ? total : 256), dim3(AH_THREADS), AH_LDS, s, Qp, Op, heads, total, 0.f, 0.f, tab + 2);
    const int64_t total8 = rows * (dim / 8);
    hipLaunchKernelGGL(ah_from_h2, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, Op, out, dim, total8, tab + 1);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

}  // namespace relax
