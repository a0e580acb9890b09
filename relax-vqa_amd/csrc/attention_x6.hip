// Attention of the DINO ViT (197 tokens, 64-d heads) on the bf16 matrix cores with fp32-grade arithmetic ("bf16x6"):
//   softmax(q k^T / 8) v          (src/extractor/visualise_vit_layer.py:93-106)
// Both contractions use the split-plane scheme of gemm_x6.hip: every fp32 operand x = hi + mid + lo (three bf16, exact), a
// product = the six partial products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// One persistent workgroup per CU walks the (image, head) items; 7 waves = 7 query tiles of 32.
//   scores   S^T[key, query] = K Q^T : the key on the MFMA row (register) axis, the query on the lane, so a lane owns one
//            query's scores: max / exp2 / sum are register-local plus one lane^32 exchange, and the probabilities already sit
//            in the B-operand layout of the next product (an accumulator tile used as the next MFMA's operand: the k order
//            inside a 16-deep step is permuted - the V^T image in LDS is stored in that same permuted key order).
//   output   O^T[d, query] = V^T P^T : V is staged TRANSPOSED ([d][key], keys contiguous) while it is split into planes.
// K and V^T share one 90 KB LDS buffer (K during the score phase, V^T during the output phase).  The fp32 rows of V arrive by
// LDS-DMA in a 50 KB staging area under the score MFMAs, those of the next item's K under the output MFMAs (no registers
// are tied up by the prefetch); splitting into planes (and, for V, transposing) is an LDS -> LDS pass between the phases.  The output leaves as split
// planes for the projection GEMM (and / or fp32), straight from the accumulators: a lane^32 exchange of 4-value runs gives
// every lane 8 consecutive d of its query, i.e. whole 16-byte units.
#include "relax_internal.h"
#include "sp3.h"
#include "h2.h"

// Phase-stamp hooks of the diagnostic build: empty in the product.  tools/abl/attention_x6_stamps.hip defines them and #includes this
// file (its flush overwrites the first floats of the fp32 output with the averages: a diagnostic library, never the product).
#ifndef A6_STAMP
#define A6_STAMP_DECL
#define A6_STAMP(i_)
#define A6_STAMP_ITEM()
#define A6_STAMP_FLUSH()
#endif

namespace relax {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float a6_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 a6_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned a6_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned a6_u32x2 __attribute__((ext_vector_type(2)));

constexpr int A6_NTOK = 197;
constexpr int A6_KT = 7;                  // key tiles of 32 (224 padded keys) = query tiles = waves
constexpr int A6_KPAD = A6_KT * 32;
[[maybe_unused]] constexpr int A6_VC = (A6_NTOK + 15) / 16;   // 16-key steps of the output phase that hold a real key: 13 (keys 208 .. 223 are padding, their probabilities exact zeros)
constexpr int A6_THREADS = A6_KT * 64;
constexpr int A6_KROW = 4 * kChunkBytes + 16;      // K image: [key][4 chunks of 16 d][3 planes][16] + pad (25 units: conflict-free)
constexpr int A6_VROW = 14 * kChunkBytes + 16;     // V^T image: [d][14 chunks of 16 keys][3 planes][16] + pad (85 units)
constexpr int A6_LDS = A6_KPAD * A6_KROW;          // 89600 B >= 64 * A6_VROW = 87040 B
static_assert(A6_LDS >= 64 * A6_VROW, "K and V^T share the buffer");
[[maybe_unused]] constexpr int A6_K_TASKS = A6_KPAD * 8 / A6_THREADS;   // (key, 8 d) tasks per thread: 4
[[maybe_unused]] constexpr int A6_V_PASSES = 4;                         // (d, 8 keys) tasks per thread: key group = pass * 7 + wave (28 >= 25)
constexpr int A6_STG_ROWS = 200;                       // staging: fp32 rows of K or V of one item, [key][64 d] (256 B), via LDS-DMA
[[maybe_unused]] constexpr int A6_STG = A6_LDS;                         // byte offset of the staging area
constexpr int A6_LDS_TOTAL = A6_LDS + A6_STG_ROWS * 256;
constexpr int A6_DMA_PIECES = A6_STG_ROWS * 256 / 1024;   // 50 pieces of 4 rows
[[maybe_unused]] constexpr int A6_DMA_PPW = (A6_DMA_PIECES + A6_KT - 1) / A6_KT;   // 8 per wave (the last wave issues 2)

__device__ inline a6_bf16x8 as_frag(const a6_u32x4 v) { return __builtin_bit_cast(a6_bf16x8, v); }

template <int OUT_PLANES, bool OUT_F32>   // OUT_PLANES: 0 none, 1 split planes (three bf16), 2 two fp16 planes of value * out_scale (csrc/h2.h)
__global__ __launch_bounds__(A6_THREADS) void attention_x6(const float* __restrict__ qkv, float* __restrict__ out,
                                                           char* __restrict__ out_sp3, int heads, int total_items, float out_scale) {
#if __HIP_DEVICE_COMPILE__   // the host pass only needs the launch stub (no __amdgpu_buffer_rsrc_t there)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    const int dim = heads * 64;
    const int ld = 3 * dim;
    const a6_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int item = blockIdx.x;
    if (item >= total_items) return;   // workgroup-uniform

    a6_u32x4 qp[4][3];                 // Q^T fragments of this lane's query: [16-deep d step][plane]
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // fp32 rows [key][64 d] of matrix which_ (1 = K, 2 = V) of an item -> staging, 4 rows per DMA instruction; rows past the
    // image's 197 tokens are beyond the buffer resource's range and arrive as zeros
#define A6_DMA(item_, which_)                                                                                        \
    {                                                                                                                \
        const float* b_ = qkv + (int64_t)((item_) / heads) * A6_NTOK * ld;                                           \
        __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_), 0, A6_NTOK * ld * 4, 0x00020000); \
        const int col_ = ((which_) * dim + ((item_) % heads) * 64) * 4 + (lane & 15) * 16;                           \
        _Pragma("unroll") for (int j = 0; j < A6_DMA_PPW; ++j) {                                                     \
            const int piece = wave_u * A6_DMA_PPW + j;                                                               \
            if (piece < A6_DMA_PIECES)                                                                               \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(smem + A6_STG + piece * 1024), \
                                                         16, (piece * 4 + (lane >> 4)) * ld * 4 + col_, 0, 0, 0);    \
        }                                                                                                            \
    }
    // staging (K rows) -> K image: split into planes; task = (key, 8 d)
#define A6_STORE_K()                                                                                                 \
    _Pragma("unroll") for (int t = 0; t < A6_K_TASKS; ++t) {                                                         \
        const int idx = t * A6_THREADS + tid;                                                                        \
        const int key = idx >> 3, d8 = idx & 7;                                                                      \
        a6_f32x4 a_ = zero4, b2_ = zero4;                                                                            \
        if (key < A6_STG_ROWS) {                                                                                     \
            a_ = *reinterpret_cast<const a6_f32x4*>(smem + A6_STG + key * 256 + d8 * 32);                            \
            b2_ = *reinterpret_cast<const a6_f32x4*>(smem + A6_STG + key * 256 + d8 * 32 + 16);                      \
        }                                                                                                            \
        sp3_u32x4 hi_, mid_, lo_;                                                                                    \
        split3_x8(a_, b2_, hi_, mid_, lo_);                                                                          \
        char* d_ = smem + key * A6_KROW + (d8 >> 1) * kChunkBytes + (d8 & 1) * 16;                                   \
        *reinterpret_cast<sp3_u32x4*>(d_) = hi_;                                                                     \
        *reinterpret_cast<sp3_u32x4*>(d_ + 32) = mid_;                                                               \
        *reinterpret_cast<sp3_u32x4*>(d_ + 64) = lo_;                                                                \
    }
    // staging (V rows) -> V^T image: thread (d = lane, key group g = pass * 7 + wave) takes keys 8g .. 8g+7 of column d.
    // Key offset ko inside its chunk of 16 sits at position 8*h + 4*(g4 >> 1) + (ko & 3) with g4 = ko >> 2, h = g4 & 1: the
    // order in which the score accumulators hold the keys (C/D map of the 32x32 MFMA), so P feeds the next MFMA unshuffled
#define A6_STORE_V()                                                                                                 \
    _Pragma("unroll") for (int ps = 0; ps < A6_V_PASSES; ++ps) {                                                     \
        const int g = ps * A6_KT + wave;                                                                             \
        float v_[8];                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                \
            v_[j] = 8 * g + j < A6_STG_ROWS ? *reinterpret_cast<const float*>(smem + A6_STG + (8 * g + j) * 256 + lane * 4) : 0.f; \
        sp3_u32x2 h0_, m0_, l0_, h1_, m1_, l1_;                                                                      \
        split3_x4((sp3_f32x4){v_[0], v_[1], v_[2], v_[3]}, h0_, m0_, l0_);                                           \
        split3_x4((sp3_f32x4){v_[4], v_[5], v_[6], v_[7]}, h1_, m1_, l1_);                                           \
        char* d_ = smem + lane * A6_VROW + (g >> 1) * kChunkBytes + (g & 1) * 8;                                     \
        *reinterpret_cast<sp3_u32x2*>(d_) = h0_;             /* keys 8(g&1) + 0..3 -> positions 4(g&1) + 0..3 */     \
        *reinterpret_cast<sp3_u32x2*>(d_ + 16) = h1_;        /* keys 8(g&1) + 4..7 -> positions 8 + 4(g&1) + 0..3 */ \
        *reinterpret_cast<sp3_u32x2*>(d_ + 32) = m0_;                                                                \
        *reinterpret_cast<sp3_u32x2*>(d_ + 48) = m1_;                                                                \
        *reinterpret_cast<sp3_u32x2*>(d_ + 64) = l0_;                                                                \
        *reinterpret_cast<sp3_u32x2*>(d_ + 80) = l1_;                                                                \
    }
#define A6_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    // Q^T fragments: lane (query li, half) needs d = 16 s + 8 half .. + 7 of its query; head_dim^-0.5 and log2(e) are folded
    // into Q (as in the fp32 kernel), so the softmax exponential is one v_exp_f32
#define A6_REQUEST_Q(item_)   /* the 64 fp32 of this lane's query half-rows into registers (requested early, split later) */ \
    {                                                                                                                \
        const int qrow_ = wave * 32 + li;                                                                            \
        const float* q_ = qkv + (int64_t)((item_) / heads) * A6_NTOK * ld + ((item_) % heads) * 64 +                 \
                          (int64_t)(qrow_ < A6_NTOK ? qrow_ : A6_NTOK - 1) * ld + 8 * half;                          \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                              \
            qraw[2 * s] = *reinterpret_cast<const a6_f32x4*>(q_ + 16 * s);                                           \
            qraw[2 * s + 1] = *reinterpret_cast<const a6_f32x4*>(q_ + 16 * s + 4);                                   \
        }                                                                                                            \
    }
#define A6_SPLIT_Q()                                                                                                 \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                  \
        sp3_u32x4 hi_, mid_, lo_;                                                                                    \
        split3_x8(qraw[2 * s] * (0.125f * 1.44269504088896341f), qraw[2 * s + 1] * (0.125f * 1.44269504088896341f), hi_, mid_, lo_); \
        qp[s][0] = hi_;                                                                                              \
        qp[s][1] = mid_;                                                                                             \
        qp[s][2] = lo_;                                                                                              \
    }

    a6_f32x4 qraw[8];
    A6_DMA(item, 1);
    A6_REQUEST_Q(item);
    A6_SPLIT_Q();
    A6_WAIT_DMA();
    __syncthreads();
    A6_STORE_K();
    __syncthreads();

    A6_STAMP_DECL;
    while (true) {
        A6_DMA(item, 2);   // V rows: in flight during the score phase
        // ---- scores: S^T tile kt = K[kt] Q^T, 4 d-steps x 6 partial products (smallest first) -------------------------
        // The K fragments of group g + 1 (one 16-deep d step of one key tile: three planes) are read while the six MFMAs of group g
        // run: left to the compiler the loop came out as read - wait - MFMA with the LDS latency exposed at every group (a wave
        // alone on its SIMD took 3 x the MFMAs' own time for this phase); __builtin_amdgcn_sched_barrier pins the pipeline.
        floatx16 sacc[A6_KT];
        a6_bf16x8 kf[2][3];
#define A6_READ_K(buf_, g_)                                                                                          \
        {                                                                                                            \
            const char* kp_ = smem + (((g_) >> 2) * 32 + li) * A6_KROW + half * 16 + ((g_) & 3) * kChunkBytes;       \
            kf[buf_][0] = *reinterpret_cast<const a6_bf16x8*>(kp_);                                                  \
            kf[buf_][1] = *reinterpret_cast<const a6_bf16x8*>(kp_ + 32);                                             \
            kf[buf_][2] = *reinterpret_cast<const a6_bf16x8*>(kp_ + 64);                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }
        A6_READ_K(0, 0);
#pragma unroll
        for (int g = 0; g < 4 * A6_KT; ++g) {
            const int kt = g >> 2, s = g & 3, b = g & 1;
            if (g + 1 < 4 * A6_KT) A6_READ_K((g + 1) & 1, g + 1);
            if (s == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
            }
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[b][2], as_frag(qp[s][0]), sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[b][1], as_frag(qp[s][1]), sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[b][0], as_frag(qp[s][2]), sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[b][1], as_frag(qp[s][0]), sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[b][0], as_frag(qp[s][1]), sacc[kt], 0, 0, 0);
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[b][0], as_frag(qp[s][0]), sacc[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef A6_READ_K
        A6_STAMP(0);   // scores
        // sacc[kt][r] = score(query li, key kt*32 + (r&3) + 8*(r>>2) + 4*half) * log2(e)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < A6_KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= A6_NTOK) sacc[kt][r] = -INFINITY;
                mx = fmaxf(mx, sacc[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < A6_KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sacc[kt][r] - mx);
                sacc[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        A6_STAMP(1);   // softmax
        A6_WAIT_DMA();
        __syncthreads();          // every wave is done with K, and the V rows have landed
        A6_STAMP(2);   // wait for V + barrier
        A6_STORE_V();             // V^T takes K's place
        __syncthreads();
        A6_STAMP(3);   // V rows -> transposed plane image + barrier
        const int next = item + gridDim.x;
        if (next < total_items) A6_DMA(next, 1);       // K rows of the next item: in flight during the output phase

        // ---- output: O^T[d, query] = V^T P^T over 13 steps of 16 keys (the 14th holds padding keys only: zeros times zeros) ---------------------------------------------------
        // Pipelined like the scores: the six V^T fragments of step c + 1 are read, and its probabilities split into planes, while
        // the twelve MFMAs of step c run.  Registers 8*(c&1) .. +7 of score tile c>>1 are keys 16c + 8*(j>>2) + 4*half + (j&3):
        // the B fragment of step c.
        floatx16 oacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
        a6_bf16x8 vf[2][2][3];
        sp3_u32x4 pp[2][3];
#define A6_READ_V(buf_, c_)                                                                                          \
        _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                           \
            const char* vp_ = smem + (dt * 32 + li) * A6_VROW + (c_) * kChunkBytes + half * 16;                      \
            vf[buf_][dt][0] = *reinterpret_cast<const a6_bf16x8*>(vp_);                                              \
            vf[buf_][dt][1] = *reinterpret_cast<const a6_bf16x8*>(vp_ + 32);                                         \
            vf[buf_][dt][2] = *reinterpret_cast<const a6_bf16x8*>(vp_ + 64);                                         \
        }
#define A6_SPLIT_P(buf_, c_)                                                                                         \
        split3_x8((sp3_f32x4){sacc[(c_) >> 1][8 * ((c_) & 1) + 0], sacc[(c_) >> 1][8 * ((c_) & 1) + 1],              \
                              sacc[(c_) >> 1][8 * ((c_) & 1) + 2], sacc[(c_) >> 1][8 * ((c_) & 1) + 3]},             \
                  (sp3_f32x4){sacc[(c_) >> 1][8 * ((c_) & 1) + 4], sacc[(c_) >> 1][8 * ((c_) & 1) + 5],              \
                              sacc[(c_) >> 1][8 * ((c_) & 1) + 6], sacc[(c_) >> 1][8 * ((c_) & 1) + 7]},             \
                  pp[buf_][0], pp[buf_][1], pp[buf_][2])
        A6_READ_V(0, 0);
        A6_SPLIT_P(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < A6_VC; ++c) {
            const int b = c & 1;
            if (c + 1 < A6_VC) {
                A6_READ_V(b ^ 1, c + 1);
                __builtin_amdgcn_sched_barrier(0);       // the reads go out first ...
                A6_SPLIT_P(b ^ 1, c + 1);                // ... the split's VALU interleaves with the MFMAs below
            }
            if (c == A6_KT && next < total_items) A6_REQUEST_Q(next);   // the next item's queries (half of the score registers are free by now)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[b][dt][2], as_frag(pp[b][0]), oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[b][dt][1], as_frag(pp[b][1]), oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[b][dt][0], as_frag(pp[b][2]), oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[b][dt][1], as_frag(pp[b][0]), oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[b][dt][0], as_frag(pp[b][1]), oacc[dt], 0, 0, 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[b][dt][0], as_frag(pp[b][0]), oacc[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef A6_READ_V
#undef A6_SPLIT_P
        A6_STAMP(4);   // P V
        // oacc[dt][r] = O(query li, d = dt*32 + (r&3) + 8*(r>>2) + 4*half): runs of 4 consecutive d.  The lane^32 partner
        // holds the runs in between: after swapping two runs per tile each lane owns two units of 8 consecutive d
        //   half 0: d = dt*32 + 0..7 and 16..23        half 1: d = dt*32 + 8..15 and 24..31
        const int q = wave * 32 + li;
        const int64_t orow = (int64_t)(item / heads) * A6_NTOK + q;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            float own[16], got[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) own[r] = oacc[dt][r] * inv;
#pragma unroll
            for (int u = 0; u < 2; ++u)        // unit u: half 0 sends run 2u+1 and keeps run 2u, half 1 sends run 2u and keeps 2u+1
#pragma unroll
                for (int j = 0; j < 4; ++j) got[4 * u + j] = __shfl_xor(half ? own[8 * u + j] : own[8 * u + 4 + j], 32);
            if (q < A6_NTOK) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    a6_f32x4 lo4, hi4;   // d0 .. d0+3, d0+4 .. d0+7
                    if (half == 0) {
                        lo4 = (a6_f32x4){own[8 * u], own[8 * u + 1], own[8 * u + 2], own[8 * u + 3]};
                        hi4 = (a6_f32x4){got[4 * u], got[4 * u + 1], got[4 * u + 2], got[4 * u + 3]};
                    } else {
                        lo4 = (a6_f32x4){got[4 * u], got[4 * u + 1], got[4 * u + 2], got[4 * u + 3]};
                        hi4 = (a6_f32x4){own[8 * u + 4], own[8 * u + 5], own[8 * u + 6], own[8 * u + 7]};
                    }
                    const int d0 = (item % heads) * 64 + dt * 32 + 16 * u + 8 * half;
                    if (OUT_PLANES == 1) store_sp3_x8(out_sp3 + orow * ((int64_t)dim * 6), d0, lo4, hi4);
                    if (OUT_PLANES == 2) store_h2_x8(out_sp3 + orow * ((int64_t)dim * 4), d0, lo4, hi4, out_scale);
                    if (OUT_F32) {
                        *reinterpret_cast<a6_f32x4*>(out + orow * dim + d0) = lo4;
                        *reinterpret_cast<a6_f32x4*>(out + orow * dim + d0 + 4) = hi4;
                    }
                }
            }
        }
        A6_STAMP(5);   // epilogue
        A6_STAMP_ITEM();
        if (next >= total_items) break;
        A6_SPLIT_Q();
        A6_STAMP(6);   // split of the next Q
        A6_WAIT_DMA();
        __syncthreads();          // every wave is done with V^T, and the K rows have landed
        A6_STAMP(7);   // wait for K + barrier
        A6_STORE_K();             // next item's K
        __syncthreads();
        A6_STAMP(8);   // K rows -> plane image + barrier
        item = next;
    }
    A6_STAMP_FLUSH();
#undef A6_DMA
#undef A6_WAIT_DMA
#undef A6_STORE_K
#undef A6_STORE_V
#undef A6_REQUEST_Q
#undef A6_SPLIT_Q
#endif
}

int launch_attention_x6(relax_handle* h, const float* qkv, float* out, void* out_planes, int Nimg, int heads, hipStream_t s,
                        float out_h2_scale) {
    RELAX_REQUIRE(h, Nimg > 0 && heads > 0 && (out || out_planes), "attention_x6: Nimg=%d heads=%d", Nimg, heads);
    RELAX_REQUIRE(h, !(out_h2_scale > 0.f) || (out_planes && !out), "attention_x6: the fp16-plane output goes alone");
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_x6<1, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, A6_LDS_TOTAL));
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_x6<0, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, A6_LDS_TOTAL));
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_x6<1, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, A6_LDS_TOTAL));
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_x6<2, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, A6_LDS_TOTAL));
        attr_set[h->device] = true;
    }
    const int total = Nimg * heads;
    const int grid = total < 256 ? total : 256;   // one persistent workgroup (7 waves) per CU
    char* o6 = static_cast<char*>(out_planes);
    if (out_h2_scale > 0.f)
        hipLaunchKernelGGL((attention_x6<2, false>), dim3(grid), dim3(A6_THREADS), A6_LDS_TOTAL, s, qkv, out, o6, heads, total, out_h2_scale);
    else if (out && o6)
        hipLaunchKernelGGL((attention_x6<1, true>), dim3(grid), dim3(A6_THREADS), A6_LDS_TOTAL, s, qkv, out, o6, heads, total, 1.f);
    else if (o6)
        hipLaunchKernelGGL((attention_x6<1, false>), dim3(grid), dim3(A6_THREADS), A6_LDS_TOTAL, s, qkv, out, o6, heads, total, 1.f);
    else
        hipLaunchKernelGGL((attention_x6<0, true>), dim3(grid), dim3(A6_THREADS), A6_LDS_TOTAL, s, qkv, out, o6, heads, total, 1.f);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

}  // namespace relax
