// Two fp16 planes ("h2"): an fp32 value x held as TWO fp16 numbers of the scaled value  x * s = hi + lo,  s = 2^e a power of two,
// hi = round-to-nearest-even(x * s), lo = round-to-nearest-even(x * s - hi): 22 significant bits (|error| <= 2^-22 |x|) wherever
// |x * s| >= 2^-3, an absolute error of at most 2^-25 / s below that (lo is then an fp16 subnormal; the f16 MFMA of gfx950 keeps
// subnormal operands - tools/micro/mfma_f16x2.hip, part 2), and NO representation above |x * s| = 65504: the scale must be chosen
// from a bound on the tensor.  Multiplying by a power of two is exact, so the scale changes no bit of the 22 that are kept.
// A row of K values is stored as K/16 chunks of 64 bytes,
//        chunk = [16 x fp16 hi][16 x fp16 lo]
// so one 16-deep K step of one row is 64 contiguous bytes whose four 16-byte units are MFMA fragments of v_mfma_f32_16x16x32_f16
// (gemm_h2.hip: the 32-deep K of that instruction takes BOTH planes of a chunk, so all four partial products ah bh, al bh, ah bl,
// al bl come out of two instructions).  4 bytes per value (split planes of three bf16: 6).
// Where the scales come from (gemm_h2.hip has the whole argument):
//   weights            per output row n: 2^t_n with max_k |W[n,k]| * 2^t_n in [2^14, 2^15)  (made at load time)
//   ViT activations    one STATIC power of two per tensor from a bound that holds for every input (LayerNorm output: |z_i| <=
//                      sqrt(dim - 1); GEMM outputs: Cauchy-Schwarz with the weight rows; attention output: a convex combination of
//                      V rows; GELU(x): |GELU(x)| <= |x|): no data-dependent scale, so every row's bits are independent of the batch
//   operator level     per row of A, from the row's own maximum (to_h2_rows): batch-invariant as well.  ONE exception: relax_op_attention
//                      (attention_h2.hip, launch_attention_h2_op) takes one scale from the maximum of the WHOLE qkv tensor it is handed, so an
//                      image's output bits there depend on what else is in the call - it is the test entry of the kernel; the engine's ViT
//                      path feeds the same kernel planes written with the static Cauchy-Schwarz scale of the qkv GEMM (vit.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace relax {

typedef float h2_f32x4 __attribute__((ext_vector_type(4)));
typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned h2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned h2_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kH2ChunkBytes = 64;   // one 16-deep K step of one row

// (x, y), already scaled -> packed fp16 pairs hi, lo (round to nearest even at both steps; x - hi is exact in fp32)
__device__ inline void split2_pair(float x, float y, unsigned& hi, unsigned& lo) {
    const h2_f16x2 h = __builtin_convertvector((h2_f32x2){x, y}, h2_f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const h2_f32x2 back = __builtin_convertvector(h, h2_f32x2);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((h2_f32x2){x - back.x, y - back.y}, h2_f16x2));
}

// 8 consecutive scaled values (half a chunk) -> the 16-byte units of the two planes
__device__ inline void split2_x8(const h2_f32x4 a, const h2_f32x4 b, h2_u32x4& hi, h2_u32x4& lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2_pair(a.x, a.y, h0, l0);
    split2_pair(a.z, a.w, h1, l1);
    split2_pair(b.x, b.y, h2, l2);
    split2_pair(b.z, b.w, h3, l3);
    hi = (h2_u32x4){h0, h1, h2, h3};
    lo = (h2_u32x4){l0, l1, l2, l3};
}

// 4 consecutive scaled values -> 8 bytes of each plane
__device__ inline void split2_x4(const h2_f32x4 a, h2_u32x2& hi, h2_u32x2& lo) {
    unsigned h0, h1, l0, l1;
    split2_pair(a.x, a.y, h0, l0);
    split2_pair(a.z, a.w, h1, l1);
    hi = (h2_u32x2){h0, h1};
    lo = (h2_u32x2){l0, l1};
}

// byte offset of the hi unit of the 8 values k .. k+7 (k % 8 == 0) inside an h2 row; the lo unit sits 32 bytes further
__device__ __host__ inline int64_t h2_offset(int k) { return (int64_t)(k >> 4) * kH2ChunkBytes + (k & 8) * 2; }

// store 8 consecutive values k .. k+7 of a row, scaled by `scale` (a power of two)
__device__ inline void store_h2_x8(char* row, int k, const h2_f32x4 a, const h2_f32x4 b, float scale) {
    h2_u32x4 hi, lo;
    split2_x8(a * scale, b * scale, hi, lo);
    char* d = row + h2_offset(k);
    *reinterpret_cast<h2_u32x4*>(d) = hi;
    *reinterpret_cast<h2_u32x4*>(d + 32) = lo;
}

// the power of two that puts a maximum magnitude `amax` into [2^14, 2^15) (zero / non-finite rows: 1)
__device__ __host__ inline float h2_scale_for(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);          // amax = m * 2^e, m in [0.5, 1)  ->  amax in [2^(e-1), 2^e)
    int sh = 15 - e;                 // amax * 2^sh in [2^14, 2^15)
    sh = sh > 120 ? 120 : (sh < -120 ? -120 : sh);
    return ldexpf(1.f, sh);
}

}  // namespace relax
