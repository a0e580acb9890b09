// Split planes ("sp3"): an fp32 value x held as three bf16 numbers x = hi + mid + lo, each a round-to-nearest of what is
// left (3 x 8 = 24 significant bits: the split is exact).  A row of K values is stored as K/16 chunks of 96 bytes,
//        chunk = [16 x bf16 hi][16 x bf16 mid][16 x bf16 lo]
// so that one 16-deep K step of one row is 96 contiguous bytes whose 16-byte units are MFMA fragments (gemm_x6.hip).
// Range: exact for finite |x| <= 3.3895e38 (the largest bf16; beyond it the round-to-nearest of `hi` is +-inf and the value
// becomes NaN - the 0.4 % of the fp32 range above it is not reachable by normalised pixels, LayerNorm outputs or trained
// weights, and relax_set_option(h, "gemm_precision", 0) runs the exact-fp32 kernels for data that needs it); the `lo` plane of
// values below 2^-110 is a bf16 subnormal (the product it would contribute is below 2^-133 of the operand scale).  NaN / inf
// inputs stay NaN / inf (v_cvt_pk_bf16_f32 keeps them), as on the fp32 path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace relax {

typedef float sp3_f32x4 __attribute__((ext_vector_type(4)));
typedef float sp3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sp3_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned sp3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sp3_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kChunkBytes = 96;   // one 16-deep K step of one row

// (x, y) -> packed bf16 pairs hi, mid, lo with x = hi + mid + lo exactly (round to nearest at every step)
__device__ inline void split3_pair(float x, float y, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector((sp3_f32x2){x, y}, sp3_bf16x2));
    const float rx = x - __uint_as_float(hi << 16), ry = y - __uint_as_float(hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector((sp3_f32x2){rx, ry}, sp3_bf16x2));
    const float sx = rx - __uint_as_float(mid << 16), sy = ry - __uint_as_float(mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((sp3_f32x2){sx, sy}, sp3_bf16x2));
}

// 4 consecutive values -> 8 bytes of each plane
__device__ inline void split3_x4(const sp3_f32x4 a, sp3_u32x2& hi, sp3_u32x2& mid, sp3_u32x2& lo) {
    unsigned h0, h1, m0, m1, l0, l1;
    split3_pair(a.x, a.y, h0, m0, l0);
    split3_pair(a.z, a.w, h1, m1, l1);
    hi = (sp3_u32x2){h0, h1};
    mid = (sp3_u32x2){m0, m1};
    lo = (sp3_u32x2){l0, l1};
}

// 8 consecutive values (half a chunk) -> the three 16-byte units of its planes
__device__ inline void split3_x8(const sp3_f32x4 a, const sp3_f32x4 b, sp3_u32x4& hi, sp3_u32x4& mid, sp3_u32x4& lo) {
    unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    split3_pair(a.x, a.y, h0, m0, l0);
    split3_pair(a.z, a.w, h1, m1, l1);
    split3_pair(b.x, b.y, h2, m2, l2);
    split3_pair(b.z, b.w, h3, m3, l3);
    hi = (sp3_u32x4){h0, h1, h2, h3};
    mid = (sp3_u32x4){m0, m1, m2, m3};
    lo = (sp3_u32x4){l0, l1, l2, l3};
}

// byte offset of the 8 values k .. k+7 (k % 8 == 0) of plane 0 inside an sp3 row
__device__ __host__ inline int64_t sp3_offset(int k) { return (int64_t)(k >> 4) * kChunkBytes + (k & 8) * 2; }

__device__ inline void store_sp3_x8(char* row, int k, const sp3_f32x4 a, const sp3_f32x4 b) {
    sp3_u32x4 hi, mid, lo;
    split3_x8(a, b, hi, mid, lo);
    char* d = row + sp3_offset(k);
    *reinterpret_cast<sp3_u32x4*>(d) = hi;
    *reinterpret_cast<sp3_u32x4*>(d + 32) = mid;
    *reinterpret_cast<sp3_u32x4*>(d + 64) = lo;
}

// 4 values k .. k+3 (k % 4 == 0): 8 bytes of each plane
__device__ inline void store_sp3_x4(char* row, int k, const sp3_f32x4 a) {
    sp3_u32x2 hi, mid, lo;
    split3_x4(a, hi, mid, lo);
    char* d = row + (int64_t)(k >> 4) * kChunkBytes + (k & 15) * 2;
    *reinterpret_cast<sp3_u32x2*>(d) = hi;
    *reinterpret_cast<sp3_u32x2*>(d + 32) = mid;
    *reinterpret_cast<sp3_u32x2*>(d + 64) = lo;
}

}  // namespace relax
