// Exact (erf) GELU of the ViT MLP (reference: nn.GELU() in src/extractor/visualise_vit_layer.py:63-76), branch-free.
//
//   GELU(v) = v * Phi(v) = v * (1 - erfc(t) / 2)   for v >= 0,      v * erfc(t) / 2   for v < 0,        t = |v| / sqrt(2)
//   erfc(t) = 2^(-t * q(t)),  q = a degree-8 polynomial fitted to -log2(erfc(t)) / t on [0, 5]   (t is clamped to 5:
//   erfc(5) = 1.5e-12, below the last bit of anything it is added to or multiplied with here)
//
// One v_exp_f32 and ten multiply-adds per value, no lane-divergent branches (libm's erff takes two range branches, and a
// wave runs both: ~45 instructions per value, which made the fc1 epilogue VALU-bound), and the four values of a vector
// go through packed fp32 instructions.  Writing the negative side as v * erfc / 2 also avoids the cancellation of
// 1 + erf(v / sqrt 2): against the fp64 function the largest error is 3.9e-7 (one rounding of the product at |v| ~ 5),
// 4.4e-8 rms - torch's own fp32 GELU: 1.2e-6 / 1.1e-7 (tools/gelu_fit.py fits the polynomial and prints both).
#pragma once
#include <hip/hip_runtime.h>

namespace relax {

typedef float gelu_f32x4 __attribute__((ext_vector_type(4)));

__device__ inline gelu_f32x4 gelu_erf4(const gelu_f32x4 v) {
    const gelu_f32x4 z = v * 0.70710678118654752440f;
    const gelu_f32x4 t = __builtin_elementwise_min(__builtin_elementwise_abs(z), (gelu_f32x4)(5.0f));
    gelu_f32x4 q = (gelu_f32x4)(-5.936020443186862e-06f);
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(9.409502672497183e-05f));
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(-0.0005975606618449092f));
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(0.0017074682982638478f));
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(0.0006545027135871351f));
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(-0.028285665437579155f));
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(0.14854633808135986f));
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(0.9183921813964844f));
    q = __builtin_elementwise_fma(q, t, (gelu_f32x4)(1.6279112100601196f));
    const gelu_f32x4 u = -(t * q);                                         // log2 erfc(t), in [-39.3, 0]: no denormals
    const gelu_f32x4 h = 0.5f * (gelu_f32x4){__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y), __builtin_amdgcn_exp2f(u.z),
                                             __builtin_amdgcn_exp2f(u.w)};   // erfc(t) / 2
    const gelu_f32x4 g = 1.0f - h;
    return v * (gelu_f32x4){z.x >= 0.f ? g.x : h.x, z.y >= 0.f ? g.y : h.y, z.z >= 0.f ? g.z : h.z, z.w >= 0.f ? g.w : h.w};
}

// the same operations on one value (bit-identical to a lane of gelu_erf4)
__device__ inline float gelu_erf(float v) {
    const float z = v * 0.70710678118654752440f;
    const float t = fminf(fabsf(z), 5.0f);
    float q = -5.936020443186862e-06f;
    q = fmaf(q, t, 9.409502672497183e-05f);
    q = fmaf(q, t, -0.0005975606618449092f);
    q = fmaf(q, t, 0.0017074682982638478f);
    q = fmaf(q, t, 0.0006545027135871351f);
    q = fmaf(q, t, -0.028285665437579155f);
    q = fmaf(q, t, 0.14854633808135986f);
    q = fmaf(q, t, 0.9183921813964844f);
    q = fmaf(q, t, 1.6279112100601196f);
    const float h = 0.5f * __builtin_amdgcn_exp2f(-(t * q));
    return v * (z >= 0.f ? 1.0f - h : h);
}

}  // namespace relax
