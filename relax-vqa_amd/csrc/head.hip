// The quality head at inference: imputer + min-max scaler + 3-layer MLP on the all-gathered
// [n_clips, F] feature matrix.
//   src/demo_test.py:177-181      imputer.transform (NaN -> column mean), scaler.transform (x*scale_+min_, float64), to fp32
//   src/model_regression.py:37-58 Mlp: fc1 -> BatchNorm1d(eval) -> GELU -> fc2 -> GELU -> fc3   (dropout is identity in eval)
//   src/demo_test.py:25-35        fix_state_dict: 'module.' prefix stripped, 'n_averaged' dropped
// fc1/fc2 run on the contraction kernel (BatchNorm folded into fc1, GELU in the epilogue); fc3 is a 128-long dot product.
#include <cmath>

#include "relax_internal.h"
#include "host_logic.h"

namespace relax {

// x [n,F] fp32 -> xp [n,Fpad] fp32: NaN -> stats[f]; (double) x * scale + min -> float; zero padding
__global__ __launch_bounds__(256) void head_preprocess(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const double* __restrict__ scale, const double* __restrict__ mn,
                                                       float* __restrict__ xp, int F, int Fpad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % Fpad);
    const int64_t r = i / Fpad;
    float o = 0.f;
    if (f < F) {
        double v = (double)x[r * F + f];
        if (stats && v != v) v = stats[f];
        o = (float)(v * scale[f] + mn[f]);
    }
    xp[i] = o;
}

__global__ __launch_bounds__(128) void head_fc3(const float* __restrict__ hid, const float* __restrict__ w, float bias,
                                                float* __restrict__ out, int n, int K) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(hid[(int64_t)r * K + k], w[k], s);
    out[r] = s + bias;
}

void free_head(relax_handle* h) {
    for (void* p : h->head.allocs) (void)hipFree(p);
    h->head = HeadW();
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_load_mlp_head(relax_handle* h, const float* const* tensors, const char* const* names, const int64_t* numels,
                        int n, const double* imputer_statistics, const double* scaler_scale, const double* scaler_min,
                        int input_features) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, tensors && names && numels && n > 0 && scaler_scale && scaler_min && input_features > 0,
                  "relax_load_mlp_head: bad arguments");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    free_head(h);
    host::StateDict sd;
    for (int i = 0; i < n; ++i) {
        if (names[i] && std::string(names[i]) == "n_averaged") continue;
        sd.add(names[i], tensors[i], numels[i], /*strip_module=*/true);
    }
    auto get = [&](const char* key, int64_t numel) -> const float* {
        std::string err;
        const float* p = sd.get(key, numel, err, "mlp head state dict");
        if (!p) set_error(h, "%s", err.c_str());
        return p;
    };
    const int F = input_features;
    const int64_t nw1 = sd.numel("fc1.weight");
    RELAX_REQUIRE(h, nw1 > 0 && nw1 % F == 0, "mlp head: fc1.weight missing or not [hidden,%d]", F);
    const int H1 = (int)(nw1 / F);
    const int H2 = H1 / 2;
    RELAX_REQUIRE(h, H1 % 64 == 0 && H2 % 64 == 0, "mlp head: hidden sizes %d/%d must be multiples of 64", H1, H2);
    const float *w1 = get("fc1.weight", (int64_t)H1 * F), *b1 = get("fc1.bias", H1);
    const float *g = get("bn1.weight", H1), *be = get("bn1.bias", H1), *mu = get("bn1.running_mean", H1),
                *var = get("bn1.running_var", H1);
    const float *w2 = get("fc2.weight", (int64_t)H2 * H1), *b2 = get("fc2.bias", H2);
    const float *w3 = get("fc3.weight", H2), *b3 = get("fc3.bias", 1);
    if (!w1 || !b1 || !g || !be || !mu || !var || !w2 || !b2 || !w3 || !b3) return RELAX_ERR_INVALID;
    HeadW& hw = h->head;
    hw.F = F;
    hw.Fpad = ((F + 31) / 32) * 32;
    hw.H1 = H1;
    hw.H2 = H2;
    // fold BatchNorm1d (eval) into fc1: y = (x W^T + b - mu) * s + beta, s = gamma / sqrt(var + eps)   (host_logic.cpp)
    std::vector<float> w1p((size_t)H1 * hw.Fpad), b1p(H1);
    host::fold_fc_bn(w1, b1, g, be, mu, var, 1e-5f, H1, F, hw.Fpad, w1p.data(), b1p.data());
    int rc = upload(h, w1p.data(), w1p.size(), &hw.w1, hw.allocs);
    if (rc == RELAX_OK) rc = upload(h, b1p.data(), H1, &hw.b1, hw.allocs);
    if (rc == RELAX_OK) rc = upload(h, w2, (size_t)H2 * H1, &hw.w2, hw.allocs);
    if (rc == RELAX_OK) rc = upload(h, b2, H2, &hw.b2, hw.allocs);
    if (rc == RELAX_OK) rc = upload(h, w3, H2, &hw.w3, hw.allocs);
    hw.b3 = b3[0];
    auto up_d = [&](const double* src, double** dst) {
        if (rc != RELAX_OK) return;
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(double) * F) != hipSuccess) {
            set_error(h, "mlp head: hipMalloc failed");
            rc = RELAX_ERR_NOMEM;
            return;
        }
        hw.allocs.push_back(p);
        if (hipMemcpy(p, src, sizeof(double) * F, hipMemcpyHostToDevice) != hipSuccess) {
            set_error(h, "mlp head: hipMemcpy failed");
            rc = RELAX_ERR_HIP;
            return;
        }
        *dst = static_cast<double*>(p);
    };
    if (imputer_statistics) up_d(imputer_statistics, &hw.stats);
    up_d(scaler_scale, &hw.scale);
    up_d(scaler_min, &hw.mn);
    if (rc != RELAX_OK) {
        free_head(h);
        return rc;
    }
    hw.loaded = true;
    return RELAX_OK;
}

int relax_mlp_head(relax_handle* h, const float* features, int n, float* scores, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, h->head.loaded, "relax_mlp_head: call relax_load_mlp_head first");
    RELAX_REQUIRE(h, features && scores && n > 0, "relax_mlp_head: bad arguments");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const HeadW& hw = h->head;
    const size_t need = sizeof(float) * (size_t)n * ((size_t)hw.Fpad + hw.H1 + hw.H2);
    RELAX_TRY(ensure_buf(h, h->head_ws, need));
    float* xp = static_cast<float*>(h->head_ws.p);
    float* a1 = xp + (size_t)n * hw.Fpad;
    float* a2 = a1 + (size_t)n * hw.H1;
    const int64_t total = (int64_t)n * hw.Fpad;
    hipLaunchKernelGGL(head_preprocess, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, features, hw.stats, hw.scale,
                       hw.mn, xp, hw.F, hw.Fpad, total);
    RELAX_TRY(launch_gemm(h, xp, hw.w1, hw.b1, nullptr, a1, n, hw.H1, hw.Fpad, 2, s));
    RELAX_TRY(launch_gemm(h, a1, hw.w2, hw.b2, nullptr, a2, n, hw.H2, hw.H1, 2, s));
    hipLaunchKernelGGL(head_fc3, dim3((n + 127) / 128), dim3(128), 0, s, a2, hw.w3, hw.b3, scores, n, hw.H2);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

}  // extern "C"
