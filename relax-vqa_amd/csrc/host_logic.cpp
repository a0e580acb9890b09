#include "host_logic.h"

#include <cmath>
#include <cstdio>
#include <vector>

namespace relax {
namespace host {

void StateDict::add(const char* name, const float* data, int64_t n, bool strip_module) {
    if (!name) return;
    std::string k(name);
    if (strip_module && k.rfind("module.", 0) == 0) k = k.substr(7);
    t[k] = {data, n};
}

int64_t StateDict::numel(const std::string& key) const {
    auto it = t.find(key);
    return it == t.end() ? -1 : it->second.second;
}

const float* StateDict::get(const std::string& key, int64_t n, std::string& err, const char* what) const {
    char buf[512];
    auto it = t.find(key);
    if (it == t.end()) {
        snprintf(buf, sizeof(buf), "%s: missing key '%s'", what, key.c_str());
        err = buf;
        return nullptr;
    }
    if (n > 0 && it->second.second != n) {
        snprintf(buf, sizeof(buf), "%s: key '%s' has %lld elements, expected %lld", what, key.c_str(),
                 (long long)it->second.second, (long long)n);
        err = buf;
        return nullptr;
    }
    if (!it->second.first) {
        snprintf(buf, sizeof(buf), "%s: key '%s' has a NULL data pointer", what, key.c_str());
        err = buf;
        return nullptr;
    }
    return it->second.first;
}

void fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int channels, float* scale,
             float* shift) {
    for (int o = 0; o < channels; ++o) {
        scale[o] = gamma[o] / std::sqrt(var[o] + eps);
        shift[o] = beta[o] - mean[o] * scale[o];
    }
}

int conv_kpad(int k, int cin_pad) { return ((k * k * cin_pad + 31) / 32) * 32; }

void pack_conv_oihw(const float* w, const float* scale, int cout, int cin, int cin_pad, int k, int kpad, float* out) {
    for (size_t i = 0, n = (size_t)cout * kpad; i < n; ++i) out[i] = 0.f;
    for (int o = 0; o < cout; ++o) {
        const float sc = scale ? scale[o] : 1.f;
        for (int c = 0; c < cin; ++c)
            for (int dy = 0; dy < k; ++dy)
                for (int dx = 0; dx < k; ++dx)
                    out[(size_t)o * kpad + (size_t)(dy * k + dx) * cin_pad + c] = w[(((size_t)o * cin + c) * k + dy) * k + dx] * sc;
    }
}

void fold_fc_bn(const float* w1, const float* b1, const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, int h1, int f, int fpad, float* w1p, float* b1p) {
    for (int o = 0; o < h1; ++o) {
        const float s = gamma[o] / std::sqrt(var[o] + eps);
        for (int i = 0; i < f; ++i) w1p[(size_t)o * fpad + i] = w1[(size_t)o * f + i] * s;
        for (int i = f; i < fpad; ++i) w1p[(size_t)o * fpad + i] = 0.f;
        b1p[o] = (b1[o] - mean[o]) * s + beta[o];
    }
}

TailSplit choose_tail_split(int ntiles, int slots, int nk, int min_steps, bool can_split) {
    TailSplit r{ntiles, 1};
    if (!can_split || ntiles <= 0 || slots <= 0 || min_steps <= 0) return r;
    const int rem = ntiles % slots;
    if (rem == 0) return r;
    int best_s = 1;
    double best = 1.0;
    const int smax = nk / min_steps < 16 ? nk / min_steps : 16;
    for (int S = 2; S <= smax; ++S) {
        const double t = (double)((rem * S + slots - 1) / slots) / S + 0.04 * S;
        if (t < best - 0.05) {
            best = t;
            best_s = S;
        }
    }
    if (best_s >= 2) {
        r.full_tiles = ntiles - rem;
        r.nsplit = best_s;
    }
    return r;
}

float h2_scale_for_bound(double amax) {
    if (!(amax > 0.0) || !(amax < 3.0e38)) return 1.f;
    int e;
    (void)std::frexp(amax, &e);   // amax in [2^(e-1), 2^e)
    int sh = 15 - e;
    sh = sh > 120 ? 120 : (sh < -120 ? -120 : sh);
    return std::ldexp(1.f, sh);
}

void h2_weight_row_scales(const float* W, int rows, int K, float* scale) {
    for (int n = 0; n < rows; ++n) {
        float m = 0.f;
        const float* r = W + (size_t)n * K;
        for (int k = 0; k < K; ++k) {
            const float a = std::fabs(r[k]);
            if (a > m) m = a;     // (a NaN never wins the comparison: such a row keeps the scale of its finite values)
        }
        scale[n] = h2_scale_for_bound((double)m);
    }
}

double layernorm_out_bound(const float* gamma, const float* beta, int dim) {
    const double zmax = std::sqrt((double)(dim > 1 ? dim - 1 : 1));
    double m = 0.0;
    for (int i = 0; i < dim; ++i) {
        const double v = std::fabs((double)gamma[i]) * zmax + std::fabs((double)beta[i]);
        if (v > m) m = v;
    }
    return m;
}

double linear_of_layernorm_bound(const float* W, const float* b, const float* gamma, const float* beta, int dim, int n0, int n1) {
    const double zn = std::sqrt((double)dim);
    double m = 0.0;
    for (int n = n0; n < n1; ++n) {
        const float* r = W + (size_t)n * dim;
        double q = 0.0, c = b ? (double)b[n] : 0.0;
        for (int i = 0; i < dim; ++i) {
            const double gw = (double)gamma[i] * (double)r[i];
            q += gw * gw;
            c += (double)beta[i] * (double)r[i];
        }
        const double v = zn * std::sqrt(q) + std::fabs(c);
        if (v > m) m = v;
    }
    return m;
}

}  // namespace host
}  // namespace relax

#ifdef RELAX_HOST_TEST_API
// ---- C entry points of the host half alone (librelax_host_san.so of tests/test_host_logic_sanitized.py; compiled only
// with -DRELAX_HOST_TEST_API: the product library does not export them) ---------------------------------------------------------------------------------
extern "C" {

// 0 and the folded / packed weights if every key is present with the right size; -1 and a message otherwise.
// out must hold cout * conv_kpad(k, cin_pad) floats, shift cout floats (shift may be NULL when bn is NULL).
int relax_host_pack_conv(const float* const* tensors, const char* const* names, const int64_t* numels, int n, const char* conv,
                         const char* bn, int cout, int cin, int cin_pad, int k, float* out, float* shift, char* err, int err_len) {
    using namespace relax::host;
    StateDict sd;
    for (int i = 0; i < n; ++i) sd.add(names[i], tensors[i], numels[i]);
    std::string e;
    auto fail = [&]() {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", e.c_str());
        return -1;
    };
    const std::string c(conv);
    const float* w = sd.get(c + ".weight", (int64_t)cout * cin * k * k, e);
    if (!w) return fail();
    float* scale = nullptr;
    std::vector<float> storage;
    if (bn && *bn) {
        const std::string b(bn);
        const float* g = sd.get(b + ".weight", cout, e);
        const float* be = g ? sd.get(b + ".bias", cout, e) : nullptr;
        const float* mu = be ? sd.get(b + ".running_mean", cout, e) : nullptr;
        const float* var = mu ? sd.get(b + ".running_var", cout, e) : nullptr;
        if (!var) return fail();
        storage.resize((size_t)cout);
        scale = storage.data();
        fold_bn(g, be, mu, var, 1e-5f, cout, scale, shift);
    }
    pack_conv_oihw(w, scale, cout, cin, cin_pad, k, conv_kpad(k, cin_pad), out);
    return 0;
}

int relax_host_conv_kpad(int k, int cin_pad) { return relax::host::conv_kpad(k, cin_pad); }

int relax_host_fold_fc_bn(const float* const* tensors, const char* const* names, const int64_t* numels, int n, int f, int fpad,
                          float* w1p, float* b1p, int* h1_out, char* err, int err_len) {
    using namespace relax::host;
    StateDict sd;
    for (int i = 0; i < n; ++i) {
        if (names[i] && std::string(names[i]) == "n_averaged") continue;
        sd.add(names[i], tensors[i], numels[i], true);
    }
    std::string e;
    auto fail = [&]() {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", e.c_str());
        return -1;
    };
    const int64_t nw = sd.numel("fc1.weight");
    if (nw <= 0 || f <= 0 || nw % f != 0) {
        e = "mlp head: fc1.weight missing or not [hidden, input_features]";
        return fail();
    }
    const int h1 = (int)(nw / f);
    if (h1_out) *h1_out = h1;
    if (!w1p) return 0;   // size query
    const float* w1 = sd.get("fc1.weight", nw, e, "mlp head state dict");
    const float* b1 = w1 ? sd.get("fc1.bias", h1, e, "mlp head state dict") : nullptr;
    const float* g = b1 ? sd.get("bn1.weight", h1, e, "mlp head state dict") : nullptr;
    const float* be = g ? sd.get("bn1.bias", h1, e, "mlp head state dict") : nullptr;
    const float* mu = be ? sd.get("bn1.running_mean", h1, e, "mlp head state dict") : nullptr;
    const float* var = mu ? sd.get("bn1.running_var", h1, e, "mlp head state dict") : nullptr;
    if (!var) return fail();
    fold_fc_bn(w1, b1, g, be, mu, var, 1e-5f, h1, f, fpad, w1p, b1p);
    return 0;
}

void relax_host_tail_split(int ntiles, int slots, int nk, int min_steps, int can_split, int* full_tiles, int* nsplit) {
    const relax::host::TailSplit r = relax::host::choose_tail_split(ntiles, slots, nk, min_steps, can_split != 0);
    if (full_tiles) *full_tiles = r.full_tiles;
    if (nsplit) *nsplit = r.nsplit;
}

// the f16x2 scale helpers (csrc/h2.h): scales[rows], and the two bounds
void relax_host_h2_weight_row_scales(const float* w, int rows, int k, float* scale) { relax::host::h2_weight_row_scales(w, rows, k, scale); }
float relax_host_h2_scale_for_bound(double amax) { return relax::host::h2_scale_for_bound(amax); }
double relax_host_layernorm_out_bound(const float* gamma, const float* beta, int dim) { return relax::host::layernorm_out_bound(gamma, beta, dim); }
double relax_host_linear_of_layernorm_bound(const float* w, const float* b, const float* gamma, const float* beta, int dim, int n0, int n1) {
    return relax::host::linear_of_layernorm_bound(w, b, gamma, beta, dim, n0, n1);
}

}  // extern "C"
#endif  // RELAX_HOST_TEST_API
