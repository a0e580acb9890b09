// Pure-host half of librelax_hip.so: everything the model loaders and the launchers compute on the CPU before a byte goes
// to the GPU - state-dict key matching, BatchNorm folding, OIHW -> packed [Cout][K] weight layout, the quality head's
// fc1 + BatchNorm1d fold, and the tail split-K cost model of the contraction launchers.  No HIP type appears here, so the file
// builds with plain g++ and runs under AddressSanitizer / UBSan on a CPU box (tests/test_host_logic_sanitized.py drives it with the
// synthetic and the deliberately malformed state dicts; sanitizers are never run on the GPU).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>

namespace relax {
namespace host {

// name -> (host pointer, element count) of a checkpoint as the C-ABI receives it (relax_load_resnet50 / _vit / _mlp_head)
struct StateDict {
    std::map<std::string, std::pair<const float*, int64_t>> t;
    // strip_module: drop a leading "module." (the reference's fix_state_dict, src/demo_test.py:25-35)
    void add(const char* name, const float* data, int64_t numel, bool strip_module = false);
    // the tensor under `key` if it has exactly `numel` elements (numel <= 0: any size); otherwise nullptr and a message in err
    const float* get(const std::string& key, int64_t numel, std::string& err, const char* what = "state dict") const;
    int64_t numel(const std::string& key) const;   // -1 if absent
};

// eval-mode BatchNorm as y = x * scale + shift:  scale = gamma / sqrt(var + eps),  shift = beta - mean * scale
void fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int channels, float* scale,
             float* shift);

// K of a packed convolution weight row: KH*KW*cin_pad rounded up to a multiple of 32
int conv_kpad(int k, int cin_pad);

// OIHW [cout][cin][k][k] -> [cout][kpad], column (dy*k + dx)*cin_pad + c, rows scaled by scale[o] (nullptr: 1), padding zero.
// out must hold cout * kpad floats.
void pack_conv_oihw(const float* w, const float* scale, int cout, int cin, int cin_pad, int k, int kpad, float* out);

// fc1 [h1][f] + BatchNorm1d(h1) -> w1p [h1][fpad] (zero padded), b1p [h1]:  y = (x W^T + b - mu) * s + beta
void fold_fc_bn(const float* w1, const float* b1, const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, int h1, int f, int fpad, float* w1p, float* b1p);

// Tail split-K of a contraction launch.  `ntiles` output tiles over `slots` resident workgroups: the last, partial round
// (rem = ntiles % slots tiles) would leave most CUs idle, so its tiles may be cut along K into S slices.  Time of the tail in
// rounds = ceil(rem*S/slots)/S plus 4 % of a round per slice for writing and re-reading the partial tiles; S is kept at 1 unless
// splitting wins by more than 5 % of a round.  S <= 16 and every slice keeps at least `min_steps` K steps of the nk.
struct TailSplit {
    int full_tiles;   // tiles that run unsplit (launched first)
    int nsplit;       // slices per remaining tile (1 = no split)
};
TailSplit choose_tail_split(int ntiles, int slots, int nk, int min_steps, bool can_split);

// ---- scales of the two-plane fp16 format (csrc/h2.h): powers of two from RIGOROUS bounds, so no value can leave the fp16 range --------
// The power of two that puts `amax` into [2^14, 2^15) (so twice the bound still fits below 65504); 1 for zero / non-finite.
float h2_scale_for_bound(double amax);
// Per-row weight scales: scale[n] = h2_scale_for_bound(max_k |W[n, k]|), W [rows][K] row-major.
void h2_weight_row_scales(const float* W, int rows, int K, float* scale);
// LayerNorm output bound: y_i = gamma_i z_i + beta_i with |z_i| <= sqrt(dim - 1) for EVERY input row (z has mean 0 and
// sum z^2 <= dim)  ->  max_i (|gamma_i| sqrt(dim - 1) + |beta_i|).
double layernorm_out_bound(const float* gamma, const float* beta, int dim);
// Bound of the outputs n in [n0, n1) of Linear(LayerNorm(x)):  |sum_i z_i gamma_i W[n,i] + (beta . W[n,:] + b[n])| <=
// sqrt(dim) * ||gamma * W[n,:]||_2 + |beta . W[n,:] + b[n]|   (Cauchy-Schwarz with ||z||_2 <= sqrt(dim)); the maximum over the range.
// Everything that is a convex combination or a contraction of such outputs (attention output: rows of V; GELU: |GELU(x)| <= |x|)
// inherits the bound.  W [N][dim] row-major, b may be null.
double linear_of_layernorm_bound(const float* W, const float* b, const float* gamma, const float* beta, int dim, int n0, int n1);

}  // namespace host
}  // namespace relax
