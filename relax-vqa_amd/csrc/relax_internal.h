// Internal declarations shared by the translation units of librelax_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "relax_hip.h"

namespace relax {

constexpr int kMaxDevices = 64;   // relax_create refuses device ids beyond this (per-device launch-attribute flags)

// ---- error plumbing -------------------------------------------------------------------------
void set_error(relax_handle* h, const char* fmt, ...);

#define RELAX_HIP_CHECK(h, expr)                                                             \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            relax::set_error((h), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RELAX_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)

#define RELAX_REQUIRE(h, cond, ...)                                                          \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            relax::set_error((h), __VA_ARGS__);                                              \
            return RELAX_ERR_INVALID;                                                        \
        }                                                                                    \
    } while (0)

#define RELAX_TRY(expr)                                                                      \
    do {                                                                                     \
        int rc_ = (expr);                                                                    \
        if (rc_ != RELAX_OK) return rc_;                                                     \
    } while (0)

// ---- device buffers -------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// ---- contraction kernel description -----------------------------------------------------------
struct ConvDesc {
    // A operand: NHWC activation, gathered on the fly (implicit GEMM)
    const float* in;
    int Nimg, H, W, Cin;
    int Ho, Wo;
    int KH, KW, stride, pad;
    // B operand: weights [Cout][Kpad], k = (dy*KW+dx)*Cin + c, zero padded to Kpad (% 32 == 0)
    const float* w;
    int Cout, Kpad;
    // epilogue
    const float* bias;      // [Cout] or null
    const float* residual;  // [M, Cout] or null
    float* out;             // [M, Cout]
    int act;                // 0 none, 1 relu, 2 gelu(erf)
    double flops;           // algorithmic FLOPs for the profiler (0 = 2*M*N*KH*KW*Cin)
};

// sp3 ("split planes") operands of the bf16x6 kernel (gemm_x6.hip): every fp32 value as bf16 hi + mid + lo, a row of K
// values = K/16 chunks of [16 hi][16 mid][16 lo] (96 bytes)
struct ConvDescX6 {
    const void* in;         // sp3 NHWC activation [Nimg*H*W][Cin*6 B] (plain GEMM: H = 1, W = M rows of Cin = K values)
    int Nimg, H, W, Cin;
    int Ho, Wo;
    int KH, KW, stride, pad;
    const void* w;          // sp3 weights [Cout][KH*KW*Cin*6 B], k = (dy*KW+dx)*Cin + c  (with in2: [Cout][(Cin + Cin2)*6 B])
    int Cout;
    // optional second activation source of a 1x1 contraction (ResNet downsample branch folded into conv3: one accumulator
    // sums W3 t2 + Wd x): sp3 NHWC [Nimg*H2*W2][Cin2*6 B], sampled at (oy, ox) * stride2; the weight rows are concatenated
    const void* in2;
    int H2, W2, Cin2, stride2;
    int in_f32;             // 1: `in` is plain fp32 [M][Cin] rows (split into planes inside the K loop): 1x1 stride-1, Cout = 64 / 128 only
    // f16x2 on the four-wave tiles (the 3x3 convolutions of layer1 / layer2): `in` and `w` are two fp16 planes (csrc/h2.h: [..][Cin*4 B],
    // [Cout][K*4 B]), image i's activations scaled by 1 / img_in_inv[i], weight row n by 1 / colscale[n]; Cout = 64 / 128, K >= 256
    int in_h2;
    const float* colscale;
    const float* img_in_inv;
    // ... and with in_f32 (1x1, Cout = 64 / 128, K >= 256): `in` stays fp32 rows, `w` is fp16 planes; the rows are split into planes inside the K loop
    // with image i's scale img_in_scale[i] (from its MEASURED maximum: launch_h2_image_scales with la = 1), img_in_inv = 1 / that
    const float* img_in_scale;
    int out_rows, gap_rows; // rows below these limits get the fp32 output / the group sums (0 = all rows)
    bool no_split;          // never cut tail tiles along K (a launch whose bits must not depend on which outputs are requested)
    const float* bias;      // [Cout] or null
    const float* residual;  // fp32 [M, Cout] or null
    const void* residual_sp3;  // the residual as split planes instead (exact), or null
    float* gap_groups;      // fused spatial mean, stage 1: sums over aligned 16-row groups [M/16][Cout] (Ho*Wo % 16 != 0: 4-row groups [M/4][Cout]), or null
    float* out;             // fp32 [M, Cout] or null
    void* out_sp3;          // sp3 [M][Cout*6 B] or null (at least one output)
    // the hand-over to the f16x2 convolutions of layer3 (gemm_h2.hip, "per-image scales"): the outputs also as two fp16 planes with
    // image i's rows scaled by img_out_scale[i], and / or the per-image maximum of the (non-negative) outputs; both run unsplit
    void* out_h2;
    const float* img_out_scale;
    unsigned* amax_out;
    int act;                // 0 none, 1 relu, 2 gelu(erf)
    double flops;           // algorithmic FLOPs for the profiler (0 = 2*M*N*K)
    // back-to-back form ("rn_fuse", with in_h2): the 3x3 convolution's output tile (bias + ReLU applied) never leaves the CU - it is the A
    // operand of the block's conv3, a 1x1 onto Cout3 columns with K = Cout.  `bias` is then the 3x3's, act must be 1, and residual /
    // out / out_rows / out_sp3 / out_h2 / img_out_scale / amax_out / gap_groups / gap_rows describe the conv3 output [M][Cout3]
    // (ReLU behind the residual add).  w3: h2 weights [Cout3][Cout*4 B] whose K axis is in the accumulator order of the 3x3's
    // transposed tile (launch_b2b_permute_k), colscale3 their inverse row scales, bias3 [Cout3].
    const void* w3;
    const float* colscale3;
    const float* bias3;
    int Cout3;
    const float* x2;        // back-to-back form of a layer's FIRST block (64-wide): the block input as fp32 rows [M][64] - conv3 then contracts over
                            // K = [the 3x3's 64 channels | these 64] with w3 = [conv3 (permuted K) | downsample (natural K)] rows of 128 and bias3 = the sum of
                            // the two shifts; no residual
    int sp3_sub;            // back-to-back form: 2 = out_sp3 receives only the pixels with even (oy, ox), as compact [Nimg][Ho/2][Wo/2] rows - the sample the
                            // next block's stride-2 downsample branch reads (the full map travels as fp32 rows); 0 / 1 = every pixel
};

// h2 ("two fp16 planes", csrc/h2.h) operands of the f16x2 kernel (gemm_h2.hip): plain GEMM, N % 256 == 0
struct GemmDescH2 {
    const void* a;          // h2 activations [M][K*4 B], values scaled by a power of two
    const void* w;          // h2 weights [N][K*4 B], row n scaled by 2^t_n
    const float* colscale;  // [N]: the inverse of (weight row scale x static activation scale)
    const float* rowscale;  // [M]: the inverse of a per-row activation scale, or null
    const float* bias;      // [N] or null
    const float* residual;  // fp32 [M][N] or null
    float* out;             // fp32 [M][N] or null
    void* out_h2;           // h2 [M][N*4 B] or null (at least one output), values scaled by out_scale (a power of two from a bound)
    float out_scale;
    int M, N, K;
    int act;                // 0 none, 1 relu, 2 gelu(erf)
    bool no_split;
    // convolution form (gemm_h3 as implicit GEMM; ResNet-50 layer3 / layer4): a = NHWC planes [Nimg*H*W][Cin*4 B], M = Nimg*Ho*Wo,
    // K = KH*KW*Cin, weight rows k = (dy*KW + dx)*Cin + c; pixels = 0: plain GEMM
    int pixels, Nimg, H, W, Cin, Ho, Wo, KH, KW, stride, pad;
    // per-image scales (image = row / rows_per_img) and the per-image maximum of the outputs (see h2_image_scales)
    int rows_per_img;
    const float* img_in_inv;
    const float* img_out_scale;
    unsigned* amax_out;
    const void* residual_h2;      // the residual as fp16 planes [M][N*4 B] with img_res_inv[image] (instead of `residual`), or null
    const float* img_res_inv;
    float* gap_groups;            // fused spatial mean, stage 1: sums over aligned 4-row groups [M/4][N], or null
    int out_rows, gap_rows;       // rows below these limits get the fp32 output / the group sums (0 = all rows)
};

// ---- model weights ------------------------------------------------------------------------------
struct ConvW {          // one folded conv (+BN) of ResNet-50
    float* w = nullptr;     // device [Cout][Kpad]
    void* w_sp3 = nullptr;  // the same as split planes (bf16x6 kernel); conv1: [64][224], k = ky*32 + kx*3 + c (conv1_x6.hip)
    void* w_h2 = nullptr;   // layer3 / layer4: the same as two fp16 planes, row n scaled by 2^t_n (gemm_h2.hip)
    float* w_inv = nullptr; // [Cout]: 2^-t_n
    void* w_h2p = nullptr;  // conv3 of a fused layer1 / layer2 block: fp16 planes with the K axis permuted for the back-to-back form (gemm_x6.hip, B2B)
    float* w_invp = nullptr;
    float l1max = 0.f;      // max_n sum_k |W[n,k]| of the folded weights and max_n |bias[n]|: Hoelder bound of the outputs from the
    float bmax = 0.f;       // measured maximum of the inputs (per-image scales, gemm_h2.hip)
    float* bias = nullptr;  // device [Cout] (null for the raw conv1)
    int Cin = 0, Cout = 0, KH = 1, KW = 1, stride = 1, pad = 0, Kpad = 0;
};

struct Bottleneck {
    ConvW c1, c2, c3, down;
    // bf16x6 path, blocks with a downsample branch: conv3 and the downsample convolution as ONE contraction over the concatenated
    // K = [conv2 output | block input] (weights [Cout][(width + Cin) * 6 B] as split planes, bias = the sum of the two folded shifts)
    void* c3d_w_sp3 = nullptr;
    float* c3d_bias = nullptr;
    void* c3d_w_h2p = nullptr;   // back-to-back form of a 64-wide first block: [conv3 (K permuted) | downsample] rows as fp16 planes + their inverse row scales
    float* c3d_w_invp = nullptr;
    bool has_down = false;
    int tap = -1;  // layer-stack tap index, -1 if not tapped
};

struct ResNet50W {
    bool loaded = false;
    ConvW conv1;                 // Cin padded 3 -> 4, K 196 -> 224, no bias (raw tap precedes bn1)
    float* bn1_scale = nullptr;  // [64]
    float* bn1_shift = nullptr;  // [64]
    std::vector<Bottleneck> blocks;  // 16
    std::vector<void*> allocs;
};

struct LinearW {
    float* w = nullptr;  // device [out][in]
    void* w_sp3 = nullptr;  // the same matrix as split planes (bf16 hi + mid + lo, gemm_x6.hip), made once at load time
    void* w_h2 = nullptr;   // the same matrix as two fp16 planes, row n scaled by 2^t_n (gemm_h2.hip), made once at load time
    float* colscale = nullptr;   // [out]: 2^-t_n / (static scale of the activation tensor this layer reads)
    float* b = nullptr;  // device [out]
    int in = 0, out = 0;
};

struct VitBlockW {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    LinearW qkv, proj, fc1, fc2;
    // f16x2: static power-of-two scales of the activation tensors that travel as fp16 planes (host_logic.h: from bounds that hold
    // for every input): LayerNorm 1 output, attention output, LayerNorm 2 output, GELU(fc1) output
    float s_ln1 = 1.f, s_att = 1.f, s_ln2 = 1.f, s_hid = 1.f;
    float s_qkv = 1.f;   // ... and the whole qkv output (attention_h2.hip reads q, k, v as planes)
};

struct VitW {
    bool loaded = false;
    int dim = 0, depth = 0, heads = 0;
    float* cls = nullptr;   // [dim]
    float* pos = nullptr;   // [197][dim]
    LinearW patch;          // [dim][3*16*16], k = c*256 + py*16 + px (c in RGB order)
    std::vector<VitBlockW> blocks;
    float *norm_g = nullptr, *norm_b = nullptr;
    std::vector<void*> allocs;
};

// ---- quality head (imputer + scaler + MLP), BatchNorm folded into fc1 ----------------------------------------
struct HeadW {
    bool loaded = false;
    int F = 0, Fpad = 0, H1 = 0, H2 = 0;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *w3 = nullptr;
    float b3 = 0.f;
    double *stats = nullptr, *scale = nullptr, *mn = nullptr;
    std::vector<void*> allocs;
};

// ---- resize coefficient tables (Pillow-exact), cached per (input size, filter) ------------------------------
struct ResizeTable {
    int in_size = 0, filt = 0, ksize = 0;
    int32_t* bounds = nullptr;  // device [224][2]: first tap, tap count
    int32_t* coeffs = nullptr;  // device [224][ksize]: 22-bit fixed point
};

// ---- event profiling ----------------------------------------------------------------------------
struct ProfSpan {
    hipEvent_t start, stop;
    double work;
    double bytes;  // algorithmic HBM bytes of the launch (contraction kernel), 0 otherwise
    int kind;
    bool ended;    // prof_end has re-recorded `stop` (an open span - the Farneback stage around its kernels' spans - is never reaped)
};

struct Profiler {
    bool on = false;
    std::vector<ProfSpan> spans;   // span_idx handed out by prof_begin = span_base + position here
    int span_base = 0;             // spans reaped so far (prof_begin folds finished spans into the totals once kReapAt have piled up:
                                   // a pass of thousands of clips with the profiler on keeps a bounded number of HIP events)
    static constexpr size_t kReapAt = 2048;
    std::vector<hipEvent_t> pool;
    // span kinds: 0 fp32 / bf16x3 contraction, 1 patch score, 2 bf16x6 contraction, 3 Farneback iteration kernel, 4 the whole Farneback stage,
    // 5 f16x2 contraction, convolution form (gemm_h2.hip, gemm_x6<H2>, conv1_x6<H2>, the fused blocks), 6 f16x2 plain GEMMs (the ViT's: the dominant kernel)
    static constexpr int kKinds = 7;
    double total_ms[kKinds] = {};
    double total_work[kKinds] = {};
    double total_bytes[kKinds] = {};
    int64_t launches[kKinds] = {};
};

}  // namespace relax

namespace relax {
// Tuning / reproducibility switches of the contraction kernel (relax_set_option; env defaults RELAX_GEMM_*).
struct GemmOptions {
    int precision = 3; // "gemm_precision": 3 = f16x2 (default,  fp32-grade: two fp16 planes, four products in two MFMAs - gemm_h2.hip - for the
                       // plain GEMMs with N % 256 == 0, i.e. the whole ViT; everything else as under 2), 2 = bf16x6 (fp32-grade), 0 = exact fp32
                       // MFMA, 1 = bf16x3 split products (~1e-5 relative)
    int h2_stages = 3; // "h2_stages": LDS stages of the 16-k form of the f16x2 kernel (3 or 4; same bits)
    int h2_form = 1;   // "h2_form": 1 = 32-k steps, three products (al bl dropped) for K >= 256, four below; 0 = 16-k steps, four products;
                       // 2 = 32-k steps, four products
    int split_k = 1;   // "gemm_split_k": tail split-K on (1) / off (0: K sums are batch-invariant bit for bit)
    int variant = -1;  // "gemm_variant": exact-fp32 kernel only: pin the tile variant for N % 128 == 0 problems, -1 = automatic
    int variant_n64 = -1;  // "gemm_variant_n64": same for N % 128 != 0 (N = 64 layers)
    int group_m = 8;   // "gemm_group_m": row-tiles per L2 group
    int flow_max_pairs = 0;  // "flow_max_pairs": cap on the pairs per optical-flow chunk (0 = by workspace size only)
    int flow_seg_rows = 0;       // "flow_seg_rows": rows per block of the Farneback iteration kernels, 0 = by the level's geometry (tests: the segmentation
                                 // restarts the running column sums - the flow must not depend on it)
    int flow_pyramid_fused = 1;  // "flow_pyramid_fused": 1 = the four pyramid-level inputs of a frame in one pass over its bytes (pyramid_fused; frames whose
                                 // height and width are multiples of 8), 0 = gray plane + per-level blur / resize kernels (any size) - same bits
    int flow_fused = 1;      // "flow_fused": 1 = one kernel per Farneback iteration (flow_iteration: M never leaves the chip); 0 = update_matrices_k +
                             // box_solve_fused (M through HBM) - same bits, the A/B switch of a test
    int rn_h2 = 1;         // "rn_h2": under "gemm_precision" 3, ResNet-50's layer3 / layer4 (the matrix-pipe-bound third of its time) run f16x2 with
                           // per-image scales; 0 = the whole network on bf16x6 (the A/B switch of a test)
    int att_h2 = 1;        // "att_h2": under "gemm_precision" 3 the ViT's attention runs on fp16 planes too (attention_h2.hip: the qkv GEMM writes planes,
                           // three products, no conversion passes); 0 = attention_x6 on the fp32 qkv output (three bf16 planes, six products)
    int rn_h2_early = 1;   // "rn_h2_early": with "rn_h2", the stem and the 3x3 convolutions of layer1 / layer2 (the MFMA-bound launches in front of layer3) run f16x2 too,
                           // on the four-wave tiles of gemm_x6.hip (conv1 writes its output as fp16 planes with the image's Hoelder scale); 0 = bf16x6 there
    int rn_c1_h2 = 1;      // "rn_c1_h2": with "rn_h2_early", the conv1 (1x1) of the layer1 / layer2 blocks whose input travels as fp32 rows runs f16x2 too: the rows
                           // are split into two fp16 planes in the K loop with the image's scale (from its measured maximum); 0 = bf16x6 (three planes, six products)
    int rn_fuse = 1;       // "rn_fuse": with "rn_h2_early", the blocks of layer1 / layer2 without a downsample branch run conv2 and conv3 back to back in ONE
                           // launch (the 3x3's output tile stays in registers as the A operand of the 1x1: no write and re-read of it, conv3 on f16x2
                           // with one scale per pixel row); 0 = two launches, conv3 on bf16x6 (the A/B switch of a test)
    int b2b_rows = 256;    // "b2b_rows": rows per tile of the back-to-back launches: 256 (two workgroups of four waves per CU) or 128 (three: measured
                           // slower, 2.49 against 2.13 ms per launch of layer1) - same bits
    int fp32_rows = 1;     // "x6_fp32_rows": bf16x6 contractions onto 64 / 128 columns take fp32 activation rows and split them in the K loop
                           // (ResNet-50 layer1 / layer2 block outputs travel as fp32); 0 = split planes everywhere (same bits, more bytes: the A/B switch of a test)
    int debug_poison = 0;  // "debug_poison": fill every workspace with 0xFF bytes when it is requested (test mode: reads of unwritten workspace surface as NaN)
};
}  // namespace relax

struct relax_handle {
    int device = 0;
    relax::GemmOptions gemm;
    std::string last_error;
    relax::DevBuf arena;        // activation workspace shared by both backbones
    int reserved_images = 0;
    relax::DevBuf scratch;      // stage-A scratch (scores)
    relax::DevBuf splitk_ws;    // split-K partial tiles of the contraction kernel
    relax::DevBuf sp3_ws;       // operand conversions of the operator-level entry points under "bf16x6"
    relax::DevBuf resize_ws;    // uint8 intermediates of the two-pass resize
    relax::DevBuf flow_ws;      // optical-flow pyramid workspace
    relax::DevBuf head_ws;      // scaled features + hidden activations of the quality head
    relax::HeadW head;
    std::vector<relax::ResizeTable> resize_tables;
    relax::ResNet50W rn;
    relax::VitW vit;
    relax::Profiler prof;
};

namespace relax {

int ensure_buf(relax_handle* h, DevBuf& b, size_t bytes);
int upload(relax_handle* h, const float* host, size_t n, float** dev, std::vector<void*>& allocs);

// profiling helpers: call around a launch; no-ops when profiling is off
int prof_begin(relax_handle* h, hipStream_t s, int kind, double work, int* span_idx, double bytes = 0);
int prof_end(relax_handle* h, hipStream_t s, int span_idx);
void prof_set_work(relax_handle* h, int span_idx, double work);   // the work of an open span, known only at its end (span ids are not vector positions)
void prof_abort(relax_handle* h, int span_idx);   // a launch failed between begin and end: drop the half-recorded span

// contraction kernel launcher (gemm.hip)
int launch_conv(relax_handle* h, const ConvDesc& d, hipStream_t s);
inline int launch_gemm(relax_handle* h, const float* A, const float* W, const float* bias, const float* residual,
                       float* out, int M, int N, int K, int act, hipStream_t s) {
    ConvDesc d{};
    d.in = A; d.Nimg = 1; d.H = 1; d.W = M; d.Cin = K; d.Ho = 1; d.Wo = M;
    d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
    d.w = W; d.Cout = N; d.Kpad = K; d.bias = bias; d.residual = residual; d.out = out; d.act = act;
    return launch_conv(h, d, s);
}

// bf16x6 contraction kernel (gemm_x6.hip)
int launch_conv_x6(relax_handle* h, const ConvDescX6& d, hipStream_t s);
int launch_b2b_permute_k(relax_handle* h, const float* w, float* wp, int rows, int K, hipStream_t s);   // gemm_x6.hip: K order of the back-to-back form
int launch_to_sp3(relax_handle* h, const float* x, int64_t ld, void* y, int64_t rows, int K, hipStream_t s);
// conv1_x6.hip: ResNet-50 conv1 on the bf16x6 arithmetic, straight from the uint8 fragments
int make_conv1_x6_weights(relax_handle* h, const float* w_packed, int kpad, void** w_sp3_out, std::vector<void*>& allocs);
int launch_conv1_x6(relax_handle* h, const uint8_t* frags, const void* w_sp3, float* out, float* gap_groups, int N, hipStream_t s,
                    const float* w_inv = nullptr);   // w_inv: `w_sp3` holds fp16 planes and the f16x2 form runs
int make_conv1_h2_weights(relax_handle* h, const float* w_packed, int kpad, void** w_h2_out, float** w_inv_out, std::vector<void*>& allocs);
inline int launch_gemm_x6(relax_handle* h, const void* A_sp3, const void* W_sp3, const float* bias, const float* residual,
                          float* out, void* out_sp3, int M, int N, int K, int act, hipStream_t s) {
    ConvDescX6 d{};
    d.in = A_sp3; d.Nimg = 1; d.H = 1; d.W = M; d.Cin = K; d.Ho = 1; d.Wo = M;
    d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
    d.w = W_sp3; d.Cout = N; d.bias = bias; d.residual = residual; d.out = out; d.out_sp3 = out_sp3; d.act = act;
    return launch_conv_x6(h, d, s);
}

// f16x2 contraction kernel (gemm_h2.hip)
int launch_gemm_h2(relax_handle* h, const GemmDescH2& d, hipStream_t s);
int launch_to_h2(relax_handle* h, const float* x, int64_t ld, void* y, int64_t rows, int K, float scale, const float* row_scale, hipStream_t s,
                 int rows_per_scale = 1);
int launch_h2_image_scales(relax_handle* h, const unsigned* amax_a, float la, const unsigned* amax_b, float lb, const unsigned* amax_r,
                           float bmax, float* scale, float* inv, int n, hipStream_t s);
int launch_image_absmax(relax_handle* h, const float* x, int64_t per_image, int n_images, unsigned* amax, hipStream_t s);
int launch_to_h2_rows(relax_handle* h, const float* x, int64_t ld, void* y, int rows, int K, float* inv_scale, hipStream_t s);

// small kernels (layers.hip)
int launch_layernorm(relax_handle* h, const float* x, const float* g, const float* b, float* y, int rows, int dim,
                     float eps, hipStream_t s);
int launch_layernorm_sp3(relax_handle* h, const float* x, const float* g, const float* b, void* y_sp3, int rows, int dim,
                         float eps, hipStream_t s);
int launch_attention(relax_handle* h, const float* qkv, float* out, int Nimg, int heads, hipStream_t s);
int launch_layernorm_h2(relax_handle* h, const float* x, const float* g, const float* b, void* y_h2, float scale, int rows, int dim,
                        float eps, hipStream_t s);
// out_planes: split planes (out_h2_scale == 0) or two fp16 planes scaled by out_h2_scale (> 0)
int launch_attention_x6(relax_handle* h, const float* qkv, float* out, void* out_planes, int Nimg, int heads, hipStream_t s,
                        float out_h2_scale = 0.f);
// attention on the fp16 planes the qkv GEMM wrote (attention_h2.hip): qkv_planes [Nimg*197][3*dim*4 B] of qkv * s_qkv -> out_planes [..][dim*4 B] of out * out_scale
int launch_attention_h2(relax_handle* h, const void* qkv_planes, float s_qkv, void* out_planes, float out_scale, int Nimg, int heads, hipStream_t s);
int launch_attention_h2_op(relax_handle* h, const float* qkv, float* out, int Nimg, int heads, hipStream_t s);   // fp32 in / out (relax_op_attention)
int launch_bn_relu_maxpool(relax_handle* h, const float* x, const float* scale, const float* shift, float* y,
                           int Nimg, int H, int W, int C, hipStream_t s);
int launch_bn_relu_maxpool_f32(relax_handle* h, const float* x, const float* scale, const float* shift, float* y, int Nimg, int H, int W, int C,
                               hipStream_t s, unsigned* amax_out, unsigned* block_ws);   // fp32 rows out + the per-image maxima
int launch_bn_relu_maxpool_sp3(relax_handle* h, const float* x, const float* scale, const float* shift, void* y_sp3,
                               int Nimg, int H, int W, int C, hipStream_t s, unsigned* amax_out = nullptr, unsigned* block_ws = nullptr);
int launch_gap_groups_finish(relax_handle* h, const float* groups, float* out, int Nimg, int HW, int C, int64_t out_stride,
                             hipStream_t s);
int launch_gap(relax_handle* h, const float* x, float* out, int Nimg, int HW, int C, int64_t out_stride,
               hipStream_t s);
int launch_nhwc_to_nchw(relax_handle* h, const float* x, float* y, int Nimg, int HW, int C, hipStream_t s);

// model drivers
void free_resnet(relax_handle* h);
void free_vit(relax_handle* h);
void free_resize(relax_handle* h);
void free_head(relax_handle* h);
size_t resnet_arena_bytes(int n_images);
size_t vit_arena_bytes(const VitW& v, int n_images);

}  // namespace relax
