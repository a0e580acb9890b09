// ResNet-50 (torchvision v1.5 layout) feature extractor on gfx950: one forward per fragment yields the
// 15 layer-stack taps' spatial means (13120-d) and the avgpool vector + stats (2051-d).
//
// Reference semantics (file:line in xinyiW915/ReLaX-VQA):
//   src/extractor/visualise_resnet.py:40-50      preprocess: PNG(BGR->RGB), ToTensor (/255), Normalize(mean,std)
//   src/extractor/visualise_resnet.py:21,83-106  resnet50, one hooked forward per tap
//   src/main_fragment_layerstack.py:91-99        tap list; 'resnet50.conv1' is the RAW conv output (before bn1)
//   src/main_fragment_layerstack.py:134-149      spatial mean per tap / avgpool + (mean,max,std)
// Layout: activations NHWC fp32 (channel = GEMM N axis, contiguous), weights [Cout][KH*KW*Cin] with
// eval-mode BatchNorm folded in (scale into the weights, shift as bias); bn1 stays separate because
// the conv1 tap is taken before it.
#include <cmath>

#include "relax_internal.h"
#include "host_logic.h"

namespace relax {

int launch_gap_ws(relax_handle* h, const float* x, float* out, int Nimg, int HW, int C, int64_t out_stride,
                  float* partial_ws, hipStream_t s);

static const int kTapChannels[RELAX_RN50_NUM_TAPS] = {64, 256, 256, 256, 512, 512, 512, 512,
                                                      1024, 1024, 1024, 1024, 2048, 2048, 2048};
static const int kTapHW[RELAX_RN50_NUM_TAPS] = {112, 56, 56, 56, 28, 28, 28, 28, 14, 14, 14, 14, 7, 7, 7};

// ---- kernels ---------------------------------------------------------------------------------------
// uint8 BGR [N,224,224,3] -> fp32 NHWC4 RGB0, ((x/255) - mean) / std   (ToTensor + Normalize)
__global__ __launch_bounds__(256) void rn_preprocess(const uint8_t* __restrict__ frag, float* __restrict__ x,
                                                     int64_t npix) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint8_t* p = frag + i * 3;
    const float b = (float)p[0] / 255.0f, g = (float)p[1] / 255.0f, r = (float)p[2] / 255.0f;
    float4 o;
    o.x = (r - 0.485f) / 0.229f;
    o.y = (g - 0.456f) / 0.224f;
    o.z = (b - 0.406f) / 0.225f;
    o.w = 0.f;
    reinterpret_cast<float4*>(x)[i] = o;
}

// avgpool vector v[2048] -> out[0:2048] = v, out[2048..2050] = mean, max, population std
__global__ __launch_bounds__(256) void rn_pool_stats(const float* __restrict__ avg, int64_t avg_stride,
                                                     float* __restrict__ out) {
    __shared__ float red[256];
    __shared__ float s_mean;
    const int n = blockIdx.x, t = threadIdx.x;
    const float* v = avg + (int64_t)n * avg_stride;
    float* o = out + (int64_t)n * RELAX_RN50_POOL_DIM;
    float vals[8];
    float s = 0.f, m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        vals[j] = v[t + 256 * j];
        o[t + 256 * j] = vals[j];
        s += vals[j];
        m = fmaxf(m, vals[j]);
    }
    red[t] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) s_mean = red[0] / 2048.0f;
    __syncthreads();
    const float mean = s_mean;
    __syncthreads();
    red[t] = m;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) red[t] = fmaxf(red[t], red[t + w]);
        __syncthreads();
    }
    const float mx = red[0];
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) q += (vals[j] - mean) * (vals[j] - mean);
    red[t] = q;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) {
        o[2048] = mean;
        o[2049] = mx;
        o[2050] = sqrtf(red[0] / 2048.0f);
    }
}

// ---- weights ---------------------------------------------------------------------------------------
// (key matching, BatchNorm folding and the OIHW -> [Cout][K] packing are host_logic.cpp: pure C++, sanitizer-tested on the CPU)
using HostSD = host::StateDict;

static const float* sd_get(relax_handle* h, const HostSD& sd, const std::string& k, int64_t numel) {
    std::string err;
    const float* p = sd.get(k, numel, err);
    if (!p) set_error(h, "%s", err.c_str());
    return p;
}

static constexpr float kBnEps = 1e-5f;

// conv (OIHW) [+ BN] -> device [Cout][Kpad] (+ bias).  cin_pad >= cin (conv1: 3 -> 4).
static int make_conv(relax_handle* h, const HostSD& sd, const std::string& conv, const std::string& bn, int cout,
                     int cin, int cin_pad, int k, int stride, int pad, ConvW* out, std::vector<void*>& allocs,
                     std::vector<float>* shift_out = nullptr) {
    const float* w = sd_get(h, sd, conv + ".weight", (int64_t)cout * cin * k * k);
    if (!w) return RELAX_ERR_INVALID;
    std::vector<float> scale(cout, 1.f), shift(cout, 0.f);
    if (!bn.empty()) {
        const float* g = sd_get(h, sd, bn + ".weight", cout);
        const float* b = g ? sd_get(h, sd, bn + ".bias", cout) : nullptr;
        const float* mu = b ? sd_get(h, sd, bn + ".running_mean", cout) : nullptr;
        const float* var = mu ? sd_get(h, sd, bn + ".running_var", cout) : nullptr;
        if (!var) return RELAX_ERR_INVALID;
        host::fold_bn(g, b, mu, var, kBnEps, cout, scale.data(), shift.data());
    }
    const int kpad = host::conv_kpad(k, cin_pad);
    std::vector<float> packed((size_t)cout * kpad);
    host::pack_conv_oihw(w, scale.data(), cout, cin, cin_pad, k, kpad, packed.data());
    out->Cin = cin_pad; out->Cout = cout; out->KH = k; out->KW = k; out->stride = stride; out->pad = pad;
    out->Kpad = kpad;
    // Hoelder constants of the folded convolution: |out[n]| <= l1max * max |in| + bmax  (per-image scales of the f16x2 layers, gemm_h2.hip)
    out->l1max = 0.f;
    out->bmax = 0.f;
    for (int o = 0; o < cout; ++o) {
        double l1 = 0.0;
        for (int kk = 0; kk < kpad; ++kk) l1 += std::fabs((double)packed[(size_t)o * kpad + kk]);
        out->l1max = std::fmax(out->l1max, (float)(l1 * (1.0 + 1e-6)));
        if (!bn.empty()) out->bmax = std::fmax(out->bmax, std::fabs(shift[o]));
    }
    RELAX_TRY(upload(h, packed.data(), packed.size(), &out->w, allocs));
    if (!bn.empty()) RELAX_TRY(upload(h, shift.data(), shift.size(), &out->bias, allocs));
    else out->bias = nullptr;
    if (shift_out) *shift_out = shift;
    return RELAX_OK;
}

void free_resnet(relax_handle* h) {
    for (void* p : h->rn.allocs) (void)hipFree(p);
    h->rn = ResNet50W();
}

// floats per image of the activation arena (see relax_resnet50_features)
static constexpr size_t kX0 = 224 * 224 * 4;        // preprocessed input, NHWC4
static constexpr size_t kBig = 112 * 112 * 64;      // == 56*56*256: largest block in/out and conv1 raw
static constexpr size_t kT1 = 56 * 56 * 128;        // largest conv1-of-block output (layer2.0 before the stride)
static constexpr size_t kT2 = 56 * 56 * 64;         // largest conv2 output
static constexpr size_t kGapWs = 196 * 256;         // GAP partial sums (16 x 2048 two-stage kernel; 3136/16 x 256 fused group sums)
static constexpr size_t kAvg = 2048;
static constexpr size_t kFirstH2Block = 7;          // layer3[0]: from here on the blocks run f16x2 under "gemm_precision" 3 (with "rn_h2")
static constexpr size_t kImgSlots = 64;             // per-image tables of the f16x2 blocks: slot t = {maximum, scale, 1 / scale} x images
static constexpr size_t kRnFloatsPerImage = kX0 + 3 * kBig + kT1 + kT2 + kGapWs + kAvg;
// bf16x6 path: block inputs / outputs exist twice (fp32 for the residual add and the taps, split planes = 1.5 floats per
// value for the next convolutions), the intermediates of a block only as split planes
static constexpr size_t kBlockMax = 256;            // block maxima of the max-pool (196 blocks of 16 output pixels per image)
static constexpr size_t kRnFloatsPerImageX6 = kX0 + 3 * kBig + 2 * (kBig * 3 / 2) + (kT1 + kT2) * 3 / 2 + kGapWs + kAvg + 3 * kImgSlots + kBlockMax;

size_t resnet_arena_bytes(int n) {
    return sizeof(float) * (kRnFloatsPerImage > kRnFloatsPerImageX6 ? kRnFloatsPerImage : kRnFloatsPerImageX6) * (size_t)n;
}

static int run_conv_x6(relax_handle* h, const ConvW& c, const void* in_sp3, int Nimg, int H, int W, const float* residual,
                       float* out, void* out_sp3, int act, hipStream_t s, const void* residual_sp3 = nullptr,
                       float* gap_groups = nullptr) {
    ConvDescX6 d{};
    d.in = in_sp3; d.Nimg = Nimg; d.H = H; d.W = W; d.Cin = c.Cin;
    d.Ho = (H + 2 * c.pad - c.KH) / c.stride + 1;
    d.Wo = (W + 2 * c.pad - c.KW) / c.stride + 1;
    d.KH = c.KH; d.KW = c.KW; d.stride = c.stride; d.pad = c.pad;
    d.w = c.w_sp3; d.Cout = c.Cout;
    d.bias = c.bias; d.residual = residual; d.out = out; d.out_sp3 = out_sp3; d.act = act;
    d.residual_sp3 = residual_sp3; d.gap_groups = gap_groups;
    return launch_conv_x6(h, d, s);
}

static int run_conv(relax_handle* h, const ConvW& c, const float* in, int Nimg, int H, int W, const float* residual,
                    float* out, int act, hipStream_t s, double flops = 0) {
    ConvDesc d{};
    d.in = in; d.Nimg = Nimg; d.H = H; d.W = W; d.Cin = c.Cin;
    d.Ho = (H + 2 * c.pad - c.KH) / c.stride + 1;
    d.Wo = (W + 2 * c.pad - c.KW) / c.stride + 1;
    d.KH = c.KH; d.KW = c.KW; d.stride = c.stride; d.pad = c.pad;
    d.w = c.w; d.Cout = c.Cout; d.Kpad = c.Kpad;
    d.bias = c.bias; d.residual = residual; d.out = out; d.act = act; d.flops = flops;
    return launch_conv(h, d, s);
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_load_resnet50(relax_handle* h, const float* const* tensors, const char* const* names,
                        const int64_t* numels, int n) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, tensors && names && numels && n > 0, "relax_load_resnet50: bad arguments");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    free_resnet(h);
    HostSD sd;
    for (int i = 0; i < n; ++i) sd.add(names[i], tensors[i], numels[i]);
    ResNet50W& rn = h->rn;
    int rc = make_conv(h, sd, "conv1", "", 64, 3, 4, 7, 2, 3, &rn.conv1, rn.allocs);
    if (rc != RELAX_OK) { free_resnet(h); return rc; }
    {
        const float* g = sd_get(h, sd, "bn1.weight", 64);
        const float* b = g ? sd_get(h, sd, "bn1.bias", 64) : nullptr;
        const float* mu = b ? sd_get(h, sd, "bn1.running_mean", 64) : nullptr;
        const float* var = mu ? sd_get(h, sd, "bn1.running_var", 64) : nullptr;
        if (!var) { free_resnet(h); return RELAX_ERR_INVALID; }
        std::vector<float> sc(64), sh(64);
        host::fold_bn(g, b, mu, var, kBnEps, 64, sc.data(), sh.data());
        rc = upload(h, sc.data(), 64, &rn.bn1_scale, rn.allocs);
        if (rc == RELAX_OK) rc = upload(h, sh.data(), 64, &rn.bn1_shift, rn.allocs);
        if (rc != RELAX_OK) { free_resnet(h); return rc; }
    }
    const int stage_blocks[4] = {3, 4, 6, 3};
    const int stage_width[4] = {64, 128, 256, 512};
    const int stage_taps[4] = {3, 4, 4, 3};  // layer3 blocks 4,5 are not tapped
    int cin = 64, tap = 1;
    for (int st = 0; st < 4; ++st) {
        for (int b = 0; b < stage_blocks[st]; ++b) {
            Bottleneck blk;
            const int width = stage_width[st];
            const int stride = (b == 0 && st > 0) ? 2 : 1;
            char pfx[64];
            snprintf(pfx, sizeof(pfx), "layer%d.%d", st + 1, b);
            const std::string p(pfx);
            rc = make_conv(h, sd, p + ".conv1", p + ".bn1", width, cin, cin, 1, 1, 0, &blk.c1, rn.allocs);
            if (rc == RELAX_OK)
                rc = make_conv(h, sd, p + ".conv2", p + ".bn2", width, width, width, 3, stride, 1, &blk.c2, rn.allocs);
            std::vector<float> shift3, shiftd;
            if (rc == RELAX_OK)
                rc = make_conv(h, sd, p + ".conv3", p + ".bn3", width * 4, width, width, 1, 1, 0, &blk.c3, rn.allocs, &shift3);
            blk.has_down = (b == 0);
            if (rc == RELAX_OK && blk.has_down)
                rc = make_conv(h, sd, p + ".downsample.0", p + ".downsample.1", width * 4, cin, cin, 1, stride, 0,
                               &blk.down, rn.allocs, &shiftd);
            if (rc == RELAX_OK && blk.has_down) {   // bias of the fused conv3 + downsample contraction
                for (size_t o = 0; o < shift3.size(); ++o) shift3[o] += shiftd[o];
                rc = upload(h, shift3.data(), shift3.size(), &blk.c3d_bias, rn.allocs);
            }
            if (rc != RELAX_OK) { free_resnet(h); return rc; }
            blk.tap = b < stage_taps[st] ? tap++ : -1;
            rn.blocks.push_back(blk);
            cin = width * 4;
        }
    }
    // split planes for the bf16x6 kernels (made on the device from the packed fp32 copies): conv1 in its own K layout ...
    rc = make_conv1_x6_weights(h, rn.conv1.w, rn.conv1.Kpad, &rn.conv1.w_sp3, rn.allocs);
    if (rc == RELAX_OK) rc = make_conv1_h2_weights(h, rn.conv1.w, rn.conv1.Kpad, &rn.conv1.w_h2, &rn.conv1.w_inv, rn.allocs);
    if (rc != RELAX_OK) { free_resnet(h); return rc; }
    // ... every other convolution as [Cout][K] rows
    for (Bottleneck& blk : rn.blocks) {
        for (ConvW* c : {&blk.c1, &blk.c2, &blk.c3, &blk.down}) {
            if (!c->w) continue;
            const int K = c->KH * c->KW * c->Cin;
            if (K != c->Kpad || c->Cin % 16 != 0) {
                set_error(h, "resnet50: conv K=%d (padded %d) does not fit the split-plane layout", K, c->Kpad);
                free_resnet(h);
                return RELAX_ERR_INVALID;
            }
            void* q = nullptr;
            if (hipMalloc(&q, (size_t)c->Cout * K * 6) != hipSuccess) {
                set_error(h, "resnet50: hipMalloc of split-plane weights failed");
                free_resnet(h);
                return RELAX_ERR_NOMEM;
            }
            rn.allocs.push_back(q);
            c->w_sp3 = q;
            rc = launch_to_sp3(h, c->w, K, q, c->Cout, K, nullptr);
            if (rc != RELAX_OK) { free_resnet(h); return rc; }
        }
    }
    // ... the convolutions of layer3 / layer4 (blocks 7 .. 15: 256 / 512-wide, every Cout a multiple of 256, every Cin of 32) also as two
    // fp16 planes with one power-of-two scale per output row (gemm_h2.hip; w_inv = the inverse scales, the epilogue's colscale)
    // ... and the 3x3 convolutions of layer1 / layer2 (64 / 128 columns: the four-wave f16x2 form of gemm_x6.hip, "rn_h2_early")
    for (size_t b = 0; b < rn.blocks.size(); ++b) {
        Bottleneck& blk = rn.blocks[b];
        const bool early = b < kFirstH2Block;
        for (ConvW* c : {&blk.c1, &blk.c2, &blk.c3, &blk.down}) {
            // (early blocks: the 3x3 - and conv1, whose fp32 input rows are split in the K loop under "rn_c1_h2")
            // ... and the downsample convolution of a stride-2 first block (layer2[0]), which runs as a launch of its own in front of that block's
            // back-to-back launch: its output is the fused conv3's fp32 residual)
            const bool early_down = early && c == &blk.down && blk.has_down && c->stride == 2 && c->Cin % 16 == 0 && c->Cout % 128 == 0 && c->Cin >= 256;
            if (!c->w || (early && !early_down && ((c != &blk.c2 && c != &blk.c1) || c->Cin % 16 != 0 || c->Cout % 64 != 0 || c->Cout % 256 == 0))) continue;
            const int K = c->KH * c->KW * c->Cin;
            void* q = nullptr;
            float* inv = nullptr;
            if ((!early && (c->Cin % 32 != 0 || c->Cout % 256 != 0)) || hipMalloc(&q, (size_t)c->Cout * K * 4) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&inv), sizeof(float) * (size_t)c->Cout) != hipSuccess) {
                if (q) (void)hipFree(q);
                set_error(h, "resnet50: fp16-plane weights of block %zu (Cin %d, Cout %d) could not be made", b, c->Cin, c->Cout);
                free_resnet(h);
                return RELAX_ERR_NOMEM;
            }
            rn.allocs.push_back(q);
            rn.allocs.push_back(inv);
            c->w_h2 = q;
            c->w_inv = inv;
            rc = launch_to_h2_rows(h, c->w, K, q, c->Cout, K, inv, nullptr);
            if (rc != RELAX_OK) { free_resnet(h); return rc; }
        }
    }
    // ... the conv3 of the layer1 / layer2 blocks without a downsample branch once more as fp16 planes with the K axis in the order of the
    // back-to-back form ("rn_fuse": gemm_x6.hip, B2B - the 3x3's transposed accumulator tile is the A operand)
    for (size_t b = 0; b < kFirstH2Block && b < rn.blocks.size(); ++b) {
        Bottleneck& blk = rn.blocks[b];
        // (64-wide blocks: the four-wave form; 128-wide: eight waves; a 128-wide FIRST block too - layer2[0]: its downsample branch arrives as a residual)
        if ((blk.has_down && !(blk.c3.Cin == 128 && blk.down.stride == 2)) || (blk.c3.Cin != 64 && blk.c3.Cin != 128)) continue;
        const int K = blk.c3.Cin, Co = blk.c3.Cout;
        float* perm = nullptr;
        void* q = nullptr;
        float* inv = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&perm), sizeof(float) * (size_t)Co * K) != hipSuccess ||
            hipMalloc(&q, (size_t)Co * K * 4) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&inv), sizeof(float) * (size_t)Co) != hipSuccess) {
            if (perm) (void)hipFree(perm);
            if (q) (void)hipFree(q);
            set_error(h, "resnet50: hipMalloc of the back-to-back conv3 weights failed");
            free_resnet(h);
            return RELAX_ERR_NOMEM;
        }
        rn.allocs.push_back(q);
        rn.allocs.push_back(inv);
        rc = launch_b2b_permute_k(h, blk.c3.w, perm, Co, K, nullptr);
        if (rc == RELAX_OK) rc = launch_to_h2_rows(h, perm, K, q, Co, K, inv, nullptr);
        if (rc == RELAX_OK && hipDeviceSynchronize() != hipSuccess) rc = RELAX_ERR_HIP;
        (void)hipFree(perm);
        if (rc != RELAX_OK) {
            set_error(h, "resnet50: building the back-to-back conv3 weights failed");
            free_resnet(h);
            return rc;
        }
        blk.c3.w_h2p = q;
        blk.c3.w_invp = inv;
    }
    // ... and for a 64-wide FIRST block (layer1[0]: downsample without a stride, 64 input channels) [conv3 (K permuted) | downsample (natural K)] rows of
    // 128 as fp16 planes: its back-to-back form contracts conv3 and the downsample convolution in one accumulator (gemm_x6.hip, B2B == 2)
    for (size_t b = 0; b < kFirstH2Block && b < rn.blocks.size(); ++b) {
        Bottleneck& blk = rn.blocks[b];
        if (!blk.has_down || blk.c3.Cin != 64 || blk.down.Cin != 64 || blk.down.stride != 1 || blk.c2.stride != 1) continue;
        const int Co = blk.c3.Cout;
        float *perm = nullptr, *cat = nullptr, *inv = nullptr;
        void* q = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&perm), sizeof(float) * (size_t)Co * 64) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&cat), sizeof(float) * (size_t)Co * 128) != hipSuccess ||
            hipMalloc(&q, (size_t)Co * 128 * 4) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&inv), sizeof(float) * (size_t)Co) != hipSuccess) {
            if (perm) (void)hipFree(perm);
            if (cat) (void)hipFree(cat);
            if (q) (void)hipFree(q);
            set_error(h, "resnet50: hipMalloc of the two-source back-to-back weights failed");
            free_resnet(h);
            return RELAX_ERR_NOMEM;
        }
        rn.allocs.push_back(q);
        rn.allocs.push_back(inv);
        rc = launch_b2b_permute_k(h, blk.c3.w, perm, Co, 64, nullptr);
        hipError_t e = rc == RELAX_OK ? hipDeviceSynchronize() : hipErrorUnknown;
        if (e == hipSuccess) e = hipMemcpy2D(cat, sizeof(float) * 128, perm, sizeof(float) * 64, sizeof(float) * 64, Co, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy2D(cat + 64, sizeof(float) * 128, blk.down.w, sizeof(float) * 64, sizeof(float) * 64, Co, hipMemcpyDeviceToDevice);
        rc = e == hipSuccess ? launch_to_h2_rows(h, cat, 128, q, Co, 128, inv, nullptr) : RELAX_ERR_HIP;
        if (rc == RELAX_OK && hipDeviceSynchronize() != hipSuccess) rc = RELAX_ERR_HIP;
        (void)hipFree(perm);
        (void)hipFree(cat);
        if (rc != RELAX_OK) {
            set_error(h, "resnet50: building the two-source back-to-back weights failed");
            free_resnet(h);
            return rc;
        }
        blk.c3d_w_h2p = q;
        blk.c3d_w_invp = inv;
    }
    // ... and, for the four blocks with a downsample branch, [conv3 | downsample] rows side by side
    for (Bottleneck& blk : rn.blocks) {
        if (!blk.has_down) continue;
        const int K1 = blk.c3.Cin, K2 = blk.down.Cin, Co = blk.c3.Cout;
        float* cat = nullptr;
        void* q = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&cat), sizeof(float) * (size_t)Co * (K1 + K2)) != hipSuccess ||
            hipMalloc(&q, (size_t)Co * (K1 + K2) * 6) != hipSuccess) {
            if (cat) (void)hipFree(cat);
            set_error(h, "resnet50: hipMalloc of the fused conv3 + downsample weights failed");
            free_resnet(h);
            return RELAX_ERR_NOMEM;
        }
        rn.allocs.push_back(q);
        blk.c3d_w_sp3 = q;
        hipError_t e = hipMemcpy2D(cat, sizeof(float) * (K1 + K2), blk.c3.w, sizeof(float) * K1, sizeof(float) * K1, Co, hipMemcpyDeviceToDevice);
        if (e == hipSuccess)
            e = hipMemcpy2D(cat + K1, sizeof(float) * (K1 + K2), blk.down.w, sizeof(float) * K2, sizeof(float) * K2, Co, hipMemcpyDeviceToDevice);
        rc = e == hipSuccess ? launch_to_sp3(h, cat, K1 + K2, q, Co, K1 + K2, nullptr) : RELAX_ERR_HIP;
        if (rc == RELAX_OK && hipDeviceSynchronize() != hipSuccess) rc = RELAX_ERR_HIP;
        (void)hipFree(cat);
        if (rc != RELAX_OK) {
            set_error(h, "resnet50: building the fused conv3 + downsample weights failed");
            free_resnet(h);
            return rc;
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        set_error(h, "resnet50: weight conversion failed");
        free_resnet(h);
        return RELAX_ERR_HIP;
    }
    rn.loaded = true;
    return RELAX_OK;
}

// One forward over N images.  Images [0, n_ls) get layer-stack rows (n_ls = 0: none), images [pool_from, N) get pool rows
// (pool == nullptr: none); taps_nchw (all N images) as in relax_resnet50_features.
static int resnet_forward(relax_handle* h, const uint8_t* frags, int N, int n_ls, int pool_from, float* layer_stack, float* pool,
                          float* const* taps_nchw, hipStream_t s) {
    RELAX_TRY(ensure_buf(h, h->arena, resnet_arena_bytes(N)));
    float* base = static_cast<float*>(h->arena.p);
    const size_t n = (size_t)N;
    float* X0 = base;
    float* bufA = X0 + kX0 * n;
    float* bufB = bufA + kBig * n;
    float* bufD = bufB + kBig * n;
    float* T1 = bufD + kBig * n;
    float* T2 = T1 + kT1 * n;
    float* gapws = T2 + kT2 * n;
    float* avg = gapws + kGapWs * n;
    const ResNet50W& rn = h->rn;
    if (n_ls == 0) layer_stack = nullptr;
    const int n_pool = pool ? N - pool_from : 0;
    // the pool vector of an image that is also in the layer stack is the last 2048 columns of its layer-stack row
    const bool pool_from_stack = pool && layer_stack && pool_from == 0 && n_ls == N;

    auto tap_offset = [](int tap) {
        int off = 0;
        for (int t = 0; t < tap; ++t) off += kTapChannels[t];
        return off;
    };
    auto pool_tail = [&](const float* last32, float* avg_ws, float* gap_scratch) -> int {   // last32: fp32 [N,49,2048] of the last block
        if (!pool) return RELAX_OK;
        const float* avg_src;
        int64_t avg_stride;
        if (pool_from_stack) {      // (layer_stack is non-null here; no pointer arithmetic on it otherwise)
            avg_src = layer_stack + (RELAX_RN50_LAYER_STACK_DIM - 2048);
            avg_stride = RELAX_RN50_LAYER_STACK_DIM;
        } else {
            RELAX_TRY(launch_gap_ws(h, last32 + (size_t)pool_from * 49 * 2048, avg_ws, n_pool, 49, 2048, 2048, gap_scratch, s));
            avg_src = avg_ws;
            avg_stride = 2048;
        }
        hipLaunchKernelGGL(rn_pool_stats, dim3(n_pool), dim3(256), 0, s, avg_src, avg_stride, pool);
        RELAX_HIP_CHECK(h, hipGetLastError());
        return RELAX_OK;
    };

    if (h->gemm.precision >= 2) {   // (3 = f16x2: layer3 / layer4 and the 3x3 convolutions of layer1 / layer2 on fp16 planes - "rn_h2", "rn_h2_early" - the rest bf16x6)
        // bf16x6.  conv1 7x7/2 (raw) straight from the uint8 fragments (conv1_x6.hip: preprocess, im2col, split and contraction in one
        // kernel, the 16-pixel sums of the tap's spatial mean formed in its epilogue); from the max-pool on, every convolution input
        // travels as split planes written by its producer
        float* gap0 = T2 + kT2 * n;   // = the fp32 carving's gapws: free until the blocks carve the arena anew below
        const bool conv1_h2 = h->gemm.precision == 3 && h->gemm.rn_h2 && h->gemm.rn_h2_early && rn.conv1.w_h2;   // the stem on f16x2 as well
        RELAX_TRY(launch_conv1_x6(h, frags, conv1_h2 ? rn.conv1.w_h2 : rn.conv1.w_sp3, bufA, layer_stack ? gap0 : nullptr, N, s,
                                  conv1_h2 ? rn.conv1.w_inv : nullptr));
        if (layer_stack) RELAX_TRY(launch_gap_groups_finish(h, gap0, layer_stack, n_ls, 112 * 112, 64, RELAX_RN50_LAYER_STACK_DIM, s));
        if (taps_nchw && taps_nchw[0]) RELAX_TRY(launch_nhwc_to_nchw(h, bufA, taps_nchw[0], N, 112 * 112, 64, s));
        float* f32a = bufB;                                   // block outputs as fp32, where something needs them (ping-pong with f32b)
        float* f32b = bufD;
        char* spa = reinterpret_cast<char*>(T1);              // carve the rest of the arena anew
        char* spb = spa + sizeof(float) * (kBig * 3 / 2) * n;
        char* T1s = spb + sizeof(float) * (kBig * 3 / 2) * n;
        char* T2s = T1s + sizeof(float) * (kT1 * 3 / 2) * n;
        float* gapws6 = reinterpret_cast<float*>(T2s + sizeof(float) * (kT2 * 3 / 2) * n);
        float* avg6 = gapws6 + kGapWs * n;
        gapws = gapws6;
        // f16x2 for layer3 / layer4 ("gemm_precision" 3 with "rn_h2"): per-image tables {maximum, scale, 1 / scale}, one slot per tensor
        const bool use_h2 = h->gemm.precision == 3 && h->gemm.rn_h2 && rn.blocks.size() == 16;
        float* imgtab = avg6 + kAvg * n;
        int next_slot = 0;
        auto slot_amax = [&](int t) { return reinterpret_cast<unsigned*>(imgtab + (size_t)(3 * t) * n); };
        auto slot_scale = [&](int t) { return imgtab + (size_t)(3 * t + 1) * n; };
        auto slot_inv = [&](int t) { return imgtab + (size_t)(3 * t + 2) * n; };
        if (use_h2) RELAX_HIP_CHECK(h, hipMemsetAsync(imgtab, 0, sizeof(float) * 3 * kImgSlots * n, s));
        int slot_x = -1;          // the current block input's slot, once it exists as fp16 planes
        int slot_prev_out = -1;   // maximum of the previous block's output (the residual of the hand-over block)
        // "rn_h2_early": the 3x3 convolutions of layer1 / layer2 on f16x2 as well (gemm_x6.hip, H2 form).  Their input (conv1's output)
        // is written as fp16 planes with the image's Hoelder scale  l1max(conv1) max|block input| + max|bias|; the block input's maximum
        // is measured by its producer: the max-pool (block maxima, reduced per image) or the previous block's conv3 epilogue.
        bool use_early = use_h2 && h->gemm.rn_h2_early;
        for (size_t b = 0; b < kFirstH2Block && use_early; ++b) use_early = rn.blocks[b].c2.w_h2 != nullptr;
        int slot_xin = -1;        // (early) the slot that holds the maximum of the current block input
        if (use_early) slot_xin = next_slot++;
        // "rn_fuse": layer1[0] back to back too, with the downsample convolution folded into its conv3 - the block input then travels as fp32 rows
        // (4 B per value instead of 6: conv1 splits them in its K loop, the fused launch reads each pixel's row as conv3's second source)
        const bool fuse0 = use_early && h->gemm.rn_fuse && h->gemm.fp32_rows && rn.blocks[0].c3d_w_h2p != nullptr;
        if (fuse0)
            RELAX_TRY(launch_bn_relu_maxpool_f32(h, bufA, rn.bn1_scale, rn.bn1_shift, bufD, N, 112, 112, 64, s, slot_amax(slot_xin),
                                                 reinterpret_cast<unsigned*>(imgtab + 3 * kImgSlots * n)));
        else
        RELAX_TRY(launch_bn_relu_maxpool_sp3(h, bufA, rn.bn1_scale, rn.bn1_shift, spa, N, 112, 112, 64, s,
                                             use_early ? slot_amax(slot_xin) : nullptr,
                                             reinterpret_cast<unsigned*>(imgtab + 3 * kImgSlots * n)));
        // A block output exists as split planes (next convolutions, next residual: hi + mid + lo is the fp32 value, exactly) and as
        // fp32 only where something needs it, and only for the images that need it: the tap export, the spatial mean of the 7x7
        // taps of the layer-stack images (49 rows per image do not divide into the 16- or 4-row groups of the mean fused into
        // the epilogue), the last block's map of the pool images.
        // The block outputs inside layer1 and layer2 (56x56x256 and 28x28x512: the widest tensors, their consumers HBM-bound) travel
        // as plain fp32 instead, 4 bytes per value where the planes take 6: the next block's conv1 (64 / 128 output columns, one
        // column tile, so every value is split exactly once, as the producer's epilogue would have) splits them inside its K loop
        // and its conv3 adds them as an fp32 residual - the same values bit for bit.  A layer's last block writes planes: the next
        // layer's first conv3 reads them as its second activation source.
        const float* cur32 = fuse0 ? bufD : nullptr;   // (f32b: block 0 writes f32a, block 1 - which overwrites f32b - runs when block 0 is through)
        char* cursp = spa;
        char* othersp = spb;
        float* out32 = f32a;
        bool cur_is_f32 = fuse0;   // the current block input exists as fp32 rows only
        const char* dr_planes = nullptr;    // ... or as fp16 planes with the per-image scale of slot dr_slot (for a downsample launch of its own)
        int dr_slot = -1, dr_slot_next = -1;
        const char* compact_in = nullptr;   // ... plus the planes of its stride-2 sample, compacted (what the previous block left for a downsample branch)
        int H = 56;
        for (size_t b = 0; b < rn.blocks.size(); ++b) {
            const Bottleneck& blk = rn.blocks[b];
            const int Ho = H / blk.c2.stride;
            const int HWo = Ho * Ho, Cout = blk.c3.Cout;
            const bool tapped = blk.tap >= 0;
            const bool want_mean = tapped && layer_stack;
            const bool want_export = tapped && taps_nchw && taps_nchw[blk.tap];
            const bool is_last = b + 1 == rn.blocks.size();
            const bool fuse_mean = want_mean && HWo % 4 == 0;
            const bool pool_needs32 = is_last && pool && !pool_from_stack;
            // "rn_fuse" (below): conv2 and conv3 back to back in one launch
            const bool fuse_x2 = b == 0 && fuse0;      // the first block: conv3 + downsample in the fused launch's one contraction
            // layer2[0] (stride 2, 256 -> 512 downsample: too many channels for a second source in registers): the downsample convolution as a
            // launch of its own (f16x2, the fp32 rows sampled with the stride and split in the K loop) whose fp32 output is the residual of the
            // block's back-to-back launch (3x3 with the stride -> conv3) - instead of the 3x3 + the two-source bf16x6 conv3
            auto down_as_residual = [&](const Bottleneck& k) {
                return use_early && h->gemm.rn_fuse && h->gemm.rn_c1_h2 && h->gemm.fp32_rows && k.has_down && k.c2.stride == 2 && k.c3.w_h2p && k.down.w_h2 &&
                       k.c1.w_h2 && k.c1.Cout % 128 == 0 && k.c1.Cin >= 256;
            };
            const bool fuse_dr = b > 0 && b < kFirstH2Block && cur_is_f32 && down_as_residual(blk) && Ho * Ho >= 256 && dr_planes != nullptr;
            const bool fuse = fuse_x2 || fuse_dr || (use_early && b < kFirstH2Block && h->gemm.rn_fuse && !blk.has_down && blk.c3.w_h2p && cur_is_f32 && blk.c2.stride == 1 && H * H >= 256);
            // a layer's last block in front of a downsample block: its output travelled as three bf16 planes (6 B per value: the next block's conv1 and
            // the second source of its conv3 read planes).  Back to back it leaves as fp32 rows like the others (conv1 splits in its K loop) PLUS the
            // planes of the stride-2 sample only - all the downsample branch reads -, compacted: 4 + 1.5 bytes per value instead of 6, and conv1 reads 4
            const bool next_down = !is_last && rn.blocks[b + 1].has_down;
            const bool next_dr = next_down && b + 1 < kFirstH2Block && fuse && down_as_residual(rn.blocks[b + 1]);   // the next block reads fp32 rows only
            const bool compact = fuse && next_down && !next_dr && b + 1 < kFirstH2Block && h->gemm.fp32_rows && Cout <= 512 && Ho % 2 == 0 && rn.blocks[b + 1].c2.stride == 2;
            const bool out_is_f32 = h->gemm.fp32_rows && Cout <= 512 && !is_last && (!next_down || compact || next_dr);
            const bool need32 = out_is_f32 || want_export || (want_mean && !fuse_mean) || pool_needs32;
            // fp32 rows: every image for the next block, an export or the pool images behind the layer-stack ones, else the
            // layer-stack images only
            const int rows32 = (out_is_f32 || want_export || pool_needs32) ? N * HWo : n_ls * HWo;
            if (use_h2 && b >= kFirstH2Block) {
                // ---- an f16x2 block (gemm_h2.hip).  Every tensor that feeds a convolution travels as two fp16 planes with one scale per
                // image; the scale of a tensor is fixed BEFORE it is written, from Hoelder's bound on its producer (measured maxima of
                // the producer's inputs, l1max / bmax of its weights).  cursp = the block input's planes (slot_x).
                const int s1 = next_slot++, s2 = next_slot++, sy = next_slot++;
                // fp32 copies only where something reads them: an export, the spatial mean of the 7x7 taps (49 rows do not divide into the
                // 4-row groups of the fused mean), the last block's map of the pool images - and only for the images that need them.  The
                // residual of a block without a downsample branch is read from the block input's PLANES (hi + lo, the stored 22-bit value).
                const bool fuse_h2 = want_mean && HWo % 4 == 0 && HWo % 16 != 0;   // 14x14 maps: 4-row groups (gemm_h3's fused mean)
                const bool need32_h2 = want_export || (want_mean && !fuse_h2) || pool_needs32;
                const int rows32_h2 = (want_export || pool_needs32) ? N * HWo : n_ls * HWo;
                auto conv_h2 = [&](const ConvW& c, const void* in, int Hin, int slot_in, GemmDescH2 g, int slot_out, int act) {
                    g.a = in; g.w = c.w_h2; g.colscale = c.w_inv; g.bias = c.bias; g.act = act;
                    g.pixels = 1; g.Nimg = N; g.H = Hin; g.W = Hin; g.Cin = c.Cin;
                    g.Ho = (Hin + 2 * c.pad - c.KH) / c.stride + 1; g.Wo = g.Ho;
                    g.KH = c.KH; g.KW = c.KW; g.stride = c.stride; g.pad = c.pad;
                    g.M = N * g.Ho * g.Wo; g.N = c.Cout; g.K = c.KH * c.KW * c.Cin;
                    g.rows_per_img = g.Ho * g.Wo; g.img_in_inv = slot_inv(slot_in);
                    if (g.out_h2) { g.img_out_scale = slot_scale(slot_out); g.amax_out = slot_amax(slot_out); }
                    return launch_gemm_h2(h, g, s);
                };
                // conv1 1x1 + ReLU
                RELAX_TRY(launch_h2_image_scales(h, slot_amax(slot_x), blk.c1.l1max, nullptr, 0.f, nullptr, blk.c1.bmax, slot_scale(s1), slot_inv(s1), N, s));
                { GemmDescH2 g{}; g.out_h2 = T1s; RELAX_TRY(conv_h2(blk.c1, cursp, H, slot_x, g, s1, 1)); }
                // conv2 3x3 (stride) + ReLU
                RELAX_TRY(launch_h2_image_scales(h, slot_amax(s1), blk.c2.l1max, nullptr, 0.f, nullptr, blk.c2.bmax, slot_scale(s2), slot_inv(s2), N, s));
                { GemmDescH2 g{}; g.out_h2 = T2s; RELAX_TRY(conv_h2(blk.c2, T1s, H, s1, g, s2, 1)); }
                // conv3 1x1 + identity + ReLU: the identity is the block input (its planes) or the downsample convolution of it (fp32, no
                // activation; bounded by its own Hoelder term)
                GemmDescH2 g3{};
                if (blk.has_down) {
                    GemmDescH2 gd{};
                    gd.out = bufA;
                    RELAX_TRY(conv_h2(blk.down, cursp, H, slot_x, gd, -1, 0));
                    g3.residual = bufA;
                    RELAX_TRY(launch_h2_image_scales(h, slot_amax(s2), blk.c3.l1max, slot_amax(slot_x), blk.down.l1max, nullptr,
                                                     blk.c3.bmax + blk.down.bmax, slot_scale(sy), slot_inv(sy), N, s));
                } else {
                    g3.residual_h2 = cursp; g3.img_res_inv = slot_inv(slot_x);
                    RELAX_TRY(launch_h2_image_scales(h, slot_amax(s2), blk.c3.l1max, nullptr, 0.f, slot_amax(slot_x), blk.c3.bmax, slot_scale(sy),
                                                     slot_inv(sy), N, s));
                }
                g3.out = need32_h2 ? out32 : nullptr; g3.out_rows = rows32_h2;
                g3.out_h2 = is_last ? nullptr : othersp;
                g3.gap_groups = fuse_h2 ? gapws : nullptr; g3.gap_rows = n_ls * HWo;
                g3.no_split = tapped && HWo % 4 == 0 && HWo % 16 != 0;   // a launch that fuses the mean when the layer stack is asked for runs
                                                                         // unsplit either way: the pool vector's bits do not depend on the request
                RELAX_TRY(conv_h2(blk.c3, T2s, Ho, s2, g3, sy, 1));
                cur32 = need32_h2 ? out32 : nullptr;
                if (need32_h2) out32 = out32 == f32a ? f32b : f32a;
                { char* t = cursp; cursp = othersp; othersp = t; }
                cur_is_f32 = false;
                slot_x = sy;
                H = Ho;
                if (tapped) {
                    const int off = tap_offset(blk.tap);
                    if (fuse_h2)
                        RELAX_TRY(launch_gap_groups_finish(h, gapws, layer_stack + off, n_ls, HWo, Cout, RELAX_RN50_LAYER_STACK_DIM, s));
                    else if (want_mean)
                        RELAX_TRY(launch_gap_ws(h, cur32, layer_stack + off, n_ls, HWo, Cout, RELAX_RN50_LAYER_STACK_DIM, gapws, s));
                    if (want_export) RELAX_TRY(launch_nhwc_to_nchw(h, cur32, taps_nchw[blk.tap], N, HWo, Cout, s));
                }
                continue;
            }
            // the two blocks in front of the f16x2 ones hand over: block 5's output maximum (the residual of block 6), block 6's conv2
            // maximum, and block 6's output as fp32 + fp16 planes with its Hoelder scale + its maximum
            const bool handover = use_h2 && b + 1 == kFirstH2Block;
            const bool pre_handover = use_h2 && b + 2 == kFirstH2Block;
            int slot_c2 = -1, slot_y = -1, slot_t1 = -1, slot_o = -1, slot_t1m = -1;
            // ("rn_fuse", `fuse` above: the 3x3's tile never leaves the CU; conv3 on f16x2 with one scale per pixel row)
            if ((handover || next_dr) && fuse) slot_t1m = next_slot++;   // the MEASURED maximum of conv1's output: what the scale of an output that leaves as fp16 planes is bounded from (below)
            if (handover) { slot_c2 = next_slot++; slot_y = next_slot++; }
            if (use_early) slot_t1 = next_slot++;
            // the maximum of this block's output: the next block's conv1 scale (early), the residual term of the hand-over block
            if (handover) slot_o = slot_y;
            else if (use_early || pre_handover) slot_o = next_slot++;
            if (pre_handover) slot_prev_out = slot_o;
            {
                ConvDescX6 d{};
                d.in = cur_is_f32 ? static_cast<const void*>(cur32) : cursp; d.in_f32 = cur_is_f32;
                d.Nimg = N; d.H = H; d.W = H; d.Cin = blk.c1.Cin; d.Ho = H; d.Wo = H;
                d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
                d.w = blk.c1.w_sp3; d.Cout = blk.c1.Cout; d.bias = blk.c1.bias; d.act = 1;
                if (use_early && cur_is_f32 && h->gemm.rn_c1_h2 && blk.c1.w_h2 && blk.c1.Cin >= 256 && blk.c1.Cout % 128 == 0) {   // (layer2: MFMA-bound on six products)
                    // f16x2: the rows are split into fp16 planes in the K loop, with the image's scale from the MEASURED maximum of the block input
                    const int sx = next_slot++;
                    RELAX_TRY(launch_h2_image_scales(h, slot_amax(slot_xin), 1.f, nullptr, 0.f, nullptr, 0.f, slot_scale(sx), slot_inv(sx), N, s));
                    d.w = blk.c1.w_h2; d.colscale = blk.c1.w_inv; d.img_in_scale = slot_scale(sx); d.img_in_inv = slot_inv(sx);
                }
                if (use_early) {   // fp16 planes for the f16x2 conv2, scaled by the image's bound
                    RELAX_TRY(launch_h2_image_scales(h, slot_amax(slot_xin), blk.c1.l1max, nullptr, 0.f, nullptr, blk.c1.bmax, slot_scale(slot_t1),
                                                     slot_inv(slot_t1), N, s));
                    d.out_h2 = T1s; d.img_out_scale = slot_scale(slot_t1);
                    if (slot_t1m >= 0) d.amax_out = slot_amax(slot_t1m);
                } else {
                    d.out_sp3 = T1s;
                }
                RELAX_TRY(launch_conv_x6(h, d, s));
            }
            if (handover || use_early) {
                ConvDescX6 d2{};
                d2.in = T1s; d2.Nimg = N; d2.H = H; d2.W = H; d2.Cin = blk.c2.Cin;
                d2.Ho = (H + 2 * blk.c2.pad - blk.c2.KH) / blk.c2.stride + 1; d2.Wo = d2.Ho;
                d2.KH = blk.c2.KH; d2.KW = blk.c2.KW; d2.stride = blk.c2.stride; d2.pad = blk.c2.pad;
                d2.w = blk.c2.w_sp3; d2.Cout = blk.c2.Cout; d2.bias = blk.c2.bias; d2.out_sp3 = T2s; d2.act = 1;
                if (use_early) { d2.in_h2 = 1; d2.w = blk.c2.w_h2; d2.colscale = blk.c2.w_inv; d2.img_in_inv = slot_inv(slot_t1); }
                if (handover && !fuse) d2.amax_out = slot_amax(slot_c2);
                if (fuse) {
                    d2.out_sp3 = nullptr;
                    d2.w3 = blk.c3.w_h2p; d2.colscale3 = blk.c3.w_invp; d2.bias3 = blk.c3.bias; d2.Cout3 = Cout;
                    d2.residual = cur32;
                    if (fuse_dr) {
                        // the downsample branch first: fp32 [N*Ho*Ho][Cout] into bufA (the stem's raw output: dead since the max-pool), on gemm_h3's
                        // 1x1 form from the compact fp16 planes of the input's stride-2 sample the previous block left (its per-image scale: dr_slot)
                        GemmDescH2 g{};
                        g.a = dr_planes; g.w = blk.down.w_h2; g.colscale = blk.down.w_inv; g.bias = blk.down.bias; g.act = 0;
                        g.pixels = 1; g.Nimg = N; g.H = Ho; g.W = Ho; g.Cin = blk.down.Cin; g.Ho = Ho; g.Wo = Ho;
                        g.KH = 1; g.KW = 1; g.stride = 1; g.pad = 0;
                        g.M = N * Ho * Ho; g.N = blk.down.Cout; g.K = blk.down.Cin;
                        g.rows_per_img = Ho * Ho; g.img_in_inv = slot_inv(dr_slot);
                        g.out = bufA;
                        RELAX_TRY(launch_gemm_h2(h, g, s));
                        d2.residual = bufA;
                    }
                    if (next_dr) {
                        // this block's output leaves as fp32 rows AND as the fp16 planes of its stride-2 sample (compacted): what the next block's
                        // downsample launch reads.  Their per-image scale: Hoelder from the measured maxima of conv1's output and of the block input
                        // (as the hand-over block below)
                        const int sy = next_slot++;
                        RELAX_TRY(launch_h2_image_scales(h, slot_amax(slot_t1m), blk.c3.l1max * blk.c2.l1max, nullptr, 0.f, slot_amax(slot_xin),
                                                         blk.c3.l1max * blk.c2.bmax + blk.c3.bmax, slot_scale(sy), slot_inv(sy), N, s));
                        d2.out_h2 = othersp; d2.img_out_scale = slot_scale(sy); d2.sp3_sub = 2;
                        dr_slot_next = sy;
                    }
                    if (fuse_x2) { d2.w3 = blk.c3d_w_h2p; d2.colscale3 = blk.c3d_w_invp; d2.bias3 = blk.c3d_bias; d2.residual = nullptr; d2.x2 = cur32; }
                    d2.out = need32 ? out32 : nullptr; d2.out_rows = rows32;
                    d2.out_sp3 = (compact || !out_is_f32) ? othersp : nullptr;
                    d2.sp3_sub = (compact || next_dr) ? 2 : 1;
                    d2.gap_groups = fuse_mean ? gapws : nullptr; d2.gap_rows = n_ls * HWo;
                    d2.amax_out = slot_o >= 0 ? slot_amax(slot_o) : nullptr;
                    if (handover) {
                        // the block's output leaves as fp16 planes with a per-image scale from a bound.  Two launches: Hoelder on conv3 with the
                        // MEASURED maximum of conv2's output; back to back that tensor never exists, so it is bounded in turn from the measured
                        // maximum of conv1's output:  |y| <= l1(c3) (l1(c2) max|t1| + b2) + b3 + max|x|  - one more L1-to-max ratio of looseness
                        // (2^5 - 2^7 of fp16's 19 binades), still never compounding beyond this block
                        RELAX_TRY(launch_h2_image_scales(h, slot_amax(slot_t1m), blk.c3.l1max * blk.c2.l1max, nullptr, 0.f, slot_amax(slot_prev_out),
                                                         blk.c3.l1max * blk.c2.bmax + blk.c3.bmax, slot_scale(slot_y), slot_inv(slot_y), N, s));
                        d2.out_sp3 = nullptr;
                        d2.out_h2 = othersp; d2.img_out_scale = slot_scale(slot_y); d2.amax_out = slot_amax(slot_y);
                    }
                }
                RELAX_TRY(launch_conv_x6(h, d2, s));
            } else {
                RELAX_TRY(run_conv_x6(h, blk.c2, T1s, N, H, H, nullptr, nullptr, T2s, 1, s));
            }
            if (!fuse) {
            ConvDescX6 d{};
            d.in = T2s; d.Nimg = N; d.H = Ho; d.W = Ho; d.Cin = blk.c3.Cin; d.Ho = Ho; d.Wo = Ho;
            d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
            d.Cout = Cout; d.act = 1;
            d.out = need32 ? out32 : nullptr; d.out_rows = rows32;
            d.out_sp3 = out_is_f32 ? nullptr : othersp;
            d.gap_groups = fuse_mean ? gapws : nullptr; d.gap_rows = n_ls * HWo;
            d.no_split = tapped && HWo % 4 == 0;   // a launch that fuses the mean when the layer stack is asked for runs unsplit either
                                                    // way: the pool vector's bits do not depend on whether the layer stack is requested
            if (blk.has_down) {
                // conv3 and the downsample convolution in ONE contraction over K = [conv2 output | block input sampled with the
                // block's stride]: no fp32 copy of the branch is written and read back (layer1.0: 6.6 GB per 1024 images)
                d.w = blk.c3d_w_sp3; d.bias = blk.c3d_bias;
                d.in2 = cursp; d.H2 = H; d.W2 = H; d.Cin2 = blk.down.Cin; d.stride2 = blk.down.stride;
                if (compact_in) { d.in2 = compact_in; d.H2 = Ho; d.W2 = Ho; d.stride2 = 1; }   // (the previous block left the stride-2 sample only)
            } else {
                d.w = blk.c3.w_sp3; d.bias = blk.c3.bias;
                if (cur_is_f32) d.residual = cur32; else d.residual_sp3 = cursp;
            }
            if (slot_o >= 0 && !handover) d.amax_out = slot_amax(slot_o);
            if (handover) {
                // (block 6 has no downsample branch and its input is fp32 rows: residual = cur32; its output leaves as fp16 planes with the
                // Hoelder scale of conv3 + identity)
                RELAX_TRY(launch_h2_image_scales(h, slot_amax(slot_c2), blk.c3.l1max, nullptr, 0.f, slot_amax(slot_prev_out), blk.c3.bmax,
                                                 slot_scale(slot_y), slot_inv(slot_y), N, s));
                d.out_sp3 = nullptr;       // (fp32 rows as the tap logic above decided; block 7 takes its identity from its downsample branch)
                d.out_h2 = othersp; d.img_out_scale = slot_scale(slot_y); d.amax_out = slot_amax(slot_y);
            }
            RELAX_TRY(launch_conv_x6(h, d, s));
            }
            if (handover) {
                cur32 = need32 ? out32 : nullptr;
                if (need32) out32 = out32 == f32a ? f32b : f32a;
                { char* t = cursp; cursp = othersp; othersp = t; }
                cur_is_f32 = false;
                slot_x = slot_y;
                H = Ho;
                if (tapped) {
                    const int off = tap_offset(blk.tap);
                    if (fuse_mean)
                        RELAX_TRY(launch_gap_groups_finish(h, gapws, layer_stack + off, n_ls, HWo, Cout, RELAX_RN50_LAYER_STACK_DIM, s));
                    else if (want_mean)
                        RELAX_TRY(launch_gap_ws(h, cur32, layer_stack + off, n_ls, HWo, Cout, RELAX_RN50_LAYER_STACK_DIM, gapws, s));
                    if (want_export) RELAX_TRY(launch_nhwc_to_nchw(h, cur32, taps_nchw[blk.tap], N, HWo, Cout, s));
                }
                continue;
            }
            cur32 = need32 ? out32 : nullptr;
            if (need32) out32 = out32 == f32a ? f32b : f32a;
            if (!out_is_f32) { char* t = cursp; cursp = othersp; othersp = t; }
            compact_in = compact ? othersp : nullptr;
            dr_planes = (fuse && next_dr) ? othersp : nullptr;
            dr_slot = dr_slot_next;
            dr_slot_next = -1;
            cur_is_f32 = out_is_f32;
            slot_xin = slot_o;
            H = Ho;
            if (tapped) {
                const int off = tap_offset(blk.tap);
                if (fuse_mean)
                    RELAX_TRY(launch_gap_groups_finish(h, gapws, layer_stack + off, n_ls, HWo, Cout, RELAX_RN50_LAYER_STACK_DIM, s));
                else if (want_mean)
                    RELAX_TRY(launch_gap_ws(h, cur32, layer_stack + off, n_ls, HWo, Cout, RELAX_RN50_LAYER_STACK_DIM, gapws, s));
                if (want_export) RELAX_TRY(launch_nhwc_to_nchw(h, cur32, taps_nchw[blk.tap], N, HWo, Cout, s));
            }
        }
        RELAX_REQUIRE(h, next_slot <= (int)kImgSlots, "resnet50: %d per-image scale slots used, %d reserved", next_slot, (int)kImgSlots);
        return pool_tail(cur32, avg6, gapws);
    }
    auto emit_tap = [&](int tap, const float* act) -> int {
        const int C = kTapChannels[tap], HW = kTapHW[tap] * kTapHW[tap];
        if (layer_stack)
            RELAX_TRY(launch_gap_ws(h, act, layer_stack + tap_offset(tap), n_ls, HW, C, RELAX_RN50_LAYER_STACK_DIM, gapws, s));
        if (taps_nchw && taps_nchw[tap]) RELAX_TRY(launch_nhwc_to_nchw(h, act, taps_nchw[tap], N, HW, C, s));
        return RELAX_OK;
    };
    const int64_t npix = (int64_t)N * 224 * 224;
    hipLaunchKernelGGL(rn_preprocess, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, frags, X0, npix);
    RELAX_HIP_CHECK(h, hipGetLastError());
    // conv1 7x7/2 (raw), algorithmic FLOPs use the real 3 input channels
    RELAX_TRY(run_conv(h, rn.conv1, X0, N, 224, 224, nullptr, bufA, 0, s, 2.0 * N * 112.0 * 112.0 * 64.0 * 147.0));
    RELAX_TRY(emit_tap(0, bufA));
    RELAX_TRY(launch_bn_relu_maxpool(h, bufA, rn.bn1_scale, rn.bn1_shift, bufB, N, 112, 112, 64, s));
    float* cur = bufB;
    float* other = bufA;
    int H = 56;
    for (const Bottleneck& blk : rn.blocks) {
        const int Ho = H / blk.c2.stride;
        RELAX_TRY(run_conv(h, blk.c1, cur, N, H, H, nullptr, T1, 1, s));
        RELAX_TRY(run_conv(h, blk.c2, T1, N, H, H, nullptr, T2, 1, s));
        const float* identity = cur;
        if (blk.has_down) {
            RELAX_TRY(run_conv(h, blk.down, cur, N, H, H, nullptr, bufD, 0, s));
            identity = bufD;
        }
        RELAX_TRY(run_conv(h, blk.c3, T2, N, Ho, Ho, identity, other, 1, s));
        float* t = cur; cur = other; other = t;
        H = Ho;
        if (blk.tap >= 0) RELAX_TRY(emit_tap(blk.tap, cur));
    }
    return pool_tail(cur, avg, gapws);
}

int relax_resnet50_features(relax_handle* h, const uint8_t* frags, int N, float* layer_stack, float* pool,
                            float* const* taps_nchw, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, h->rn.loaded, "relax_resnet50_features: call relax_load_resnet50 first");
    RELAX_REQUIRE(h, frags && N > 0, "relax_resnet50_features: bad arguments");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    return resnet_forward(h, frags, N, layer_stack ? N : 0, 0, layer_stack, pool, taps_nchw, static_cast<hipStream_t>(stream));
}

int relax_resnet50_clip_features(relax_handle* h, const uint8_t* frags, int N, int n_layer_stack, float* layer_stack, float* pool,
                                 relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, h->rn.loaded, "relax_resnet50_clip_features: call relax_load_resnet50 first");
    RELAX_REQUIRE(h, frags && N > 0 && n_layer_stack >= 0 && n_layer_stack <= N, "relax_resnet50_clip_features: bad arguments");
    RELAX_REQUIRE(h, (n_layer_stack == 0 || layer_stack) && (n_layer_stack == N || pool),
                  "relax_resnet50_clip_features: NULL output for a non-empty group of images");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    return resnet_forward(h, frags, N, n_layer_stack, n_layer_stack, layer_stack, n_layer_stack < N ? pool : nullptr, nullptr,
                          static_cast<hipStream_t>(stream));
}

}  // extern "C"
