// DINO ViT (patch 16, 224x224 input -> 197 tokens, 64-d heads) feature extractor on gfx950.
//
// Reference semantics (file:line in xinyiW915/ReLaX-VQA, src/extractor/visualise_vit_layer.py):
//   :466-470,339-342,492-494  input: PIL RGB, /255, no mean/std normalisation
//   :132-149  PatchEmbed conv 16x16/16            -> patchify kernel + GEMM (K = 3*16*16 = 768)
//   :221-232  cls token prepend, + pos_embed       -> vit_assemble
//   :93-129   pre-LN blocks: qkv, softmax(q k^T/8) v, proj, MLP with exact-erf GELU
//   :234-239  final LayerNorm (eps 1e-6, :287-289), patch tokens x[:,1:]
//   src/main_fragment_pool.py:124-133  mean / max / population-std over the 196 tokens
#include "relax_internal.h"
#include "host_logic.h"
#include "sp3.h"
#include "h2.h"

namespace relax {

constexpr int NTOK = 197;
constexpr int NPATCH = 196;
constexpr int PATCH_K = 3 * 16 * 16;
constexpr float kLnEps = 1e-6f;
constexpr float kPatchScale = 16384.f;   // patch values are value/255 in [0, 1]: as fp16 planes of value * 2^14 (csrc/h2.h)

// uint8 BGR [N,224,224,3] -> fp32 [N*196, 768], k = c*256 + py*16 + px with c in RGB order, value/255
__global__ __launch_bounds__(256) void vit_patchify(const uint8_t* __restrict__ frag, float* __restrict__ P,
                                                    int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % PATCH_K);
    const int64_t row = i / PATCH_K;
    const int p = (int)(row % NPATCH);
    const int64_t n = row / NPATCH;
    const int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
    const int y = (p / 14) * 16 + py, x = (p % 14) * 16 + px;
    P[i] = (float)frag[((n * 224 + y) * 224 + x) * 3 + (2 - c)] / 255.0f;
}

// the same patches as split planes (bf16 hi + mid + lo of value/255, gemm_x6.hip): one thread per 8 k (8 pixels of a patch row)
__global__ __launch_bounds__(256) void vit_patchify_sp3(const uint8_t* __restrict__ frag, char* __restrict__ P, int64_t total8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int k = (int)(i % (PATCH_K / 8)) * 8;
    const int64_t row = i / (PATCH_K / 8);
    const int p = (int)(row % NPATCH);
    const int64_t n = row / NPATCH;
    const int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
    const int y = (p / 14) * 16 + py, x = (p % 14) * 16 + px;
    const uint8_t* src = frag + ((n * 224 + y) * 224 + x) * 3 + (2 - c);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)src[3 * j] / 255.0f;
    store_sp3_x8(P + row * (PATCH_K * 6), k, (sp3_f32x4){v[0], v[1], v[2], v[3]}, (sp3_f32x4){v[4], v[5], v[6], v[7]});
}

// the same patches as two fp16 planes of value/255 * scale (csrc/h2.h; the values are in [0, 1]: scale 2^14)
__global__ __launch_bounds__(256) void vit_patchify_h2(const uint8_t* __restrict__ frag, char* __restrict__ P, int64_t total8, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int k = (int)(i % (PATCH_K / 8)) * 8;
    const int64_t row = i / (PATCH_K / 8);
    const int p = (int)(row % NPATCH);
    const int64_t n = row / NPATCH;
    const int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
    const int y = (p / 14) * 16 + py, x = (p % 14) * 16 + px;
    const uint8_t* src = frag + ((n * 224 + y) * 224 + x) * 3 + (2 - c);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)src[3 * j] / 255.0f;
    store_h2_x8(P + row * (PATCH_K * 4), k, (h2_f32x4){v[0], v[1], v[2], v[3]}, (h2_f32x4){v[4], v[5], v[6], v[7]}, scale);
}

// X[n,0,:] = cls + pos[0];  X[n,1+p,:] = PE[n*196+p,:] + pos[1+p]
__global__ __launch_bounds__(256) void vit_assemble(const float* __restrict__ PE, const float* __restrict__ cls,
                                                    const float* __restrict__ pos, float* __restrict__ X, int dim4,
                                                    int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % dim4);
    const int64_t row = i / dim4;
    const int tok = (int)(row % NTOK);
    const int64_t n = row / NTOK;
    const float4 pp = reinterpret_cast<const float4*>(pos)[(int64_t)tok * dim4 + d];
    const float4 v = tok == 0 ? reinterpret_cast<const float4*>(cls)[d]
                              : reinterpret_cast<const float4*>(PE)[(n * NPATCH + (tok - 1)) * dim4 + d];
    reinterpret_cast<float4*>(X)[i] = make_float4(v.x + pp.x, v.y + pp.y, v.z + pp.z, v.w + pp.w);
}

// Y [N,197,dim] -> tokens [N,196,dim] (drop cls)
__global__ __launch_bounds__(256) void vit_drop_cls(const float* __restrict__ Y, float* __restrict__ T, int dim4,
                                                    int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % dim4);
    const int64_t row = i / dim4;
    const int p = (int)(row % NPATCH);
    const int64_t n = row / NPATCH;
    reinterpret_cast<float4*>(T)[i] = reinterpret_cast<const float4*>(Y)[(n * NTOK + 1 + p) * dim4 + d];
}

// per (image, channel): mean, max, population std over tokens 1..196 -> out[n, 0:dim | dim:2dim | 2dim:3dim]
__global__ __launch_bounds__(256) void vit_token_stats(const float* __restrict__ Y, float* __restrict__ out, int dim,
                                                       int tok_per_img, int first, int count) {
    __shared__ float red[4][64];
    __shared__ float s_mean[64];
    const int n = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;  // 4 groups stride the tokens
    const float* yb = Y + ((int64_t)n * tok_per_img + first) * dim + c;
    float s = 0.f, m = -INFINITY;
    for (int p = grp; p < count; p += 4) {
        const float v = yb[(int64_t)p * dim];
        s += v;
        m = fmaxf(m, v);
    }
    red[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0) s_mean[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)count;
    __syncthreads();
    const float mean = s_mean[threadIdx.x & 63];
    __syncthreads();
    red[grp][threadIdx.x & 63] = m;
    __syncthreads();
    float mx = 0.f;
    if (grp == 0) mx = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
    __syncthreads();
    float q = 0.f;
    for (int p = grp; p < count; p += 4) {
        const float dv = yb[(int64_t)p * dim] - mean;
        q += dv * dv;
    }
    red[grp][threadIdx.x & 63] = q;
    __syncthreads();
    if (grp == 0) {
        const float var = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)count;
        float* o = out + (int64_t)n * 3 * dim;
        o[c] = mean;
        o[dim + c] = mx;
        o[2 * dim + c] = sqrtf(var);
    }
}

void free_vit(relax_handle* h) {
    for (void* p : h->vit.allocs) (void)hipFree(p);
    h->vit = VitW();
}

static size_t vit_floats_per_image(int dim) {
    return (size_t)NPATCH * PATCH_K        // P   patches
           + (size_t)NPATCH * dim          // PE  patch-embed output
           + (size_t)NTOK * dim * 2        // X, Y
           + (size_t)NTOK * dim * 3        // QKV
           + (size_t)NTOK * dim * 4;       // Hid
}

// bf16x6 path: sp3 operands take 6 bytes per value (1.5 floats)
static size_t vit_floats_per_image_x6(int dim) {
    return (size_t)NPATCH * PATCH_K * 3 / 2     // P    patches, sp3
           + (size_t)NPATCH * dim               // PE   patch-embed output
           + (size_t)NTOK * dim                 // X    residual stream
           + (size_t)NTOK * dim * 3 / 2         // Y    LayerNorm / attention output, sp3
           + (size_t)NTOK * dim * 3             // QKV
           + (size_t)NTOK * dim                 // ATT  attention output / final LayerNorm, fp32
           + (size_t)NTOK * dim * 4 * 3 / 2;    // Hid  GELU(fc1), sp3
}

size_t vit_arena_bytes(const VitW& v, int n) {
    const size_t a = vit_floats_per_image(v.dim), b = vit_floats_per_image_x6(v.dim);
    return sizeof(float) * (a > b ? a : b) * (size_t)n;
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_load_vit(relax_handle* h, const float* const* tensors, const char* const* names, const int64_t* numels,
                   int n, int dim, int depth, int heads) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, tensors && names && numels && n > 0, "relax_load_vit: bad arguments");
    RELAX_REQUIRE(h, heads > 0 && dim == heads * 64, "relax_load_vit: dim=%d must be heads*64 (heads=%d)", dim, heads);
    RELAX_REQUIRE(h, dim <= 768 && depth > 0, "relax_load_vit: dim=%d depth=%d unsupported", dim, depth);
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    free_vit(h);
    host::StateDict sd;
    for (int i = 0; i < n; ++i) sd.add(names[i], tensors[i], numels[i]);
    VitW& v = h->vit;
    int rc = RELAX_OK;
    auto up = [&](const std::string& key, int64_t numel, float** dst) {
        if (rc != RELAX_OK) return;
        std::string err;
        const float* src = sd.get(key, numel, err, "vit state dict");
        if (!src) {
            set_error(h, "%s", err.c_str());
            rc = RELAX_ERR_INVALID;
            return;
        }
        rc = upload(h, src, (size_t)numel, dst, v.allocs);
    };
    auto lin = [&](const std::string& p, int in, int out, LinearW* l) {
        l->in = in;
        l->out = out;
        up(p + ".weight", (int64_t)in * out, &l->w);
        up(p + ".bias", out, &l->b);
    };
    v.dim = dim; v.depth = depth; v.heads = heads;
    up("cls_token", dim, &v.cls);
    up("pos_embed", (int64_t)NTOK * dim, &v.pos);
    lin("patch_embed.proj", PATCH_K, dim, &v.patch);  // OIHW [dim][3][16][16] is already [dim][c*256+py*16+px]
    v.blocks.resize(depth);
    for (int i = 0; i < depth; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        VitBlockW& b = v.blocks[i];
        up(p + "norm1.weight", dim, &b.ln1_g);
        up(p + "norm1.bias", dim, &b.ln1_b);
        lin(p + "attn.qkv", dim, 3 * dim, &b.qkv);
        lin(p + "attn.proj", dim, dim, &b.proj);
        up(p + "norm2.weight", dim, &b.ln2_g);
        up(p + "norm2.bias", dim, &b.ln2_b);
        lin(p + "mlp.fc1", dim, 4 * dim, &b.fc1);
        lin(p + "mlp.fc2", 4 * dim, dim, &b.fc2);
    }
    up("norm.weight", dim, &v.norm_g);
    up("norm.bias", dim, &v.norm_b);
    // split planes of every GEMM weight for the bf16x6 kernel (made on the device from the uploaded fp32 copy)
    auto sp3 = [&](LinearW* l) {
        if (rc != RELAX_OK) return;
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)l->in * l->out * 6) != hipSuccess) {
            set_error(h, "vit: hipMalloc of split-plane weights failed");
            rc = RELAX_ERR_NOMEM;
            return;
        }
        v.allocs.push_back(q);
        l->w_sp3 = q;
        rc = launch_to_sp3(h, l->w, l->in, q, l->out, l->in, nullptr);
    };
    sp3(&v.patch);
    for (VitBlockW& b : v.blocks) {
        sp3(&b.qkv);
        sp3(&b.proj);
        sp3(&b.fc1);
        sp3(&b.fc2);
    }
    // two fp16 planes of every GEMM weight for the f16x2 kernel: row n scaled by 2^t_n (from the row's maximum); colscale[n] =
    // 2^-t_n / (the static scale of the activation tensor the layer reads).  The activation scales come from bounds that hold for
    // EVERY input (host_logic.h), evaluated here in double on the host copies of the weights.
    auto host_of = [&](const std::string& key, int64_t numel) -> const float* {
        std::string err;
        return sd.get(key, numel, err, "vit state dict");   // (present: `up` has already checked every key)
    };
    std::vector<float> tmp_scale, tmp_col;
    auto h2w = [&](LinearW* l, const std::string& p, float act_scale) {
        if (rc != RELAX_OK) return;
        const float* hw = host_of(p + ".weight", (int64_t)l->in * l->out);
        tmp_scale.resize((size_t)l->out);
        tmp_col.resize((size_t)l->out);
        host::h2_weight_row_scales(hw, l->out, l->in, tmp_scale.data());
        for (int n = 0; n < l->out; ++n) tmp_col[(size_t)n] = (1.f / tmp_scale[(size_t)n]) * (1.f / act_scale);   // powers of two: exact
        float* d_scale = nullptr;
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)l->in * l->out * 4) != hipSuccess) {
            set_error(h, "vit: hipMalloc of fp16-plane weights failed");
            rc = RELAX_ERR_NOMEM;
            return;
        }
        v.allocs.push_back(q);
        l->w_h2 = q;
        rc = upload(h, tmp_scale.data(), (size_t)l->out, &d_scale, v.allocs);
        if (rc == RELAX_OK) rc = upload(h, tmp_col.data(), (size_t)l->out, &l->colscale, v.allocs);
        if (rc == RELAX_OK) rc = launch_to_h2(h, l->w, l->in, q, l->out, l->in, 1.f, d_scale, nullptr);
    };
    if (rc == RELAX_OK && dim % 256 == 0) {   // (the f16x2 tile takes N % 256 == 0: ViT-B; smaller models run bf16x6 under "gemm_precision" 3)
        h2w(&v.patch, "patch_embed.proj", kPatchScale);
        for (int i = 0; i < depth && rc == RELAX_OK; ++i) {
            const std::string p = "blocks." + std::to_string(i) + ".";
            VitBlockW& b = v.blocks[i];
            const float *g1 = host_of(p + "norm1.weight", dim), *b1 = host_of(p + "norm1.bias", dim);
            const float *g2 = host_of(p + "norm2.weight", dim), *b2 = host_of(p + "norm2.bias", dim);
            b.s_ln1 = host::h2_scale_for_bound(host::layernorm_out_bound(g1, b1, dim));
            b.s_ln2 = host::h2_scale_for_bound(host::layernorm_out_bound(g2, b2, dim));
            // attention output = convex combinations of the V rows of qkv(LayerNorm1(x)): columns 2 dim .. 3 dim of the qkv Linear
            b.s_att = host::h2_scale_for_bound(host::linear_of_layernorm_bound(host_of(p + "attn.qkv.weight", (int64_t)3 * dim * dim),
                                                                               host_of(p + "attn.qkv.bias", 3 * dim), g1, b1, dim, 2 * dim, 3 * dim));
            b.s_qkv = host::h2_scale_for_bound(host::linear_of_layernorm_bound(host_of(p + "attn.qkv.weight", (int64_t)3 * dim * dim),
                                                                               host_of(p + "attn.qkv.bias", 3 * dim), g1, b1, dim, 0, 3 * dim));
            // |GELU(x)| <= |x|, x = fc1(LayerNorm2(.))
            b.s_hid = host::h2_scale_for_bound(host::linear_of_layernorm_bound(host_of(p + "mlp.fc1.weight", (int64_t)4 * dim * dim),
                                                                               host_of(p + "mlp.fc1.bias", 4 * dim), g2, b2, dim, 0, 4 * dim));
            h2w(&b.qkv, p + "attn.qkv", b.s_ln1);
            h2w(&b.proj, p + "attn.proj", b.s_att);
            h2w(&b.fc1, p + "mlp.fc1", b.s_ln2);
            h2w(&b.fc2, p + "mlp.fc2", b.s_hid);
        }
    }
    if (rc == RELAX_OK && hipDeviceSynchronize() != hipSuccess) {
        set_error(h, "vit: weight conversion failed");
        rc = RELAX_ERR_HIP;
    }
    if (rc != RELAX_OK) {
        free_vit(h);
        return rc;
    }
    v.loaded = true;
    return RELAX_OK;
}

int relax_vit_features(relax_handle* h, const uint8_t* frags, int N, float* tokens, float* pooled,
                       relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, h->vit.loaded, "relax_vit_features: call relax_load_vit first");
    RELAX_REQUIRE(h, frags && N > 0, "relax_vit_features: bad arguments");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const VitW& v = h->vit;
    const int dim = v.dim;
    RELAX_TRY(ensure_buf(h, h->arena, vit_arena_bytes(v, N)));
    const size_t n = (size_t)N;
    float* P = static_cast<float*>(h->arena.p);
    float* PE = P + n * NPATCH * PATCH_K;
    float* X = PE + n * NPATCH * dim;
    float* Y = X + n * NTOK * dim;
    float* QKV = Y + n * NTOK * dim;
    float* Hid = QKV + n * NTOK * dim * 3;
    const int rows = N * NTOK;

    if (h->gemm.precision == 3 && dim % 256 == 0) {
        // f16x2: GEMM inputs travel as two fp16 planes of (value x a static power of two), written by the kernel that produces them;
        // GEMM outputs that feed LayerNorm / attention / the residual stream stay fp32.  The buffers of the bf16x6 layout are reused
        // (4-byte plane values fit its 6-byte slots).  Attention keeps its bf16x6 arithmetic inside (attention_x6.hip: 4 % of the FLOPs).
        float* base = static_cast<float*>(h->arena.p);
        char* Ps = reinterpret_cast<char*>(base);
        float* PEx = base + n * NPATCH * PATCH_K * 3 / 2;
        float* Xx = PEx + n * NPATCH * dim;
        char* Ys = reinterpret_cast<char*>(Xx + n * NTOK * dim);
        float* QKVx = reinterpret_cast<float*>(Ys) + n * NTOK * dim * 3 / 2;
        float* ATT = QKVx + n * NTOK * dim * 3;
        char* Hs = reinterpret_cast<char*>(ATT + n * NTOK * dim);
        auto gemm = [&](const void* A, const LinearW& l, const float* residual, float* out, void* out_h2, float out_scale, int M, int act) {
            GemmDescH2 d{};
            d.a = A; d.w = l.w_h2; d.colscale = l.colscale; d.bias = l.b; d.residual = residual; d.out = out; d.out_h2 = out_h2;
            d.out_scale = out_scale; d.M = M; d.N = l.out; d.K = l.in; d.act = act;
            return launch_gemm_h2(h, d, s);
        };
        const int64_t p8 = (int64_t)N * NPATCH * (PATCH_K / 8);
        hipLaunchKernelGGL(vit_patchify_h2, dim3((unsigned)((p8 + 255) / 256)), dim3(256), 0, s, frags, Ps, p8, kPatchScale);
        RELAX_TRY(gemm(Ps, v.patch, nullptr, PEx, nullptr, 0.f, N * NPATCH, 0));
        const int64_t at = (int64_t)rows * (dim / 4);
        hipLaunchKernelGGL(vit_assemble, dim3((unsigned)((at + 255) / 256)), dim3(256), 0, s, PEx, v.cls, v.pos, Xx, dim / 4, at);
        RELAX_HIP_CHECK(h, hipGetLastError());
        for (const VitBlockW& b : v.blocks) {
            RELAX_TRY(launch_layernorm_h2(h, Xx, b.ln1_g, b.ln1_b, Ys, b.s_ln1, rows, dim, kLnEps, s));
            if (h->gemm.att_h2) {   // q, k, v leave the GEMM as fp16 planes (the same 4 bytes per value) and attention reads them as they are
                RELAX_TRY(gemm(Ys, b.qkv, nullptr, nullptr, QKVx, b.s_qkv, rows, 0));
                RELAX_TRY(launch_attention_h2(h, QKVx, b.s_qkv, Ys, b.s_att, N, v.heads, s));
            } else {
                RELAX_TRY(gemm(Ys, b.qkv, nullptr, QKVx, nullptr, 0.f, rows, 0));
                RELAX_TRY(launch_attention_x6(h, QKVx, nullptr, Ys, N, v.heads, s, b.s_att));   // output straight into fp16 planes
            }
            RELAX_TRY(gemm(Ys, b.proj, Xx, Xx, nullptr, 0.f, rows, 0));                     // x += proj(attn)
            RELAX_TRY(launch_layernorm_h2(h, Xx, b.ln2_g, b.ln2_b, Ys, b.s_ln2, rows, dim, kLnEps, s));
            RELAX_TRY(gemm(Ys, b.fc1, nullptr, nullptr, Hs, b.s_hid, rows, 2));              // GELU(erf) -> fp16 planes
            RELAX_TRY(gemm(Hs, b.fc2, Xx, Xx, nullptr, 0.f, rows, 0));                      // x += mlp
        }
        RELAX_TRY(launch_layernorm(h, Xx, v.norm_g, v.norm_b, ATT, rows, dim, kLnEps, s));
        if (tokens) {
            const int64_t t = (int64_t)N * NPATCH * (dim / 4);
            hipLaunchKernelGGL(vit_drop_cls, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, ATT, tokens, dim / 4, t);
        }
        if (pooled) hipLaunchKernelGGL(vit_token_stats, dim3(dim / 64, N), dim3(256), 0, s, ATT, pooled, dim, NTOK, 1, NPATCH);
        RELAX_HIP_CHECK(h, hipGetLastError());
        return RELAX_OK;
    }

    if (h->gemm.precision >= 2) {   // (3 with a dim the f16x2 tile does not take, e.g. vit_tiny's 192: bf16x6)
        // bf16x6: GEMM inputs travel as split planes (written by the kernel that produces them), GEMM outputs that feed
        // LayerNorm / attention / the residual stream stay fp32
        float* base = static_cast<float*>(h->arena.p);
        char* Ps = reinterpret_cast<char*>(base);
        float* PEx = base + n * NPATCH * PATCH_K * 3 / 2;
        float* Xx = PEx + n * NPATCH * dim;
        char* Ys = reinterpret_cast<char*>(Xx + n * NTOK * dim);
        float* QKVx = reinterpret_cast<float*>(Ys) + n * NTOK * dim * 3 / 2;
        float* ATT = QKVx + n * NTOK * dim * 3;
        char* Hs = reinterpret_cast<char*>(ATT + n * NTOK * dim);
        const int64_t p8 = (int64_t)N * NPATCH * (PATCH_K / 8);
        hipLaunchKernelGGL(vit_patchify_sp3, dim3((unsigned)((p8 + 255) / 256)), dim3(256), 0, s, frags, Ps, p8);
        RELAX_TRY(launch_gemm_x6(h, Ps, v.patch.w_sp3, v.patch.b, nullptr, PEx, nullptr, N * NPATCH, dim, PATCH_K, 0, s));
        const int64_t at = (int64_t)rows * (dim / 4);
        hipLaunchKernelGGL(vit_assemble, dim3((unsigned)((at + 255) / 256)), dim3(256), 0, s, PEx, v.cls, v.pos, Xx, dim / 4, at);
        RELAX_HIP_CHECK(h, hipGetLastError());
        for (const VitBlockW& b : v.blocks) {
            RELAX_TRY(launch_layernorm_sp3(h, Xx, b.ln1_g, b.ln1_b, Ys, rows, dim, kLnEps, s));
            RELAX_TRY(launch_gemm_x6(h, Ys, b.qkv.w_sp3, b.qkv.b, nullptr, QKVx, nullptr, rows, 3 * dim, dim, 0, s));
            RELAX_TRY(launch_attention_x6(h, QKVx, nullptr, Ys, N, v.heads, s));   // output straight into split planes
            RELAX_TRY(launch_gemm_x6(h, Ys, b.proj.w_sp3, b.proj.b, Xx, Xx, nullptr, rows, dim, dim, 0, s));      // x += proj(attn)
            RELAX_TRY(launch_layernorm_sp3(h, Xx, b.ln2_g, b.ln2_b, Ys, rows, dim, kLnEps, s));
            RELAX_TRY(launch_gemm_x6(h, Ys, b.fc1.w_sp3, b.fc1.b, nullptr, nullptr, Hs, rows, 4 * dim, dim, 2, s)); // GELU(erf) -> sp3
            RELAX_TRY(launch_gemm_x6(h, Hs, b.fc2.w_sp3, b.fc2.b, Xx, Xx, nullptr, rows, dim, 4 * dim, 0, s));     // x += mlp
        }
        RELAX_TRY(launch_layernorm(h, Xx, v.norm_g, v.norm_b, ATT, rows, dim, kLnEps, s));
        if (tokens) {
            const int64_t t = (int64_t)N * NPATCH * (dim / 4);
            hipLaunchKernelGGL(vit_drop_cls, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, ATT, tokens, dim / 4, t);
        }
        if (pooled) hipLaunchKernelGGL(vit_token_stats, dim3(dim / 64, N), dim3(256), 0, s, ATT, pooled, dim, NTOK, 1, NPATCH);
        RELAX_HIP_CHECK(h, hipGetLastError());
        return RELAX_OK;
    }

    const int64_t ptotal = (int64_t)N * NPATCH * PATCH_K;
    hipLaunchKernelGGL(vit_patchify, dim3((unsigned)((ptotal + 255) / 256)), dim3(256), 0, s, frags, P, ptotal);
    RELAX_TRY(launch_gemm(h, P, v.patch.w, v.patch.b, nullptr, PE, N * NPATCH, dim, PATCH_K, 0, s));
    const int64_t atotal = (int64_t)rows * (dim / 4);
    hipLaunchKernelGGL(vit_assemble, dim3((unsigned)((atotal + 255) / 256)), dim3(256), 0, s, PE, v.cls, v.pos, X,
                       dim / 4, atotal);
    RELAX_HIP_CHECK(h, hipGetLastError());
    for (const VitBlockW& b : v.blocks) {
        RELAX_TRY(launch_layernorm(h, X, b.ln1_g, b.ln1_b, Y, rows, dim, kLnEps, s));
        RELAX_TRY(launch_gemm(h, Y, b.qkv.w, b.qkv.b, nullptr, QKV, rows, 3 * dim, dim, 0, s));
        RELAX_TRY(launch_attention(h, QKV, Y, N, v.heads, s));
        RELAX_TRY(launch_gemm(h, Y, b.proj.w, b.proj.b, X, X, rows, dim, dim, 0, s));       // x += proj(attn)
        RELAX_TRY(launch_layernorm(h, X, b.ln2_g, b.ln2_b, Y, rows, dim, kLnEps, s));
        RELAX_TRY(launch_gemm(h, Y, b.fc1.w, b.fc1.b, nullptr, Hid, rows, 4 * dim, dim, 2, s));   // GELU(erf)
        RELAX_TRY(launch_gemm(h, Hid, b.fc2.w, b.fc2.b, X, X, rows, dim, 4 * dim, 0, s));   // x += mlp
    }
    RELAX_TRY(launch_layernorm(h, X, v.norm_g, v.norm_b, Y, rows, dim, kLnEps, s));
    if (tokens) {
        const int64_t t = (int64_t)N * NPATCH * (dim / 4);
        hipLaunchKernelGGL(vit_drop_cls, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, Y, tokens, dim / 4, t);
    }
    if (pooled) hipLaunchKernelGGL(vit_token_stats, dim3(dim / 64, N), dim3(256), 0, s, Y, pooled, dim, NTOK, 1, NPATCH);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}


int relax_op_token_stats(relax_handle* h, const float* x, float* out, int Nimg, int tokens, int dim,
                         relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, x && out && Nimg > 0 && tokens > 0 && dim > 0 && dim % 64 == 0,
                  "relax_op_token_stats: bad arguments (dim must be a multiple of 64)");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(vit_token_stats, dim3(dim / 64, Nimg), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, dim,
                       tokens, 0, tokens);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

}  // extern "C"
