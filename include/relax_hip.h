/*
 * relax_hip.h — C-ABI of librelax_hip.so, the MI355X (gfx950) engine for the
 * ReLaX-VQA feature-extraction hot path.
 *
 * The reference (xinyiW915/ReLaX-VQA) has no FFI: its de-facto operator API for
 * this path is a set of module-level Python functions that pass PNG paths and
 * numpy arrays.  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference repo).  The Python host layer
 * (relax-vqa_amd/) keeps those function names and binds this header through
 * ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative relax_status otherwise;
 *     relax_last_error() gives the message (Python raises RuntimeError).
 *   - caller allocates: every in/out buffer is a DEVICE pointer (e.g.
 *     torch.Tensor.data_ptr() under PyTorch-ROCm) unless the name says host.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  All work
 *     is enqueued on it; no hidden synchronisation.
 *   - a handle owns only weights + workspace; it is NOT thread-safe; one handle
 *     per (process, device).  The workspace is reused from call to call, so calls on
 *     one handle must be ordered: issue them on ONE stream (or order the streams with
 *     events); two backbone calls running concurrently on different streams would race.
 *   - images are uint8 HWC **BGR** exactly as cv2.imread holds them
 *     (src/main_fragment_layerstack.py:295-296); fragments are 224x224x3.
 */
#ifndef RELAX_HIP_H
#define RELAX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELAX_ABI_VERSION 1

#define RELAX_PATCH 16           /* patch_size  (main_fragment_layerstack.py:298) */
#define RELAX_TARGET 224         /* target_size (main_fragment_layerstack.py:297) */
#define RELAX_TOP_N 196          /* top_n = (224/16)^2 (main_fragment_layerstack.py:299) */
#define RELAX_FRAG_BYTES (224 * 224 * 3)
#define RELAX_RN50_LAYER_STACK_DIM 13120 /* 64+3*256+4*512+4*1024+3*2048 */
#define RELAX_RN50_POOL_DIM 2051         /* 2048 + mean,max,std */
#define RELAX_RN50_NUM_TAPS 15

typedef enum relax_status {
    RELAX_OK = 0,
    RELAX_ERR_INVALID = -1,   /* bad argument */
    RELAX_ERR_HIP = -2,       /* a HIP runtime call failed */
    RELAX_ERR_STATE = -3,     /* weights not loaded / workspace too small */
    RELAX_ERR_NOMEM = -4
} relax_status;

typedef struct relax_handle relax_handle;
typedef void* relax_stream;

/* ---- lifetime ------------------------------------------------------------------------------ */
int relax_abi_version(void);
int relax_create(int device, relax_handle** out);
int relax_destroy(relax_handle* h);
/* message of the last failing call on this handle (h may be NULL: last create error) */
const char* relax_last_error(const relax_handle* h);

/* Size the activation workspace for batches of up to max_images fragments
 * (ResNet-50 and ViT share one arena).  Called implicitly (growing) by the
 * backbone entry points; call it up front to keep allocation out of timed code. */
int relax_reserve(relax_handle* h, int max_images);

/* Integer options.  "gemm_precision": 3 (default) = "f16x2" (fp32-grade): every fp32 operand is held as two fp16 numbers of a
 * power-of-two multiple of itself, x * s = hi + lo (22 bits), and a*b = ah*bh + ah*bl + al*bh on v_mfma_f32_16x16x32_f16 with fp32
 * accumulation (al*bl, 2^-22 of the product, is added for K < 256 only); every scale comes from a bound, never from the data of a
 * batch - weights per output row, ViT activations one static power of two per tensor, ResNet-50 activations one per image from
 * Hoelder's inequality on the measured per-image maxima of the producer's inputs - so nothing can overflow and no row depends on its
 * batch (csrc/gemm_h2.hip, csrc/h2.h, tests/test_gpu_h2.py).  It covers the plain GEMMs with N % 256 == 0 (the whole ViT-B/16;
 * relax_op_gemm), the convolutions of ResNet-50's layer3 / layer4 (relax_op_conv2d_nhwc with Cin % 32 == 0, Cout % 256 == 0) and the
 * 3x3 convolutions of its layer1 / layer2 (K x K filters onto 64 / 128 channels, K >= 256: v_mfma_f32_32x32x16_f16 on the four-wave
 * tiles of csrc/gemm_x6.hip, the two small products in an accumulator of their own), its stem (csrc/conv1_x6.hip) and the ViT's attention (csrc/attention_h2.hip;
 * relax_op_attention); everything else runs as under 2.  0 = exact fp32 products (v_mfma_f32_32x32x2_f32); 2 = "bf16x6" (fp32-grade):
 * every fp32 operand is held as three bf16 numbers hi + mid + lo (exact) and a*b = the six partial products of weight
 * >= 2^-16 on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, two products per instruction; 32x32x16 on the 64 / 128-column
 * tiles) with fp32 accumulation - as close to the exact sum as the fp32 FMA chain
 * (csrc/gemm_x6.hip, tests/test_gpu_x6.py); used by the ViT / ResNet drivers and by relax_op_gemm; 1 = "bf16x3":
 * every fp32 operand is split on the fly into bf16 hi + lo and a*b = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16
 * with fp32 accumulation (about 2^-16 relative error per product; measured features within ~1e-5 of the fp32 path).
 * "gemm_split_k" (default 1): cut the tail tiles of a contraction along K so the last round fills
 * the chip; results stay deterministic for a given batch, but the K-summation order of tail tiles then depends on the
 * batch size - set 0 when features must be bit-identical across batch compositions (e.g. comparing sharded runs).
 * "gemm_variant", "gemm_variant_n64" (exact-fp32 kernel only), "gemm_group_m": tuning knobs (tile variants are listed in
 * csrc/gemm.hip; none of them changes results beyond fp32 rounding).  "flow_max_pairs": cap on the pairs one optical-flow launch takes
 * (0 = by workspace size).  "x6_fp32_rows" (default 1): bf16x6 contractions onto 64 / 128 output columns read fp32 activation rows and
 * split them in the K loop, and the ResNet-50 block outputs inside layer1 / layer2 travel as fp32 (4 bytes per value instead of 6);
 * 0 = split planes everywhere: the same bits, more bytes (kept as the A/B switch of tests/test_gpu_x6.py).
 * "h2_form" (default 1): 1 = the f16x2 loop with 32-deep K steps and three products for K >= 256; 0 = 16-deep steps, four products
 * ("h2_stages" = 3 or 4 LDS stages, same bits); 2 = 32-deep steps, four products at every K.  "rn_h2" (default 1): under
 * "gemm_precision" 3 ResNet-50's layer3 / layer4 run f16x2; 0 = the whole network on bf16x6 (the A/B switch of tests/test_gpu_h2.py).
 * "rn_h2_early" (default 1): with "rn_h2", the stem and the 3x3 convolutions of layer1 / layer2 run f16x2 as well; 0 = bf16x6 there.
 * "att_h2" (default 1): under "gemm_precision" 3 the ViT's attention runs on fp16 planes too (csrc/attention_h2.hip: the qkv GEMM writes planes,
 * three partial products, K / V by LDS-DMA into the fragment images); 0 = the bf16x6 attention kernel on an fp32 qkv output (A/B switch).
 * "rn_fuse" (default 1): with "rn_h2_early", the layer1 / layer2 blocks without a downsample branch run their 3x3 and their conv3 back to back in ONE
 * launch (csrc/gemm_x6.hip, B2B: the 3x3's output tile stays in registers as the A operand of the 1x1, conv3 on f16x2 with one scale
 * per pixel row); 0 = two launches, conv3 on bf16x6 (A/B switch).  "b2b_rows" (256 or 128): rows per tile of layer1's such launches, same bits.
 * "rn_c1_h2" (default 1): layer2's conv1 (1x1, fp32 rows in) on f16x2 - the rows are split into two fp16 planes in the K loop with the image's scale -;
 * 0 = bf16x6 (A/B switch).
 * "debug_poison" (test mode, default 0): every workspace request fills the buffer with 0xFF bytes
 * first (synchronously), so a read of workspace that was not written in the same call shows up in the results. */
int relax_set_option(relax_handle* h, const char* key, int value);
/* Reads an option back (bench.py reports the arithmetic the ENGINE is in, not the one its command line asked for).  Two read-only
 * counters for leak checks of long passes (tests/test_gpu_soak.py): "profile_events" = HIP events the handle owns (pooled + in open spans),
 * "workspace_mib" = MiB of its device workspaces (they grow to the largest batch seen and stay). */
int relax_get_option(relax_handle* h, const char* key, int* value);

/* ---- weights ------------------------------------------------------------------------------- */
/* Replaces `models.resnet50(pretrained=True)` (src/extractor/visualise_resnet.py:21,
 * visualise_resnet_layer.py:20).  names[i] are torchvision state-dict keys
 * ("conv1.weight", "layer1.0.bn1.running_var", ...); tensors[i] are HOST fp32
 * pointers in PyTorch layout (OIHW for convs); numels[i] their element counts.
 * BatchNorm (eps 1e-5, eval) is folded into the following conv's weights/bias,
 * except bn1 which is applied after the raw `conv1` tap. "fc.*" is ignored. */
int relax_load_resnet50(relax_handle* h, const float* const* tensors, const char* const* names,
                        const int64_t* numels, int n);

/* Replaces VitGenerator(name_model, patch_size=16, ...) + load_state_dict
 * (src/extractor/visualise_vit_layer.py:263-329).  DINO state-dict keys
 * ("cls_token", "blocks.0.attn.qkv.weight", ...).  dim/depth/heads: 768/12/12
 * for vit_base (:287-289); head_dim must be 64, input is always 224x224 (197 tokens). */
int relax_load_vit(relax_handle* h, const float* const* tensors, const char* const* names,
                   const int64_t* numels, int n, int dim, int depth, int heads);

/* ---- stage A: residual -> patch score -> top-n -> fragments (bit-exact integer path) -------- */
/* Replaces, per (frame, next) pair: cv2.absdiff (main_fragment_layerstack.py:302),
 * process_patches('frame_diff') = get_patch_diff + extract_important_patches (:232-240,
 * :177-210) and get_original_frame_patches (:212-230).
 *   orig, next : uint8 [H,W,3] per pair; pair t starts at orig + t*pair_stride (bytes)
 *   positions  : int32 [T,196,2] (y,x) patch coordinates in raster order, (-1,-1) past counts[t]
 *   counts     : int32 [T]  = min(top_n, (H/16)*(W/16))
 *   ori_frag   : uint8 [T,224,224,3] patches of `orig`   (may be NULL)
 *   diff_frag  : uint8 [T,224,224,3] patches of |next-orig| (may be NULL)
 *   scores     : uint32 [T,(H/16)*(W/16)] patch sums (may be NULL -> internal scratch)
 * Tie rule: higher score first, then lower flat patch index. */
int relax_fragment_pairs(relax_handle* h, const uint8_t* orig, const uint8_t* next, int64_t pair_stride,
                         int T, int H, int W, int top_n, int32_t* positions, int32_t* counts,
                         uint8_t* ori_frag, uint8_t* diff_frag, uint32_t* scores, relax_stream stream);

/* Same selection on an already-computed residual image (the optical-flow image of
 * process_patches('optical_flow'), main_fragment_layerstack.py:319; main_residual_fragment.py:206-214).
 *   image: uint8 [H,W,3] per item, item t at image + t*item_stride. */
int relax_fragment_image(relax_handle* h, const uint8_t* image, int64_t item_stride, int T, int H, int W,
                         int top_n, int32_t* positions, int32_t* counts, uint8_t* frag, uint32_t* scores,
                         relax_stream stream);

/* get_original_frame_patches (main_fragment_layerstack.py:212-230) with given positions. */
int relax_gather_patches(relax_handle* h, const uint8_t* image, int64_t item_stride, int T, int H, int W,
                         const int32_t* positions, const int32_t* counts, uint8_t* frag, relax_stream stream);

/* merge_fragments = cv2.addWeighted(a,.5,b,.5,0) (main_fragment_layerstack.py:242-245):
 * round-half-to-even(0.5a+0.5b) on uint8. n_bytes elements. */
int relax_merge_fragments(relax_handle* h, const uint8_t* a, const uint8_t* b, uint8_t* out, int64_t n_bytes,
                          relax_stream stream);

/* ---- optical flow --------------------------------------------------------------- */
/* Replaces cv2.calcOpticalFlowFarneback(gray(orig), gray(next), None, 0.5, 3, 15, 3, 5, 1.2, 0) and flow_to_rgb
 * (src/main_fragment_layerstack.py:313-316, 162-175; src/main_residual_fragment.py:283-287).  Parameters are the
 * reference's literals.  orig/next as in relax_fragment_pairs.
 *   flow     : fp32 [T,H,W,2] (dx,dy)                       (may be NULL)
 *   flow_bgr : uint8 [T,H,W,3] the visualisation image that process_patches('optical_flow', ...) consumes (may be NULL)
 * OpenCV's algorithm restated (not ported); parity is by tolerance against the reference's example flow PNGs. */
int relax_optical_flow(relax_handle* h, const uint8_t* orig, const uint8_t* next, int64_t pair_stride, int T, int H, int W,
                       float* flow, uint8_t* flow_bgr, relax_stream stream);
/* flow_to_rgb alone: fp32 [T,H,W,2] -> uint8 [T,H,W,3] (BGR, despite the reference's name). */
int relax_flow_to_rgb(relax_handle* h, const float* flow, int T, int H, int W, uint8_t* flow_bgr, relax_stream stream);

/* ---- whole-frame front-end ------------------------------------------------------ */
/* H x W -> 224 x 224, bit-identical to Pillow's 8-bit resample:
 *   out_bilinear : what transforms.Resize((224,224)) gives a PIL image (src/extractor/visualise_resnet.py:40-47)
 *   out_lanczos  : img.resize((224,224), Image.Resampling.LANCZOS)     (src/extractor/visualise_vit_layer.py:466-469)
 * frames uint8 [H,W,3] per item (item n at frames + n*item_stride; channel order is irrelevant), outputs
 * uint8 [N,224,224,3]; either output may be NULL.  One read of each frame serves both filters. */
int relax_resize_frames(relax_handle* h, const uint8_t* frames, int64_t item_stride, int N, int H, int W,
                        uint8_t* out_bilinear, uint8_t* out_lanczos, relax_stream stream);

/* ---- stage B: backbones ---------------------------------------------------------------------- */
/* ResNet-50 on N fragments (uint8 [N,224,224,3] BGR).  One forward per image yields everything
 * the reference gets from 15 hooked forwards + 1 avgpool forward:
 *   layer_stack : fp32 [N,13120] = get_deep_feature(..,'layer_stack') + process_video_feature(..,'layer_stack')
 *                 (main_fragment_layerstack.py:91-96,134-140; extractor/visualise_resnet.py:62-109)   (may be NULL)
 *   pool        : fp32 [N,2051]  = get_deep_feature(..,'pool') + process_video_feature(..,'pool')
 *                 (main_fragment_layerstack.py:97-99,141-149; extractor/visualise_resnet_layer.py:62-102) (may be NULL)
 *   taps_nchw   : NULL, or RELAX_RN50_NUM_TAPS device pointers (each NULL or fp32 [N,C,H,W]) receiving the
 *                 hooked activations themselves (visualise_resnet.process_video_frame's dict values). */
int relax_resnet50_features(relax_handle* h, const uint8_t* frags, int N, float* layer_stack, float* pool,
                            float* const* taps_nchw, relax_stream stream);

/* The clip path asks different things of the two groups of its batch: the layer stack of the ORIGINAL fragments and the pool
 * vector of the RESIDUAL fragments (src/main_fragment_layerstack.py:327-328: get_deep_feature(.., original_frag_path, ..,
 * 'layer_stack') and get_deep_feature(.., merged_frag_path, .., 'pool'); :340-341).  ONE forward over all N images:
 *   images [0, n_layer_stack)  -> layer_stack fp32 [n_layer_stack, 13120]
 *   images [n_layer_stack, N)  -> pool        fp32 [N - n_layer_stack, 2051]
 * Same values as relax_resnet50_features on the respective images; the taps of the second group are neither reduced nor written
 * as fp32, the pool statistics of the first group are not formed.  Either group may be empty (its pointer may then be NULL). */
int relax_resnet50_clip_features(relax_handle* h, const uint8_t* frags, int N, int n_layer_stack, float* layer_stack, float* pool,
                                 relax_stream stream);

/* ViT on N fragments.  tokens: fp32 [N,196,dim] final-norm patch tokens
 * (visualise_vit_layer.process_video_frame, :447-500) (may be NULL);
 * pooled: fp32 [N,3*dim] mean|max|std over tokens (main_fragment_pool.py:124-133) (may be NULL). */
int relax_vit_features(relax_handle* h, const uint8_t* frags, int N, float* tokens, float* pooled,
                       relax_stream stream);

/* ---- quality head at inference --------------------------------------------------- */
/* Replaces imputer.transform + scaler.transform + Mlp.forward (src/demo_test.py:177-208, src/model_regression.py:37-58).
 * State-dict keys of the reference's Mlp (fc1/bn1/fc2/fc3; a 'module.' prefix is stripped and 'n_averaged' ignored as
 * fix_state_dict does, demo_test.py:25-35); HOST pointers.  imputer_statistics = SimpleImputer.statistics_ (may be
 * NULL), scaler_scale / scaler_min = MinMaxScaler.scale_ / .min_, all HOST float64 [input_features]. */
int relax_load_mlp_head(relax_handle* h, const float* const* tensors, const char* const* names, const int64_t* numels,
                        int n, const double* imputer_statistics, const double* scaler_scale, const double* scaler_min,
                        int input_features);
/* features: device fp32 [n, input_features] (the all-gathered per-clip vectors) -> scores: device fp32 [n]. */
int relax_mlp_head(relax_handle* h, const float* features, int n, float* scores, relax_stream stream);

/* ---- operator level (what the backbones are built from; parity-tested one by one) ------------ */
/* out[M,N] = act(A[M,K] * W[N,K]^T + bias[N] + residual[M,N]);  act: 0 none, 1 relu, 2 gelu(erf).
 * fp32 in, fp32 MFMA accumulate.  K % 32 == 0 (bf16x6: K % 16 == 0), N % 64 == 0.  bias/residual may be NULL; every pointer
 * 16-byte aligned.  These operator-level entry points are test / bench paths: under "gemm_precision" 3 (default; N % 256 == 0)
 * both operands are converted to fp16 planes on every call, each row with the power-of-two scale of its own maximum; under 2 (and under
 * 3 for the other shapes) the operands are converted to split planes on every call (two extra kernels, (M+N)*K*6 bytes; the model drivers keep weights
 * and activations in that format instead; where N % 256 != 0 only W is converted: the 64 / 128-column form of the kernel
 * splits the fp32 rows of A in its K loop, as ResNet-50's layer1 / layer2 do), into a workspace the handle owns - like every entry point they must not run
 * concurrently on two streams of one handle.  Finite operands up to 3.38e38 (csrc/sp3.h); beyond: "gemm_precision" 0. */
int relax_op_gemm(relax_handle* h, const float* A, const float* W, const float* bias, const float* residual,
                  float* out, int M, int N, int K, int act, relax_stream stream);
/* NHWC conv as implicit GEMM: in [Nimg,H,W,Cin], w [Cout, KH*KW*Cin (padded to %32)] (k = (dy*KW+dx)*Cin+c),
 * out [Nimg,Ho,Wo,Cout].  Cin a power of two >= 4 when KH*KW > 1.  Under "gemm_precision" 2 the bf16x6 kernel takes
 * geometries with Cin % 16 == 0, KH*KW <= 32 and Cout % 64 == 0; every other one runs on the exact-fp32 kernel. */
int relax_op_conv2d_nhwc(relax_handle* h, const float* in, const float* w, const float* bias,
                         const float* residual, float* out, int Nimg, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, int act, relax_stream stream);
/* rows of `dim` floats: y = (x-mean)/sqrt(var+eps)*gamma+beta */
int relax_op_layernorm(relax_handle* h, const float* x, const float* gamma, const float* beta, float* y,
                       int rows, int dim, float eps, relax_stream stream);
/* qkv [Nimg*197, 3*heads*64] -> out [Nimg*197, heads*64]; softmax(q k^T / 8) v per (image, head) */
int relax_op_attention(relax_handle* h, const float* qkv, float* out, int Nimg, int heads, relax_stream stream);
/* relu(x*scale[c]+shift[c]) then 3x3/s2/p1 max-pool: [Nimg,H,W,C] -> [Nimg,H/2,W/2,C] */
int relax_op_bn_relu_maxpool(relax_handle* h, const float* x, const float* scale, const float* shift, float* y,
                             int Nimg, int H, int W, int C, relax_stream stream);
/* spatial mean: x [Nimg,HW,C] -> out[n*out_stride + c] */
int relax_op_gap(relax_handle* h, const float* x, float* out, int Nimg, int HW, int C, int64_t out_stride,
                 relax_stream stream);

/* x [Nimg, tokens, dim] -> out [Nimg, 3*dim] = per-channel mean | max | population std over tokens
 * (process_video_feature, vit branch: src/main_residual_fragment.py:128-136; src/main_fragment_pool.py:124-133) */
int relax_op_token_stats(relax_handle* h, const float* x, float* out, int Nimg, int tokens, int dim,
                         relax_stream stream);

/* Device-to-device copy on the caller's stream (the fragment batch of one backbone duplicated for the other). */
int relax_copy_bytes(relax_handle* h, const void* src, void* dst, int64_t n_bytes, relax_stream stream);

/* Per-clip mean over frames (src/demo_test.py:171-175, src/data_processing/extract_npy2mat.py:121-126) of one column block
 * of a per-frame feature matrix: dst[s, dst_col0 + c] = mean of src[row0 + r, c] over r in [seg_offsets[s], seg_offsets[s+1]),
 * c < ncols.  seg_offsets: HOST int32 [nseg + 1] (prefix sums of the frames per clip; passed to the kernel by value, so the
 * call stays capturable into a HIP graph).  Rows are summed in order. */
int relax_segment_mean(relax_handle* h, const float* src, int64_t src_stride, int ncols, int row0, const int32_t* seg_offsets,
                       int nseg, float* dst, int64_t dst_stride, int dst_col0, relax_stream stream);

/* ---- measurement ----------------------------------------------------------------------------- */
/* While enabled, every launch of the contraction kernel (GEMM / implicit-GEMM conv) and of the patch-score
 * kernel is bracketed by HIP events on the caller's stream.  relax_profile_read synchronises those events and
 * returns totals since the last enable: kind 0 = fp32 / bf16x3 contraction launches (work = algorithmic FLOPs), kind 1 =
 * patch score (work = bytes), kind 2 = kind 0 again with work = algorithmic HBM bytes (operands and results touched once),
 * kind 3 / 4 = the same two views of the bf16x6 contraction launches, kind 7 / 8 = those of the f16x2 launches, kind 9 / 10 = those of the
 * plain f16x2 GEMMs alone (gemm_h3 without the convolution form: the ViT's GEMMs, the dominant kernel of the headline), kind 5 = flow_iteration, the dominant kernel of the
 * Farneback stage (one launch per iteration; work = algorithmic bytes: 56 per pixel, level and iteration), kind 6 = the whole
 * Farneback stage of a relax_optical_flow chunk, first launch to last (work = the algorithmic bytes of all its kernels: every
 * kernel's inputs read once and outputs written once; launches = chunks). */
int relax_profile_enable(relax_handle* h, int on);
int relax_profile_read(relax_handle* h, int kind, double* total_ms, double* total_work, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* RELAX_HIP_H */
