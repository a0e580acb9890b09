// "bf16x6": the fp32-grade contraction kernel on the bf16 matrix cores of gfx950.
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] + residual[m, n] )          (same contract as gemm.hip)
//
// Every fp32 operand x is held as THREE bf16 numbers x = hi + mid + lo (each a round-to-nearest of what is left:
// 3 x 8 = 24 significant bits, the split is exact), and a*b is evaluated with the six partial products whose weight is
// at least 2^-16 of the full product,
//        a*b ~= al*bh + am*bm + ah*bl + am*bh + ah*bm + ah*bh          (dropped: am*bl, al*bm ~2^-24, al*bl ~2^-32)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: the error per product is of the size of one fp32 rounding, and
// the accumulator is rounded once per 16-deep MFMA instead of once per product, so the result is as close to the
// exact sum as the fp32 FMA chain of gemm.hip (measured, tests/test_gpu_x6.py) at 6/16 of its matrix-pipe cycles.
//
// Operand format "sp3" (split planes): a row of K fp32 values is stored as K/16 chunks of 96 bytes,
//        chunk = [16 x bf16 hi][16 x bf16 mid][16 x bf16 lo]
// so one 16-deep K step of one row is 96 contiguous bytes whose 16-byte units are exactly the MFMA fragments
// (lane (r, h) of a 32x32x16 MFMA holds k = 8h .. 8h+7 of row r).  The split happens ONCE where a value is produced
// (weights at load time, activations in the producing kernel's epilogue / LayerNorm / patchify / max-pool), never in the
// K loop: the loop has no VALU work, and tiles go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) without
// touching registers.  (One exception, the AF32 form below.)
//
// Structure: BK = 16, LDS stages of (BM + BN) rows x 96 B filled by DMA, one raw s_barrier per K step; the K loop is rotated into
// barrier-to-barrier "regions" that read the fragments of the next step ahead of their MFMAs; counted s_waitcnt vmcnt keeps
// loads in flight across barriers.  Two forms of the loop:
//   * 256x256 tile on 8 waves (2 x 4, 128x64 per wave, one workgroup per CU), template flag M16: the six products on
//     v_mfma_f32_16x16x32_bf16, TWO per instruction (the 32-deep K of the instruction takes two planes of the chunk), B forms
//     double-buffered in registers, A fragments streamed; THREE stages (144 KB) with the DMA pieces of step k+3 issued one per
//     MFMA group (plain GEMMs; the implicit-GEMM and two-source instantiations: two stages).  See the comments at X6_MFMAS16 /
//     X6_REGION16 for why (the chip's clock on this shape; the texture-address FIFO).
//   * 256x128 / 256x64 tiles on 4 waves (two workgroups per CU) for N = 128 / 64: v_mfma_f32_32x32x16_bf16, one product per
//     instruction, two stages, X operand double-buffered in registers, Y in halves; LDS rows hold the two 16-byte halves of a
//     plane swapped in rows with bit 3 set, which makes these ds_read_b128 fragment reads bank-conflict free (the permutation
//     is applied on the SOURCE offset of the DMA; the M16 form uses the linear image, conflict-free for ITS lane map).
//     AF32 (plain GEMMs on these tiles): the activation rows arrive as fp32 (64 B per row and K step, unit q of row r in slot
//     q ^ ((r >> 2) & 3) of the row: conflict-free ds_read_b128) and the lane splits its 8 values into the three planes right after
//     the fragment read.  N = 64 / 128 means ONE column tile, so every activation is split exactly once either way; these launches
//     wait on memory, and their operands (ResNet-50's widest tensors) cost 4 bytes per value instead of 6.  Same values, same
//     products, same order as split planes: bit-identical results (tests/test_gpu_x6.py).
//     H2 (the 3x3 convolutions of ResNet-50's layer1 / layer2 under f16x2): both operands as two fp16 planes (csrc/h2.h: 64 B per row
//     and K step, LDS rows in the AF32 geometry: unit (half h, plane q) of row r in slot (2h + q) ^ ((r >> 2) & 3)), the three
//     partial products al bh, ah bl, ah bh on v_mfma_f32_32x32x16_f16 (K >= 256: al bl is below the accumulator's rounding, as in
//     gemm_h2.hip), half the matrix instructions and two thirds of the staged bytes of the sp3 form; the epilogue undoes the scales
//     (weight row x image: powers of two).  These launches are the only MFMA-bound ones of the two layers.
// Rows beyond M and the padding taps of an implicit-GEMM convolution use an out-of-range buffer offset, which the buffer unit
// answers with zeros.  The last, partial round of tiles is split along K (splitk_finish_x6 adds the slices in a fixed order;
// cost model: host_logic.cpp).  DUAL: a second activation source for the K steps past K1 (ResNet conv3 + downsample in one
// contraction).  The epilogue goes through LDS in 64-row passes so that every store instruction writes whole 512-byte row
// segments, and can add bias / an sp3 or fp32 residual, apply ReLU / GELU, write fp32 (optionally only the first out_rows rows)
// and / or sp3, and emit 16-row column sums for the global-average-pool taps.  What was measured and NOT adopted (staggered
// starts, pinned instruction order of the 32x32 loop, non-temporal stores, 128x256 tiles on two workgroups per CU, K-slice phase
// starts, static wave priorities) is recorded in LAB_NOTES.md sections 3.2 / 3.2.1; none of that code is kept here.
#include "relax_internal.h"
#include "host_logic.h"
#include "sp3.h"
#include "h2.h"
#include "gelu.h"

// Phase-stamp hooks of the diagnostic build: empty in the product.  tools/abl/gemm_x6_stamps.hip defines them (and the report) and
// #includes this file; nothing in this translation unit records or prints a stamp.
#ifndef X6_STAMP
#define X6_STAMP(i_)
#define X6_STAMP_IDS()
#define X6_STAMPS_BEFORE_LAUNCH(h_, p_, units_)
#define X6_STAMPS_AFTER_LAUNCH(BM_, BN_, h_, p_, units_, s_)
#endif

// Hooks of the diagnostic builds tools/abl/gemm_x6_b2b_abl.hip (tools/build_ablations.sh b2babl / b2bpf / x6stg2), neutral here - the product
// compiles these defaults: X6_B2B_ABL (timing-only ablations of the back-to-back tail: bit 0 no fp32 stores, 1 no residual loads, 2 no conv3),
// X6_B2B_PREFETCH_B (same bits: conv3's B fragments requested per K chunk right before their MFMAs - 0, 2.07 ms per launch of layer1 -, a whole
// pass ahead - 1: 16 spilled registers, 2.21 ms -, or half a pass ahead of this pass's stores - 2: 14 spilled, 2.17 ms), X6_H2_STAGES (same bits:
// LDS stages of the f16x2 3x3 loop)
#ifndef X6_B2B_RES_EARLY
#define X6_B2B_RES_EARLY 0   // 1: the residual rows of a pass requested AHEAD of its MFMA phase (128-wide tiles; the 64-wide ones have no registers for it).  Measured slower,
                             // layer2's blocks 1.50 -> 1.55 ms: memory operations return in order, so the first wait for a weight fragment of the phase now waits for the residual rows
#endif
#ifndef X6_B2B_RSETS
#define X6_B2B_RSETS 0   // residual register sets of the back-to-back tail: 0 = a whole pass (default), 4 = the rows in two halves (the old form)
#endif
#ifndef X6_B2B_PREFETCH_B
#define X6_B2B_PREFETCH_B 0
#endif
#ifndef X6_B2B_ABL
#define X6_B2B_ABL 0
#endif
#ifndef X6_H2_STAGES
#define X6_H2_STAGES 3
#endif

namespace relax {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// fp32 [rows][K] (row stride ld floats) -> sp3 [rows][K*6 bytes]; one thread per 8 values
__global__ __launch_bounds__(256) void to_sp3_kernel(const float* __restrict__ x, int64_t ld, char* __restrict__ y, int K,
                                                     int64_t total8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int k8 = K >> 3;
    const int64_t row = i / k8;
    const int k = (int)(i - row * k8) * 8;
    const float* s = x + row * ld + k;
    store_sp3_x8(y + row * (int64_t)K * 6, k, *reinterpret_cast<const f32x4*>(s), *reinterpret_cast<const f32x4*>(s + 4));
}

int launch_to_sp3(relax_handle* h, const float* x, int64_t ld, void* y, int64_t rows, int K, hipStream_t s) {
    RELAX_REQUIRE(h, K % 16 == 0 && ld % 4 == 0 && rows > 0, "to_sp3: K=%d must be a multiple of 16 (ld %lld)", K, (long long)ld);
    const int64_t total8 = rows * (K / 8);
    hipLaunchKernelGGL(to_sp3_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, x, ld, static_cast<char*>(y), K,
                       total8);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- the kernel ------------------------------------------------------------------------------------------------------
struct X6Params {
    const char* a;        // sp3 activations: [M][K*6 B] (plain) or NHWC pixels [Nimg*H*W][Cin*6 B] (implicit GEMM)
    const char* a2;       // DUAL: a second activation source, NHWC pixels [Nimg*H2*W2][Cin2*6 B] sampled with stride2 at the output
                          // pixels: the K steps past K1 read it (K = K1 + Cin2, weight rows concatenated the same way), or null
    const char* w;        // sp3 weights [N][K*6 B], k = (dy*KW+dx)*Cin + c (the K loop visits the 16-channel chunks tap-innermost)
    const float* bias;
    const float* residual;   // fp32 [M][N] or null
    const char* residual_sp3;   // the residual as split planes [M][N*6 B] (exact: hi + mid + lo), or null
    float* gap;              // fused spatial mean, stage 1: sums of the outputs over aligned groups of 16 (or 4: gap_shift) rows [M/16][N], or null
    float* out;              // fp32 [M][N] or null
    char* out_sp3;           // sp3 [M][N*6 B] or null
    char* out_h2;            // the outputs also as two fp16 planes [M][N*4 B], image i's rows scaled by img_out_scale[i] (csrc/h2.h), or null
    const float* img_out_scale;
    const float* colscale;   // H2: [N] inverse weight-row scales
    const float* img_in_inv; // H2: [images] inverse activation scales (image = row / (Ho*Wo))
    const float* img_in_scale;   // H2 + AF32: [images] the scales themselves (the fp32 rows are split into planes inside the K loop)
    unsigned* amax_out;      // [images]: atomicMax of the bits of the (non-negative) outputs of each image, or null (image = row / (Ho*Wo))
    float* partial;          // split-K partial tiles
    int M, N, K;
    int K1, H2, W2, Cin2, stride2;   // DUAL only
    int out_rows, gap_rows;          // rows below these limits get the fp32 output / the group sums (default M)
    int gap_shift;                   // log2 of the rows per group of the fused spatial mean: 4, or 2 for the 14x14 maps (196 rows per image)
    int no_split;
    int H, W, Cin, Ho, Wo, KW, stride, pad;
    int act;
    int tiles_m, tiles_n, ntiles, group_m;
    int full_tiles, nsplit;
    unsigned long long* stamps;   // diagnostic builds only (tools/abl/gemm_x6_stamps.hip); null in the product
    // B2B (back to back): the conv3 of the block - h2 weights [N3][N*4 B] in the permuted K order, their inverse row scales, bias; the
    // output-side fields above (residual, out, out_sp3, out_h2, gap, amax_out, out_rows, gap_rows) then describe the [M][N3] result
    const char* w3;
    const float* colscale3;
    const float* bias3;
    int N3;
    const float* x2;   // B2B == 2: the second source of conv3 (the downsample branch of a block's first bottleneck folded in): fp32 rows [M][64], K = N + 64
    int sp3_sub;   // B2B: 2 = out_sp3 holds only the pixels with even (oy, ox), compacted to [Nimg][Ho/2][Wo/2] rows (what a stride-2 consumer samples); 1 = all
};

__device__ inline int xcd_remap6(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

__device__ inline f32x4 act_gelu4(const f32x4 v) { return gelu_erf4(v); }

// global -> LDS without registers: 16 bytes per lane to LDS address M0 + 16 * lane; the source is base(rsrc) + voff + soff,
// and a lane whose voff is beyond the resource's byte count gets zeros (rows past M, padding taps)
#define X6_DMA(rsrc_, lds_off_, voff_, soff_)                                                                             \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_, (__attribute__((address_space(3))) void*)(smem + (lds_off_)), 16,     \
                                             voff_, soff_, 0, 0)
[[maybe_unused]] constexpr unsigned kOutOfRange = 0x80000000u;   // > every num_records below: the DMA writes zeros for that lane
constexpr int64_t kMaxRecords = 0x7ffffff0;

// Tile BM x BN on WM x WN waves (256x256 on 2 x 4: one workgroup per CU; 256x128 / 256x64 on four waves: two workgroups per
// CU), BK = 16, two LDS stages.  The fragment reads are software-pipelined one half-step ahead of the MFMAs that use them:
//   step k:  [ M0(k): MFMAs on Y-half 0 ]  while reading Y-half 1 of stage k
//            wait own DMA of step k+1, barrier          (stage k is now fully in registers -> free; stage k+1 has landed)
//            issue the DMA of step k+2 into stage k,  read X and Y-half 0 of stage k+1
//            [ M1(k): MFMAs on Y-half 1 ]
// (the loop body is rotated: one "region" = barrier .. next barrier = M1(k) + M0(k+1) is one basic block, which the compiler's
// scheduler interleaves better than any order pinned by hand)
// X = the operand with fewer fragments per wave (double-buffered in registers), Y = the other one (two halves).
// a float at a workgroup-uniform address, read through the constant address space: s_load_dword into a scalar register (memory that no
// launch in flight writes: tables an earlier launch left)
__device__ inline float x6_uniform_load(const float* q) {
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<uintptr_t>(q));
}

template <int BM, int BN, int WM, int WN, bool TAPS, bool M16 = false, bool DUAL = false, bool AF32 = false, bool H2 = false, int B2B = 0>
__global__ __launch_bounds__(WM * WN * 64, (B2B && BM == 128) ? 3 : 2) void gemm_x6(const X6Params p) {   // 2 waves per SIMD: 2 x 4 waves or 1 x 8 (B2B on 128 rows: 3 x 4)
#if __HIP_DEVICE_COMPILE__   // the host pass only needs the launch stub (it has no __amdgpu_buffer_rsrc_t and emits no stub for a body that names it)
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr bool SPLIT_B = TN >= TM;
    constexpr int XT = SPLIT_B ? TM : TN;
    constexpr int YT = SPLIT_B ? TN : TM;
    constexpr int YH = YT / 2;
    constexpr int ROWS = BM + BN;
    constexpr int CH = H2 ? kH2ChunkBytes : kChunkBytes;   // bytes of one 16-deep K step of one operand row (two fp16 / three bf16 planes)
    constexpr int NP = H2 ? 2 : 3;                         // planes
    constexpr int A_ROW = AF32 ? 64 : CH;   // LDS bytes of one activation row of a stage (AF32: 16 fp32 values, split in registers)
    constexpr int A_BYTES = BM * A_ROW;
    constexpr int STAGE = A_BYTES + BN * CH;
    constexpr int PIECES = STAGE / 1024;             // 1 KiB DMA pieces per stage
    constexpr int A_PIECES = A_BYTES / 1024;
    constexpr int PPW = (PIECES + NW - 1) / NW;   // pieces per wave (a wave short of real pieces issues dummy ones)
    constexpr int A_PPW = A_PIECES / NW;
    // LDS stages (three: the 16x16x32 loop of the plain GEMMs, see X6_REGION16 - its implicit-GEMM and two-source forms would spill -, and the
    // f16x2 3x3 form, whose K loop waits for L2: a piece then has two steps to land instead of one, see X6_REGION3)
    constexpr int NSTG = ((M16 && !TAPS && !DUAL) || (H2 && X6_H2_STAGES == 3)) ? 3 : 2;
    constexpr int DUMMY = NSTG * STAGE;           // 1 KiB nobody reads
    static_assert(ROWS % 32 == 0 && TM >= 1 && TN >= 1 && YT % 2 == 0 && A_PIECES % NW == 0, "tile / wave layout mismatch");
    static_assert(!M16 || (TM == 4 && TN == 2 && PPW <= 7), "the 16x16x32 loop is written for 128 x 64 per wave");
    static_assert(!DUAL || !TAPS, "a second activation source goes with 1x1 contractions");
    static_assert(!AF32 || (!M16 && !TAPS && !DUAL), "fp32 activation rows: the plain-GEMM form of the four-wave tiles only");
    static_assert(!H2 || (!M16 && !DUAL), "fp16 planes: the four-wave tiles only (the 256 x 256 tile of f16x2 is gemm_h3)");
    static_assert(!(H2 && AF32) || !B2B, "fp32 rows split into fp16 planes in the loop: the 1x1 form");
    static_assert(!B2B || (H2 && TAPS && (TM == 1 || TM == 2) && TN == 2 && (WN == 1 || (WN == 2 && TM == 2))),
                  "back to back: the f16x2 3x3 form on 4 x 1 waves of 32 or 64 rows x 64 columns, or on 4 x 2 waves of 64 x 64");
    static_assert(B2B != 2 || (WN == 1 && TM == 2), "back to back with a second source: the four-wave form of 64 x 64 per wave");
    static_assert(STAGE % 1024 == 0 && A_BYTES % 1024 == 0, "stage regions are whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // work unit -> (tile, K slice)
    int tile, kt_begin, kt_end, slice = -1, split_tile = 0;
    {
        const int nk_all = p.K >> 4;
        const int b = blockIdx.x;
        if (b < p.full_tiles) {
            tile = xcd_remap6(b, p.full_tiles);
            kt_begin = 0;
            kt_end = nk_all;
        } else {
            const int u = b - p.full_tiles;
            split_tile = u / p.nsplit;
            slice = u - split_tile * p.nsplit;
            tile = p.full_tiles + split_tile;
            kt_begin = (int)((int64_t)nk_all * slice / p.nsplit);
            kt_end = (int)((int64_t)nk_all * (slice + 1) / p.nsplit);
        }
    }
    int tm, tn;
    {
        const int per_group = p.group_m * p.tiles_n;
        const int g = tile / per_group;
        const int first = g * p.group_m;
        const int gsz = p.tiles_m - first < p.group_m ? p.tiles_m - first : p.group_m;
        const int w = tile - g * per_group;
        tm = first + w % gsz;
        tn = w / gsz;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    X6_STAMP(0);
    X6_STAMP_IDS();

    // ---- DMA descriptors.  Piece j of this wave (global piece wave + NW*j) covers LDS bytes [piece*1024, +1024) of a
    // stage; lane l fills unit u = piece*64 + l = (tile row u/6, physical unit u%6); physical unit (plane q, half h')
    // holds logical half h = h' ^ bit3(row).  Pieces j < A_PPW belong to the activation rows, the rest to the weight rows.
    const bool pixels = TAPS || p.stride != 1;     // rows are output pixels of an NHWC image (else: plain matrix rows)
    constexpr int VB = H2 ? 4 : 6;                                    // bytes per value of a row in planes
    const int64_t row_bytes = (int64_t)p.K * VB;                      // weight rows
    const int64_t arow_bytes = (int64_t)(DUAL ? p.K1 : p.K) * (AF32 ? 4 : VB);   // plain activation rows (DUAL: the first source's K1 values)
    const int64_t pix_bytes = (int64_t)p.Cin * VB;
    const int img0 = (pixels || DUAL) ? m0 / (p.Ho * p.Wo) : 0;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w, rsrc_a2;
    {
        const int64_t base = pixels ? (int64_t)img0 * p.H * p.W * pix_bytes : (int64_t)m0 * arow_bytes;
        const int64_t total = pixels ? (int64_t)(p.M / (p.Ho * p.Wo)) * p.H * p.W * pix_bytes : (int64_t)p.M * arow_bytes;
        const int64_t left = total - base;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a + base), 0, (int)(left < kMaxRecords ? left : kMaxRecords),
                                                   0x00020000);
        rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w + (int64_t)n0 * row_bytes), 0, (int)(BN * row_bytes), 0x00020000);
        if (DUAL) {
            const int64_t pix2 = (int64_t)p.Cin2 * 6;
            const int64_t base2 = (int64_t)img0 * p.H2 * p.W2 * pix2;
            const int64_t left2 = (int64_t)(p.M / (p.Ho * p.Wo)) * p.H2 * p.W2 * pix2 - base2;
            rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a2 + base2), 0, (int)(left2 < kMaxRecords ? left2 : kMaxRecords),
                                                        0x00020000);
        }
    }
    unsigned voff[PPW];
    unsigned voff2[DUAL ? A_PPW : 1];   // DUAL: the same rows in the second source (pixels (oy, ox) * stride2 of their image)
    unsigned a_taps[A_PPW > 0 ? A_PPW : 1];   // implicit GEMM: bit t set = tap t of this piece's pixel lies inside the image (KH*KW <= 32)
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = wave + NW * j;
        int trow, unit_off;
        if (H2 && !(AF32 && j < A_PPW)) {
            // fp16 planes: 4 units per row as well, every row of the stage (activation rows, then weight rows); slot s of row r holds
            // the unit v = s ^ ((r >> 2) & 3) = (half v >> 1, plane v & 1) of the chunk, i.e. source bytes plane * 32 + half * 16: a lane's hi
            // and lo fragments sit where AF32 keeps the two fp32 units of its half (the same conflict-free reads)
            const int u = piece * 64 + lane;
            trow = u >> 2;
            const int v = (u & 3) ^ ((trow >> 2) & 3);
            unit_off = (v & 1) * 32 + (v >> 1) * 16;
        } else if (AF32 && j < A_PPW) {
            // fp32 activation rows: 4 units of 16 bytes per row; unit q of row r sits in slot q ^ ((r >> 2) & 3), which makes the
            // two ds_read_b128 of a fragment (lanes = 32 consecutive rows at one q) bank-conflict free
            const int u = piece * 64 + lane;
            trow = u >> 2;
            unit_off = (((u & 3) ^ ((trow >> 2) & 3)) << 4);
        } else {
            const int u = (piece - (AF32 ? A_PIECES : 0)) * 64 + lane;
            trow = u / 6;
            const int c = u - trow * 6;
            unit_off = (c >> 1) * 32 + (((c & 1) ^ (M16 ? 0 : ((trow >> 3) & 1))) << 4);   // (M16: linear image, see below)
            if (AF32) trow += BM;
        }
        if (j < A_PPW) {
            const int m = m0 + trow;
            a_taps[j] = 0u;
            if (m >= p.M) {
                voff[j] = kOutOfRange;
            } else if (pixels) {
                const int img = m / (p.Ho * p.Wo);
                const int rem = m - img * (p.Ho * p.Wo);
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                const int iy = oy * p.stride - p.pad, ix = ox * p.stride - p.pad;
                if (TAPS) {
                    const int ntaps_ = p.K / p.Cin;
                    for (int t = 0, dy = 0, dx = 0; t < ntaps_; ++t) {
                        if ((unsigned)(iy + dy) < (unsigned)p.H && (unsigned)(ix + dx) < (unsigned)p.W) a_taps[j] |= 1u << t;
                        if (++dx == p.KW) { dx = 0; ++dy; }
                    }
                }
                voff[j] = (unsigned)((((img - img0) * p.H + iy) * p.W + ix) * (int)pix_bytes + unit_off);
            } else {
                voff[j] = (unsigned)(trow * (int)arow_bytes + unit_off);
            }
            if (DUAL) {
                if (m >= p.M) {
                    voff2[j] = kOutOfRange;
                } else {
                    const int img = m / (p.Ho * p.Wo);
                    const int rem = m - img * (p.Ho * p.Wo);
                    const int oy = rem / p.Wo;
                    const int ox = rem - oy * p.Wo;
                    voff2[j] = (unsigned)((((img - img0) * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * (p.Cin2 * 6) + unit_off);
                }
            }
        } else {
            voff[j] = piece < PIECES ? (unsigned)((trow - BM) * (int)row_bytes + unit_off) : kOutOfRange;
        }
    }
    // K-step iterator of the DMA issue (steps are issued in order): plain = a byte offset; taps = (channel chunk, dy, dx) with the
    // TAP innermost: the KH*KW steps of one 16-channel chunk read the same 96-byte pieces of neighbouring pixels, so all but the
    // first find them in L2.  (Tap-outermost - the order of the weight rows - streams every pixel's whole channel vector once per
    // tap, KH*KW times from HBM: 10 GB per launch for the 3x3 convolutions of layer1 against 2.5 GB algorithmic.)  The weight
    // rows keep their (tap, channel) order; the step reads chunk tap * cin_chunks + cc of them.
    int d_kt = kt_begin;
    int d_dy = 0, d_dx = 0, d_cc = 0, d_tap = 0;
    const int cin_chunks = p.Cin >> 4;
    const int ntaps = p.K / p.Cin;   // KH * KW
    [[maybe_unused]] const int k1_chunks = DUAL ? p.K1 >> 4 : 0;
    if (TAPS) {
        d_cc = kt_begin / ntaps;
        d_tap = kt_begin - d_cc * ntaps;
        d_dy = d_tap / p.KW;
        d_dx = d_tap - d_dy * p.KW;
    }

#define X6_ISSUE_PIECE(st_, j_)                                                                                         \
    {                                                                                                                   \
        const int lin_ = (TAPS ? d_tap * cin_chunks + d_cc : d_kt) * CH;                                                \
        const int lin_a_ = AF32 ? d_kt * 64 : lin_;                                                                     \
        const int tapoff_ = TAPS ? (d_dy * p.W + d_dx) * (int)pix_bytes + d_cc * CH : 0;                                 \
        const int piece_ = wave + NW * (j_);                                                                            \
        const int dst_ = piece_ < PIECES ? (st_) * STAGE + piece_ * 1024 : DUMMY;                                       \
        if ((j_) < A_PPW) {                                                                                             \
            if (TAPS) {                                                                                                 \
                const int jj_ = (j_) < A_PPW ? (j_) : 0;                                                                \
                const bool ok_ = (a_taps[jj_] >> d_tap) & 1u;                                                           \
                X6_DMA(rsrc_a, dst_, ok_ ? voff[j_] + (unsigned)tapoff_ : kOutOfRange, 0);                              \
            } else if (DUAL && d_kt >= k1_chunks) {   /* workgroup-uniform: the steps past K1 read the second source */    \
                const int jj_ = (j_) < A_PPW ? (j_) : 0;                                                                \
                X6_DMA(rsrc_a2, dst_, voff2[jj_], lin_ - k1_chunks * CH);                                               \
            } else {                                                                                                    \
                X6_DMA(rsrc_a, dst_, voff[j_], lin_a_);                                                                 \
            }                                                                                                           \
        } else {                                                                                                        \
            X6_DMA(rsrc_w, dst_, voff[j_], lin_);                                                                       \
        }                                                                                                               \
    }
#define X6_ISSUE_ADVANCE()                                                                                              \
    {                                                                                                                   \
        ++d_kt;                                                                                                         \
        if (TAPS) {                                                                                                     \
            ++d_tap;                                                                                                    \
            ++d_dx;                                                                                                     \
            const bool wx_ = d_dx == p.KW;                                                                              \
            d_dx = wx_ ? 0 : d_dx;                                                                                      \
            d_dy += wx_ ? 1 : 0;                                                                                        \
            const bool wt_ = d_tap == ntaps;                                                                            \
            d_tap = wt_ ? 0 : d_tap;                                                                                    \
            d_dy = wt_ ? 0 : d_dy;                                                                                      \
            d_cc += wt_ ? 1 : 0;                                                                                        \
        }                                                                                                               \
    }
#define X6_ISSUE(st_)                                                                                                   \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) X6_ISSUE_PIECE(st_, j);                                         \
        X6_ISSUE_ADVANCE();                                                                                             \
    }

    floatx16 acc[TM][TN];
    // H2: the two small products (al bh, ah bl: 2^-11 of the sum) have an accumulator of their own, added once at the end - the main one is
    // rounded once per 16-deep step instead of three times (its rounding is what the result's error consists of: measured against fp64,
    // one accumulator 1.25 x the exact-fp32 kernel's error on these 3x3 geometries, two 0.8 x)
    floatx16 accs[H2 ? TM : 1][H2 ? TN : 1];
#pragma unroll
    for (int i = 0; i < (H2 ? TM : 1); ++i)
#pragma unroll
        for (int j = 0; j < (H2 ? TN : 1); ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[i][j][r] = 0.f;
    floatx4 acc16[M16 ? 8 : 1][4];   // M16: 16-row A fragments x 16-column B fragments of the wave's 128 x 64
#pragma unroll
    for (int i = 0; i < (M16 ? 8 : 1); ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: lane (r, h) reads the unit of half h of row r; rows with bit 3 set hold the halves swapped
    const int r32 = lane & 31;
    // (H2: the hi fragment of lane (r, h) = slot (2h) ^ ((r >> 2) & 3) of its 64-byte row, the lo fragment = slot (2h + 1) ^ ...: FLIP apart)
    const int frag = H2 ? r32 * CH + ((((lane >> 5) * 2) ^ ((r32 >> 2) & 3)) << 4) : r32 * CH + ((((lane >> 5) ^ (r32 >> 3)) & 1) << 4);
    const int a_off = (wm * TM * 32) * CH + frag;
    const int b_off = A_BYTES + (wn * TN * 32) * CH + frag;
    // AF32: lane (r, h) reads the fp32 values k = 8h .. 8h+7 of row r as two 16-byte units and splits them into the three planes
    [[maybe_unused]] const int a32_0 = (wm * TM * 32 + r32) * 64 + ((((lane >> 5) * 2) ^ ((r32 >> 2) & 3)) << 4);
    [[maybe_unused]] const int a32_1 = (wm * TM * 32 + r32) * 64 + ((((lane >> 5) * 2 + 1) ^ ((r32 >> 2) & 3)) << 4);
    // H2 + AF32: the scale of the image each of this lane's activation rows belongs to (row (wm*TM + i)*32 + r32 of the tile)
    [[maybe_unused]] float asc[TM];
    if constexpr (H2 && AF32) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + r32;
            asc[i] = p.img_in_scale[(m < p.M ? m : p.M - 1) / (p.Ho * p.Wo)];
        }
    }
    const int x_off = SPLIT_B ? a_off : b_off;
    const int y_off = SPLIT_B ? b_off : a_off;

    bf16x8 xf[2][XT][NP], yf0[YH][NP], yf1[YH][NP];   // [.][fragment][plane hi, mid, lo] (H2: the bits of 8 fp16, planes hi, lo)
    // H2: the lo fragment sits in the neighbouring slot of the swizzled row: XOR 16 on the byte offset (slots 2h and 2h + 1 differ in bit 0,
    // the row's swizzle is the same for both)
#define X6_PLANE(off_, pl_) (H2 ? ((off_) ^ ((pl_) << 4)) : ((off_) + (pl_) * 32))
#define X6_READ_A32(dst_, idx_, sp_)                                                                                    \
    {                                                                                                                   \
        const f32x4 v0_ = *reinterpret_cast<const f32x4*>((sp_) + a32_0 + (idx_) * 32 * 64);                            \
        const f32x4 v1_ = *reinterpret_cast<const f32x4*>((sp_) + a32_1 + (idx_) * 32 * 64);                            \
        if constexpr (H2) {   /* two fp16 planes of x * (the image's scale): 22 bits, csrc/h2.h */                      \
            u32x4 h_, l_;                                                                                               \
            split2_x8(v0_ * asc[(idx_) < TM ? (idx_) : 0], v1_ * asc[(idx_) < TM ? (idx_) : 0], h_, l_);                \
            dst_[0] = __builtin_bit_cast(bf16x8, h_);                                                                   \
            dst_[1] = __builtin_bit_cast(bf16x8, l_);                                                                   \
        } else {                                                                                                        \
            u32x4 h_, m_, l_;                                                                                           \
            split3_x8(v0_, v1_, h_, m_, l_);                                                                            \
            dst_[0] = __builtin_bit_cast(bf16x8, h_);                                                                   \
            dst_[1] = __builtin_bit_cast(bf16x8, m_);                                                                   \
            dst_[NP - 1] = __builtin_bit_cast(bf16x8, l_);                                                              \
        }                                                                                                               \
    }
#define X6_READ_X(set_, sp_)                                                                                            \
    _Pragma("unroll") for (int x = 0; x < XT; ++x) {                                                                    \
        if constexpr (AF32 && SPLIT_B) {                                                                                \
            X6_READ_A32(xf[set_][x], x, sp_);                                                                           \
        } else {                                                                                                        \
            _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                                           \
                xf[set_][x][pl] = *reinterpret_cast<const bf16x8*>((sp_) + X6_PLANE(x_off, pl) + x * 32 * CH);          \
        }                                                                                                               \
    }
#define X6_READ_Y(yf_, half_, sp_)                                                                                      \
    _Pragma("unroll") for (int y = 0; y < YH; ++y) {                                                                    \
        if constexpr (AF32 && !SPLIT_B) {                                                                               \
            X6_READ_A32(yf_[y], (half_) * YH + y, sp_);                                                                 \
        } else {                                                                                                        \
            _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                                           \
                yf_[y][pl] = *reinterpret_cast<const bf16x8*>((sp_) + X6_PLANE(y_off, pl) + ((half_) * YH + y) * 32 * CH);   \
        }                                                                                                               \
    }
    // the six partial products, smallest first: (A plane, B plane) = (lo,hi) (mid,mid) (hi,lo) (mid,hi) (hi,mid) (hi,hi);
    // product type outermost: XT*YH independent accumulators between two MFMAs on the same one
    // H2: (lo,hi) (hi,lo) (hi,hi) on the fp16 instruction
#define X6_MFMA1(a_, b_, c_)                                                                                            \
    (H2 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0) \
        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0))
#define X6_MFMAS(set_, yf_, half_)                                                                                      \
    _Pragma("unroll") for (int t = 0; t < (H2 ? 3 : 6); ++t) {                                                          \
        const int pa = H2 ? (t == 0 ? 1 : 0) : t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0;                                 \
        const int pb = H2 ? (t == 1 ? 1 : 0) : t == 2 ? 2 : (t == 1 || t == 4) ? 1 : 0;                                 \
        _Pragma("unroll") for (int x = 0; x < XT; ++x) _Pragma("unroll") for (int y = 0; y < YH; ++y) {                 \
            if (B2B && t < 2) { /* the TRANSPOSED tile (weights as the A operand): channels along the accumulator registers */ \
                accs[H2 ? x : 0][H2 ? (half_) * YH + y : 0] =                                                           \
                    X6_MFMA1(yf_[y][pb], xf[set_][x][pa], accs[H2 ? x : 0][H2 ? (half_) * YH + y : 0]);                 \
            } else if (B2B) {                                                                                           \
                acc[x][(half_) * YH + y] = X6_MFMA1(yf_[y][pb], xf[set_][x][pa], acc[x][(half_) * YH + y]);             \
            } else if (H2 && t < 2) {                                                                                   \
                if (SPLIT_B)                                                                                            \
                    accs[H2 ? x : 0][H2 ? (half_) * YH + y : 0] =                                                       \
                        X6_MFMA1(xf[set_][x][pa], yf_[y][pb], accs[H2 ? x : 0][H2 ? (half_) * YH + y : 0]);             \
                else                                                                                                    \
                    accs[H2 ? (half_) * YH + y : 0][H2 ? x : 0] =                                                       \
                        X6_MFMA1(yf_[y][pa], xf[set_][x][pb], accs[H2 ? (half_) * YH + y : 0][H2 ? x : 0]);             \
            } else if (SPLIT_B)                                                                                         \
                acc[x][(half_) * YH + y] = X6_MFMA1(xf[set_][x][pa], yf_[y][pb], acc[x][(half_) * YH + y]);             \
            else                                                                                                        \
                acc[(half_) * YH + y][x] = X6_MFMA1(yf_[y][pa], xf[set_][x][pb], acc[(half_) * YH + y][x]);             \
        }                                                                                                               \
    }
    // One pipeline region = everything between two barriers: [wait, barrier, DMA of step k+2, reads of X and Y-half-0 of step
    // k+1, M1(k), reads of Y-half-1 of step k+1, M0(k+1)].  xs_ = parity (= LDS stage) of step k.  has_next_ / has_d_ are
    // literal `true` in the steady state (branch-free, one basic block per region: the compiler's own interleaving of the
    // block measured faster than any order pinned with sched_group_barrier).
#define X6_REGION(xs_, has_next_, has_d_)                                                                               \
    {                                                                                                                   \
        const char* sn_ = smem + ((xs_) ^ 1) * STAGE;                                                                   \
        if (has_next_) {                                                                                                \
            /* own DMA pieces of step k+1 have landed, own fragment reads are done */                                   \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
            __builtin_amdgcn_s_barrier();                                                                               \
            /* every wave has stage k in registers: it is free for step k + 2 */                                        \
            if (has_d_) X6_ISSUE(xs_);                                                                                  \
            X6_READ_X((xs_) ^ 1, sn_);                                                                                  \
            X6_READ_Y(yf0, 0, sn_);                                                                                     \
        }                                                                                                               \
        X6_MFMAS(xs_, yf1, 1);                                                                                          \
        if (has_next_) {                                                                                                \
            X6_READ_Y(yf1, 1, sn_);                                                                                     \
            X6_MFMAS((xs_) ^ 1, yf0, 0);                                                                                \
        }                                                                                                               \
    }

    // The same region with THREE stages (H2): step k + 3 is requested in region k into the stage step k just left, and region k waits only for
    // the pieces of step k + 1 - those of step k + 2, this wave's newest PPW, stay in flight (vmcnt counts in issue order).  st_ = k % 3.
#define X6_REGION3(xs_, st_, has_next_, has_n2_, has_d_)                                                                \
    {                                                                                                                   \
        const char* sn_ = smem + (((st_) + 1) % 3) * STAGE;                                                             \
        if (has_next_) {                                                                                                \
            if (has_n2_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");                          \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
            __builtin_amdgcn_s_barrier();                                                                               \
            if (has_d_) X6_ISSUE(st_);                                                                                  \
            X6_READ_X((xs_) ^ 1, sn_);                                                                                  \
            X6_READ_Y(yf0, 0, sn_);                                                                                     \
        }                                                                                                               \
        X6_MFMAS(xs_, yf1, 1);                                                                                          \
        if (has_next_) {                                                                                                \
            X6_READ_Y(yf1, 1, sn_);                                                                                     \
            X6_MFMAS((xs_) ^ 1, yf0, 0);                                                                                \
        }                                                                                                               \
    }

    // ---- the same loop on v_mfma_f32_16x16x32_bf16 (M16).  The 32-deep K of that instruction takes TWO planes of a 16-deep
    // chunk: lanes with k group g = lane >> 4 < 2 read the halves of the first plane of a pair, g >= 2 those of the second, so
    //     A[lo|hi] B[hi|lo] = al bh + ah bl,   A[mid|hi] B[hi|mid] = am bh + ah bm,   A[hi|mid] B[hi|mid] = ah bh + am bm
    // are the six partial products in three instructions per 16 x 16 outputs (same FLOPs, same order smallest first; measured
    // error against fp64 equal to the 32x32x16 form: tools/micro/mfma_shape.hip).  The chip holds a higher clock on this shape
    // on real data: the K loop takes 3.8 - 4.3 k cycles per step here against 3.55 k on the 32x32x16 loop and still finishes a
    // ViT pass 2.7 % sooner (profiles/r03_mfma_shape.txt, LAB_NOTES.md section 3.2).  B (2 forms x 4 fragments) is double-buffered in registers across steps, A (3 forms) streams one 16-row
    // fragment at a time, two buffers.  The LDS image is linear here (no half swap: these reads are conflict-free on it).
    const int r16 = lane & 15, g16 = lane >> 4, second16 = g16 >> 1;
    const int rowoff16 = r16 * kChunkBytes + ((g16 & 1) << 4);
    const int a16 = (wm * TM * 32) * kChunkBytes + rowoff16;
    const int b16 = (BM + wn * TN * 32) * kChunkBytes + rowoff16;
    const int a_hm = a16 + (second16 ? 32 : 0), a_mh = a16 + (second16 ? 0 : 32), a_lh = a16 + (second16 ? 0 : 64);
    const int b_hm = b16 + (second16 ? 32 : 0), b_hl = b16 + (second16 ? 64 : 0);
    bf16x8 xb[2][4][2], ya[2][3];
#define X6_READ_XB(set_, sp_)                                                                                           \
    _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                                                     \
        xb[set_][y][0] = *reinterpret_cast<const bf16x8*>((sp_) + b_hm + y * 16 * kChunkBytes);                         \
        xb[set_][y][1] = *reinterpret_cast<const bf16x8*>((sp_) + b_hl + y * 16 * kChunkBytes);                         \
    }
#define X6_READ_A16(buf_, e_, sp_)                                                                                      \
    {                                                                                                                   \
        ya[buf_][0] = *reinterpret_cast<const bf16x8*>((sp_) + a_hm + (e_) * 16 * kChunkBytes);                         \
        ya[buf_][1] = *reinterpret_cast<const bf16x8*>((sp_) + a_mh + (e_) * 16 * kChunkBytes);                         \
        ya[buf_][2] = *reinterpret_cast<const bf16x8*>((sp_) + a_lh + (e_) * 16 * kChunkBytes);                         \
        __builtin_amdgcn_sched_barrier(0);   /* issued HERE, a whole MFMA group ahead of their use */                  \
    }
#define X6_MFMAS16(set_, buf_, e_)                                                                                      \
    {                                                                                                                   \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc16[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[buf_][2], xb[set_][y][1], acc16[e_][y], 0, 0, 0); \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc16[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[buf_][1], xb[set_][y][0], acc16[e_][y], 0, 0, 0); \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc16[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[buf_][0], xb[set_][y][0], acc16[e_][y], 0, 0, 0); \
        /* nothing moves across: left alone, the scheduler regroups the loop by form (one read, four MFMAs, wait, ...) to save */ \
        /* registers and exposes the LDS latency every 64 cycles */                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }
    // region k: [wait, barrier, B forms and A fragment 0 of step k+1, fragment 7 of step k, then fragments 0..6 of step k+1 each
    // behind the read of the next one]; xs_ = k & 1 (register set of B), st_ = k % NSTG (LDS stage of step k).
    // Where the DMA goes (cycles per K step from the stamped build, ViT-B GEMMs, 3072 = the matrix pipes' own time):
    //   two stages, all six pieces right behind the barrier (the 32x32x16 loop's place) ....... both waves of a SIMD stall in issue
    //   two stages, one piece per MFMA group ............ 3.78 - 4.36 k (the last pieces have a third of a step to land)
    //   two stages, three pieces in each of the first two groups ............ 3.71 - 3.90 k
    //   THREE stages (DMA two steps ahead, counted vmcnt), three + three ..... 3.71 - 3.90 k: no change - latency is not the cost
    //   three stages, one piece per MFMA group ........... 3.43 - 3.48 k = the loop with every DMA hitting L2 (3.43 k; no DMA: 3.32 k)
    // PMC: with real traffic the TA address FIFO is full 3 x as long as VMEM instructions are active (SQ_VMEM_TA_ADDR_FIFO_FULL): a
    // burst of pieces blocks the issuing wave - and its MFMAs, issue is in order - until the texture path has taken them, so the pieces
    // are issued one at a time, a MFMA group apart, and the third stage gives the late ones a whole extra step to land.
#define X6_REGION16(xs_, st_, has_next_, has_d_)                                                                        \
    {                                                                                                                   \
        const char* sn_ = smem + (((st_) + 1) % NSTG) * STAGE;                                                          \
        if (has_next_) {                                                                                                \
            /* own pieces of step k+1 have landed (three stages: those of step k+2, the newest PPW, stay in flight) */   \
            if (NSTG == 3 && (has_d_)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");            \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
            __builtin_amdgcn_s_barrier();                                                                               \
            X6_READ_XB((xs_) ^ 1, sn_);                                                                                 \
            X6_READ_A16(0, 0, sn_);                                                                                     \
            if (NSTG == 2 && (has_d_)) { _Pragma("unroll") for (int j = 0; j < (PPW + 1) / 2; ++j) X6_ISSUE_PIECE(st_, j); } \
        }                                                                                                               \
        X6_MFMAS16(xs_, 1, 7);                                                                                          \
        if (has_next_) {                                                                                                \
            _Pragma("unroll") for (int e = 0; e < 7; ++e) {                                                             \
                X6_READ_A16((e + 1) & 1, e + 1, sn_);                                                                   \
                if (NSTG == 2 && (has_d_) && e == 0) { _Pragma("unroll") for (int j = (PPW + 1) / 2; j < PPW; ++j) X6_ISSUE_PIECE(st_, j); } \
                if (NSTG == 3 && (has_d_) && e < PPW) X6_ISSUE_PIECE(st_, e);   /* the stage step k just left */             \
                X6_MFMAS16((xs_) ^ 1, e & 1, e);                                                                        \
            }                                                                                                           \
            if (has_d_) X6_ISSUE_ADVANCE();                                                                             \
        }                                                                                                               \
    }

    const int nk = kt_end - kt_begin;
    X6_ISSUE(0);
    if (nk > 1) X6_ISSUE(1);
    if (NSTG == 3 && nk > 2) X6_ISSUE(2);
    // step 0 has landed; the younger steps stay in flight (vmcnt counts this wave's pieces in issue order)
    if (NSTG == 3 && nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (M16) {
        X6_READ_XB(0, smem);
        X6_READ_A16(0, 0, smem);
        X6_STAMP(1);
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            X6_READ_A16((e + 1) & 1, e + 1, smem);
            X6_MFMAS16(0, e & 1, e);
        }
        // has_next = step k+1 exists; has_d = step k + NSTG exists (its DMA is issued in region k); literal `true` in the steady state
        int k = 0;
        if constexpr (NSTG == 3) {   // six regions = one period of (register set, stage)
            for (; k + 6 + 3 <= nk; k += 6) {
                X6_REGION16(0, 0, true, true);
                X6_REGION16(1, 1, true, true);
                X6_REGION16(0, 2, true, true);
                X6_REGION16(1, 0, true, true);
                X6_REGION16(0, 1, true, true);
                X6_REGION16(1, 2, true, true);
            }
            for (; k < nk; k += 6) {
                X6_REGION16(0, 0, k + 1 < nk, k + 3 < nk);
                if (k + 1 < nk) X6_REGION16(1, 1, k + 2 < nk, k + 4 < nk);
                if (k + 2 < nk) X6_REGION16(0, 2, k + 3 < nk, k + 5 < nk);
                if (k + 3 < nk) X6_REGION16(1, 0, k + 4 < nk, k + 6 < nk);
                if (k + 4 < nk) X6_REGION16(0, 1, k + 5 < nk, k + 7 < nk);
                if (k + 5 < nk) X6_REGION16(1, 2, k + 6 < nk, k + 8 < nk);
            }
        } else {
            for (; k + 3 < nk; k += 2) {
                X6_REGION16(0, 0, true, true);
                X6_REGION16(1, 1, true, true);
            }
            for (; k < nk; k += 2) {
                X6_REGION16(0, 0, k + 1 < nk, k + 2 < nk);
                if (k + 1 < nk) X6_REGION16(1, 1, k + 2 < nk, k + 3 < nk);
            }
        }
    } else {
    X6_READ_X(0, smem);
    X6_READ_Y(yf0, 0, smem);
    X6_STAMP(1);
    X6_STAMP(6);
    X6_READ_Y(yf1, 1, smem);
    X6_MFMAS(0, yf0, 0);          // M0 of step 0
    int k = 0;
    if constexpr (NSTG == 3) {   // six regions = one period of (register set, stage)
        for (; k + 6 + 3 <= nk; k += 6) {
            X6_REGION3(0, 0, true, true, true);
            X6_REGION3(1, 1, true, true, true);
            X6_REGION3(0, 2, true, true, true);
            X6_REGION3(1, 0, true, true, true);
            X6_REGION3(0, 1, true, true, true);
            X6_REGION3(1, 2, true, true, true);
        }
        for (; k < nk; k += 6) {
            X6_REGION3(0, 0, k + 1 < nk, k + 2 < nk, k + 3 < nk);
            if (k + 1 < nk) X6_REGION3(1, 1, k + 2 < nk, k + 3 < nk, k + 4 < nk);
            if (k + 2 < nk) X6_REGION3(0, 2, k + 3 < nk, k + 4 < nk, k + 5 < nk);
            if (k + 3 < nk) X6_REGION3(1, 0, k + 4 < nk, k + 5 < nk, k + 6 < nk);
            if (k + 4 < nk) X6_REGION3(0, 1, k + 5 < nk, k + 6 < nk, k + 7 < nk);
            if (k + 5 < nk) X6_REGION3(1, 2, k + 6 < nk, k + 7 < nk, k + 8 < nk);
        }
    } else {
    // region k: has_next = step k+1 exists; has_d = step k+2 exists (its DMA is issued here)
    for (; k + 3 < nk; k += 2) {
        X6_REGION(0, true, true);
        X6_REGION(1, true, true);
    }
    for (; k < nk; k += 2) {
        X6_REGION(0, k + 1 < nk, k + 2 < nk);
        if (k + 1 < nk) X6_REGION(1, k + 2 < nk, k + 3 < nk);
    }
    }
    }
#undef X6_REGION3
#undef X6_REGION16
#undef X6_MFMAS16
#undef X6_READ_A16
#undef X6_READ_XB
#undef X6_REGION
#undef X6_MFMAS
#undef X6_MFMA1
#undef X6_PLANE
#undef X6_READ_X
#undef X6_READ_A32
#undef X6_READ_Y
#undef X6_ISSUE
#undef X6_ISSUE_PIECE
#undef X6_ISSUE_ADVANCE
    X6_STAMP(2);
    if (H2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += accs[H2 ? i : 0][H2 ? j : 0];
    }
    __syncthreads();   // no DMA is in flight and every fragment is in registers: the stages become the epilogue staging

    if constexpr (B2B) {
        // ---- back to back: acc[x][y] holds the TRANSPOSED tile of the 3x3 convolution - lane (c, half) has pixel  (wm*TM + x)*32 + c  of the
        // tile and the 16 channels  y*32 + 4*half + (r & 3) + 8*(r >> 2), r = 0 .. 15.  That is the A-operand layout of v_mfma_f32_32x32x16_f16
        // for a contraction over channels in the order  k = 16q + 8h + e  <->  channel 16q + 4h + (e & 3) + 8 (e >> 2)  (the order w3 is packed
        // in): bias, ReLU, ONE scale per pixel row from the row's own maximum (measured, batch-invariant: csrc/h2.h "operator level"), the split
        // into two fp16 planes - all in registers -, and the tile is the A operand of the block's conv3 without leaving the CU.
        const int c32 = lane & 31, half = lane >> 5;
        constexpr int B2B_STG = NW * TM * 32 * 68 * 4 > 2 * STAGE ? NW * TM * 32 * 68 * 4 : 2 * STAGE;   // the waves' staging rows (TM * 32 x 68 floats each), over the two stages
        // WN == 2 (128-wide blocks: layer2): a wave holds 64 of the 128 channels of its pixels.  conv3's K is split over the two waves of a
        // pair (wm, 0), (wm, 1): each contracts ITS 64 channels - with its own scale per pixel row, from the maximum over its channels - over all
        // N3 columns, both stage their partial tiles, and after a barrier each wave finishes 32 of the pair's 64 rows:
        // (partial0 / scale0 + partial1 / scale1) * colscale + bias + residual, a fixed order.
        float* rowinv = reinterpret_cast<float*>(smem + B2B_STG);              // [WN][BM]: 1 / scale of every pixel row of the tile (per K half)
        unsigned* simg = reinterpret_cast<unsigned*>(smem + B2B_STG + WN * BM * 4);   // two per-image maxima of the tile
        if (tid < 2) simg[tid] = 0u;
        // B2B == 2 (the first block of layer1: conv3 + the downsample convolution in ONE contraction, K = [the 3x3's 64 channels | the block input's 64]):
        // the lane also loads ITS pixel's row of the block input (fp32, k = 16q + 8 half + e: the natural order the downsample weights keep), the
        // row maximum runs over both halves of K, and ONE scale per pixel row covers both - a single accumulator, no two-launch round trip
        constexpr int KQ = 2 * TN + (B2B == 2 ? 4 : 0);      // 16-deep chunks of conv3's K
        u32x4 ah[TM][KQ], al[TM][KQ];
#pragma unroll
        for (int x = 0; x < TM; ++x) {
            const int m = m0 + (wm * TM + x) * 32 + c32;
            const float rs = p.img_in_inv[(m < p.M ? m : p.M - 1) / (p.Ho * p.Wo)];
            float mx = 0.f;
#pragma unroll
            for (int y = 0; y < TN; ++y)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch0 = n0 + (wn * TN + y) * 32 + 4 * half + 8 * g;
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(p.colscale + ch0);
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + ch0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaxf(acc[x][y][4 * g + e] * (cs[e] * rs) + bb[e], 0.f);   // (cs * rs: powers of two, exact)
                        acc[x][y][4 * g + e] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            [[maybe_unused]] f32x4 xa[4], xb[4];
            if constexpr (B2B == 2) {
                const float* xr = p.x2 + (int64_t)(m < p.M ? m : p.M - 1) * 64 + 8 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xa[q] = *reinterpret_cast<const f32x4*>(xr + 16 * q);
                    xb[q] = *reinterpret_cast<const f32x4*>(xr + 16 * q + 4);
                    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(xa[q].x), fabsf(xa[q].y)), fmaxf(fabsf(xa[q].z), fabsf(xa[q].w))),
                                         fmaxf(fmaxf(fabsf(xb[q].x), fabsf(xb[q].y)), fmaxf(fabsf(xb[q].z), fabsf(xb[q].w)))));
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float sc = h2_scale_for(mx);
            if (half == 0) rowinv[wn * BM + (wm * TM + x) * 32 + c32] = 1.f / sc;       // (a power of two: exact)
            if constexpr (B2B == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) split2_x8(xa[q] * sc, xb[q] * sc, ah[x][2 * TN + q], al[x][2 * TN + q]);
            }
#pragma unroll
            for (int y = 0; y < TN; ++y)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const f32x4 lo4 = {acc[x][y][8 * sub + 0], acc[x][y][8 * sub + 1], acc[x][y][8 * sub + 2], acc[x][y][8 * sub + 3]};
                    const f32x4 hi4 = {acc[x][y][8 * sub + 4], acc[x][y][8 * sub + 5], acc[x][y][8 * sub + 6], acc[x][y][8 * sub + 7]};
                    split2_x8(lo4 * sc, hi4 * sc, ah[x][2 * y + sub], al[x][2 * y + sub]);
                }
        }
        // ---- conv3: [64 pixels of this wave] x [K = BN channels] x [N3 columns] in passes of 64 columns, B fragments straight from
        // global memory (64 KB shared by every workgroup of the launch: L2 / L1 hits); four products (K < 256), one accumulator, smallest first
        constexpr int LDC2 = 68;
        float* stgw = reinterpret_cast<float*>(smem) + wave * (TM * 32 * LDC2);    // this wave's own 64 staging rows (WN == 1: no workgroup barrier in the passes)
        [[maybe_unused]] const float* stg0 = reinterpret_cast<const float*>(smem) + (wm * WN) * (TM * 32 * LDC2);       // WN == 2: the pair's two partial tiles
        [[maybe_unused]] const float* stg1 = reinterpret_cast<const float*>(smem) + (wm * WN + 1) * (TM * 32 * LDC2);
        float* stgf = WN == 1 ? stgw : const_cast<float*>(stg0);                   // where finished rows go back for the group sums
        const int lr = lane >> 3, lc = (lane & 7) * 8;
        const int img_first = m0 / (p.Ho * p.Wo);
        const int r0 = wm * TM * 32;                                               // the pair's (wave's) first row of the tile
        constexpr int RW = TM * 32 / WN;                                           // rows this wave finishes: 64, or 32 of the pair's 64
        const int rw0 = wn * RW;                                                   // ... starting at this row of the 64
        float tmax0 = 0.f, tmax1 = 0.f;
        const int64_t w3row = (int64_t)(BN + (B2B == 2 ? 64 : 0)) * 4;       // conv3's K values x 4 bytes of planes
        // The memory operations of a wave complete in issue order (vmcnt), so the order below is the pipeline: the B fragments of pass p + 1
        // are requested BEFORE the stores of pass p (a wait for them must not wait for those stores' acknowledgements); the residual rows
        // come 32 at a time, requested while the accumulators are on their way through the LDS.  Residual, fp32 output and weights go
        // through buffer resources: ONE offset register per lane for the whole tile (row lr, columns lc .. lc + 7 of the wave's rows), the
        // pass / row-group part in a scalar; rows beyond M (beyond out_rows) read zeros (are not stored) by the resource's byte count.
        const int npass = p.N3 / 64;
        const int64_t wave_row0 = (int64_t)(m0 + r0 + rw0) * p.N3 * 4;           // byte offset of the wave's first row in an [M][N3] fp32 matrix
        auto bytes_left = [&](int rows) {
            const int64_t left = (int64_t)rows * p.N3 * 4 - wave_row0;
            return (int)(left < 0 ? 0 : (left < kMaxRecords ? left : kMaxRecords));
        };
        // (no residual - a block with a downsample branch -: a resource of zero bytes, every load answers zeros)
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.residual)) + (p.residual ? wave_row0 : 0), 0,
                                                                                p.residual ? bytes_left(p.M) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (p.out ? wave_row0 : 0), 0,
                                                                                p.out ? bytes_left(p.out_rows < p.M ? p.out_rows : p.M) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w3), 0, (int)((int64_t)p.N3 * w3row), 0x00020000);
        const int voff_row = (lr * p.N3 + lc) * 4;
        const int voff_w3 = c32 * (int)w3row + half * 16;
        f16x8 bh[X6_B2B_PREFETCH_B ? KQ : 1][2], bl[X6_B2B_PREFETCH_B ? KQ : 1][2];
        constexpr int QPRE = X6_B2B_PREFETCH_B == 2 ? KQ / 2 : KQ;   // chunks of the next pass requested ahead of this pass's stores (mode 2: the first half)
        // residual rows of a whole pass (RW / 8 lane-rows: 8 register sets on the 64-wide tiles): ALL of them are requested before the pass's
        // first store.  With four sets the second half was requested behind the stores of the first, and a wait for those loads is a wait
        // for every older operation of the wave - the write acknowledgements of the pass's first 32 rows (X6_B2B_RSETS=4: the old form)
        constexpr int RSETS = (X6_B2B_RSETS) ? (X6_B2B_RSETS) : RW / 8;
        [[maybe_unused]] u32x4 ra[RSETS], rb[RSETS];
        constexpr bool RES_EARLY = (X6_B2B_RES_EARLY) && WN == 2;   // (the 64-wide tiles have no registers for it: 120 bytes of scratch per lane)
#define X6_B2B_LOAD_BQ(pass_, q_, slot_)                                                                                \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                                  \
        const int so_ = ((pass_) * 64 + nt * 32) * (int)w3row + (wn * KQ + (q_)) * 64;   /* (this wave's K half) */        \
        bh[slot_][nt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, voff_w3, so_, 0));       \
        bl[slot_][nt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, voff_w3 + 32, so_, 0));  \
    }
#define X6_B2B_LOAD_B(pass_) _Pragma("unroll") for (int q = 0; q < KQ; ++q) X6_B2B_LOAD_BQ(pass_, q, q)
#define X6_B2B_LOAD_RES(pass_, it0_)                                                                                    \
    if constexpr (B2B != 2) _Pragma("unroll") for (int it = (it0_); it < (it0_) + 4; ++it) {   /* (B2B == 2: no residual) */ \
        const int so_ = (it * 8 * p.N3 + (pass_) * 64) * 4;                                                             \
        if (X6_B2B_ABL & 2) { ra[it % RSETS] = (u32x4){0u, 0u, 0u, 0u}; rb[it % RSETS] = ra[it % RSETS]; continue; }                \
        ra[it % RSETS] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff_row, so_, 0);                                   \
        rb[it % RSETS] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff_row + 16, so_, 0);                              \
    }
#define X6_B2B_ROWS(it0_)                                                                                               \
    _Pragma("unroll") for (int it = (it0_); it < (it0_) + 4; ++it) {                                                    \
        const int row = rw0 + it * 8 + lr;                                                                              \
        const int m = m0 + r0 + row;                                                                                    \
        f32x4 va, vb;                                                                                                   \
        if constexpr (WN == 1) {                                                                                        \
            va = *reinterpret_cast<const f32x4*>(stgw + row * LDC2 + lc);                                               \
            vb = *reinterpret_cast<const f32x4*>(stgw + row * LDC2 + lc + 4);                                           \
            const float ri = rowinv[r0 + row];                                                                          \
            va = va * (cs_a * ri) + b3_a;         /* (powers of two: exact) */                                          \
            vb = vb * (cs_b * ri) + b3_b;                                                                               \
        } else {                                                                                                        \
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg0 + row * LDC2 + lc), b0 = *reinterpret_cast<const f32x4*>(stg0 + row * LDC2 + lc + 4); \
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg1 + row * LDC2 + lc), b1 = *reinterpret_cast<const f32x4*>(stg1 + row * LDC2 + lc + 4); \
            const float ri0 = rowinv[r0 + row], ri1 = rowinv[BM + r0 + row];                                            \
            va = (a0 * (cs_a * ri0) + a1 * (cs_a * ri1)) + b3_a;   /* the two K halves in a fixed order */              \
            vb = (b0 * (cs_b * ri0) + b1 * (cs_b * ri1)) + b3_b;                                                        \
        }                                                                                                               \
        if constexpr (B2B != 2) {                                                                                       \
            va += __builtin_bit_cast(f32x4, ra[it % RSETS]);   /* (acc + bias) + residual: the order of the two-launch path */ \
            vb += __builtin_bit_cast(f32x4, rb[it % RSETS]);                                                                \
        }                                                                                                               \
        va = (f32x4){fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f)};                           \
        vb = (f32x4){fmaxf(vb.x, 0.f), fmaxf(vb.y, 0.f), fmaxf(vb.z, 0.f), fmaxf(vb.w, 0.f)};                           \
        if (p.gap) {                                                                                                    \
            *reinterpret_cast<f32x4*>(stgf + row * LDC2 + lc) = va;                                                     \
            *reinterpret_cast<f32x4*>(stgf + row * LDC2 + lc + 4) = vb;                                                 \
        }                                                                                                               \
        if (!(X6_B2B_ABL & 1)) {                                                                                        \
            const int so_ = (it * 8 * p.N3 + n0p) * 4;                                                                  \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, va), rs_out, voff_row, so_, 0);            \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vb), rs_out, voff_row + 16, so_, 0);       \
        }                                                                                                               \
        if (m >= p.M) continue;                                                                                         \
        if (p.out_sp3 || p.out_h2 || p.amax_out) {                                                                      \
            const int img = m / (p.Ho * p.Wo);                                                                          \
            int64_t ms_ = m;                                                                                            \
            bool keep_ = true;                                                                                          \
            if (p.sp3_sub == 2) {   /* the planes hold the stride-2 sample of the map only, compacted */                \
                const int rem_ = m - img * (p.Ho * p.Wo);                                                               \
                const int oy_ = rem_ / p.Wo, ox_ = rem_ - oy_ * p.Wo;                                                   \
                keep_ = ((oy_ | ox_) & 1) == 0;                                                                         \
                ms_ = ((int64_t)img * (p.Ho >> 1) + (oy_ >> 1)) * (p.Wo >> 1) + (ox_ >> 1);                             \
            }                                                                                                           \
            if (p.out_sp3 && keep_) store_sp3_x8(p.out_sp3 + ms_ * ((int64_t)p.N3 * 6), n0p + lc, va, vb);              \
            if (p.out_h2 && keep_) store_h2_x8(p.out_h2 + ms_ * ((int64_t)p.N3 * 4), n0p + lc, va, vb, p.img_out_scale[img]); \
            const float mx = fmaxf(fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w)), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w))); \
            if (img == img_first) tmax0 = fmaxf(tmax0, mx);      /* (a tile of 256 rows spans at most two images of >= 256 rows) */ \
            else tmax1 = fmaxf(tmax1, mx);                                                                              \
        }                                                                                                               \
    }
        if (X6_B2B_ABL & 4) return;
        if (X6_B2B_PREFETCH_B) { _Pragma("unroll") for (int q = 0; q < QPRE; ++q) X6_B2B_LOAD_BQ(0, q, q); }
        for (int pass = 0; pass < npass; ++pass) {
            const int n0p = pass * 64;
            if constexpr (RES_EARLY) {   // the pass's residual rows requested ahead of its MFMA phase: they land under it
                X6_B2B_LOAD_RES(pass, 0);
                if constexpr (RW == 64 && RSETS == 8) X6_B2B_LOAD_RES(pass, 4);
            }
            {
                floatx16 c2[TM][2];
#pragma unroll
                for (int x = 0; x < TM; ++x)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) c2[x][nt][r] = 0.f;
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    constexpr int slot = X6_B2B_PREFETCH_B ? -1 : 0;
                    if (!X6_B2B_PREFETCH_B) X6_B2B_LOAD_BQ(pass, q, 0);
                    if (X6_B2B_PREFETCH_B == 2 && q == 0) { _Pragma("unroll") for (int q2 = QPRE; q2 < KQ; ++q2) X6_B2B_LOAD_BQ(pass, q2, q2); }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int x = 0; x < TM; ++x)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                c2[x][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (t == 0 || t == 1) ? al[x][q] : ah[x][q]),
                                                                                   (t == 0 || t == 2) ? bl[slot < 0 ? q : 0][nt] : bh[slot < 0 ? q : 0][nt], c2[x][nt], 0, 0, 0);
                }
#pragma unroll
                for (int x = 0; x < TM; ++x)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) stgw[(x * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * LDC2 + nt * 32 + c32] = c2[x][nt][r];
            }
            const f32x4 cs_a = *reinterpret_cast<const f32x4*>(p.colscale3 + n0p + lc), cs_b = *reinterpret_cast<const f32x4*>(p.colscale3 + n0p + lc + 4);
            const f32x4 b3_a = *reinterpret_cast<const f32x4*>(p.bias3 + n0p + lc), b3_b = *reinterpret_cast<const f32x4*>(p.bias3 + n0p + lc + 4);
            if (X6_B2B_PREFETCH_B && pass + 1 < npass) { _Pragma("unroll") for (int q = 0; q < QPRE; ++q) X6_B2B_LOAD_BQ(pass + 1, q, q); }
            if constexpr (!RES_EARLY) {
                X6_B2B_LOAD_RES(pass, 0);
                if constexpr (RW == 64 && RSETS == 8) X6_B2B_LOAD_RES(pass, 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's stores have reached the LDS before its loads ...
            if constexpr (WN == 1) __builtin_amdgcn_wave_barrier();
            else __builtin_amdgcn_s_barrier();                   // ... and (WN == 2) the partner's partial tile is there too
            X6_B2B_ROWS(0);
            if constexpr (RW == 64) {
                if constexpr (RSETS != 8) X6_B2B_LOAD_RES(pass, 4);
                X6_B2B_ROWS(4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (p.gap) {   // sums over the aligned 16-row groups of the rows this wave finished, rows added in order (batch-invariant)
#pragma unroll
                for (int g = 0; g < RW / 16; ++g) {
                    const int mg = m0 + r0 + rw0 + g * 16;
                    float t = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += stgf[(rw0 + g * 16 + r) * LDC2 + lane];
                    if (mg < p.gap_rows) p.gap[(int64_t)(mg >> 4) * p.N3 + n0p + lane] = t;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (WN == 2) {                             // the partner has read this wave's partial tile: the next pass may overwrite it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
#undef X6_B2B_ROWS
#undef X6_B2B_LOAD_RES
#undef X6_B2B_LOAD_B
#undef X6_B2B_LOAD_BQ
        if (p.amax_out) {   // (workgroup-uniform) outputs are >= 0: integer max of the bits = float max, order-free
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                tmax0 = fmaxf(tmax0, __shfl_xor(tmax0, o));
                tmax1 = fmaxf(tmax1, __shfl_xor(tmax1, o));
            }
            __syncthreads();
            if (lane == 0) {
                atomicMax(simg, __float_as_uint(tmax0));
                atomicMax(simg + 1, __float_as_uint(tmax1));
            }
            __syncthreads();
            if (tid < 2 && simg[tid] != 0u) atomicMax(p.amax_out + img_first + tid, simg[tid]);
        }
        X6_STAMP(3);
        return;
    }

    // ---- epilogue, staged through LDS in 64-row chunks (C/D map of the 32x32 MFMA: col = lane & 31,
    // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); a thread owns two groups of 4 columns, at lcA and lcB: bias / residual /
    // activation fused, 16-byte stores.  With split planes on either side (sp3 output or residual) the groups are adjacent
    // (8 consecutive columns = whole 16-byte plane units); for plain fp32 traffic group B sits half a tile row further, so
    // that the 64 lanes of ONE store / load instruction cover contiguous 512-byte row segments (adjacent groups made every
    // instruction touch 16 of each 32 bytes: twice the memory requests, and the 256x256 epilogue is bound by them)
    float* stg = reinterpret_cast<float*>(smem);
    constexpr int LDC = BN + 4;
    constexpr int EP_ROWS = 64;
    constexpr int C8 = BN / 8;
    static_assert(NT % C8 == 0 && (EP_ROWS * C8) % NT == 0, "epilogue chunk must divide over the workgroup");
    static_assert(EP_ROWS * LDC * 4 <= 2 * STAGE, "epilogue chunk must fit the staging LDS");   // (the first two stages)
    constexpr int EP_STEP = NT / C8;
    constexpr int EP_ITERS = EP_ROWS / EP_STEP;
    const int half = lane >> 5;
    const bool planes = p.out_sp3 != nullptr || p.residual_sp3 != nullptr || p.out_h2 != nullptr;   // workgroup-uniform
    const int lcA = planes ? (tid % C8) * 8 : (tid % C8) * 4;
    const int lcB = planes ? lcA + 4 : lcA + BN / 2;
    const int lr0 = tid / C8;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bias_a = zero4, bias_b = zero4;
    if (slice < 0 && p.bias) {
        bias_a = *reinterpret_cast<const f32x4*>(p.bias + n0 + lcA);
        bias_b = *reinterpret_cast<const f32x4*>(p.bias + n0 + lcB);
    }
    [[maybe_unused]] f32x4 cs_a = zero4, cs_b = zero4;   // H2: the inverse weight-row scales of this thread's columns
    if (H2) {
        cs_a = *reinterpret_cast<const f32x4*>(p.colscale + n0 + lcA);
        cs_b = *reinterpret_cast<const f32x4*>(p.colscale + n0 + lcB);
    }
    const bool interior = m0 + BM <= p.M;   // workgroup-uniform
    // per-image maxima of this tile (amax_out), collected in LDS right behind the staging rows: one global atomic per image and tile (a
    // global atomic per row segment - 1.6 M per launch of layer2's conv3, hundreds per address - cost 1.7 ms of a 1.1 ms launch)
    static_assert(EP_ROWS * LDC * 4 + 64 <= 2 * STAGE + 1024, "no room for the per-image maxima behind the staging rows");
    unsigned* simg = reinterpret_cast<unsigned*>(smem + EP_ROWS * LDC * 4);
    const bool per_img = (H2 || p.out_h2 != nullptr || p.amax_out != nullptr) && p.Ho * p.Wo > 0;   // workgroup-uniform
    const int img_first = per_img ? m0 / (p.Ho * p.Wo) : 0;
    const bool lds_amax = p.amax_out && slice < 0 && p.Ho * p.Wo >= 18;   // at most 16 images under the tile's 256 rows
    // The per-image scalars of the rows (1 / input scale, output scale).  Images of at least BM rows (every launch of the ResNet engine): the
    // tile's TWO images' values are loaded here, once, and a row selects.  (Loaded per row inside the row loop, each load sat behind the
    // stores of the rows before it, and the wait for a load is a wait for every older operation of the wave: one write acknowledgement
    // per row.)  The loads are unconditional - a launch without the table reads the first word of the weights and ignores it - so that no
    // value is a phi with a default that the compiler may resolve by a copy behind the load.
    const bool two_sc = per_img && p.Ho * p.Wo >= BM;
    const int img_second = per_img && (int64_t)(img_first + 1) * (p.Ho * p.Wo) < p.M ? img_first + 1 : img_first;
    const float* inv_tab = H2 ? p.img_in_inv : reinterpret_cast<const float*>(p.w);
    const float* osc_tab = p.out_h2 ? p.img_out_scale : reinterpret_cast<const float*>(p.w);
    // (uniform addresses read through the constant address space: scalar loads into scalar registers - the tables were written by earlier launches)
    const float inv0 = x6_uniform_load(inv_tab + (H2 && two_sc ? img_first : 0)), inv1 = x6_uniform_load(inv_tab + (H2 && two_sc ? img_second : 0));
    const float osc0 = x6_uniform_load(osc_tab + (p.out_h2 && two_sc ? img_first : 0)), osc1 = x6_uniform_load(osc_tab + (p.out_h2 && two_sc ? img_second : 0));
    if (lds_amax && tid < 16) simg[tid] = 0u;   // (ordered before the first use by the barrier of the first pass)
    // images of at least BM rows (layer1 / layer2): a tile spans at most TWO images - a thread keeps one running maximum for each in
    // registers and the workgroup reduces them once, after the last pass (no shuffles or LDS atomics per row segment)
    const bool two_img = lds_amax && p.Ho * p.Wo >= BM;
    float tmax0 = 0.f, tmax1 = 0.f;
#pragma unroll
    for (int pass = 0; pass < BM / EP_ROWS; ++pass) {
        if (pass > 0) __syncthreads();
        // residual rows of this pass: requested before the accumulators are staged, in flight under the LDS traffic.  (Requesting them
        // one pass ahead changes nothing: the epilogue takes the time the XCD's memory path needs for the tile's bytes - 17.7 k cycles
        // for 256 KB of stores, 34 k with the 256 KB residual, chip-wide 7 TB/s - not the latency of these loads.)
        f32x4 ra[EP_ITERS], rb[EP_ITERS];
        if (slice < 0 && p.residual) {
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it) {
                const int m = m0 + pass * EP_ROWS + it * EP_STEP + lr0;
                ra[it] = zero4;
                rb[it] = zero4;
                if (interior || m < p.M) {
                    const float* r = p.residual + (int64_t)m * p.N + n0;
                    ra[it] = *reinterpret_cast<const f32x4*>(r + lcA);
                    rb[it] = *reinterpret_cast<const f32x4*>(r + lcB);
                }
            }
        } else if (slice < 0 && p.residual_sp3) {
            // the residual in split planes: x = (hi + mid) + lo reconstructs the fp32 value exactly
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it) {
                const int m = m0 + pass * EP_ROWS + it * EP_STEP + lr0;
                ra[it] = zero4;
                rb[it] = zero4;
                if (interior || m < p.M) {
                    const char* r = p.residual_sp3 + (int64_t)m * ((int64_t)p.N * 6) + sp3_offset(n0 + lcA);
                    const u32x4 hi = *reinterpret_cast<const u32x4*>(r), mid = *reinterpret_cast<const u32x4*>(r + 32),
                                lo = *reinterpret_cast<const u32x4*>(r + 64);
#define X6_UNSPLIT(h_, m_, l_, sh_) ((__uint_as_float(sh_ ? (h_) & 0xffff0000u : (h_) << 16) + __uint_as_float(sh_ ? (m_) & 0xffff0000u : (m_) << 16)) + \
                                      __uint_as_float(sh_ ? (l_) & 0xffff0000u : (l_) << 16))
                    ra[it] = (f32x4){X6_UNSPLIT(hi.x, mid.x, lo.x, 0), X6_UNSPLIT(hi.x, mid.x, lo.x, 1), X6_UNSPLIT(hi.y, mid.y, lo.y, 0),
                                     X6_UNSPLIT(hi.y, mid.y, lo.y, 1)};
                    rb[it] = (f32x4){X6_UNSPLIT(hi.z, mid.z, lo.z, 0), X6_UNSPLIT(hi.z, mid.z, lo.z, 1), X6_UNSPLIT(hi.w, mid.w, lo.w, 0),
                                     X6_UNSPLIT(hi.w, mid.w, lo.w, 1)};
#undef X6_UNSPLIT
                }
            }
        }
        if constexpr (M16) {   // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int rb0 = wm * TM * 32 + x * 16;
                if (rb0 / EP_ROWS == pass) {
#pragma unroll
                    for (int y = 0; y < 4; ++y)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            stg[(rb0 % EP_ROWS + 4 * g16 + r) * LDC + wn * TN * 32 + y * 16 + r16] = acc16[x][y][r];
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb0 = (wm * TM + i) * 32;
            if (rb0 / EP_ROWS == pass) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stg[(rb0 % EP_ROWS + 4 * half + (r & 3) + 8 * (r >> 2)) * LDC + (wn * TN + j) * 32 + (lane & 31)] = acc[i][j][r];
            }
        }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int lr = it * EP_STEP + lr0;
            const int trow = pass * EP_ROWS + lr;
            const int m = m0 + trow;
            if (!interior && m >= p.M) continue;
            f32x4 va = *reinterpret_cast<const f32x4*>(stg + lr * LDC + lcA);
            f32x4 vb = *reinterpret_cast<const f32x4*>(stg + lr * LDC + lcB);
            if (slice >= 0) {   // K slice of a split tile: raw partial sums, the epilogue runs in splitk_finish_x6
                float* o = p.partial + ((int64_t)split_tile * p.nsplit + slice) * (BM * BN) + (int64_t)trow * BN;
                *reinterpret_cast<f32x4*>(o + lcA) = va;
                *reinterpret_cast<f32x4*>(o + lcB) = vb;
                continue;
            }
            const int img = per_img ? m / (p.Ho * p.Wo) : 0;
            if (H2) {   // powers of two: exact
                const float rs = two_sc ? (img == img_first ? inv0 : inv1) : p.img_in_inv[img];
                va = va * (cs_a * rs);
                vb = vb * (cs_b * rs);
            }
            va += bias_a;
            vb += bias_b;
            if (p.residual || p.residual_sp3) {   // (acc + bias) + residual: the same order on every path
                va += ra[it];
                vb += rb[it];
            }
            if (p.act == 1) {
                va = (f32x4){fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f)};
                vb = (f32x4){fmaxf(vb.x, 0.f), fmaxf(vb.y, 0.f), fmaxf(vb.z, 0.f), fmaxf(vb.w, 0.f)};
            } else if (p.act == 2) {
                va = act_gelu4(va);
                vb = act_gelu4(vb);
            }
            const int64_t o = (int64_t)m * p.N + n0;
            if (p.out && m < p.out_rows) {
                *reinterpret_cast<f32x4*>(p.out + o + lcA) = va;
                *reinterpret_cast<f32x4*>(p.out + o + lcB) = vb;
            }
            if (p.out_sp3) store_sp3_x8(p.out_sp3 + (int64_t)m * ((int64_t)p.N * 6), n0 + lcA, va, vb);   // (planes: lcB = lcA + 4)
            if (p.out_h2 || p.amax_out) {   // hand-over to the f16x2 layers: fp16 planes with the image's scale, the image's maximum
                if (p.out_h2) store_h2_x8(p.out_h2 + (int64_t)m * ((int64_t)p.N * 4), n0 + lcA, va, vb,
                                          two_sc ? (img == img_first ? osc0 : osc1) : p.img_out_scale[img]);
                if (p.amax_out) {   // outputs are >= 0 (ReLU): integer max of the bits = float max, order-free; per tile in LDS first
                    float mx = fmaxf(fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w)), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
                    if (two_img) {
                        if (img == img_first) tmax0 = fmaxf(tmax0, mx);
                        else tmax1 = fmaxf(tmax1, mx);
                    } else {
                    _Pragma("unroll") for (int o = C8 / 2; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                    if ((tid % C8) == 0) {
                        if (lds_amax) atomicMax(simg + (img - img_first), __float_as_uint(mx));
                        else atomicMax(p.amax_out + img, __float_as_uint(mx));
                    }
                    }
                }
            }
            if (p.gap) {   // the finished values go back to the staging rows for the group sums below
                *reinterpret_cast<f32x4*>(stg + lr * LDC + lcA) = va;
                *reinterpret_cast<f32x4*>(stg + lr * LDC + lcB) = vb;
            }
        }
        if (p.gap && slice < 0) {
            // fused spatial mean, stage 1: sums over the aligned 16-row (14x14 maps: 4-row) groups of this chunk, rows added in order.
            // Images start at multiples of the group (the launcher checks Ho*Wo % group == 0), so a group never spans two images and
            // the grouping does not depend on where in the batch an image sits: bits are batch-invariant.
            __syncthreads();
            if (p.gap_shift == 4) {
                for (int e = tid; e < (EP_ROWS / 16) * BN; e += NT) {
                    const int g = e / BN, col = e - g * BN;
                    const int mg = m0 + pass * EP_ROWS + g * 16;
                    if (mg < p.gap_rows) {
                        float t = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) t += stg[(g * 16 + r) * LDC + col];
                        p.gap[(int64_t)(mg >> 4) * p.N + n0 + col] = t;
                    }
                }
            } else {
                for (int e = tid; e < (EP_ROWS / 4) * BN; e += NT) {
                    const int g = e / BN, col = e - g * BN;
                    const int mg = m0 + pass * EP_ROWS + g * 4;
                    if (mg < p.gap_rows) {
                        float t = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) t += stg[(g * 4 + r) * LDC + col];
                        p.gap[(int64_t)(mg >> 2) * p.N + n0 + col] = t;
                    }
                }
            }
        }
    }
    if (two_img) {   // (workgroup-uniform)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            tmax0 = fmaxf(tmax0, __shfl_xor(tmax0, o));
            tmax1 = fmaxf(tmax1, __shfl_xor(tmax1, o));
        }
        if (lane == 0) {
            atomicMax(simg, __float_as_uint(tmax0));
            atomicMax(simg + 1, __float_as_uint(tmax1));
        }
    }
    if (lds_amax) {
        __syncthreads();
        if (tid < 16 && simg[tid] != 0u) atomicMax(p.amax_out + img_first + tid, simg[tid]);
    }
    X6_STAMP(3);
#endif
}

// Sums the K slices of the split tiles in slice order (deterministic) and applies the epilogue.
template <int BM, int BN>
__global__ __launch_bounds__(256) void splitk_finish_x6(const X6Params p) {
    const int split_tile = blockIdx.y;
    const int tile = p.full_tiles + split_tile;
    int tm, tn;
    {
        const int per_group = p.group_m * p.tiles_n;
        const int g = tile / per_group;
        const int first = g * p.group_m;
        const int gsz = p.tiles_m - first < p.group_m ? p.tiles_m - first : p.group_m;
        const int w = tile - g * per_group;
        tm = first + w % gsz;
        tn = w / gsz;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    const int e8 = blockIdx.x * 256 + threadIdx.x;  // 8-column group index inside the tile
    if (e8 >= BM * BN / 8) return;
    const int lr = e8 / (BN / 8), lc = (e8 % (BN / 8)) * 8;
    const int m = m0 + lr;
    if (m >= p.M) return;
    const float* pt = p.partial + (int64_t)split_tile * p.nsplit * (BM * BN) + lr * BN + lc;
    f32x4 va = *reinterpret_cast<const f32x4*>(pt), vb = *reinterpret_cast<const f32x4*>(pt + 4);
    for (int k = 1; k < p.nsplit; ++k) {
        va += *reinterpret_cast<const f32x4*>(pt + (int64_t)k * (BM * BN));
        vb += *reinterpret_cast<const f32x4*>(pt + (int64_t)k * (BM * BN) + 4);
    }
    const int64_t o = (int64_t)m * p.N + n0 + lc;
    if (p.bias) {
        va += *reinterpret_cast<const f32x4*>(p.bias + n0 + lc);
        vb += *reinterpret_cast<const f32x4*>(p.bias + n0 + lc + 4);
    }
    if (p.residual) {
        va += *reinterpret_cast<const f32x4*>(p.residual + o);
        vb += *reinterpret_cast<const f32x4*>(p.residual + o + 4);
    }
    if (p.act == 1) {
        va = (f32x4){fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f)};
        vb = (f32x4){fmaxf(vb.x, 0.f), fmaxf(vb.y, 0.f), fmaxf(vb.z, 0.f), fmaxf(vb.w, 0.f)};
    } else if (p.act == 2) {
        va = act_gelu4(va);
        vb = act_gelu4(vb);
    }
    if (p.out && m < p.out_rows) {
        *reinterpret_cast<f32x4*>(p.out + o) = va;
        *reinterpret_cast<f32x4*>(p.out + o + 4) = vb;
    }
    if (p.out_sp3) store_sp3_x8(p.out_sp3 + (int64_t)m * ((int64_t)p.N * 6), n0 + lc, va, vb);
}

template <int BM, int BN, int WM, int WN, bool TAPS, bool M16 = false, bool DUAL = false, bool AF32 = false, bool H2 = false, int B2B = 0>
static int launch_x6_variant(relax_handle* h, X6Params& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr int WG_PER_CU = NT == 256 ? 2 : 1;
    p.tiles_n = p.N / BN;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.ntiles = p.tiles_m * p.tiles_n;
    p.group_m = h->gemm.group_m;
    p.partial = nullptr;
    // Tail split-K: the last, partial round of tiles is cut along K (cost model: host_logic.cpp, shared with gemm.hip);
    // splitk_finish_x6 knows neither the fused group sums nor a split-plane residual, so those launches run unsplit
    const bool can_split = h->gemm.split_k && !p.gap && !p.residual_sp3 && !p.no_split && !p.out_h2 && !p.amax_out && !H2;
    const host::TailSplit ts = host::choose_tail_split(p.ntiles, 256 * WG_PER_CU, p.K / 16, 8, can_split);
    p.full_tiles = ts.full_tiles;
    p.nsplit = ts.nsplit;
    if (p.nsplit > 1) {
        const size_t need = sizeof(float) * (size_t)(p.ntiles - p.full_tiles) * p.nsplit * BM * BN;
        RELAX_TRY(ensure_buf(h, h->splitk_ws, need < (size_t)(64 << 20) ? (size_t)(64 << 20) : need));
        p.partial = static_cast<float*>(h->splitk_ws.p);
    }
    constexpr int CH = H2 ? kH2ChunkBytes : kChunkBytes;
    // (B2B: the waves' own staging rows of the conv3 passes, 64 x 68 floats each, + the row scales + two maxima: 70.7 KB, two workgroups per CU)
    constexpr size_t stages_lds = (((M16 && !TAPS && !DUAL) || (H2 && X6_H2_STAGES == 3)) ? 3 : 2) * (size_t)(BM * (AF32 ? 64 : CH) + BN * CH);
    constexpr size_t b2b_stg = (size_t)(WM * WN) * (BM / WM) * 68 * 4;     // the waves' own staging rows of the conv3 passes
    constexpr size_t lds = B2B ? (b2b_stg > stages_lds ? b2b_stg : stages_lds) + (size_t)WN * BM * 4 + 64 : stages_lds + 1024;
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6<BM, BN, WM, WN, TAPS, M16, DUAL, AF32, H2, B2B>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[h->device] = true;
    }
    const int units = p.full_tiles + (p.ntiles - p.full_tiles) * p.nsplit;
    X6_STAMPS_BEFORE_LAUNCH(h, p, units);
    hipLaunchKernelGGL((gemm_x6<BM, BN, WM, WN, TAPS, M16, DUAL, AF32, H2, B2B>), dim3(units), dim3(NT), lds, s, p);
    X6_STAMPS_AFTER_LAUNCH(BM, BN, h, p, units, s);
    if (p.nsplit > 1)
        hipLaunchKernelGGL((splitk_finish_x6<BM, BN>), dim3(BM * BN / 8 / 256, p.ntiles - p.full_tiles), dim3(256), 0, s, p);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// K order of the back-to-back form (B2B above): wp[n][16q + 8h + e] = w[n][16q + 4h + (e & 3) + 8 (e >> 2)] - position (q, h, e) of an
// MFMA A fragment holds the channel that accumulator register 8 (q & 1) + e of lane half h holds in the transposed tile of the 3x3
__global__ __launch_bounds__(256) void b2b_permute_k_kernel(const float* __restrict__ w, float* __restrict__ wp, int K, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t n = i / K;
    const int k = (int)(i - n * K);
    const int q = k >> 4, hh = (k >> 3) & 1, e = k & 7;
    wp[i] = w[n * K + 16 * q + 4 * hh + (e & 3) + 8 * (e >> 2)];
}

int launch_b2b_permute_k(relax_handle* h, const float* w, float* wp, int rows, int K, hipStream_t s) {
    RELAX_REQUIRE(h, w && wp && rows > 0 && K % 16 == 0, "b2b_permute_k: bad arguments");
    const int64_t total = (int64_t)rows * K;
    hipLaunchKernelGGL(b2b_permute_k_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wp, K, total);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_conv_x6(relax_handle* h, const ConvDescX6& d, hipStream_t s) {
    X6Params p{};
    p.a = static_cast<const char*>(d.in);
    p.w = static_cast<const char*>(d.w);
    p.bias = d.bias; p.residual = d.residual; p.out = d.out; p.out_sp3 = static_cast<char*>(d.out_sp3);
    p.residual_sp3 = static_cast<const char*>(d.residual_sp3);
    p.out_h2 = static_cast<char*>(d.out_h2); p.img_out_scale = d.img_out_scale; p.amax_out = d.amax_out;
    p.colscale = d.colscale; p.img_in_inv = d.img_in_inv; p.img_in_scale = d.img_in_scale;
    p.gap = d.gap_groups;
    p.M = d.Nimg * d.Ho * d.Wo;
    p.N = d.Cout;
    p.K = d.KH * d.KW * d.Cin + (d.in2 ? d.Cin2 : 0);
    p.a2 = static_cast<const char*>(d.in2);
    p.K1 = d.KH * d.KW * d.Cin; p.H2 = d.H2; p.W2 = d.W2; p.Cin2 = d.Cin2; p.stride2 = d.stride2;
    p.no_split = d.no_split;
    p.out_rows = d.out_rows > 0 && d.out_rows < p.M ? d.out_rows : p.M;
    p.gap_rows = d.gap_rows > 0 && d.gap_rows < p.M ? d.gap_rows : p.M;
    p.gap_shift = (d.Ho * d.Wo) % 16 == 0 ? 4 : 2;
    p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Ho = d.Ho; p.Wo = d.Wo;
    p.KW = d.KW; p.stride = d.stride; p.pad = d.pad;
    p.act = d.act;
    const bool taps = d.KH * d.KW > 1;
    const bool b2b = d.w3 != nullptr;
    p.w3 = static_cast<const char*>(d.w3); p.colscale3 = d.colscale3; p.bias3 = d.bias3; p.N3 = d.Cout3; p.sp3_sub = d.sp3_sub == 2 ? 2 : 1;
    p.x2 = d.x2;
    RELAX_REQUIRE(h, !d.x2 || (b2b && d.Cout == 64 && !d.residual), "x6 conv: a second conv3 source (fp32 rows of 64 channels) goes with the 64-wide back-to-back form, without a residual");
    RELAX_REQUIRE(h, d.sp3_sub != 2 || (b2b && (d.out_sp3 || d.out_h2) && d.Ho % 2 == 0 && d.Wo % 2 == 0), "x6 conv: the stride-2 plane output goes with the back-to-back form and even maps");
    RELAX_REQUIRE(h, !b2b || (d.in_h2 && taps && (d.Cout == 64 || d.Cout == 128) && d.Cout3 % 64 == 0 && d.Cout3 > 0 && d.colscale3 && d.bias3 && d.bias && d.act == 1 &&
                              (d.residual || d.x2) && !d.residual_sp3 && !d.in2 && (d.Ho * d.Wo) % 16 == 0 && d.Ho * d.Wo >= 256),
                  "x6 conv: the back-to-back form needs the f16x2 3x3 onto 64 or 128 columns, ReLU, both biases, an fp32 residual (or the second source) and images of >= 256 pixels (a multiple of 16)");
    RELAX_REQUIRE(h, p.M > 0 && p.N > 0 && p.K > 0, "x6 conv/gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    RELAX_REQUIRE(h, d.Cin % 16 == 0, "x6 conv/gemm: Cin=%d must be a multiple of 16", d.Cin);
    RELAX_REQUIRE(h, p.N % 64 == 0, "x6 conv/gemm: N=%d must be a multiple of 64", p.N);
    RELAX_REQUIRE(h, (int64_t)p.K * 6 * 256 < kMaxRecords, "x6 conv/gemm: K=%d too large", p.K);
    RELAX_REQUIRE(h, !d.in_h2 || (d.colscale && d.img_in_inv && !d.in2 && !d.in_f32 && p.N % 256 != 0 && p.K >= 256 && taps),
                  "x6 conv: fp16-plane operands go with a KxK convolution of 64 or 128 output columns, K >= 256, and need both scale tables");
    RELAX_REQUIRE(h, d.out || d.out_sp3 || d.gap_groups || d.out_h2, "x6 conv/gemm: no output requested");
    RELAX_REQUIRE(h, !(d.residual && d.residual_sp3), "x6 conv/gemm: two residuals");
    RELAX_REQUIRE(h, (!d.out_h2 || d.img_out_scale) && (!(d.out_h2 || d.amax_out) || d.act == 1),
                  "x6 conv/gemm: the fp16-plane output needs its per-image scales, and it and the per-image maximum go with a ReLU");
    RELAX_REQUIRE(h, !d.gap_groups || (d.Ho * d.Wo) % 4 == 0, "x6 conv: the fused spatial mean needs Ho*Wo %% 4 == 0");
    RELAX_REQUIRE(h, !taps || (d.pad >= 0 && d.KH * d.KW <= 32), "x6 conv: bad padding, or more than 32 taps (%dx%d)", d.KH, d.KW);
    RELAX_REQUIRE(h, taps || d.pad == 0, "x6 conv: 1x1 with padding is not supported");
    RELAX_REQUIRE(h, !d.in2 || (!taps && d.stride == 1 && d.Cin2 % 16 == 0 && d.stride2 >= 1 && p.N % 256 == 0 &&
                                (d.H2 - 1) / d.stride2 + 1 == d.Ho && (d.W2 - 1) / d.stride2 + 1 == d.Wo),
                  "x6 conv: a second activation source needs a 1x1 stride-1 first source, Cout %% 256 == 0 and a matching output grid");
    const bool f32_h2 = d.in_f32 && d.img_in_scale != nullptr;   // fp32 rows split into fp16 planes in the K loop, weights as fp16 planes: three products
    RELAX_REQUIRE(h, !f32_h2 || (d.colscale && d.img_in_inv && p.K >= 256 && !d.in_h2 && p.N % 128 == 0), "x6 conv: fp32 rows on f16x2 need the weights' inverse "
                  "row scales, both per-image scale tables, K >= 256 (three products) and N %% 128 == 0");
    RELAX_REQUIRE(h, !d.in_f32 || (!taps && !d.in2 && ((d.stride == 1 && p.N % 256 != 0) || f32_h2)),
                  "x6 conv/gemm: fp32 activation rows go with a 1x1 stride-1 contraction of 64 or 128 (not a multiple of 256) output columns");
    // every operand is read / written in 16-byte units (LDS-DMA pieces, f32x4 bias / residual / output accesses, plane units)
    auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    RELAX_REQUIRE(h, aligned16(d.in) && aligned16(d.w) && aligned16(d.bias) && aligned16(d.residual) && aligned16(d.residual_sp3) &&
                         aligned16(d.out) && aligned16(d.out_sp3) && aligned16(d.gap_groups) && aligned16(d.in2),
                  "x6 conv/gemm: every operand pointer must be 16-byte aligned");
    const double flops = d.flops > 0 ? d.flops : 2.0 * p.M * ((double)p.N * (double)p.K + (b2b ? (double)(p.N + (d.x2 ? 64 : 0)) * d.Cout3 : 0.0));
    // algorithmic HBM bytes: activation in (6 B per value), weights, outputs (+ residual), each touched once
    const double n_out = b2b ? d.Cout3 : p.N;    // (back to back: the outputs and the residual are the conv3's)
    const double bytes = ((d.in_f32 || d.in_h2) ? 4.0 : 6.0) * ((double)d.Nimg * d.H * d.W * d.Cin) +
                         ((d.in_h2 || f32_h2) ? 4.0 : 6.0) * ((double)p.N * p.K + (d.in2 ? (double)p.M * d.Cin2 : 0.0)) + (b2b ? 4.0 * (p.N + (d.x2 ? 64 : 0)) * d.Cout3 : 0.0) + (d.x2 ? 4.0 * p.M * 64 : 0.0) +
                         (double)p.M * n_out * ((d.out ? 4.0 : 0.0) + (d.out_sp3 ? (d.sp3_sub == 2 ? 1.5 : 6.0) : 0.0) + (d.out_h2 ? 4.0 : 0.0) + (d.residual ? 4.0 : 0.0) + (d.residual_sp3 ? 6.0 : 0.0));
    int span;
    RELAX_TRY(prof_begin(h, s, (d.in_h2 || f32_h2) ? 5 : 2, flops, &span, bytes));   // (kind 5 = f16x2: three executed products per fp32 product)
    int rc;
    // N % 256 == 0: one 8-wave workgroup per CU on a 256x256 tile (fewest staged bytes per MFMA); N = 128 / 64 layers: the same
    // loop on four waves, two workgroups per CU
    if (b2b && d.Cout == 128)   // ... of a 128-wide block: eight waves, conv3's K split over the two column waves
        rc = launch_x6_variant<256, 128, 4, 2, true, false, false, false, true, true>(h, p, s);
    else if (b2b && d.x2)   // ... of a first block: conv3's K = [the 3x3's channels | the block input's]: the downsample convolution folded in
        rc = launch_x6_variant<256, 64, 4, 1, true, false, false, false, true, 2>(h, p, s);
    else if (b2b)   // the 3x3 and the block's conv3 back to back (the 3x3's tile stays in registers)
        rc = h->gemm.b2b_rows == 256 ? launch_x6_variant<256, 64, 4, 1, true, false, false, false, true, true>(h, p, s)
                                     : launch_x6_variant<128, 64, 4, 1, true, false, false, false, true, true>(h, p, s);
    else if (d.in_h2)    // two fp16 planes on both sides: the 3x3 convolutions of layer1 / layer2 under f16x2
        // (N = 128 on EIGHT waves of 64 x 64: the second accumulator does not fit the 128 x 64 wave tile of the four-wave form)
        rc = p.N % 128 == 0 ? launch_x6_variant<256, 128, 4, 2, true, false, false, false, true>(h, p, s)
                            : launch_x6_variant<256, 64, 4, 1, true, false, false, false, true>(h, p, s);
    else if (f32_h2)     // fp32 activation rows split into two fp16 planes in the K loop (the image's scale from its measured maximum), weights as planes
        // (N = 128 on EIGHT waves of 64 x 64, as the 3x3 form: the second accumulator does not fit a 128 x 64 wave tile; the 64-column
        // launches - layer1's conv1 - wait on memory either way and stay on bf16x6: their f16x2 instantiation spills)
        rc = launch_x6_variant<256, 128, 4, 2, false, false, false, true, true>(h, p, s);
    else if (d.in_f32)   // fp32 activation rows, split in the K loop (ResNet-50: the block outputs of layer1 / layer2 travel as fp32)
        rc = p.N % 128 == 0 ? launch_x6_variant<256, 128, 2, 2, false, false, false, true>(h, p, s)
                            : launch_x6_variant<256, 64, 4, 1, false, false, false, true>(h, p, s);
    else if (d.in2)
        rc = launch_x6_variant<256, 256, 2, 4, false, true, true>(h, p, s);
    else if (p.N % 256 == 0)   // the 8-wave tile runs the 16x16x32 form of the loop
        rc = taps ? launch_x6_variant<256, 256, 2, 4, true, true>(h, p, s) : launch_x6_variant<256, 256, 2, 4, false, true>(h, p, s);
    else if (p.N % 128 == 0)
        rc = taps ? launch_x6_variant<256, 128, 2, 2, true>(h, p, s) : launch_x6_variant<256, 128, 2, 2, false>(h, p, s);
    else
        rc = taps ? launch_x6_variant<256, 64, 4, 1, true>(h, p, s) : launch_x6_variant<256, 64, 4, 1, false>(h, p, s);
    if (rc != RELAX_OK) { prof_abort(h, span); return rc; }
    RELAX_TRY(prof_end(h, s, span));
    return RELAX_OK;
}

}  // namespace relax
