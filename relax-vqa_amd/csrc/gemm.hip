// The contraction kernel of the backbones: fp32 implicit-GEMM convolution / GEMM on the gfx950
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] + residual[m, n] )
//
//   m = output pixel (image, oy, ox) of an NHWC activation  (or a plain GEMM row)
//   k = (dy*KW + dx)*Cin + c, gathered on the fly from the NHWC input (never materialised)
//   n = output channel; W is stored [Cout][Kpad] (K fastest) so A and W tiles stage identically
//
// Tiling for 64-wide wavefronts: a workgroup of WM x WN waves owns a BM x BN tile, each wave a (TM*32) x (TN*32) sub-tile
// held in TM*TN 32x32 MFMA accumulators.  The K loop runs BK-deep steps through two LDS buffers (register-staged prefetch of
// step k+1 under the MFMAs of step k, one barrier per step).  LDS rows are padded to BK + 4 floats so the ds_read_b128
// fragment reads are bank-conflict free.  Because the MFMA sums over k, the k order inside a step is free: lane (i, h)
// reads the 4 consecutive floats k = 8q+4h .. 8q+4h+3 of row i with ONE 128-bit LDS read and feeds them to 4 MFMAs.
// Workgroup ids are remapped so that the workgroups sharing one XCD (and its L2) walk neighbouring tiles.
// Defaults (launch_conv): fp32 128x128 tile, 4 waves, BK = 16, three workgroups per CU; 128x64 for N % 128 != 0;
// opt-in bf16x3 precision: 256x256 tile on 8 waves for large plain GEMMs, 128-wide tiles otherwise (LAB_NOTES.md 3.1 / 3.2).
#include "relax_internal.h"
#include "host_logic.h"
#include "gelu.h"

namespace relax {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));  // native vector: stays in VGPRs (HIP's float4 struct did not)


struct GemmParams {
    const float* in;
    const float* w;
    const float* bias;
    const float* residual;
    float* out;
    float* partial;  // split-K partial tiles [split_tile][slice][BM*BN], or null
    int M, N, Kpad;
    int H, W, Cin, cin_log2, Ho, Wo, KW, stride, pad, ntaps;
    int act;
    int tiles_m, tiles_n, ntiles, group_m;
    int full_tiles;  // tiles [0, full_tiles) run the whole K loop and the epilogue in-kernel
    int nsplit;      // tiles [full_tiles, ntiles) are cut into nsplit K slices (raw partial sums)
};

__device__ inline int xcd_remap(int b, int nwg) {
    // blocks b, b+8, ... share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles.
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// x = hi + lo with hi = bf16_rne(x) and lo = bf16_rne(x - hi): 16 mantissa bits kept.  Both roundings are to nearest, so
// the dropped lo*lo term has no preferred sign (a truncating hi makes lo >= 0 relative to x and the error drifts: measured
// 1.2e-4 instead of 5e-6 on the 13120-d layer-stack vector).  3 VALU per element (half a cvt_pk twice, shift/and, sub).
__device__ inline void split_bf16x4(const f32x4 v, uint2* hi, uint2* lo) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v.x, v.y}, bf16x2_t));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v.z, v.w}, bf16x2_t));
    const float l0 = v.x - __uint_as_float(h01 << 16), l1 = v.y - __uint_as_float(h01 & 0xffff0000u);
    const float l2 = v.z - __uint_as_float(h23 << 16), l3 = v.w - __uint_as_float(h23 & 0xffff0000u);
    hi->x = h01;
    hi->y = h23;
    lo->x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){l0, l1}, bf16x2_t));
    lo->y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){l2, l3}, bf16x2_t));
}

__device__ inline float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return gelu_erf(v);
    return v;
}

// PREC 0: exact fp32 products on v_mfma_f32_32x32x2_f32.
// PREC 1 ("bf16x3", opt-in): every fp32 operand is split on the fly into two bf16 values x = hi + lo (16 mantissa bits
// kept), and a*b is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2^-16
// relative error per product (features stay within ~1e-5 of the fp32 path, bar 1e-3) at 3/16 of the fp32 MFMA cycles.
template <int BM, int BN, int WM, int WN, int BK, int OCC, bool TAPS, int PREC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void conv_gemm_f32(const GemmParams p) {
    static_assert(PREC == 0 || BK == 32 || BK == 16, "bf16x3 LDS row image: [BK x bf16 hi][BK x bf16 lo][16 B pad]");
    constexpr int NT = WM * WN * 64;
    constexpr int LDK = BK + 4;   // padded LDS row: conflict-free b128 writes and reads for BK = 32 (36) and 16 (20)
    constexpr int KL = BK / 4;    // lanes (float4) per K step of a row
    constexpr int PASS = NT / KL; // tile rows staged per pass
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_LOADS = BM / PASS;
    constexpr int B_LOADS = BN / PASS;
    constexpr int STAGE = (BM + BN) * LDK;
    static_assert(BM % PASS == 0 && BN % PASS == 0 && TM >= 1 && TN >= 1, "tile / wave layout mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid / KL;  // row within a staging pass
    const int c4 = tid % KL;    // which float4 of the K step

    // work unit -> (tile, K slice)
    int tile, kt_begin, kt_end, slice = -1, split_tile = 0;
    const int nk = p.Kpad / BK;
    if ((int)blockIdx.x < p.full_tiles) {
        tile = xcd_remap(blockIdx.x, p.full_tiles);
        kt_begin = 0;
        kt_end = nk;
    } else {
        const int u = blockIdx.x - p.full_tiles;
        split_tile = u / p.nsplit;
        slice = u - split_tile * p.nsplit;
        tile = p.full_tiles + split_tile;
        kt_begin = (int)((int64_t)nk * slice / p.nsplit);
        kt_end = (int)((int64_t)nk * (slice + 1) / p.nsplit);
    }
    // grouped order: consecutive tiles walk group_m row-tiles x all column tiles column by column, so the ~64
    // workgroups resident on one XCD share both a few activation row-tiles and a few weight column-tiles in L2
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

    // per-thread row descriptors of the A operand (fixed for the whole K loop)
    int64_t a_base[A_LOADS];
    int a_iy[A_LOADS], a_ix[A_LOADS];
    bool a_ok[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + i * PASS + lrow;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if (TAPS || p.stride != 1) {
            const int img = mm / (p.Ho * p.Wo);
            const int rem = mm - img * (p.Ho * p.Wo);
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            a_base[i] = (((int64_t)img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.Cin;
        } else {
            a_iy[i] = 0;
            a_ix[i] = 0;
            a_base[i] = (int64_t)mm * p.Cin;
        }
    }
    const float* wrow0 = p.w + (int64_t)(n0 + lrow) * p.Kpad + c4 * 4;
    const int64_t wpass = (int64_t)PASS * p.Kpad;  // weight rows per staging pass

    f32x4 ra0[A_LOADS], rb0[B_LOADS];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define RELAX_LOAD_TILE_R(k0_, ra, rb)                                                                        \
    {                                                                                                 \
        const int k_ = (k0_) + c4 * 4;                                                                \
        if (TAPS) {                                                                                   \
            const int tap_ = k_ >> p.cin_log2;                                                        \
            const int c_ = k_ & (p.Cin - 1);                                                          \
            const int dy_ = tap_ / p.KW;                                                              \
            const int dx_ = tap_ - dy_ * p.KW;                                                        \
            const int koff_ = (dy_ * p.W + dx_) * p.Cin + c_;                                         \
            const bool tap_ok_ = tap_ < p.ntaps;                                                      \
            _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                     \
                const bool ok_ = a_ok[i] && tap_ok_ && (unsigned)(a_iy[i] + dy_) < (unsigned)p.H &&   \
                                 (unsigned)(a_ix[i] + dx_) < (unsigned)p.W;                           \
                f32x4 v_ = zero4;                                                                     \
                if (ok_) v_ = *reinterpret_cast<const f32x4*>(p.in + a_base[i] + koff_);              \
                ra[i] = v_;                                                                           \
            }                                                                                         \
        } else {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                     \
                f32x4 v_ = zero4;                                                                     \
                /* rows past M were clamped to row 0: loading them unconditionally keeps the loop branch-free */ \
                if (PREC == 1 || a_ok[i]) v_ = *reinterpret_cast<const f32x4*>(p.in + a_base[i] + k_); \
                ra[i] = v_;                                                                           \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                           \
            rb[i] = *reinterpret_cast<const f32x4*>(wrow0 + i * wpass + (k0_));                       \
    }
#define RELAX_STORE_TILE_R(buf_, ra, rb)                                                                        \
    {                                                                                                 \
        float* As_ = smem + (buf_) * STAGE;                                                           \
        float* Bs_ = As_ + BM * LDK;                                                                  \
        if (PREC == 0) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                       \
                *reinterpret_cast<f32x4*>(As_ + (i * PASS + lrow) * LDK + c4 * 4) = ra[i];            \
            _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                       \
                *reinterpret_cast<f32x4*>(Bs_ + (i * PASS + lrow) * LDK + c4 * 4) = rb[i];            \
        } else {                                                                                      \
            /* row image: [BK x bf16 hi][BK x bf16 lo][16 B pad] = LDK * 4 bytes = LDK / 2 8-byte slots */ \
            uint2* A8_ = reinterpret_cast<uint2*>(As_);                                                 \
            uint2* B8_ = reinterpret_cast<uint2*>(Bs_);                                                 \
            _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                     \
                uint2 hi_, lo_;                                                                       \
                split_bf16x4(ra[i], &hi_, &lo_);                                                      \
                A8_[(i * PASS + lrow) * (LDK / 2) + c4] = hi_;                                               \
                A8_[(i * PASS + lrow) * (LDK / 2) + BK / 4 + c4] = lo_;                                           \
            }                                                                                         \
            _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) {                                     \
                uint2 hi_, lo_;                                                                       \
                split_bf16x4(rb[i], &hi_, &lo_);                                                      \
                B8_[(i * PASS + lrow) * (LDK / 2) + c4] = hi_;                                               \
                B8_[(i * PASS + lrow) * (LDK / 2) + BK / 4 + c4] = lo_;                                           \
            }                                                                                         \
        }                                                                                             \
    }

#define RELAX_LOAD_TILE(k0_) RELAX_LOAD_TILE_R(k0_, ra0, rb0)
#define RELAX_STORE_TILE(buf_) RELAX_STORE_TILE_R(buf_, ra0, rb0)

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * LDK + 4 * (lane >> 5);
#define RELAX_COMPUTE(cur_)  \
    {  \
        const float* As = smem + (cur_) * STAGE + (wm * TM * 32) * LDK + frag_off;  \
        const float* Bs = smem + (cur_) * STAGE + BM * LDK + (wn * TN * 32) * LDK + frag_off;  \
        if constexpr (PREC == 0) {  \
_Pragma("unroll")  \
            for (int q = 0; q < BK / 8; ++q) {  \
                f32x4 af[TM], bf[TN];  \
_Pragma("unroll")  \
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDK + q * 8);  \
_Pragma("unroll")  \
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDK + q * 8);  \
_Pragma("unroll")  \
                for (int i = 0; i < TM; ++i)  \
_Pragma("unroll")  \
                    for (int j = 0; j < TN; ++j) {  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);  \
                    }  \
            }  \
        } else {  \
  \
            const char* Ab = reinterpret_cast<const char*>(smem + (cur_) * STAGE + (wm * TM * 32 + (lane & 31)) * LDK) + 16 * (lane >> 5);  \
            const char* Bb = reinterpret_cast<const char*>(smem + (cur_) * STAGE + BM * LDK + (wn * TN * 32 + (lane & 31)) * LDK) + 16 * (lane >> 5);  \
_Pragma("unroll")  \
            for (int ks = 0; ks < BK / 16; ++ks) {  \
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];  \
_Pragma("unroll")  \
                for (int i = 0; i < TM; ++i) {  \
                    ah[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LDK * 4 + ks * 32);  \
                    al[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LDK * 4 + BK * 2 + ks * 32);  \
                }  \
_Pragma("unroll")  \
                for (int j = 0; j < TN; ++j) {  \
                    bh[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * LDK * 4 + ks * 32);  \
                    bl[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * LDK * 4 + BK * 2 + ks * 32);  \
                }  \
_Pragma("unroll")  \
                for (int t = 0; t < 3; ++t)  \
_Pragma("unroll")  \
                for (int i = 0; i < TM; ++i)  \
_Pragma("unroll")  \
                    for (int j = 0; j < TN; ++j)  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 0 ? al[i] : ah[i], t == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);  \
            }  \
        }  \
    }

    // bf16x3 steady-state step: MFMAs on LDS buffer cur_, with the staged registers (ra, rb) of the NEXT tile split and
    // stored to buffer nxt_ piece by piece between the accumulator blocks
#define RELAX_COMPUTE_STORE_X3(cur_, nxt_, ra, rb)  \
    {  \
        constexpr int PIECES = A_LOADS + B_LOADS;  \
        constexpr int BLOCKS = 3 * (BK / 16) * TM * TN;  /* MFMA slots of the step */  \
        const char* Ab = reinterpret_cast<const char*>(smem + (cur_) * STAGE + (wm * TM * 32 + (lane & 31)) * LDK) + 16 * (lane >> 5);  \
        const char* Bb = reinterpret_cast<const char*>(smem + (cur_) * STAGE + BM * LDK + (wn * TN * 32 + (lane & 31)) * LDK) + 16 * (lane >> 5);  \
        uint2* A8_ = reinterpret_cast<uint2*>(smem + (nxt_) * STAGE) + lrow * (LDK / 2) + c4;  \
        uint2* B8_ = A8_ + BM * (LDK / 2);  \
_Pragma("unroll")  \
        for (int ks = 0; ks < BK / 16; ++ks) {  \
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];  \
_Pragma("unroll")  \
            for (int i = 0; i < TM; ++i) {  \
                ah[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LDK * 4 + ks * 32);  \
                al[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LDK * 4 + BK * 2 + ks * 32);  \
            }  \
_Pragma("unroll")  \
            for (int j = 0; j < TN; ++j) {  \
                bh[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * LDK * 4 + ks * 32);  \
                bl[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * LDK * 4 + BK * 2 + ks * 32);  \
            }  \
            /* product type outermost: 8 independent accumulators between two MFMAs on the same one; per accumulator the  \
               order lo*hi, hi*lo, hi*hi is unchanged */  \
_Pragma("unroll")  \
            for (int t = 0; t < 3; ++t)  \
_Pragma("unroll")  \
            for (int i = 0; i < TM; ++i)  \
_Pragma("unroll")  \
                for (int j = 0; j < TN; ++j) {  \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 0 ? al[i] : ah[i], t == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);  \
                    const int g_ = ((ks * 3 + t) * TM + i) * TN + j;  \
_Pragma("unroll")  \
                    for (int pc = 0; pc < PIECES; ++pc)  \
                        if (pc * BLOCKS >= g_ * PIECES && pc * BLOCKS < (g_ + 1) * PIECES) {  \
                            uint2 hi_, lo_;  \
                            const f32x4 src_ = pc < A_LOADS ? ra[pc < A_LOADS ? pc : 0] : rb[pc >= A_LOADS ? pc - A_LOADS : 0];  \
                            split_bf16x4(src_, &hi_, &lo_);  \
                            uint2* dst_ = pc < A_LOADS ? A8_ + pc * PASS * (LDK / 2) : B8_ + (pc - A_LOADS) * PASS * (LDK / 2);  \
                            dst_[0] = hi_;  \
                            dst_[BK / 4] = lo_;  \
                        }  \
                }  \
        }  \
    }

    if constexpr (PREC == 0) {
        RELAX_LOAD_TILE(kt_begin * BK);
        RELAX_STORE_TILE(0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int cur = (kt - kt_begin) & 1;
            if (kt + 1 < kt_end) RELAX_LOAD_TILE((kt + 1) * BK);  // global loads in flight under the MFMAs below
            RELAX_COMPUTE(cur);
            if (kt + 1 < kt_end) RELAX_STORE_TILE(cur ^ 1);
            __syncthreads();
        }
    } else {
        // bf16x3: a K step is only 24 MFMAs (768 cycles) per wave, too short to cover a global load, so the register
        // stage runs TWO steps ahead (two named register sets, loop unrolled by two so every index is static).
        f32x4 ra1[A_LOADS], rb1[B_LOADS];
        RELAX_LOAD_TILE_R(kt_begin * BK, ra0, rb0);
        RELAX_STORE_TILE_R(0, ra0, rb0);
        if (kt_begin + 1 < kt_end) RELAX_LOAD_TILE_R((kt_begin + 1) * BK, ra1, rb1);
        __syncthreads();
        int kt = kt_begin;
        // steady state, branch-free: the split + LDS stores of tile k+1 are emitted between the MFMA blocks of tile k
        // so that one basic block holds both and the VALU work runs under the matrix pipe
        for (; kt + 3 < kt_end; kt += 2) {
            RELAX_LOAD_TILE_R((kt + 2) * BK, ra0, rb0);
            RELAX_COMPUTE_STORE_X3(0, 1, ra1, rb1);
            __syncthreads();
            RELAX_LOAD_TILE_R((kt + 3) * BK, ra1, rb1);
            RELAX_COMPUTE_STORE_X3(1, 0, ra0, rb0);
            __syncthreads();
        }
        for (; kt < kt_end; kt += 2) {
            if (kt + 2 < kt_end) RELAX_LOAD_TILE_R((kt + 2) * BK, ra0, rb0);
            RELAX_COMPUTE(0);
            if (kt + 1 < kt_end) RELAX_STORE_TILE_R(1, ra1, rb1);
            __syncthreads();
            if (kt + 1 < kt_end) {
                if (kt + 3 < kt_end) RELAX_LOAD_TILE_R((kt + 3) * BK, ra1, rb1);
                RELAX_COMPUTE(1);
                if (kt + 2 < kt_end) RELAX_STORE_TILE_R(0, ra0, rb0);
                __syncthreads();
            }
        }
    }
#undef RELAX_COMPUTE
#undef RELAX_COMPUTE_STORE_X3

    // ---- epilogue, staged through LDS -----------------------------------------------------------------------
    // The accumulators hold one column per lane and rows in registers (C/D map of the 32x32 MFMA:
    // col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); stored directly that is 64 dword stores per
    // lane.  Instead each 64-row chunk of the tile goes through the (now free) staging LDS and leaves as whole
    // 512-byte row segments: 16-byte loads/stores, 4x fewer memory instructions, bias/residual/activation fused.
    constexpr int LDC = BN + 4;
    constexpr int EP_ROWS = BM < 64 ? BM : (64 * LDC <= 2 * STAGE ? 64 : 32);
    constexpr int C4 = BN / 4;                       // float4 per tile row
    constexpr int EP_ITERS = EP_ROWS * C4 / NT;
    static_assert(EP_ROWS * LDC <= 2 * STAGE, "epilogue chunk must fit the staging LDS");
    static_assert((EP_ROWS * C4) % NT == 0, "epilogue chunk must divide over the workgroup");
    const int half = lane >> 5;
    float* outp;
    int64_t ldo;
    if (slice >= 0) {   // K slice of a split tile: raw partial sums [BM][BN]; splitk_finish applies the epilogue
        outp = p.partial + ((int64_t)split_tile * p.nsplit + slice) * (BM * BN);
        ldo = BN;
    } else {
        outp = p.out + (int64_t)m0 * p.N + n0;
        ldo = p.N;
    }
    // a thread keeps its column group through every iteration and pass (NT is a multiple of the float4s per row), so its
    // bias is ONE load, issued here, ahead of the first barrier (8 dependent L2 round trips per pass before: 6 us each)
    static_assert(NT % C4 == 0, "a thread must keep its column group");
    constexpr int EP_STEP = NT / C4;                 // tile rows covered per iteration
    const int lc = (tid % C4) * 4;
    const int lr0 = tid / C4;
    f32x4 bias4 = zero4;
    if (slice < 0 && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + lc);
    const float act_floor = p.act == 1 ? 0.f : -INFINITY;   // ReLU / identity as one v_max
    const bool interior = m0 + BM <= p.M;  // workgroup-uniform: interior tiles skip the per-row guards
#pragma unroll
    for (int pass = 0; pass < BM / EP_ROWS; ++pass) {
                if (pass > 0) __syncthreads();
                f32x4 rv[EP_ITERS];
        if (slice < 0 && interior) {
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it)   // residual rows of this pass: in flight under the LDS staging below
                rv[it] = p.residual ? *reinterpret_cast<const f32x4*>(p.residual + (int64_t)(m0 + pass * EP_ROWS + it * EP_STEP + lr0) * p.N + n0 + lc)
                                    : zero4;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb = (wm * TM + i) * 32;  // first tile row of this accumulator block (wave-uniform)
            if (rb / EP_ROWS == pass) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        smem[(rb % EP_ROWS + 4 * half + (r & 3) + 8 * (r >> 2)) * LDC + (wn * TN + j) * 32 + (lane & 31)] =
                            acc[i][j][r];
            }
        }
        __syncthreads();
                if (slice >= 0) {
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it) {
                const int lr = it * EP_STEP + lr0;
                *reinterpret_cast<f32x4*>(outp + (int64_t)(pass * EP_ROWS + lr) * ldo + lc) =
                    *reinterpret_cast<const f32x4*>(smem + lr * LDC + lc);
            }
        } else if (interior) {
            f32x4 v[EP_ITERS];
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it) {
                v[it] = *reinterpret_cast<const f32x4*>(smem + (it * EP_STEP + lr0) * LDC + lc);
                v[it] += bias4;
                v[it] += rv[it];  // (acc + bias) + residual: the same order on every path, so bits do not depend on the tile
            }
            if (p.act == 2) {
#pragma unroll
                for (int it = 0; it < EP_ITERS; ++it) {
                    f32x4 w = {apply_act(v[it].x, 2), apply_act(v[it].y, 2), apply_act(v[it].z, 2), apply_act(v[it].w, 2)};
                    *reinterpret_cast<f32x4*>(outp + (int64_t)(pass * EP_ROWS + it * EP_STEP + lr0) * ldo + lc) = w;
                }
            } else {
#pragma unroll
                for (int it = 0; it < EP_ITERS; ++it) {
                    f32x4 w = {fmaxf(v[it].x, act_floor), fmaxf(v[it].y, act_floor), fmaxf(v[it].z, act_floor), fmaxf(v[it].w, act_floor)};
                    *reinterpret_cast<f32x4*>(outp + (int64_t)(pass * EP_ROWS + it * EP_STEP + lr0) * ldo + lc) = w;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it) {
                const int lr = it * EP_STEP + lr0;
                const int trow = pass * EP_ROWS + lr;
                if (m0 + trow < p.M) {
                    const int64_t o = (int64_t)trow * ldo + lc;
                    f32x4 v = *reinterpret_cast<const f32x4*>(smem + lr * LDC + lc);
                    v += bias4;
                    if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (int64_t)m0 * p.N + n0 + o);
                    f32x4 w = {apply_act(v.x, p.act), apply_act(v.y, p.act), apply_act(v.z, p.act), apply_act(v.w, p.act)};
                    *reinterpret_cast<f32x4*>(outp + o) = w;
                }
            }
        }
    }
}

#undef RELAX_LOAD_TILE
#undef RELAX_STORE_TILE
#undef RELAX_LOAD_TILE_R
#undef RELAX_STORE_TILE_R

// Sums the K slices of the split tiles in slice order (deterministic) and applies the epilogue.
template <int BM, int BN>
__global__ __launch_bounds__(256) void splitk_finish(const GemmParams p) {
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
    const int e4 = blockIdx.x * 256 + threadIdx.x;  // float4 index inside the tile
    if (e4 >= BM * BN / 4) return;
    const int lr = e4 / (BN / 4), lc = (e4 % (BN / 4)) * 4;
    const int row = m0 + lr;
    if (row >= p.M) return;
    const float* pt = p.partial + (int64_t)split_tile * p.nsplit * (BM * BN) + lr * BN + lc;
    f32x4 s = *reinterpret_cast<const f32x4*>(pt);
    for (int k = 1; k < p.nsplit; ++k) s += *reinterpret_cast<const f32x4*>(pt + (int64_t)k * (BM * BN));
    const int64_t o = (int64_t)row * p.N + n0 + lc;
    if (p.bias) s += *reinterpret_cast<const f32x4*>(p.bias + n0 + lc);
    if (p.residual) s += *reinterpret_cast<const f32x4*>(p.residual + o);
    f32x4 v = {apply_act(s.x, p.act), apply_act(s.y, p.act), apply_act(s.z, p.act), apply_act(s.w, p.act)};
    *reinterpret_cast<f32x4*>(p.out + o) = v;
}

template <int BM, int BN, int WM, int WN, int BK, int OCC, bool TAPS, int PREC>
static int launch_variant(relax_handle* h, GemmParams& p, int blocks_per_cu, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    p.tiles_n = p.N / BN;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.ntiles = p.tiles_m * p.tiles_n;
    p.group_m = h->gemm.group_m;
    p.full_tiles = p.ntiles;
    p.nsplit = 1;
    p.partial = nullptr;
    // Tail split-K: the last partial round of tiles would leave most CUs idle; cut those tiles along K so
    // that the tail fills the chip once with short work units (partials summed in order by splitk_finish).  Cost model:
    // host_logic.cpp (shared with gemm_x6.hip).
    const host::TailSplit ts = host::choose_tail_split(p.ntiles, blocks_per_cu * 256, p.Kpad / BK, 4, h->gemm.split_k != 0);
    if (ts.nsplit > 1) {
        const size_t need = sizeof(float) * (size_t)(p.ntiles - ts.full_tiles) * ts.nsplit * BM * BN;
        RELAX_TRY(ensure_buf(h, h->splitk_ws, need < (size_t)(64 << 20) ? (size_t)(64 << 20) : need));
        p.partial = static_cast<float*>(h->splitk_ws.p);
        p.full_tiles = ts.full_tiles;
        p.nsplit = ts.nsplit;
    }
    constexpr size_t lds = sizeof(float) * 2 * (BM + BN) * (BK + 4);
    static bool attr_set[kMaxDevices] = {};   // per (kernel instantiation, device): the attribute lives on the device's code object
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_f32<BM, BN, WM, WN, BK, OCC, TAPS, PREC>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[h->device] = true;
    }
    const int units = p.full_tiles + (p.ntiles - p.full_tiles) * p.nsplit;
    hipLaunchKernelGGL((conv_gemm_f32<BM, BN, WM, WN, BK, OCC, TAPS, PREC>), dim3(units), dim3(NT), lds, s, p);
    if (p.nsplit > 1)
        hipLaunchKernelGGL((splitk_finish<BM, BN>), dim3(BM * BN / 4 / 256, p.ntiles - p.full_tiles), dim3(256), 0, s, p);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// tile variants: id -> <BM, BN, WM, WN, BK>, workgroups resident per CU (LDS / VGPR bound).  7 and 10 are the defaults; 1, 4, 5
// are the alternatives they were measured against (kept selectable: relax_set_option "gemm_variant", tools/gemm_bench.py).
// Tried and dropped: 8-wave 128x128 splits, 64x64 tiles, a register cap for 4 workgroups per CU (spilled).
#define RELAX_DISPATCH(BM_, BN_, WM_, WN_, BK_, OCC_, BPC_)                                  \
    (taps ? launch_variant<BM_, BN_, WM_, WN_, BK_, OCC_, true, 0>(h, p, BPC_, s)            \
          : launch_variant<BM_, BN_, WM_, WN_, BK_, OCC_, false, 0>(h, p, BPC_, s))
#define RELAX_DISPATCH_X3(BM_, BN_, WM_, WN_, OCC_, BPC_)                                    \
    (taps ? launch_variant<BM_, BN_, WM_, WN_, 32, OCC_, true, 1>(h, p, BPC_, s)             \
          : launch_variant<BM_, BN_, WM_, WN_, 32, OCC_, false, 1>(h, p, BPC_, s))

int launch_conv(relax_handle* h, const ConvDesc& d, hipStream_t s) {
    GemmParams p{};
    p.in = d.in; p.w = d.w; p.bias = d.bias; p.residual = d.residual; p.out = d.out;
    p.M = d.Nimg * d.Ho * d.Wo;
    p.N = d.Cout;
    p.Kpad = d.Kpad;
    p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Ho = d.Ho; p.Wo = d.Wo;
    p.KW = d.KW; p.stride = d.stride; p.pad = d.pad; p.ntaps = d.KH * d.KW;
    p.act = d.act;
    const bool taps = d.KH * d.KW > 1;
    RELAX_REQUIRE(h, p.M > 0 && p.N > 0 && p.Kpad > 0, "conv/gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.Kpad);
    RELAX_REQUIRE(h, p.Kpad % 32 == 0, "conv/gemm: K=%d must be a multiple of 32", p.Kpad);
    RELAX_REQUIRE(h, p.N % 64 == 0, "conv/gemm: N=%d must be a multiple of 64", p.N);
    RELAX_REQUIRE(h, p.Cin % 4 == 0, "conv/gemm: Cin=%d must be a multiple of 4", p.Cin);
    RELAX_REQUIRE(h, (reinterpret_cast<uintptr_t>(d.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(d.out) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(d.residual) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(d.bias) & 15) == 0,
                  "conv/gemm: operands must be 16-byte aligned");
    if (taps) {
        p.cin_log2 = ilog2_exact(d.Cin);
        RELAX_REQUIRE(h, p.cin_log2 >= 2, "conv: Cin=%d must be a power of two >= 4 for KHxKW > 1", d.Cin);
        RELAX_REQUIRE(h, p.Kpad >= d.KH * d.KW * d.Cin, "conv: Kpad too small");
    } else {
        RELAX_REQUIRE(h, d.pad == 0, "conv: 1x1 with padding is not supported");
        RELAX_REQUIRE(h, p.Kpad == d.Cin, "conv/gemm: 1x1 needs Kpad == Cin");
    }
    const double flops = d.flops > 0 ? d.flops : 2.0 * p.M * (double)p.N * (double)(d.KH * d.KW * d.Cin);
    int span;
    // algorithmic HBM bytes: activation in, weights, output (+ residual), each touched once
    const double bytes = 4.0 * ((double)d.Nimg * d.H * d.W * d.Cin + (double)p.N * p.Kpad + (double)p.M * p.N * (d.residual ? 2.0 : 1.0));
    RELAX_TRY(prof_begin(h, s, 0, flops, &span, bytes));
    // automatic choice: BK = 16 keeps three 128x128 workgroups (12 waves) per CU, which hides the prologue / epilogue of
    // one tile under the MFMAs of the others (+8 % over BK = 32 with two workgroups, measured on config 3)
    int variant = h->gemm.variant >= 0 ? h->gemm.variant : 7;
    if (p.N % 128 != 0) variant = h->gemm.variant_n64 >= 0 ? h->gemm.variant_n64 : 10;
    int rc;
    if (h->gemm.precision == 1) {   // opt-in bf16x3
        // plain GEMMs with enough rows take a 256x256 tile on 8 waves (128x64 per wave, one workgroup per CU): half the
        // L2 -> LDS bytes per MFMA of the 128x128 tile, +15 % on the ViT shapes (tools/gemm_bench.py --precision bf16x3)
        const bool big = !taps && p.N % 256 == 0 && (int64_t)((p.M + 255) / 256) * (p.N / 256) >= 256 && h->gemm.variant != 1;
        if (big) rc = launch_variant<256, 256, 2, 4, 32, 1, false, 1>(h, p, 1, s);
        else rc = (p.N % 128 == 0) ? RELAX_DISPATCH_X3(128, 128, 2, 2, 2, 2) : RELAX_DISPATCH_X3(128, 64, 2, 2, 2, 2);
        if (rc != RELAX_OK) { prof_abort(h, span); return rc; }
        RELAX_TRY(prof_end(h, s, span));
        return RELAX_OK;
    }
    switch (variant) {
        case 1: rc = RELAX_DISPATCH(128, 128, 2, 2, 32, 1, 2); break;   // 4 waves, 64x64 per wave
        case 4: rc = RELAX_DISPATCH(256, 128, 4, 2, 32, 1, 1); break;   // 8 waves, 64x64 per wave, 1 workgroup / CU
        case 5: rc = RELAX_DISPATCH(128, 64, 2, 2, 32, 1, 2); break;   // 4 waves, 64x32 per wave
        case 7: rc = RELAX_DISPATCH(128, 128, 2, 2, 16, 1, 3); break;   // BK 16: 40 KB LDS, 101 + 64 registers -> 3 workgroups / CU
        case 10: rc = RELAX_DISPATCH(128, 64, 2, 2, 16, 1, 4); break;   // N = 64 layers, BK 16   // 4 waves, 32x32 per wave
        default:
            set_error(h, "conv/gemm: unknown tile variant %d", variant);
            rc = RELAX_ERR_INVALID;
    }
    if (rc != RELAX_OK) { prof_abort(h, span); return rc; }   // no half-recorded span is left behind for relax_profile_read
    RELAX_TRY(prof_end(h, s, span));
    return RELAX_OK;
}
#undef RELAX_DISPATCH
#undef RELAX_DISPATCH_X3

}  // namespace relax

using namespace relax;

extern "C" {

int relax_op_gemm(relax_handle* h, const float* A, const float* W, const float* bias, const float* residual,
                  float* out, int M, int N, int K, int act, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, A && W && out, "relax_op_gemm: NULL operand");
    RELAX_REQUIRE(h, act >= 0 && act <= 2, "relax_op_gemm: act=%d", act);
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->gemm.precision == 3 && N % 256 == 0 && K % 16 == 0) {
        // operator-level entry under "f16x2": both operands are converted to two fp16 planes here, each ROW scaled by the power of two
        // that puts its own maximum into [2^14, 2^15) (no static bound on an arbitrary A: csrc/h2.h); the inverse scales go to the
        // epilogue as rowscale / colscale.  (The model drivers keep weights and activations in the format, with static scales.)
        const size_t a_bytes = (size_t)M * K * 4, w_bytes = (size_t)N * K * 4;
        const size_t a_al = (a_bytes + 255) & ~(size_t)255, w_al = (w_bytes + 255) & ~(size_t)255, m_al = ((size_t)M * 4 + 255) & ~(size_t)255;
        RELAX_TRY(ensure_buf(h, h->sp3_ws, a_al + w_al + m_al + (size_t)N * 4 + 256));
        char* As = static_cast<char*>(h->sp3_ws.p);
        char* Ws = As + a_al;
        float* rs = reinterpret_cast<float*>(Ws + w_al);
        float* cs = reinterpret_cast<float*>(Ws + w_al + m_al);
        RELAX_TRY(launch_to_h2_rows(h, A, K, As, M, K, rs, s));
        RELAX_TRY(launch_to_h2_rows(h, W, K, Ws, N, K, cs, s));
        GemmDescH2 d{};
        d.a = As; d.w = Ws; d.colscale = cs; d.rowscale = rs; d.bias = bias; d.residual = residual; d.out = out;
        d.M = M; d.N = N; d.K = K; d.act = act;
        return launch_gemm_h2(h, d, s);
    }
    if (h->gemm.precision >= 2) {
        // operator-level entry under "bf16x6": both operands are converted to split planes here (the model drivers
        // keep weights and activations in that format instead)
        RELAX_REQUIRE(h, K % 16 == 0, "relax_op_gemm (bf16x6): K=%d must be a multiple of 16", K);
        // (64 / 128-column problems: the kernel takes the fp32 rows of A as they are and splits them in its K loop)
        const bool a_f32 = h->gemm.fp32_rows && N % 256 != 0;
        const size_t a_bytes = a_f32 ? 0 : (size_t)M * K * 6, w_bytes = (size_t)N * K * 6;
        RELAX_TRY(ensure_buf(h, h->sp3_ws, a_bytes + w_bytes + 256));
        char* As = static_cast<char*>(h->sp3_ws.p);
        char* Ws = As + ((a_bytes + 255) & ~(size_t)255);
        if (!a_f32) RELAX_TRY(launch_to_sp3(h, A, K, As, M, K, s));
        RELAX_TRY(launch_to_sp3(h, W, K, Ws, N, K, s));
        ConvDescX6 d{};
        d.in = a_f32 ? static_cast<const void*>(A) : As; d.in_f32 = a_f32;
        d.Nimg = 1; d.H = 1; d.W = M; d.Cin = K; d.Ho = 1; d.Wo = M;
        d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
        d.w = Ws; d.Cout = N; d.bias = bias; d.residual = residual; d.out = out; d.act = act;
        return launch_conv_x6(h, d, s);
    }
    return launch_gemm(h, A, W, bias, residual, out, M, N, K, act, s);
}

int relax_op_conv2d_nhwc(relax_handle* h, const float* in, const float* w, const float* bias, const float* residual,
                         float* out, int Nimg, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                         int act, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, in && w && out, "relax_op_conv2d_nhwc: NULL operand");
    RELAX_REQUIRE(h, Nimg > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                  "relax_op_conv2d_nhwc: bad geometry");
    RELAX_REQUIRE(h, act >= 0 && act <= 2, "relax_op_conv2d_nhwc: act=%d", act);
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    ConvDesc d{};
    d.in = in; d.Nimg = Nimg; d.H = H; d.W = W; d.Cin = Cin;
    d.Ho = (H + 2 * pad - KH) / stride + 1;
    d.Wo = (W + 2 * pad - KW) / stride + 1;
    d.KH = KH; d.KW = KW; d.stride = stride; d.pad = pad;
    d.w = w; d.Cout = Cout;
    d.Kpad = ((KH * KW * Cin + 31) / 32) * 32;
    d.bias = bias; d.residual = residual; d.out = out; d.act = act;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // operator-level entry under "f16x2": the input becomes fp16 planes with one scale per IMAGE from its measured maximum, the weights with one per
    // output row; both conversions happen here (the model drivers keep the planes and derive the scales from bounds: gemm_h2.hip, resnet50.hip)
    const bool h2_wide = Cin % 32 == 0 && Cout % 256 == 0 && (KH * KW > 1 || pad == 0);                       // ResNet-50 layer3 / layer4: gemm_h3
    const bool h2_narrow = h->gemm.rn_h2_early && Cin % 16 == 0 && KH * KW > 1 && Cout % 64 == 0 && Cout % 256 != 0 && d.Kpad >= 256;   // the 3x3s of layer1 / layer2: gemm_x6<H2>
    if (h->gemm.precision == 3 && d.Kpad == KH * KW * Cin && KH * KW <= 32 && (h2_wide || h2_narrow)) {
        const size_t a_bytes = (size_t)Nimg * H * W * Cin * 4, w_bytes = (size_t)Cout * d.Kpad * 4;
        const size_t a_al = (a_bytes + 255) & ~(size_t)255, w_al = (w_bytes + 255) & ~(size_t)255, i_al = ((size_t)Nimg * 4 + 255) & ~(size_t)255;
        RELAX_TRY(ensure_buf(h, h->sp3_ws, a_al + w_al + 3 * i_al + (size_t)Cout * 4 + 256));
        char* As = static_cast<char*>(h->sp3_ws.p);
        char* Ws = As + a_al;
        unsigned* amax = reinterpret_cast<unsigned*>(Ws + w_al);
        float* sc = reinterpret_cast<float*>(Ws + w_al + i_al);
        float* inv = reinterpret_cast<float*>(Ws + w_al + 2 * i_al);
        float* cs = reinterpret_cast<float*>(Ws + w_al + 3 * i_al);
        RELAX_TRY(launch_image_absmax(h, in, (int64_t)H * W * Cin, Nimg, amax, s));
        RELAX_TRY(launch_h2_image_scales(h, amax, 1.f, nullptr, 0.f, nullptr, 0.f, sc, inv, Nimg, s));
        RELAX_TRY(launch_to_h2(h, in, Cin, As, (int64_t)Nimg * H * W, Cin, 1.f, sc, s, H * W));
        RELAX_TRY(launch_to_h2_rows(h, w, d.Kpad, Ws, Cout, d.Kpad, cs, s));
        if (h2_wide) {
            GemmDescH2 g{};
            g.a = As; g.w = Ws; g.colscale = cs; g.bias = bias; g.residual = residual; g.out = out;
            g.M = Nimg * d.Ho * d.Wo; g.N = Cout; g.K = d.Kpad; g.act = act;
            g.pixels = 1; g.Nimg = Nimg; g.H = H; g.W = W; g.Cin = Cin; g.Ho = d.Ho; g.Wo = d.Wo; g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
            g.rows_per_img = d.Ho * d.Wo; g.img_in_inv = inv;
            return launch_gemm_h2(h, g, s);
        }
        ConvDescX6 x{};
        x.in = As; x.in_h2 = 1; x.colscale = cs; x.img_in_inv = inv; x.Nimg = Nimg; x.H = H; x.W = W; x.Cin = Cin; x.Ho = d.Ho; x.Wo = d.Wo;
        x.KH = KH; x.KW = KW; x.stride = stride; x.pad = pad;
        x.w = Ws; x.Cout = Cout; x.bias = bias; x.residual = residual; x.out = out; x.act = act;
        return launch_conv_x6(h, x, s);
    }
    // bf16x6 where the split-plane kernel takes the geometry (16-channel chunks, at most 32 taps, 64-column tiles); anything
    // else - e.g. a 7x7 filter, Cin = 8 - runs on the exact-fp32 kernel as it did before bf16x6 became the default
    if (h->gemm.precision >= 2 && Cin % 16 == 0 && d.Kpad == KH * KW * Cin && KH * KW <= 32 && Cout % 64 == 0 &&
        (KH * KW > 1 || pad == 0)) {
        // operator-level entry under "bf16x6": input and weights are converted to split planes here
        // (1x1 stride-1 convolutions onto 64 / 128 channels: the kernel splits the fp32 pixels in its K loop)
        const bool a_f32 = h->gemm.fp32_rows && KH * KW == 1 && stride == 1 && Cout % 256 != 0;
        const size_t a_bytes = a_f32 ? 0 : (size_t)Nimg * H * W * Cin * 6, w_bytes = (size_t)Cout * d.Kpad * 6;
        RELAX_TRY(ensure_buf(h, h->sp3_ws, a_bytes + w_bytes + 256));
        char* As = static_cast<char*>(h->sp3_ws.p);
        char* Ws = As + ((a_bytes + 255) & ~(size_t)255);
        if (!a_f32) RELAX_TRY(launch_to_sp3(h, in, Cin, As, (int64_t)Nimg * H * W, Cin, s));
        RELAX_TRY(launch_to_sp3(h, w, d.Kpad, Ws, Cout, d.Kpad, s));
        ConvDescX6 x{};
        x.in = a_f32 ? static_cast<const void*>(in) : As; x.in_f32 = a_f32; x.Nimg = Nimg; x.H = H; x.W = W; x.Cin = Cin; x.Ho = d.Ho; x.Wo = d.Wo;
        x.KH = KH; x.KW = KW; x.stride = stride; x.pad = pad;
        x.w = Ws; x.Cout = Cout; x.bias = bias; x.residual = residual; x.out = out; x.out_sp3 = nullptr; x.act = act;
        return launch_conv_x6(h, x, s);
    }
    return launch_conv(h, d, s);
}

}  // extern "C"
