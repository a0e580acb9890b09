// Diagnostic variant of csrc/gemm_h2.hip (tools/build_ablations.sh h2stamps): thread 0 of every workgroup records cycle stamps
// (prologue / K loop / epilogue), printed per launch after a stream sync.  The product translation unit only carries empty hooks.
#include "../../relax-vqa_amd/csrc/relax_internal.h"
#include <cstdlib>
#include <vector>

#define H2_STAMP(i_) if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime()
#define H2_STAMPS_BEFORE_LAUNCH(h_, p_, units_)                                                          \
    RELAX_TRY(ensure_buf(h_, (h_)->scratch, sizeof(unsigned long long) * 8 * (size_t)(units_)));        \
    (p_).stamps = static_cast<unsigned long long*>((h_)->scratch.p)
#define H2_STAMPS_AFTER_LAUNCH(h_, p_, units_, s_) RELAX_TRY((h2_report_stamps(h_, p_, units_, s_)))

#ifdef H3_B2B_BOUND   // (tools/build_ablations.sh h3b2b) WRONG results: the conv3 launches of layer3 / layer4 never fetch their A operand and the 3x3
#define H3_ABL_NO_A(taps_, perimg_, p_) ((perimg_) && !(taps_) && ((p_).residual != nullptr || (p_).residual_h2 != nullptr))   // launches never store
#define H3_ABL_NO_OUT_H2(taps_, p_) (taps_)                                                                                  // their planes
#endif
#ifdef H3_L2HIT   // (tools/build_ablations.sh h2l2hit) every K step of a plain gemm_h3 re-reads the first four: WRONG results, the K loop with
#define H3_KSTEP(k_) ((k_) & 3)   // every DMA piece served by L2 - the bound of what prefetching into L2 could return
#endif
namespace relax {
template <class Params>
static int h2_report_stamps(relax_handle* h, const Params& p, int units, hipStream_t s) {
    RELAX_HIP_CHECK(h, hipStreamSynchronize(s));
    std::vector<unsigned long long> hs(8 * (size_t)units);
    RELAX_HIP_CHECK(h, hipMemcpy(hs.data(), p.stamps, hs.size() * sizeof(hs[0]), hipMemcpyDeviceToHost));
    double d[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int u = 0; u < p.full_tiles; ++u) {
        const unsigned long long* t = &hs[8 * (size_t)u];
        d[0] += (double)(t[1] - t[0]);
        d[1] += (double)(t[2] - t[1]);
        d[2] += (double)(t[3] - t[2]);
        d[3] += (double)(t[4] - t[2]);   // first epilogue pass (gemm_h3): what the pass needs from memory requested,
        d[4] += (double)(t[5] - t[4]);   // accumulators staged + barrier,
        d[5] += (double)(t[6] - t[5]);   // rows finished
        d[6] += (double)(t[7] - t[2]);   // the barrier behind the K loop
    }
    const double n = p.full_tiles > 0 ? p.full_tiles : 1;
    fprintf(stderr, "h2 %dx%dx%d act %d res %d h2out %d: cycles per tile: prologue %.0f, K loop %.0f (%d steps, %.0f per step of 16 k), "
            "epilogue %.0f (barrier behind the loop %.0f; first pass: fetch issued %.0f after the loop, staged %.0f, rows %.0f)\n", p.M, p.N, p.K, p.act, p.residual != nullptr, p.out_h2 != nullptr,
            d[0] / n, d[1] / n, p.K / 16, d[1] / n / (p.K / 16), d[2] / n, d[6] / n, d[3] / n, d[4] / n, d[5] / n);
    return RELAX_OK;
}
}  // namespace relax
#include "../../relax-vqa_amd/csrc/gemm_h2.hip"
