// Diagnostic variant of csrc/gemm_x6.hip (tools/build_ablations.sh x6stamps): thread 0 of every workgroup records cycle stamps
// (prologue / K loop / epilogue), printed per launch after a stream sync (tools/x6_gaps.py reads the raw dump).  The product
// translation unit only carries empty hooks; this file defines them and then #includes it.
#include "../../relax-vqa_amd/csrc/relax_internal.h"
#include <cstdlib>
#include <vector>

#define X6_STAMP(i_) if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime()
#define X6_STAMP_IDS()                                                                                                       \
    if (p.stamps && threadIdx.x == 0) {                                                                                      \
        p.stamps[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    /* HW_REG_HW_ID */   \
        p.stamps[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   /* HW_REG_XCC_ID */  \
    }
#define X6_STAMPS_BEFORE_LAUNCH(h_, p_, units_)                                                          \
    RELAX_TRY(ensure_buf(h_, (h_)->scratch, sizeof(unsigned long long) * 8 * (size_t)(units_)));        \
    (p_).stamps = static_cast<unsigned long long*>((h_)->scratch.p)
#define X6_STAMPS_AFTER_LAUNCH(BM_, BN_, h_, p_, units_, s_) RELAX_TRY((x6_report_stamps<BM_, BN_>(h_, p_, units_, s_)))

namespace relax {
template <int BM, int BN, class Params>
static int x6_report_stamps(relax_handle* h, const Params& p, int units, hipStream_t s) {
    RELAX_HIP_CHECK(h, hipStreamSynchronize(s));
    std::vector<unsigned long long> hs(8 * (size_t)units);
    RELAX_HIP_CHECK(h, hipMemcpy(hs.data(), p.stamps, hs.size() * sizeof(hs[0]), hipMemcpyDeviceToHost));
    double d[3] = {0, 0, 0};
    for (int u = 0; u < p.full_tiles; ++u) {
        const unsigned long long* t = &hs[8 * (size_t)u];
        d[0] += (double)(t[1] - t[0]);
        d[1] += (double)(t[2] - t[1]);
        d[2] += (double)(t[3] - t[2]);
    }
    if (const char* dump = getenv("RELAX_X6_STAMP_DUMP")) {   // raw records of every launch, appended
        if (FILE* f = fopen(dump, "ab")) {
            const int hdr[8] = {p.M, p.N, p.K, BM, BN, units, p.full_tiles, 0};
            fwrite(hdr, sizeof(hdr), 1, f);
            fwrite(hs.data(), sizeof(hs[0]), hs.size(), f);
            fclose(f);
        }
    }
    const double n = p.full_tiles > 0 ? p.full_tiles : 1;
    fprintf(stderr, "x6 %dx%dx%d tile %dx%d act %d res %d sp3out %d: cycles per tile: prologue %.0f, K loop %.0f (%d steps, %.0f per step), "
            "epilogue %.0f\n", p.M, p.N, p.K, BM, BN, p.act, p.residual != nullptr, p.out_sp3 != nullptr, d[0] / n, d[1] / n,
            p.K / 16, d[1] / n / (p.K / 16), d[2] / n);
    return RELAX_OK;
}
}  // namespace relax
#include "../../relax-vqa_amd/csrc/gemm_x6.hip"
