// Diagnostic variants of the back-to-back kernel of csrc/gemm_x6.hip (tools/build_ablations.sh b2babl:<mask> | b2bpf:<mode> | x6stg2): this file
// defines the hooks and #includes the product translation unit, which carries their neutral defaults only.
//   B2B_ABL_MASK   timing only, WRONG results: bit 0 no fp32 stores of the conv3 output, bit 1 no residual loads, bit 2 no conv3 phase at all
//   B2B_PF_MODE    same bits: how conv3's B fragments are requested (0 per K chunk, 1 a whole pass ahead, 2 half a pass ahead)
//   X6_STG2        same bits: two LDS stages in the f16x2 3x3 loop instead of three
#ifdef B2B_ABL_MASK
#define X6_B2B_ABL B2B_ABL_MASK
#endif
#ifdef B2B_PF_MODE
#define X6_B2B_PREFETCH_B B2B_PF_MODE
#endif
#ifdef X6_STG2
#define X6_H2_STAGES 2
#endif
#include "../../relax-vqa_amd/csrc/gemm_x6.hip"
