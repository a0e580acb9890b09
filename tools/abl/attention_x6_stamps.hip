// Diagnostic variant of csrc/attention_x6.hip (tools/build_ablations.sh a6stamps; reader: tools/attn_stamps.py): ticks per phase of an
// (image, head) item, waves 0 and 6 of workgroup 3.  The averages go out in the first floats of the fp32 output (WRONG output, timing
// only), which is why none of this lives in the product translation unit: it only carries empty hooks.
#define A6_STAMP_DECL                                                  \
    unsigned long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};           \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();             \
    int items_done = 0
#define A6_STAMP(i_)                                                   \
    {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                             \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();    \
        ph[i_] += n_ - t_;                                             \
        t_ = n_;                                                       \
        __builtin_amdgcn_sched_barrier(0);                             \
    }
#define A6_STAMP_ITEM() ++items_done
#define A6_STAMP_FLUSH()                                                                                   \
    if (OUT_F32 && blockIdx.x == 3 && (tid == 0 || tid == 6 * 64)) {                                       \
        for (int i = 0; i < 9; ++i) out[(tid ? 16 : 0) + i] = (float)(ph[i] / items_done);                 \
    }
#include "../../relax-vqa_amd/csrc/attention_x6.hip"
