// Diagnostic variant of csrc/flow.hip (tools/build_ablations.sh flowstamps; reader: tools/flow_stamps.py): ticks per phase of a step of
// flow_iteration, one producer and one box wave of one block per launch.  The product translation unit only carries empty hooks; this
// file defines them and then #includes it, so the stamped kernel exists in the diagnostic library alone.
#include <hip/hip_runtime.h>
namespace relax {
__device__ unsigned long long g_flow_stamps[16];
}
#define IT_STAMP_DECL unsigned long long st_t_ = __builtin_amdgcn_s_memtime(), st_[6] = {0, 0, 0, 0, 0, 0}
#define IT_STAMP(i_)                                                         \
    {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                   \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();          \
        st_[i_] += n_ - st_t_;                                               \
        st_t_ = n_;                                                          \
        __builtin_amdgcn_sched_barrier(0);                                   \
    }
#define IT_STAMP_WAIT_LOADS   /* everything this step consumes has landed: the wait as one number (the product waits row by row) */ \
    {                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
        IT_STAMP(5);                                                         \
    }
#define IT_STAMP_FLUSH_PRODUCER                                                                                                        \
    if (threadIdx.x == 256 && blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 &&                                                 \
        (g_flow_stamps[15] == 0 || g_flow_stamps[15] == (unsigned long long)w)) {                                                     \
        atomicAdd(&g_flow_stamps[0], st_[0]);                                                                                          \
        atomicAdd(&g_flow_stamps[1], st_[1]);                                                                                          \
        atomicAdd(&g_flow_stamps[3], st_[5]);                                                                                          \
        atomicAdd(&g_flow_stamps[2], (unsigned long long)Q);                                                                           \
    }
#define IT_STAMP_FLUSH_BOX                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 &&                                                   \
        (g_flow_stamps[15] == 0 || g_flow_stamps[15] == (unsigned long long)w)) {                                                     \
        atomicAdd(&g_flow_stamps[4], st_[2]);                                                                                          \
        atomicAdd(&g_flow_stamps[5], st_[3]);                                                                                          \
        atomicAdd(&g_flow_stamps[6], st_[4]);                                                                                          \
        atomicAdd(&g_flow_stamps[7], (unsigned long long)(Q + 2));                                                                     \
    }
#include "../../relax-vqa_amd/csrc/flow.hip"

extern "C" int relax_debug_flow_stamps(unsigned long long* out16, int reset, int only_width) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(relax::g_flow_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        z[15] = (unsigned long long)only_width;   // 0: launches of every level
        if (hipMemcpyToSymbol(HIP_SYMBOL(relax::g_flow_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
