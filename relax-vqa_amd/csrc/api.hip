// Handle lifetime, error reporting, workspace and event profiling of librelax_hip.so.
#include <cstdarg>
#include <cstdlib>

#include "relax_internal.h"

namespace relax {

static thread_local std::string g_create_error;

void set_error(relax_handle* h, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->last_error = buf;
    else g_create_error = buf;
}

// "debug_poison" (relax_set_option / RELAX_DEBUG_POISON=1): every workspace request fills the whole buffer with 0xFF bytes (NaN as
// fp32, -1 as an index) first, so a kernel that reads workspace it did not write this call shows up in the parity tests instead of
// passing on whatever the previous call left there.  Synchronous and slow: a test mode, never on in the product path.
static int poison_buf(relax_handle* h, DevBuf& b) {
    if (!h->gemm.debug_poison || !b.p) return RELAX_OK;
    RELAX_HIP_CHECK(h, hipDeviceSynchronize());
    RELAX_HIP_CHECK(h, hipMemset(b.p, 0xFF, b.bytes));
    RELAX_HIP_CHECK(h, hipDeviceSynchronize());
    return RELAX_OK;
}

int ensure_buf(relax_handle* h, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return poison_buf(h, b);
    if (b.p) {
        RELAX_HIP_CHECK(h, hipDeviceSynchronize());
        RELAX_HIP_CHECK(h, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        set_error(h, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        b.p = nullptr;
        return RELAX_ERR_NOMEM;
    }
    b.bytes = bytes;
    return poison_buf(h, b);
}

int upload(relax_handle* h, const float* host, size_t n, float** dev, std::vector<void*>& allocs) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, n * sizeof(float));
    if (e != hipSuccess) {
        set_error(h, "hipMalloc(%zu floats) failed: %s", n, hipGetErrorString(e));
        return RELAX_ERR_NOMEM;
    }
    allocs.push_back(p);
    RELAX_HIP_CHECK(h, hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice));
    *dev = static_cast<float*>(p);
    return RELAX_OK;
}

// folds the finished spans at the front of the list into the totals (in order; stops at the first span that is still open or whose stop
// event has not completed) and gives their events back to the pool
static void prof_reap(relax_handle* h) {
    Profiler& p = h->prof;
    size_t n = 0;
    while (n < p.spans.size() && p.spans[n].ended && hipEventQuery(p.spans[n].stop) == hipSuccess) {
        ProfSpan& sp = p.spans[n];
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.start, sp.stop) == hipSuccess) {
            p.total_ms[sp.kind] += ms;
            p.total_work[sp.kind] += sp.work;
            p.total_bytes[sp.kind] += sp.bytes;
            p.launches[sp.kind] += 1;
        }
        p.pool.push_back(sp.start);
        p.pool.push_back(sp.stop);
        ++n;
    }
    (void)hipGetLastError();   // (hipErrorNotReady of the query that ended the loop is not an error of the caller)
    if (n) {
        p.spans.erase(p.spans.begin(), p.spans.begin() + (long)n);
        p.span_base += (int)n;
    }
}

int prof_begin(relax_handle* h, hipStream_t s, int kind, double work, int* span_idx, double bytes) {
    *span_idx = -1;
    Profiler& p = h->prof;
    if (!p.on) return RELAX_OK;
    if (p.spans.size() >= Profiler::kReapAt) prof_reap(h);
    ProfSpan sp;
    for (hipEvent_t* ev : {&sp.start, &sp.stop}) {
        if (!p.pool.empty()) {
            *ev = p.pool.back();
            p.pool.pop_back();
        } else {
            RELAX_HIP_CHECK(h, hipEventCreate(ev));
        }
    }
    sp.work = work;
    sp.bytes = bytes;
    sp.kind = kind;
    sp.ended = false;
    RELAX_HIP_CHECK(h, hipEventRecord(sp.start, s));
    RELAX_HIP_CHECK(h, hipEventRecord(sp.stop, s));     // re-recorded by prof_end; an error path that never gets there leaves an empty span, not an unrecorded event
    p.spans.push_back(sp);
    *span_idx = p.span_base + static_cast<int>(p.spans.size()) - 1;
    return RELAX_OK;
}

int prof_end(relax_handle* h, hipStream_t s, int span_idx) {
    if (span_idx < 0) return RELAX_OK;
    Profiler& p = h->prof;
    const int pos = span_idx - p.span_base;
    if (pos < 0 || pos >= static_cast<int>(p.spans.size())) return RELAX_OK;   // (the profiler was switched off and on in between)
    RELAX_HIP_CHECK(h, hipEventRecord(p.spans[pos].stop, s));
    p.spans[pos].ended = true;
    return RELAX_OK;
}

void prof_set_work(relax_handle* h, int span_idx, double work) {
    if (span_idx < 0) return;
    Profiler& p = h->prof;
    const int pos = span_idx - p.span_base;
    if (pos >= 0 && pos < static_cast<int>(p.spans.size())) p.spans[pos].work = work;
}

void prof_abort(relax_handle* h, int span_idx) {
    Profiler& p = h->prof;
    span_idx -= p.span_base;
    if (span_idx < 0 || span_idx != static_cast<int>(p.spans.size()) - 1) return;
    p.pool.push_back(p.spans.back().start);
    p.pool.push_back(p.spans.back().stop);
    p.spans.pop_back();
}

static int prof_drain(relax_handle* h) {
    Profiler& p = h->prof;
    for (ProfSpan& sp : p.spans) {
        RELAX_HIP_CHECK(h, hipEventSynchronize(sp.stop));
        float ms = 0.f;
        RELAX_HIP_CHECK(h, hipEventElapsedTime(&ms, sp.start, sp.stop));
        p.total_ms[sp.kind] += ms;
        p.total_work[sp.kind] += sp.work;
        p.total_bytes[sp.kind] += sp.bytes;
        p.launches[sp.kind] += 1;
        p.pool.push_back(sp.start);
        p.pool.push_back(sp.stop);
    }
    p.span_base += static_cast<int>(p.spans.size());
    p.spans.clear();
    return RELAX_OK;
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_abi_version(void) { return RELAX_ABI_VERSION; }

int relax_create(int device, relax_handle** out) {
    if (!out) {
        set_error(nullptr, "relax_create: out is NULL");
        return RELAX_ERR_INVALID;
    }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error(nullptr, "relax_create: no HIP device (%s)", hipGetErrorString(e));
        return RELAX_ERR_HIP;
    }
    if (device < 0 || device >= count || device >= kMaxDevices) {
        set_error(nullptr, "relax_create: device %d out of range (have %d)", device, count);
        return RELAX_ERR_INVALID;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        set_error(nullptr, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
        return RELAX_ERR_HIP;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        set_error(nullptr, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return RELAX_ERR_HIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(nullptr, "relax_create: device %d is %s; this library is built for gfx950 only", device,
                  prop.gcnArchName);
        return RELAX_ERR_HIP;
    }
    relax_handle* h = new relax_handle();
    h->device = device;
    if (const char* e = getenv("RELAX_GEMM_SPLIT")) h->gemm.split_k = atoi(e);
    if (const char* e = getenv("RELAX_GEMM_PRECISION")) h->gemm.precision = atoi(e) >= 0 && atoi(e) <= 3 ? atoi(e) : 0;
    if (const char* e = getenv("RELAX_H2_FORM")) h->gemm.h2_form = atoi(e) >= 0 && atoi(e) <= 2 ? atoi(e) : 1;
    if (const char* e = getenv("RELAX_GEMM_VARIANT")) h->gemm.variant = atoi(e);
    if (const char* e = getenv("RELAX_GEMM_VARIANT_N64")) h->gemm.variant_n64 = atoi(e);
    if (const char* e = getenv("RELAX_GEMM_GROUP_M")) h->gemm.group_m = atoi(e) > 0 ? atoi(e) : 1;
    if (const char* e = getenv("RELAX_DEBUG_POISON")) h->gemm.debug_poison = atoi(e) != 0;
    *out = h;
    return RELAX_OK;
}

int relax_destroy(relax_handle* h) {
    if (!h) return RELAX_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_resnet(h);
    free_vit(h);
    free_resize(h);
    free_head(h);
    if (h->head_ws.p) (void)hipFree(h->head_ws.p);
    if (h->flow_ws.p) (void)hipFree(h->flow_ws.p);
    if (h->arena.p) (void)hipFree(h->arena.p);
    if (h->scratch.p) (void)hipFree(h->scratch.p);
    if (h->splitk_ws.p) (void)hipFree(h->splitk_ws.p);
    if (h->sp3_ws.p) (void)hipFree(h->sp3_ws.p);
    for (auto& sp : h->prof.spans) {
        (void)hipEventDestroy(sp.start);
        (void)hipEventDestroy(sp.stop);
    }
    for (auto ev : h->prof.pool) (void)hipEventDestroy(ev);
    delete h;
    return RELAX_OK;
}

const char* relax_last_error(const relax_handle* h) {
    return h ? h->last_error.c_str() : g_create_error.c_str();
}

int relax_reserve(relax_handle* h, int max_images) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, max_images > 0, "relax_reserve: max_images must be > 0");
    size_t need = resnet_arena_bytes(max_images);
    if (h->vit.loaded) {
        size_t v = vit_arena_bytes(h->vit, max_images);
        if (v > need) need = v;
    }
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    RELAX_TRY(ensure_buf(h, h->arena, need));
    h->reserved_images = max_images;
    return RELAX_OK;
}

int relax_set_option(relax_handle* h, const char* key, int value) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, key, "relax_set_option: key is NULL");
    const std::string k(key);
    if (k == "gemm_split_k") h->gemm.split_k = value;
    else if (k == "gemm_precision") {
        RELAX_REQUIRE(h, value >= 0 && value <= 3, "relax_set_option: gemm_precision must be 0 (fp32), 1 (bf16x3), 2 (bf16x6) or 3 (f16x2)");
        h->gemm.precision = value;
    }
    else if (k == "gemm_variant") h->gemm.variant = value;
    else if (k == "gemm_variant_n64") h->gemm.variant_n64 = value;
    else if (k == "gemm_group_m") h->gemm.group_m = value > 0 ? value : 1;
    else if (k == "flow_max_pairs") h->gemm.flow_max_pairs = value > 0 ? value : 0;
    else if (k == "flow_fused") h->gemm.flow_fused = value != 0;
    else if (k == "flow_seg_rows") h->gemm.flow_seg_rows = value > 0 ? value : 0;
    else if (k == "flow_pyramid_fused") h->gemm.flow_pyramid_fused = value != 0;
    else if (k == "x6_fp32_rows") h->gemm.fp32_rows = value != 0;
    else if (k == "rn_h2") h->gemm.rn_h2 = value != 0;
    else if (k == "rn_h2_early") h->gemm.rn_h2_early = value != 0;
    else if (k == "rn_fuse") h->gemm.rn_fuse = value != 0;
    else if (k == "rn_c1_h2") h->gemm.rn_c1_h2 = value != 0;
    else if (k == "b2b_rows") {
        RELAX_REQUIRE(h, value == 128 || value == 256, "relax_set_option: b2b_rows must be 128 or 256");
        h->gemm.b2b_rows = value;
    }
    else if (k == "att_h2") h->gemm.att_h2 = value != 0;
    else if (k == "h2_form") {
        RELAX_REQUIRE(h, value >= 0 && value <= 2, "relax_set_option: h2_form must be 0, 1 or 2");
        h->gemm.h2_form = value;
    }
    else if (k == "h2_stages") {
        RELAX_REQUIRE(h, value == 3 || value == 4, "relax_set_option: h2_stages must be 3 or 4");
        h->gemm.h2_stages = value;
    }
    else if (k == "debug_poison") h->gemm.debug_poison = value != 0;
    else {
        set_error(h, "relax_set_option: unknown option '%s'", key);
        return RELAX_ERR_INVALID;
    }
    return RELAX_OK;
}

int relax_copy_bytes(relax_handle* h, const void* src, void* dst, int64_t n_bytes, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, src && dst && n_bytes >= 0, "relax_copy_bytes: bad arguments");
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    if (n_bytes > 0)
        RELAX_HIP_CHECK(h, hipMemcpyAsync(dst, src, (size_t)n_bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return RELAX_OK;
}

int relax_get_option(relax_handle* h, const char* key, int* value) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, key && value, "relax_get_option: NULL argument");
    const std::string k(key);
    if (k == "gemm_split_k") *value = h->gemm.split_k;
    else if (k == "gemm_precision") *value = h->gemm.precision;
    else if (k == "gemm_variant") *value = h->gemm.variant;
    else if (k == "gemm_variant_n64") *value = h->gemm.variant_n64;
    else if (k == "gemm_group_m") *value = h->gemm.group_m;
    else if (k == "flow_max_pairs") *value = h->gemm.flow_max_pairs;
    else if (k == "flow_fused") *value = h->gemm.flow_fused;
    else if (k == "flow_seg_rows") *value = h->gemm.flow_seg_rows;
    else if (k == "flow_pyramid_fused") *value = h->gemm.flow_pyramid_fused;
    else if (k == "x6_fp32_rows") *value = h->gemm.fp32_rows;
    else if (k == "rn_h2") *value = h->gemm.rn_h2;
    else if (k == "rn_h2_early") *value = h->gemm.rn_h2_early;
    else if (k == "rn_fuse") *value = h->gemm.rn_fuse;
    else if (k == "rn_c1_h2") *value = h->gemm.rn_c1_h2;
    else if (k == "b2b_rows") *value = h->gemm.b2b_rows;
    else if (k == "att_h2") *value = h->gemm.att_h2;
    else if (k == "h2_stages") *value = h->gemm.h2_stages;
    else if (k == "h2_form") *value = h->gemm.h2_form;
    else if (k == "debug_poison") *value = h->gemm.debug_poison;
    // read-only counters (leak checks of a long pass): HIP events this handle owns (pooled + in spans), bytes of its workspaces in MiB
    else if (k == "profile_events") *value = (int)(h->prof.pool.size() + 2 * h->prof.spans.size());
    else if (k == "workspace_mib")
        *value = (int)((h->arena.bytes + h->scratch.bytes + h->splitk_ws.bytes + h->sp3_ws.bytes + h->resize_ws.bytes + h->flow_ws.bytes + h->head_ws.bytes) >> 20);
    else {
        set_error(h, "relax_get_option: unknown option '%s'", key);
        return RELAX_ERR_INVALID;
    }
    return RELAX_OK;
}

int relax_profile_enable(relax_handle* h, int on) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_TRY(prof_drain(h));
    h->prof.on = on != 0;
    if (on) {
        for (int k = 0; k < Profiler::kKinds; ++k) {
            h->prof.total_ms[k] = 0;
            h->prof.total_work[k] = 0;
            h->prof.total_bytes[k] = 0;
            h->prof.launches[k] = 0;
        }
    }
    return RELAX_OK;
}

int relax_profile_read(relax_handle* h, int kind, double* total_ms, double* total_work, int64_t* launches) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, kind >= 0 && kind <= 10, "relax_profile_read: kind must be 0..10");
    RELAX_TRY(prof_drain(h));
    // read kind -> (span kind, which total): 2 and 4 return the algorithmic HBM bytes of the contraction launches
    // (7 / 8: every f16x2 launch = span kinds 5 + 6; 9 / 10: the plain f16x2 GEMMs alone = span kind 6)
    static const int span_of[11] = {0, 1, 0, 2, 2, 3, 4, 5, 5, 6, 6};
    const int k = span_of[kind];
    const bool bytes = kind == 2 || kind == 4 || kind == 8 || kind == 10;
    const Profiler& p = h->prof;
    const bool both = kind == 7 || kind == 8;
    if (total_ms) *total_ms = p.total_ms[k] + (both ? p.total_ms[6] : 0.0);
    if (total_work) *total_work = bytes ? p.total_bytes[k] + (both ? p.total_bytes[6] : 0.0) : p.total_work[k] + (both ? p.total_work[6] : 0.0);
    if (launches) *launches = p.launches[k] + (both ? p.launches[6] : 0);
    return RELAX_OK;
}

}  // extern "C"
