// Runtime primitives of the C ABI (section 1 of include/bullet_amd.h): the memory / stream
// half of the reference's GpuBindings trait (crates/gpu/src/runtime/bindings.rs:76-125,
// ROCm implementation crates/gpu/src/runtime/rocm.rs:66-171), straight on the HIP runtime.
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <utility>
#include <string>
#include <vector>

#include "common.h"

static thread_local std::string g_last_error;

void bt_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

namespace {
struct ProfRec {
    int id;
    hipEvent_t a, b;
};
struct Profiler {
    bool on = false;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    std::vector<ProfRec> recs;
    hipEvent_t pending = nullptr;
    unsigned long long filter = ~0ull;  // bit i set = kernel id i < 64 is timed (bt_profile_set_filter)
    std::vector<std::string> only;      // if not empty: only kernels whose name contains one of these (bt_profile_only)
    int every = 1;                      // time every n-th launch of a kernel (bt_profile_set_sampling)
    std::vector<unsigned> seen;
    std::vector<signed char> allowed;   // per kernel id: -1 unknown, 0 filtered out, 1 timed
};
thread_local Profiler g_prof;
// kernel names in registration order (a launch site registers once, on its first launch, from any thread)
std::mutex g_names_mutex;
std::vector<std::string> &kernel_names() {
    static std::vector<std::string> names;
    return names;
}
std::string kernel_name(int id) {
    std::lock_guard<std::mutex> lock(g_names_mutex);
    return (id >= 0 && id < (int)kernel_names().size()) ? kernel_names()[id] : std::string();
}
}  // namespace

int bt_prof_register(const char *kernel_expr) {
    // the spelling of BT_LAUNCH's kernel argument -> "name<args>" as rocprofv3 prints it: no outer parentheses,
    // ", " between template arguments
    std::string n;
    for (const char *c = kernel_expr; *c; c++)
        if (*c != ' ' && *c != '(' && *c != ')') n += *c, n += (*c == ',' ? " " : "");
    std::lock_guard<std::mutex> lock(g_names_mutex);
    auto &names = kernel_names();
    for (size_t i = 0; i < names.size(); i++)
        if (names[i] == n) return (int)i;
    names.push_back(n);
    return (int)names.size() - 1;
}

bool bt_prof_active() { return g_prof.on; }

bool bt_prof_enabled(int id) {
    Profiler &p = g_prof;
    if (!p.on) return false;
    if ((size_t)id >= p.seen.size()) p.seen.resize(id + 1, 0u), p.allowed.resize(id + 1, (signed char)-1);
    if (p.allowed[id] < 0) {
        bool ok = id >= 64 || ((p.filter >> id) & 1ull);
        if (ok && !p.only.empty()) {
            const std::string n = kernel_name(id);
            ok = false;
            for (const std::string &w : p.only) ok = ok || n.find(w) != std::string::npos;
        }
        p.allowed[id] = ok ? 1 : 0;
    }
    if (!p.allowed[id]) return false;
    return (p.seen[id]++ % (unsigned)p.every) == 0;
}

void bt_prof_record(int id, hipStream_t s, bool begin) {
    Profiler &p = g_prof;
    if (p.used >= p.pool.size()) {  // pool exhausted: stop recording rather than allocate mid-run
        if (begin) p.pending = nullptr;
        return;
    }
    hipEvent_t e = p.pool[p.used++];
    (void)hipEventRecord(e, s);
    if (begin) p.pending = e;
    else if (p.pending) {
        p.recs.push_back({id, p.pending, e});
        p.pending = nullptr;
    }
}

bool bt_prof_take_pair(int id, hipEvent_t *a, hipEvent_t *b) {
    Profiler &p = g_prof;
    if (p.used + 2 > p.pool.size()) return false;  // pool exhausted: stop recording rather than allocate mid-run
    *a = p.pool[p.used++], *b = p.pool[p.used++];
    p.recs.push_back({id, *a, *b});
    return true;
}

namespace {
thread_local int tl_stop_kernel = -1;
thread_local hipEvent_t tl_stop_event = nullptr;
}  // namespace
int bt_set_max_lds(const void *kernel, size_t bytes) {
    // (the attribute belongs to the kernel on the current device; a launch is a few microseconds of host time, and so is
    // this call: it is made when a kernel first needs more dynamic LDS than it has been granted on this device)
    static std::mutex m;
    static std::map<std::pair<int, const void *>, size_t> granted;
    int dev = 0;
    BT_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(m);
    size_t &g = granted[{dev, kernel}];
    if (bytes <= g) return BT_OK;
    BT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    g = bytes;
    return BT_OK;
}

namespace {
thread_local bool tl_capturing = false;
}
bool bt_capturing() { return tl_capturing; }
void bt_set_capturing(bool on) { tl_capturing = on; }
void bt_attach_stop_event(int kernel_id, hipEvent_t ev) { tl_stop_kernel = kernel_id, tl_stop_event = ev; }
hipEvent_t bt_take_stop_event(int kernel_id) {
    if (kernel_id != tl_stop_kernel || !tl_stop_event) return nullptr;
    hipEvent_t ev = tl_stop_event;
    tl_stop_kernel = -1, tl_stop_event = nullptr;
    return ev;
}

extern "C" {

int bt_profile_start(int max_launches) {
    BT_REQUIRE(max_launches > 0, "bt_profile_start: max_launches must be positive");
    Profiler &p = g_prof;
    while (p.pool.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        BT_HIP(hipEventCreate(&e));
        p.pool.push_back(e);
    }
    p.used = 0;
    p.recs.clear();
    p.pending = nullptr;
    p.filter = ~0ull;
    p.only.clear();
    p.every = 1;
    p.seen.clear(), p.allowed.clear();
    p.on = true;
    return BT_OK;
}

int bt_profile_set_sampling(int every) {
    BT_REQUIRE(every >= 1, "bt_profile_set_sampling: every must be >= 1");
    g_prof.every = every;
    return BT_OK;
}

int bt_profile_set_filter(unsigned long long kernel_mask) {
    g_prof.filter = kernel_mask;
    g_prof.allowed.assign(g_prof.allowed.size(), (signed char)-1);
    return BT_OK;
}

int bt_profile_only(const char *name_part) {
    BT_REQUIRE(name_part && name_part[0], "bt_profile_only: empty name");
    g_prof.only.push_back(name_part);
    g_prof.allowed.assign(g_prof.allowed.size(), (signed char)-1);
    return BT_OK;
}

int bt_profile_stop(int *counts, double *total_ms, int n_ids) {
    Profiler &p = g_prof;
    p.on = false;
    BT_REQUIRE(counts && total_ms && n_ids >= bt_profile_num_kernels(),
               "bt_profile_stop: need arrays of bt_profile_num_kernels() entries");
    for (int i = 0; i < n_ids; i++) counts[i] = 0, total_ms[i] = 0.0;
    for (const ProfRec &r : p.recs) {
        BT_HIP(hipEventSynchronize(r.b));
        float ms = 0.0f;
        BT_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        counts[r.id]++;
        total_ms[r.id] += ms;
    }
    p.recs.clear();
    p.used = 0;
    return BT_OK;
}

int bt_profile_num_kernels(void) {
    std::lock_guard<std::mutex> lock(g_names_mutex);
    return (int)kernel_names().size();
}

const char *bt_profile_kernel_name(int id) {  // (valid until the library is unloaded: names are never removed)
    std::lock_guard<std::mutex> lock(g_names_mutex);
    return (id >= 0 && id < (int)kernel_names().size()) ? kernel_names()[id].c_str() : "";
}

int bt_version(void) { return 1; }

const char *bt_last_error(void) { return g_last_error.c_str(); }

int bt_device_count(int *count) {
    BT_REQUIRE(count, "bt_device_count: null argument");
    BT_HIP(hipGetDeviceCount(count));
    return BT_OK;
}

int bt_device_set(int ordinal) {
    BT_HIP(hipSetDevice(ordinal));
    return BT_OK;
}

int bt_device_props(int ordinal, char *name, int name_len, char *arch, int arch_len, int *wave_size, int *num_cus) {
    hipDeviceProp_t p;
    BT_HIP(hipGetDeviceProperties(&p, ordinal));
    if (name && name_len > 0) {
        strncpy(name, p.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (wave_size) *wave_size = p.warpSize;
    if (num_cus) *num_cus = p.multiProcessorCount;
    return BT_OK;
}

int bt_device_sync(void) {
    BT_HIP(hipDeviceSynchronize());
    return BT_OK;
}

int bt_malloc(void **ptr, size_t bytes) {
    BT_REQUIRE(ptr, "bt_malloc: null argument");
    *ptr = nullptr;
    if (bytes == 0) return BT_OK;
    BT_HIP(hipMalloc(ptr, bytes));
    // Buffer::zeroed, buffer.rs:112-131. The memset runs on the NULL stream and returns before it has executed; the
    // library's own streams are non-blocking (no implicit order against the NULL stream), so without the wait a
    // first write on such a stream -- e.g. the 4-byte upload of 1/batch in bt_train_run -- could be overtaken by the
    // zero fill (seen as a replica training with adj = 0, about 1 run in 15 with two processes on one GPU). The
    // reference syncs the device around every malloc as well (crates/gpu/src/runtime.rs:102-151).
    hipError_t e = hipMemset(*ptr, 0, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
        bt_set_error("%s: %s", hipGetErrorName(e), hipGetErrorString(e));
        return BT_ERR_HIP;
    }
    return BT_OK;
}

int bt_free(void *ptr) {
    if (!ptr) return BT_OK;
    BT_HIP(hipFree(ptr));
    return BT_OK;
}

int bt_host_alloc_pinned(void **ptr, size_t bytes) {
    BT_REQUIRE(ptr, "bt_host_alloc_pinned: null argument");
    BT_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return BT_OK;
}

int bt_host_free_pinned(void *ptr) {
    if (!ptr) return BT_OK;
    BT_HIP(hipHostFree(ptr));
    return BT_OK;
}

int bt_stream_create(void **stream) {
    BT_REQUIRE(stream, "bt_stream_create: null argument");
    hipStream_t s;
    BT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return BT_OK;
}

int bt_stream_destroy(void *stream) {
    if (!stream) return BT_OK;
    BT_HIP(hipStreamDestroy(bt_stream(stream)));
    return BT_OK;
}

int bt_stream_sync(void *stream) {
    BT_HIP(hipStreamSynchronize(bt_stream(stream)));
    return BT_OK;
}

int bt_event_create(void **event) {
    BT_REQUIRE(event, "bt_event_create: null argument");
    hipEvent_t e;
    BT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return BT_OK;
}

int bt_event_destroy(void *event) {
    if (!event) return BT_OK;
    BT_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return BT_OK;
}

int bt_event_record(void *event, void *stream) {
    BT_REQUIRE(event, "bt_event_record: null event");
    BT_HIP(hipEventRecord(static_cast<hipEvent_t>(event), bt_stream(stream)));
    return BT_OK;
}

int bt_event_sync(void *event) {
    BT_REQUIRE(event, "bt_event_sync: null event");
    BT_HIP(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return BT_OK;
}

int bt_event_query(void *event, int *done) {
    BT_REQUIRE(event && done, "bt_event_query: null argument");
    const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
    if (e != hipSuccess && e != hipErrorNotReady) {
        bt_set_error("%s: %s", hipGetErrorName(e), hipGetErrorString(e));
        return BT_ERR_HIP;
    }
    *done = e == hipSuccess;
    return BT_OK;
}

int bt_stream_wait_event(void *stream, void *event) {
    BT_REQUIRE(event, "bt_stream_wait_event: null event");
    BT_HIP(hipStreamWaitEvent(bt_stream(stream), static_cast<hipEvent_t>(event), 0));
    return BT_OK;
}

int bt_memset_async(void *ptr, int byte_value, size_t bytes, void *stream) {
    if (bytes == 0) return BT_OK;
    BT_REQUIRE(ptr, "bt_memset_async: null pointer");
    BT_HIP(hipMemsetAsync(ptr, byte_value, bytes, bt_stream(stream)));
    return BT_OK;
}

int bt_memcpy_h2d_async(void *dst, const void *src_host, size_t bytes, void *stream) {
    if (bytes == 0) return BT_OK;
    BT_REQUIRE(dst && src_host, "bt_memcpy_h2d_async: null pointer");
    BT_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, bt_stream(stream)));
    return BT_OK;
}

int bt_memcpy_d2h_async(void *dst_host, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return BT_OK;
    BT_REQUIRE(dst_host && src, "bt_memcpy_d2h_async: null pointer");
    BT_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, bt_stream(stream)));
    return BT_OK;
}

int bt_memcpy_d2d_async(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return BT_OK;
    BT_REQUIRE(dst && src, "bt_memcpy_d2d_async: null pointer");
    BT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, bt_stream(stream)));
    return BT_OK;
}

}  // extern "C"
