// Library-level entry points: error reporting, model validation, device query, timers.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <chrono>
#include <mutex>
#include <unordered_map>

#include "obe_common.h"
#include "obe_models.h"

namespace obe {

// How long the host spins on the result words before it hands the wait to hipStreamSynchronize (microseconds;
// OBE_HOST_SPIN_US overrides).  While it spins it asks the stream every 50 us whether it has failed or drained
// (hipStreamQuery, ~1 us), so a kernel that faulted or never delivered is still reported.
static double host_spin_us() {
    static const double us = getenv("OBE_HOST_SPIN_US") ? atof(getenv("OBE_HOST_SPIN_US")) : kHostWaitSpinUs;
    return us;
}

int wait_host_words(const void* h_words, int64_t n, hipStream_t st) {
    const volatile uint64_t* p = static_cast<const volatile uint64_t*>(h_words);
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = host_spin_us();
    double next_query = 50.0;
    int64_t next = n - 1;            // words are checked from the last one down; `next` is the highest still armed
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            while (next >= 0 && p[next] != kHostSentinel) --next;
            if (next < 0) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return 0;
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us > limit) break;
        if (us > next_query) {
            next_query = us + 50.0;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) break;                          // drained: the words are there, or never will be
            if (q != hipErrorNotReady) return fail(q, "kernel failed while its host results were awaited");
        }
    }
    OBE_HIP_TRY(hipStreamSynchronize(st));       // long kernel, or one that never delivered: the stream knows
    return 0;
}

int wait_host_word(const void* h_word, hipStream_t st) { return wait_host_words(h_word, 1, st); }

// Arrival counters for "the last workgroup to finish folds the partials" (obe_common.h: arrive_last):
// one zeroed 1152-byte slot of device memory per (device, stream), allocated in one small block per device
// on first use and never freed.  Kernels on one stream never overlap, the counter wraps back to zero with
// the last arrival (atomicInc), so a slot is always zero between launches — no per-call memset, and nothing
// is asked of the caller's workspace.  nullptr (table full, allocation failed): callers fold in a launch
// of their own.
namespace {
constexpr int kControlSlotsMax = 256;      // (kControlSlotWords: obe_common.h, nine 128-byte lines per stream)
struct ControlTable {
    std::mutex mu;
    std::unordered_map<uint64_t, int> slot;            // (device << 56) ^ stream -> slot index on that device
    unsigned* base[64] = {};
    int used[64] = {};
    uint64_t key_of[64][kControlSlotsMax] = {};
    uint64_t last_use[64][kControlSlotsMax] = {};
    uint64_t tick = 0;
};
ControlTable& control_table() {
    static ControlTable* t = new ControlTable;          // (leaked on purpose: no destructor order games at exit)
    return *t;
}
// slots per device; OBE_CONTROL_SLOTS (0..256) is a test aid: 0 makes every lookup fail (the refusal paths of
// the entry points that need a counter), a small number exercises the hand-over of a slot
int control_slots() {
    static const int n = [] {
        const char* e = getenv("OBE_CONTROL_SLOTS");
        const int v = e ? atoi(e) : kControlSlotsMax;
        return v < 0 ? 0 : (v > kControlSlotsMax ? kControlSlotsMax : v);
    }();
    return n;
}
}  // namespace

unsigned* stream_control_words(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    const int n_slots = control_slots();
    if (n_slots == 0) return nullptr;
    ControlTable& t = control_table();
    const uint64_t key = ((uint64_t)dev << 56) ^ (uint64_t)reinterpret_cast<uintptr_t>(st);
    std::lock_guard<std::mutex> lock(t.mu);
    auto it = t.slot.find(key);
    if (it != t.slot.end()) {
        t.last_use[dev][it->second] = ++t.tick;
        return t.base[dev] + (size_t)it->second * kControlSlotWords;
    }
    if (!t.base[dev]) {
        void* p = nullptr;
        const size_t bytes = (size_t)kControlSlotsMax * kControlSlotWords * sizeof(unsigned);
        if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        t.base[dev] = static_cast<unsigned*>(p);
    }
    int s;
    if (t.used[dev] < n_slots) {
        s = t.used[dev]++;
    } else {
        // Table full (streams come and go: torch creates one per `torch.cuda.Stream()`): the least recently used
        // slot changes hands.  Its previous owner may be a stream that still exists, with a kernel in flight that
        // counts arrivals in it: the device is drained first (a counter is zero whenever no kernel is using it),
        // and if that stream comes back later it simply gets a slot the same way.  Rare by construction — one
        // device synchronisation per NEW stream beyond the first 256.
        s = 0;
        for (int k = 1; k < n_slots; ++k)
            if (t.last_use[dev][k] < t.last_use[dev][s]) s = k;
        if (hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        t.slot.erase(t.key_of[dev][s]);
    }
    t.key_of[dev][s] = key;
    t.last_use[dev][s] = ++t.tick;
    t.slot[key] = s;
    return t.base[dev] + (size_t)s * kControlSlotWords;
}

namespace {
struct SideTable {
    std::mutex mu;
    std::unordered_map<uint64_t, SideStream> of;
};
SideTable& side_table() {
    static SideTable* t = new SideTable;
    return *t;
}
}  // namespace

bool side_stream_of(hipStream_t st, SideStream* out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    SideTable& t = side_table();
    const uint64_t key = ((uint64_t)dev << 56) ^ (uint64_t)reinterpret_cast<uintptr_t>(st);
    std::lock_guard<std::mutex> lock(t.mu);
    auto it = t.of.find(key);
    if (it != t.of.end()) {
        *out = it->second;
        return out->stream != nullptr;
    }
    SideStream s{};
    bool ok = t.of.size() < 256 && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s.entry, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s.mid, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s.done2, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s.pre, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        s = SideStream{};                 // remembered: not tried again for this stream
    }
    t.of.emplace(key, s);
    *out = s;
    return ok;
}

static thread_local std::string g_last_error;
static thread_local bool g_defer_host_sync = false;

bool defer_host_sync() { return g_defer_host_sync; }

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return static_cast<int>(e);
}

int bad_arg(const char* what) {
    g_last_error = what;
    return -1;
}

struct ModelInfo {
    int n_setdims, n_channels, n_read, n_consts;
};

}  // namespace obe

using namespace obe;

extern "C" {

int obe_abi_version(void) { return OBE_ABI_VERSION; }

#ifndef OBE_SOURCE_FINGERPRINT
#define OBE_SOURCE_FINGERPRINT "unknown"
#endif
const char* obe_source_fingerprint(void) { return OBE_SOURCE_FINGERPRINT; }

const char* obe_last_error(void) { return g_last_error.c_str(); }

int obe_defer_host_sync(int32_t on) {
    const int prev = g_defer_host_sync ? 1 : 0;
    g_defer_host_sync = on != 0;
    return prev;
}

int obe_model_validate(obe_model* m) {
    if (!m) return bad_arg("model is NULL");
    ModelInfo info{};
    int rc = dispatch_model(*m, [&](auto M) -> int {
        using Model = decltype(M);
        info.n_setdims = Model::NS;
        info.n_channels = Model::NC;
        info.n_read = Model::NREAD;
        info.n_consts = Model::NCONST;
        return 0;
    });
    if (rc) return rc;
    if (m->n_setdims == 0) m->n_setdims = info.n_setdims;
    if (m->n_channels == 0) m->n_channels = info.n_channels;
    if (m->n_setdims != info.n_setdims) return bad_arg("model: n_setdims does not match the model");
    if (m->n_channels != info.n_channels) return bad_arg("model: n_channels does not match the model");
    if (m->n_params < info.n_read || m->n_params > OBE_MAX_DIMS) return bad_arg("model: n_params out of range for the model");
    if (m->n_consts < info.n_consts || m->n_consts > OBE_MAX_CONSTS) return bad_arg("model: too few constants");
    return 0;
}

int obe_host_word_arm(void* h_pinned_word) {
    if (!h_pinned_word) return bad_arg("obe_host_word_arm: null pointer");
    arm_host_word(h_pinned_word);
    return 0;
}

int obe_host_word_wait(const void* h_pinned_word, void* stream) {
    if (!h_pinned_word) return bad_arg("obe_host_word_wait: null pointer");
    return wait_host_word(h_pinned_word, as_stream(stream));
}

int obe_host_words_arm(void* h_pinned_words, int64_t n) {
    if (!h_pinned_words || n < 1) return bad_arg("obe_host_words_arm: null pointer / empty block");
    arm_host_words(h_pinned_words, n);
    return 0;
}

int obe_host_words_wait(const void* h_pinned_words, int64_t n, void* stream) {
    if (!h_pinned_words || n < 1) return bad_arg("obe_host_words_wait: null pointer / empty block");
    return wait_host_words(h_pinned_words, n, as_stream(stream));
}

int obe_host_device_pointer(const void* h_pinned, void** d_out) {
    if (!h_pinned || !d_out) return bad_arg("obe_host_device_pointer: null pointer");
    void* d = device_view_of_host(h_pinned);
    if (!d) return bad_arg("obe_host_device_pointer: not page-locked host memory");
    *d_out = d;
    return 0;
}

int obe_device_info(char* name, int name_len, int* n_cu, int64_t* hbm_bytes) {
    int dev = 0;
    OBE_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    OBE_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (name && name_len > 0) {
        std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return bad_arg("current device is not gfx950");
    return 0;
}

struct ObeTimer {
    hipEvent_t e0, e1;
};

int obe_timer_create(void** timer) {
    if (!timer) return bad_arg("timer is NULL");
    ObeTimer* t = new ObeTimer;
    hipError_t e = hipEventCreate(&t->e0);
    if (e == hipSuccess) e = hipEventCreate(&t->e1);
    if (e != hipSuccess) {
        delete t;
        return fail(e, "hipEventCreate");
    }
    *timer = t;
    return 0;
}

int obe_timer_start(void* timer, void* stream) {
    if (!timer) return bad_arg("timer is NULL");
    OBE_HIP_TRY(hipEventRecord(static_cast<ObeTimer*>(timer)->e0, as_stream(stream)));
    return 0;
}

int obe_timer_stop(void* timer, void* stream, float* ms) {
    if (!timer || !ms) return bad_arg("timer/ms is NULL");
    ObeTimer* t = static_cast<ObeTimer*>(timer);
    OBE_HIP_TRY(hipEventRecord(t->e1, as_stream(stream)));
    OBE_HIP_TRY(hipEventSynchronize(t->e1));
    OBE_HIP_TRY(hipEventElapsedTime(ms, t->e0, t->e1));
    return 0;
}

int obe_timer_destroy(void* timer) {
    if (!timer) return 0;
    ObeTimer* t = static_cast<ObeTimer*>(timer);
    (void)hipEventDestroy(t->e0);
    (void)hipEventDestroy(t->e1);
    delete t;
    return 0;
}

}  // extern "C"
