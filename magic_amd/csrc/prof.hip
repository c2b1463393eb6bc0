// Per-kernel HIP-event timing and the small status / version entry points.
#include "common.h"
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace magic {

namespace {
struct Span {
    const char *name;
    hipEvent_t a, b;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Span> g_spans;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open_a = nullptr;
const char *g_open_name = nullptr;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

bool prof_on() { return g_on; }

// "<base><a,b>" with static storage (one string per distinct combination), so that the template
// instances of one kernel are timed apart
const char *prof_name(const char *base, int a, int b) {
    static std::map<std::string, std::string> names;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    std::string key = std::string(base) + "<" + std::to_string(a) + "," + std::to_string(b) + ">";
    auto it = names.find(key);
    if (it == names.end()) it = names.emplace(key, key).first;
    return it->second.c_str();
}

void prof_begin(const char *name, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_open_a = get_event();
    g_open_name = name;
    if (g_open_a) (void)hipEventRecord(g_open_a, s);
}

void prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_open_a) return;
    hipEvent_t b = get_event();
    if (b) {
        (void)hipEventRecord(b, s);
        g_spans.push_back({g_open_name, g_open_a, b});
    } else {
        g_pool.push_back(g_open_a);
    }
    g_open_a = nullptr;
}

int ensure_dynamic_lds(const void *kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return MAGIC_OK;  // within the default limit
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MAGIC_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(mu);
    size_t &have = granted[{kernel, dev}];
    if (bytes <= have) return MAGIC_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return MAGIC_ERR_RUNTIME;
    have = bytes;
    return MAGIC_OK;
}

}  // namespace magic

extern "C" {

int magic_abi_version(void) { return 3; }

const char *magic_status_string(int status) {
    switch (status) {
        case MAGIC_OK: return "ok";
        case MAGIC_ERR_INVALID_ARG: return "invalid argument";
        case MAGIC_ERR_UNSUPPORTED: return "unsupported head count / head width";
        case MAGIC_ERR_WORKSPACE: return "workspace missing or too small";
        case MAGIC_ERR_LAUNCH: return "kernel launch failed";
        case MAGIC_ERR_RUNTIME: return "HIP runtime call failed";
        case MAGIC_ERR_STATIC_LIMIT: return "input exceeds the static scratch of the no-allocation entry point";
        default: return "unknown status";
    }
}

int magic_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(magic::g_mu);
    magic::g_on = on != 0;
    return MAGIC_OK;
}

int magic_prof_reset(void) {
    std::lock_guard<std::mutex> lk(magic::g_mu);
    for (auto &s : magic::g_spans) {
        magic::g_pool.push_back(s.a);
        magic::g_pool.push_back(s.b);
    }
    magic::g_spans.clear();
    return MAGIC_OK;
}

int magic_prof_read(int max_entries, char *names, int64_t *launches, double *total_ms) {
    if (max_entries < 0 || (max_entries > 0 && (!names || !launches || !total_ms))) return MAGIC_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(magic::g_mu);
    std::map<std::string, std::pair<int64_t, double>> acc;
    for (auto &s : magic::g_spans) {
        if (hipEventSynchronize(s.b) != hipSuccess) return MAGIC_ERR_RUNTIME;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) return MAGIC_ERR_RUNTIME;
        auto &e = acc[s.name];
        e.first += 1;
        e.second += ms;
    }
    int n = 0;
    for (auto &kv : acc) {
        if (n >= max_entries) break;
        std::strncpy(names + (size_t)n * 64, kv.first.c_str(), 63);
        names[(size_t)n * 64 + 63] = 0;
        launches[n] = kv.second.first;
        total_ms[n] = kv.second.second;
        ++n;
    }
    return n;
}

int magic_prof_sequence(int max_entries, char *names) {
    if (max_entries < 0 || (max_entries > 0 && !names)) return MAGIC_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(magic::g_mu);
    int n = 0;
    for (auto &s : magic::g_spans) {
        if (n >= max_entries) break;
        std::strncpy(names + (size_t)n * 64, s.name, 63);
        names[(size_t)n * 64 + 63] = 0;
        ++n;
    }
    return (int)magic::g_spans.size();
}

}  // extern "C"
