// Host-side scoped timers for investigations (off unless compiled with -DEQF_HOST_PROFILE): accumulated nanoseconds and call counts per label,
// printed to stderr when the process exits. HP_SCOPE("label"); in a block times it.
#pragma once
#ifdef EQF_HOST_PROFILE
#include <chrono>
#include <cstdio>
#include <map>
#include <string>
namespace eqf_host_prof {
struct Acc {
    long long ns = 0;
    long calls = 0;
};
struct Table {
    std::map<std::string, Acc> t;
    ~Table() {
        for (const auto& kv : t)
            std::fprintf(stderr, "[host_prof] %-40s %10.3f ms  %8ld calls  %8.2f us/call\n", kv.first.c_str(), 1e-6 * kv.second.ns, kv.second.calls,
                         kv.second.calls ? 1e-3 * kv.second.ns / kv.second.calls : 0.0);
    }
};
inline Table& table() {
    static Table tb;
    return tb;
}
struct Scope {
    Acc& a;
    std::chrono::steady_clock::time_point t0;
    explicit Scope(const char* label) : a(table().t[label]), t0(std::chrono::steady_clock::now()) {}
    ~Scope() {
        a.ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        ++a.calls;
    }
};
} // namespace eqf_host_prof
#define HP_CAT2(a, b) a##b
#define HP_CAT(a, b) HP_CAT2(a, b)
#define HP_SCOPE(label) eqf_host_prof::Scope HP_CAT(hp_scope_, __LINE__)(label)
#else
#define HP_SCOPE(label) ((void)0)
#endif
