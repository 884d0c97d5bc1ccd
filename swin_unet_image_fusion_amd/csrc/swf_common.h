// Shared host-side helpers of libswinfuse (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/swinfuse.h"

namespace swf {

// thread-local error text behind swf_last_error_string()
char* err_buf();
int fail(int status, const char* fmt, ...);

inline hipStream_t as_stream(swf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Check the launch that was just enqueued (no sync: only launch-configuration errors show up here).
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SWF_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return SWF_OK;
}

// A/B switches of the dispatch (every one selects between HIP kernels of this library; there is no CPU path).  They are
// read ONLY when SWF_DEBUG_SWITCHES=1 is set as well, so that a stray variable in a production environment — or on one rank of a
// sharded job, where it would break the bit-identity of batch shards — cannot change the kernel path.  tests/test_gpu_switches.py
// runs the main fallbacks in child processes with the opt-in set.
inline const char* debug_env(const char* name) {
    static const bool on = [] { const char* e = std::getenv("SWF_DEBUG_SWITCHES"); return e && e[0] == '1'; }();
    return on ? std::getenv(name) : nullptr;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over the caller's workspace; every carve is 256-byte aligned.
struct Carver {
    char* base;
    size_t cap, used;
    Carver(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes), used(0) {}
    float* floats(int64_t n) {
        size_t off = align_up(used, 256);
        used = off + static_cast<size_t>(n) * sizeof(float);
        return reinterpret_cast<float*>(base + off);
    }
    bool ok() const { return base != nullptr ? used <= cap : used == 0; }
};
inline size_t carve_bytes(std::initializer_list<int64_t> floats) {
    size_t used = 0;
    for (int64_t n : floats) used = align_up(used, 256) + static_cast<size_t>(n) * sizeof(float);
    return align_up(used, 256);
}

}  // namespace swf

#define SWF_TRY(expr)                      \
    do {                                   \
        int _st = (expr);                  \
        if (_st != SWF_OK) return _st;     \
    } while (0)
