// Error plumbing and version of libmore4d_hip.so (the kernels live in the .hip files).
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

#include "more4d_hip.h"

static thread_local char g_err[512] = "";

extern "C" __attribute__((visibility("hidden"))) void m4d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* m4d_last_error(void) { return g_err; }

extern "C" int m4d_version(void) { return 100; }

// sha256 of csrc/ + include/ at build time (more4d_amd/build.py: source_hash); _lib.load() refuses a library whose hash is not the tree's
#ifndef M4D_SRC_HASH
#define M4D_SRC_HASH "unhashed"
#endif
static const char g_src_hash[] = "M4D_SRC_HASH=" M4D_SRC_HASH;      // the tag lets build.py read the hash out of the file without dlopen
extern "C" const char* m4d_source_hash(void) { return g_src_hash + 13; }

// ---- per-kernel-class launch counters (diagnostics; see more4d_hip.h) ----
static std::atomic<int64_t> g_launches[M4D_KC_COUNT];

extern "C" __attribute__((visibility("hidden"))) void m4d_count_launch(int kernel_class) {
    if (kernel_class >= 0 && kernel_class < M4D_KC_COUNT) g_launches[kernel_class].fetch_add(1, std::memory_order_relaxed);
}

extern "C" int64_t m4d_launch_count(int kernel_class, int reset) {
    if (kernel_class < 0) {
        if (reset)
            for (auto& c : g_launches) c.store(0, std::memory_order_relaxed);
        return 0;
    }
    if (kernel_class >= M4D_KC_COUNT) return -1;
    return reset ? g_launches[kernel_class].exchange(0, std::memory_order_relaxed) : g_launches[kernel_class].load(std::memory_order_relaxed);
}
