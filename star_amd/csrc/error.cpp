// Thread-local error string, version and the test-only variant switches of the C ABI (include/star_nugget.h).
#include <stdarg.h>
#include <stdio.h>
#include <atomic>
#include "../../include/star_nugget.h"

namespace star {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::atomic<int> g_override[STAR_OVR_COUNT];
int test_override(int which) { return g_override[which].load(std::memory_order_relaxed); }
}  // namespace star

extern "C" int star_version(void) { return 200; }
extern "C" const char* star_last_error(void) { return star::g_err; }
extern "C" int star_test_override(int which, int value) {
    if (which < 0 || which >= STAR_OVR_COUNT) {
        star::set_error("star_test_override: unknown switch %d", which);
        return STAR_ERR_ARG;
    }
    return star::g_override[which].exchange(value, std::memory_order_relaxed);
}
