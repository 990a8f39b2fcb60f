// Thread-local error string + version for the C ABI (include/star_nugget.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/star_nugget.h"

namespace star {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace star

extern "C" int star_version(void) { return 100; }
extern "C" const char* star_last_error(void) { return star::g_err; }
