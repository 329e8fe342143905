// C-ABI plumbing: version + thread-local error string.
#include <stdarg.h>

#include "common.h"

namespace dpft {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dpft

extern "C" int dpft_version(void) { return 100; }
extern "C" const char* dpft_last_error(void) { return dpft::g_err; }
