// Thread-local error text of the segk C ABI.
#include "common.h"
#include <string.h>

namespace segk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace segk

extern "C" const char* segk_last_error(void) { return segk::g_err; }
extern "C" int segk_version(void) { return 100; }
