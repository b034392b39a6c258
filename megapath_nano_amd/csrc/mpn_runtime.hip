// libmpn.so runtime glue: error string shared by every entry point.
#include "mpn_common.h"
#include <stdarg.h>

namespace mpn {
static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char *get_error() { return g_err.c_str(); }
thread_local Arena *tl_arena = nullptr;
}  // namespace mpn

extern "C" const char *mpn_last_error(void) { return mpn::get_error(); }
