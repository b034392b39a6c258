// libmpn.so runtime glue: error string shared by every entry point.
#include "mpn_common.h"
#include <stdarg.h>
#include <stdlib.h>

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
thread_local bool tl_oom = false;
std::atomic<long long> g_arena_bytes{0};
long long arena_test_budget() {
    static const long long b = []() { const char *e = getenv("MPN_TEST_ARENA_BUDGET"); return e ? atoll(e) : 0LL; }();
    return b;
}
}  // namespace mpn

// The pipelined mapper keeps 12 workers x 2 HIP streams in flight.  ROCm multiplexes streams onto GPU_MAX_HW_QUEUES
// hardware queues (4 by default), and streams that share a queue serialise: measured on MI355X, 4 -> 16 queues raised
// whole-job throughput by 15-30%, 16 -> 20 by another 2% (24, one queue per stream, loses 15%).  The variable is read when the HIP runtime initialises, so it is set when this
// library is loaded, unless the caller already chose a value; a host that initialised HIP earlier should export it.
__attribute__((constructor)) static void mpn_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "20", 0); }

extern "C" const char *mpn_last_error(void) { return mpn::get_error(); }
