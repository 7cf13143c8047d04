// The few helpers of sgl_common.h the measurement library needs (thread-local error text, tuning lookups), so that libsgl_probe.so
// stands alone next to libsgl_hip.so: it shares no state with the product library and exports its own error accessor.
#include "sgl_common.h"

#include "../../include/sgl_probe.h"

namespace sgl {

namespace {
thread_local std::string g_error;
}

void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

const char *get_error() { return g_error.c_str(); }

int64_t tuning(const char *, int64_t dflt) { return dflt; }   // the probes have no knobs

}  // namespace sgl

SGL_EXPORT const char *sgl_probe_last_error(void) { return sgl::get_error(); }
