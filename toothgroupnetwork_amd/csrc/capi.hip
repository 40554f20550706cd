// capi.hip -- library-wide state of libtgn_pointops.so: version string, per-thread error text and the
// stream used by the reference-signature entry points (which have no stream argument).
#include "tgn_common.h"

#include <stdarg.h>

namespace tgn {

static thread_local char g_error[512] = "";
static hipStream_t g_default_stream = nullptr;

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

hipStream_t default_stream() { return g_default_stream; }

}  // namespace tgn

TGN_API const char *tgn_version(void) { return "tgn_pointops 0.1.0 (gfx950)"; }
TGN_API const char *tgn_last_error(void) { return tgn::g_error; }
TGN_API void tgn_set_default_stream(tgn_stream_t stream) { tgn::g_default_stream = (hipStream_t)stream; }
