// capi.hip -- library-wide state of libtgn_pointops.so: version string, per-thread error text, the per-thread
// stream used by the reference-signature entry points (which have no stream argument) and the per-device
// index-error word of the gather family.
#include "tgn_common.h"

#include <stdarg.h>

#include <mutex>

namespace tgn {

static thread_local char g_error[512] = "";
// thread-local: two host threads driving two streams through the reference ABI do not race on it
static thread_local hipStream_t g_default_stream = nullptr;

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

hipStream_t default_stream() { return g_default_stream; }

// One error word per device, created on first use under a lock (the kernels of a device OR into the word that
// lives in that device's memory; a single static would be written across GPUs with two devices in one process).
constexpr int kMaxDevices = 64;
static int *g_err_word[kMaxDevices] = {};
static std::mutex g_err_mutex;

int *index_error_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_err_mutex);
    if (!g_err_word[dev]) {
        int *w = nullptr;
        if (hipMalloc((void **)&w, sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(w, 0, sizeof(int)) != hipSuccess) {
            (void)hipFree(w);
            return nullptr;
        }
        g_err_word[dev] = w;
    }
    return g_err_word[dev];
}

}  // namespace tgn

TGN_API const char *tgn_version(void) { return "tgn_pointops 0.3.0 (gfx950)"; }
TGN_API const char *tgn_last_error(void) { return tgn::g_error; }
TGN_API void tgn_set_default_stream(tgn_stream_t stream) { tgn::g_default_stream = (hipStream_t)stream; }

// Returns the OR of the error bits latched on the current device since the last call (and clears them):
// bit 0 = a gather / grouping saw an index outside [-N, N).  Synchronises `stream`.
TGN_API int tgn_take_index_error(tgn_stream_t stream) {
    int *w = tgn::index_error_word();
    if (!w) return 0;
    int h = 0;
    if (hipMemcpyAsync(&h, w, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 0;
    if (h) (void)hipMemsetAsync(w, 0, sizeof(int), (hipStream_t)stream);
    return h;
}

// Forget whatever earlier launches latched, in stream order (no synchronisation): a checked operator calls this in front of
// its own launch so that a bit left by an UNCHECKED launch (a planner, a captured graph, TGN_INDEX_CHECK=off sections) is not
// blamed on it.
TGN_API int tgn_clear_index_error(tgn_stream_t stream) {
    int *w = tgn::index_error_word();
    if (!w) return TGN_OK;
    return hipMemsetAsync(w, 0, sizeof(int), (hipStream_t)stream) == hipSuccess ? TGN_OK : TGN_ERR_LAUNCH;
}

// A planner's spacer: one wave that sleeps for about `microseconds` on `stream` (100 MHz wall clock).  HotPath uses it to
// hold the groupings back until the FPS level-1 workgroups have read their clouds.
namespace tgn {
__global__ void delay_kernel(unsigned ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace tgn
TGN_API int tgn_stream_delay(int microseconds, tgn_stream_t stream) {
    if (microseconds <= 0) return TGN_OK;
    hipLaunchKernelGGL(tgn::delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned)microseconds * 100u);
    return tgn::check_launch("delay_kernel");
}
