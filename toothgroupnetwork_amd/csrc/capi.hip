// capi.hip -- library-wide state of libtgn_pointops.so: version string, per-thread error text, the per-thread
// stream used by the reference-signature entry points (which have no stream argument), the per-(device, stream)
// index-error words of the gather family and the kernel-variant switches.
#include "tgn_common.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
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

// One error word per (device, stream), created on first use under a lock: the kernels launched on a stream OR into that
// stream's word, which lives in that device's memory.  (One word per device was shared by every host thread: a loader thread's
// tgn_clear_index_error could erase the bit another stream had just latched.)  Words are carved from one 1-KiB block per
// device; the 257th stream of a device shares word 0.
constexpr int kMaxDevices = 64, kWordsPerDevice = 256;
struct ErrWords {
    int *block = nullptr;
    int used = 0;
    hipStream_t owner[kWordsPerDevice];
};
static ErrWords g_err[kMaxDevices];
static std::mutex g_err_mutex;

int *index_error_word(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_err_mutex);
    ErrWords &e = g_err[dev];
    if (!e.block) {
        int *w = nullptr;
        if (hipMalloc((void **)&w, kWordsPerDevice * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(w, 0, kWordsPerDevice * sizeof(int)) != hipSuccess) {
            (void)hipFree(w);
            return nullptr;
        }
        e.block = w;
    }
    for (int i = 0; i < e.used; ++i)
        if (e.owner[i] == stream) return e.block + i;
    if (e.used == kWordsPerDevice) {
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
            fprintf(stderr, "libtgn_pointops: more than %d streams on device %d use the gather family; further streams share "
                            "index-error word 0 (a bit may be reported to, or cleared by, another stream's check)\n", kWordsPerDevice, dev);
        return e.block;
    }
    e.owner[e.used] = stream;
    return e.block + e.used++;
}

// ---- kernel-variant switches ------------------------------------------------------------------------------------------
struct TuneEntry {
    const char *key, *env;   // env: the legacy environment name that seeds the entry when the library is loaded
    int def;
};
static const TuneEntry kTune[kTuneCount] = {
    {"fps_plain", "TGN_FPS_V1", 0},           {"fps_config", "TGN_FPS_CONFIG", 0},       {"fps_bucket_config", "TGN_FPS_BUCKET_CONFIG", 0},
    {"fps_cell_bits", "TGN_FPS_CELL_BITS", 4}, {"fps_bucket_min", "TGN_FPS_BUCKET_MIN", -1}, {"ball_bitmap", "TGN_BALL_BITMAP", 2},
    {"knn_memset", "TGN_KNN_MEMSET", 0},      {"knn_grid_scale", "TGN_KNN_GRID_SCALE", 1000}, {"sa_tile", "TGN_SA_TILE", 0},
    {"gather_v4", "TGN_GATHER_V4", 5},        {"fps_lean", "TGN_FPS_LEAN", 1},
};
static std::atomic<int> g_tune[kTuneCount];
static const bool g_tune_seeded = [] {   // runs once, at load time, before any launch can read the table
    for (int i = 0; i < kTuneCount; ++i) {
        int v = kTune[i].def;
        if (const char *e = getenv(kTune[i].env)) {
            int a = 0, b = 0;
            if (i == kTuneFpsConfig || i == kTuneFpsBucketConfig)
                v = sscanf(e, "%d,%d", &a, &b) == 2 ? a * 256 + b : 0;
            else if (i == kTuneKnnGridScale)
                v = (int)(atof(e) * 1000.0 + 0.5);
            else
                v = atoi(e);
        }
        g_tune[i].store(v, std::memory_order_relaxed);
    }
    return true;
}();

int tuning(Tuning t) { return g_tune[t].load(std::memory_order_relaxed); }

}  // namespace tgn

// Select a kernel variant (experiments, A/B runs, the parity tests that must reach every variant).  Thread-safe; takes effect
// for launches enqueued after the call.  Returns TGN_ERR_INVALID_ARGUMENT for an unknown key.
TGN_API int tgn_set_tuning(const char *key, int value) {
    if (key)
        for (int i = 0; i < tgn::kTuneCount; ++i)
            if (!strcmp(key, tgn::kTune[i].key)) {
                tgn::g_tune[i].store(value, std::memory_order_relaxed);
                return TGN_OK;
            }
    tgn::set_error("tgn_set_tuning: unknown key '%s'", key ? key : "(null)");
    return TGN_ERR_INVALID_ARGUMENT;
}
// Current value of a switch; `fallback` for an unknown key.
TGN_API int tgn_get_tuning(const char *key, int fallback) {
    if (key)
        for (int i = 0; i < tgn::kTuneCount; ++i)
            if (!strcmp(key, tgn::kTune[i].key)) return tgn::tuning((tgn::Tuning)i);
    return fallback;
}

TGN_API const char *tgn_version(void) { return "tgn_pointops 0.5.0 (gfx950)"; }
TGN_API const char *tgn_last_error(void) { return tgn::g_error; }
TGN_API void tgn_set_default_stream(tgn_stream_t stream) { tgn::g_default_stream = (hipStream_t)stream; }

// Returns the OR of the error bits latched by launches on `stream` of the current device since the last call (and clears them):
// bit 0 = a gather / grouping saw an index outside [-N, N).  Synchronises `stream`.
TGN_API int tgn_take_index_error(tgn_stream_t stream) {
    int *w = tgn::index_error_word((hipStream_t)stream);
    if (!w) return 0;
    int h = 0;
    if (hipMemcpyAsync(&h, w, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 0;
    if (h) (void)hipMemsetAsync(w, 0, sizeof(int), (hipStream_t)stream);
    return h;
}

// The same over EVERY stream of the current device: synchronises the device, returns the OR of all its words and clears them.
// For callers that launch on streams of their own without checking (HotPath's three streams, captured graphs replayed on
// another stream, TGN_INDEX_CHECK=off sections) and want one answer at a synchronisation point.
TGN_API int tgn_take_index_error_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= tgn::kMaxDevices) return 0;
    int *block = nullptr;
    {
        std::lock_guard<std::mutex> lock(tgn::g_err_mutex);
        block = tgn::g_err[dev].block;
    }
    if (!block) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    int h[tgn::kWordsPerDevice];
    if (hipMemcpy(h, block, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    int any = 0;
    for (int i = 0; i < tgn::kWordsPerDevice; ++i) any |= h[i];
    if (any) (void)hipMemset(block, 0, sizeof(h));
    return any;
}

// Forget whatever earlier launches latched, in stream order (no synchronisation): a checked operator calls this in front of
// its own launch so that a bit left by an UNCHECKED launch (a planner, a captured graph, TGN_INDEX_CHECK=off sections) is not
// blamed on it.
TGN_API int tgn_clear_index_error(tgn_stream_t stream) {
    int *w = tgn::index_error_word((hipStream_t)stream);
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
namespace tgn {
// out[r, 0 .. ncols) = in[r, first .. first + ncols) for rows of `stride` floats: the coordinate block of a scan's (N, 6) rows
// (gen_utils.py:138 / pointnet_pp_model.py:16-20 slice it with torch indexing).
__global__ __launch_bounds__(256) void slice_columns_kernel(long long rows, int stride, int first, int ncols,
                                                            const float *__restrict__ in, float *__restrict__ out) {
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += nthreads) {
        const float *__restrict__ src = in + r * stride + first;
        float *__restrict__ dst = out + r * ncols;
        for (int c = 0; c < ncols; ++c) dst[c] = src[c];
    }
}
// The case that matters -- xyz out of (x, y, z, nx, ny, nz) rows, 16-byte aligned: a lane moves FOUR rows with six 16-byte loads
// and three 16-byte stores (a wave: 6 KB in, 3 KB out, both contiguous).  This runs on a copy stream beside whatever the GPU is
// doing -- typically in the one wave per SIMD an FPS level-1 workgroup leaves free, where a dword-per-lane copy is bound by
// the latency of its few loads in flight (2.8 ms for 256 scans against 0.1 ms).
__global__ __launch_bounds__(256) void slice_xyz_of_6_kernel(long long quads, const float4 *__restrict__ in, float4 *__restrict__ out) {
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += nthreads) {
        const float4 *__restrict__ src = in + q * 6;
        const float4 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4], f = src[5];
        // rows: (a.x a.y a.z | a.w b.x b.y) (b.z b.w c.x | c.y c.z c.w) (d.x d.y d.z | ...) (e.z e.w f.x | ...)
        float4 *__restrict__ dst = out + q * 3;
        dst[0] = make_float4(a.x, a.y, a.z, b.z);
        dst[1] = make_float4(b.w, c.x, d.x, d.y);
        dst[2] = make_float4(d.z, e.z, e.w, f.x);
    }
}
}  // namespace tgn
TGN_API int tgn_slice_columns(long long rows, int stride, int first, int ncols, const float *in, float *out, tgn_stream_t stream) {
    if (rows < 0 || stride <= 0 || first < 0 || ncols < 0 || first + ncols > stride) {
        tgn::set_error("tgn_slice_columns: rows %lld, stride %d, columns [%d, %d)", rows, stride, first, first + ncols);
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (rows == 0 || ncols == 0) return TGN_OK;
    if (!in || !out) {
        tgn::set_error("tgn_slice_columns: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    long long done = 0;
    if (stride == 6 && first == 0 && ncols == 3 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && rows >= 4) {
        const long long quads = rows / 4;
        long long blocks = (quads + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(tgn::slice_xyz_of_6_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, quads, (const float4 *)in,
                           (float4 *)out);
        if (int rc = tgn::check_launch("slice_xyz_of_6_kernel")) return rc;
        done = quads * 4;
        if (done == rows) return TGN_OK;
    }
    const long long rest = rows - done;
    long long blocks = (rest + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tgn::slice_columns_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rest, stride, first, ncols,
                       in + done * stride, out + done * ncols);
    return tgn::check_launch("slice_columns_kernel");
}

TGN_API int tgn_stream_delay(int microseconds, tgn_stream_t stream) {
    if (microseconds <= 0) return TGN_OK;
    hipLaunchKernelGGL(tgn::delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned)microseconds * 100u);
    return tgn::check_launch("delay_kernel");
}
