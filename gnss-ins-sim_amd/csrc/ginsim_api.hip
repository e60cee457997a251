// C ABI of libginsim.so (declared in include/ginsim.h): context handling, argument validation, error
// reporting and the host-buffer convenience entry points.  No kernels here.
#include <hip/hip_runtime.h>
#include <chrono>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ginsim.h"
#include "allan.hpp"
#include "comm.hpp"
#include "placed.hpp"

namespace ginsim {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

hipError_t launch_mc(const ginsim_mc_params& p, hipStream_t stream, char* name, size_t cap, double* proc_about);      // name: report, do not launch
hipError_t launch_mc_f32(const ginsim_mc_params& p, float* truth32, hipStream_t stream, char* name, size_t cap);
size_t mc_f32_truth_bytes(const ginsim_mc_params& p);
int mc_variant(const ginsim_mc_params& p);
bool series_path_applies(const ginsim_mc_params& p);
int series_pass_b(const ginsim_mc_params& p);
int64_t series_chunks(const ginsim_mc_params& p, int32_t* L_out);
hipError_t launch_series(const ginsim_mc_params& p, double* carry, hipStream_t stream);
int mc_variant_f32(const ginsim_mc_params& p);
hipError_t launch_gather_runs_f32(const float* series, int C, int64_t n, int64_t runs, const int64_t* ids, int nsel,
                                  double* out, hipStream_t s);
hipError_t launch_aux(const ginsim_aux_params& p, hipStream_t s);
size_t vib_psd_scratch_bytes(int64_t period, int64_t runs);
int launch_vib_psd(int device, hipStream_t stream, const double* amp, int64_t period, int64_t runs, uint64_t run_offset, uint64_t seed,
                   int sensor, int halve, void* scratch, double* out);
void vib_psd_drop_plans(hipStream_t stream);
hipError_t launch_rng_probe(uint64_t seed, uint64_t run, uint32_t stream, int64_t count, double* z0, double* z1,
                            uint32_t* words, hipStream_t stream_h);
hipError_t launch_aos_to_soa(const double* src, double* dst, int64_t R, int64_t n, int C, hipStream_t s);
hipError_t launch_runs_to_series(const double* in, double* out, int C, int64_t n, int64_t R, hipStream_t s);
hipError_t launch_normal_transform(const uint32_t* words, int64_t count, double* z0, double* z1, hipStream_t s);
hipError_t launch_gather_runs(const double* series, int C, int64_t n, int64_t runs, const int64_t* ids, int nsel,
                              double* out, hipStream_t s);
hipError_t launch_gather_series(const double* series, int C, int64_t n, const int64_t* ids, int nsel, double* out, hipStream_t s);
size_t stats_scratch_bytes(int64_t runs);
int stats_blocks(int64_t runs);
hipError_t launch_end_stats(const double* end_err, int64_t runs, void* scratch, hipStream_t s);
void stats_merge_host(const ginsim_stats* parts, int nparts, ginsim_stats* out);
hipError_t launch_process_stats(const double* traj, const double* ref, int64_t n, int64_t runs, int64_t j0, int pos_ned,
                                int run_major, double* out, hipStream_t s);
hipError_t launch_process_stats_f32(const float* traj, const double* ref, int64_t n, int64_t runs, int64_t j0, int pos_ned,
                                    int run_major, double* out, const double* origin, int64_t n_ini, uint64_t ini_first, hipStream_t s);


}  // namespace ginsim

using namespace ginsim;

struct ginsim_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    std::vector<hipEvent_t> pool;   // lazily created, indexed by slot
    void* ws[4] = {nullptr, nullptr, nullptr, nullptr};   // grow-only scratch regions (stats / allan)
    size_t ws_bytes[4] = {0, 0, 0, 0};
    double* allan_host = nullptr;                         // pinned host memory the Allan kernels write their sums into
    size_t allan_host_doubles = 0;
    ginsim::Comm* comm = nullptr;                         // RCCL communicator of this rank (ginsim_comm_init), or nullptr
    double* comm_recv = nullptr;                          // device [8 slots][nranks][28]: the gathered records
    ginsim_stats* comm_host = nullptr;                    // pinned host copy of the same
    hipEvent_t comm_ev[8] = {};
    bool comm_pending[8] = {};
    ginsim_stats* stat_slots = nullptr;                   // pinned host records of ginsim_end_stats_begin/_finish
    hipEvent_t stat_ev[8] = {};
    bool stat_pending[8] = {};
};

// grow-only scratch owned by the context: avoids a hipMalloc/hipFree pair (~100 us each) per call
static hipError_t scratch(ginsim_ctx* c, int slot, size_t bytes, void** out) {
    if (c->ws_bytes[slot] < bytes) {
        if (c->ws[slot]) {
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) return e;
            e = hipFree(c->ws[slot]);
            if (e != hipSuccess) return e;
            c->ws[slot] = nullptr;
            c->ws_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&c->ws[slot], want);
        if (e != hipSuccess) return e;
        c->ws_bytes[slot] = want;
    }
    *out = c->ws[slot];
    return hipSuccess;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (e_ == hipErrorOutOfMemory) { (void)hipGetLastError(); return GINSIM_ERR_NOMEM; } \
            return GINSIM_ERR_HIP;                                                            \
        }                                                                                     \
    } while (0)

#define REQUIRE(cond, ...)            \
    do {                              \
        if (!(cond)) {                \
            set_error(__VA_ARGS__);   \
            return GINSIM_ERR_ARG;    \
        }                             \
    } while (0)

// RAII device allocation for the host-buffer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

extern "C" {

int ginsim_abi_version(void) { return GINSIM_ABI_VERSION; }

const char* ginsim_last_error(void) { return g_err.c_str(); }

int ginsim_device_count(int* count) {
    REQUIRE(count, "device_count: NULL output");
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return GINSIM_ERR_NODEV;
    }
    *count = n;
    return GINSIM_OK;
}

int ginsim_create(int device, ginsim_ctx** out) {
    REQUIRE(out, "create: NULL output");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        set_error("no HIP device visible (the engine has no CPU fallback)");
        return GINSIM_ERR_NODEV;
    }
    REQUIRE(device >= 0 && device < n, "create: device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    ginsim_ctx* c = new ginsim_ctx();
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    ginsim::placed_context_created(device);
    *out = c;
    return GINSIM_OK;
}

int ginsim_destroy(ginsim_ctx* c) {
    if (!c) return GINSIM_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->pool)
        if (e) (void)hipEventDestroy(e);
    for (void* w : c->ws)
        if (w) (void)hipFree(w);
    for (hipEvent_t e : c->stat_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stat_slots) (void)hipHostFree(c->stat_slots);
    if (c->allan_host) (void)hipHostFree(c->allan_host);
    if (c->comm) ginsim::comm_destroy(c->comm);
    if (c->comm_recv) (void)hipFree(c->comm_recv);
    if (c->comm_host) (void)hipHostFree(c->comm_host);
    for (hipEvent_t e : c->comm_ev)
        if (e) (void)hipEventDestroy(e);
    ginsim::vib_psd_drop_plans(c->stream);            // this stream's hipFFT plans of the PSD vibration (it is idle now)
    (void)hipStreamDestroy(c->stream);
    ginsim::placed_free_owner(c->device, c);          // regions this context carved and never freed go back to the free list
    ginsim::placed_context_destroyed(c->device);      // the device's last context gives its placed arena back
    delete c;
    return GINSIM_OK;
}

int ginsim_device_name(ginsim_ctx* c, char* buf, size_t cap) {
    REQUIRE(c && buf && cap > 0, "device_name: bad arguments");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return GINSIM_OK;
}

int ginsim_malloc(ginsim_ctx* c, size_t bytes, void** dptr) {
    REQUIRE(c && dptr, "malloc: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 8));
    return GINSIM_OK;
}

// ---- ABI 7: placed device memory (csrc/placed.hip)
int ginsim_placed_configure(ginsim_ctx* c, const ginsim_placed_options* o) {
    REQUIRE(c && o, "placed_configure: bad arguments");
    return ginsim::placed_configure(c->device, *o);
}

int ginsim_placed_reserve(ginsim_ctx* c, size_t bytes) {
    REQUIRE(c, "placed_reserve: NULL context");
    return ginsim::placed_reserve(c->device, c->stream, bytes);
}

int ginsim_malloc_placed(ginsim_ctx* c, size_t bytes, void** dptr) {
    REQUIRE(c && dptr, "malloc_placed: bad arguments");
    return ginsim::placed_malloc(c->device, c->stream, bytes, c, dptr);
}

int ginsim_placed_release(ginsim_ctx* c) {
    REQUIRE(c, "placed_release: NULL context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ginsim::placed_release(c->device, false);
}

int ginsim_placed_info_get(ginsim_ctx* c, ginsim_placed_info* out) {
    REQUIRE(c && out, "placed_info_get: bad arguments");
    ginsim::placed_info(c->device, out);
    return GINSIM_OK;
}

int ginsim_mem_info(ginsim_ctx* c, size_t* free_bytes, size_t* total_bytes) {
    REQUIRE(c && free_bytes && total_bytes, "mem_info: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return GINSIM_OK;
}

int ginsim_host_alloc(ginsim_ctx* c, size_t bytes, void** hptr) {
    REQUIRE(c && hptr, "host_alloc: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 8, hipHostMallocDefault));
    return GINSIM_OK;
}

int ginsim_host_free(ginsim_ctx* c, void* hptr) {
    // c may be NULL: page-locked arrays handed to a caller can outlive the context that allocated them (hipHostFree needs
    // no stream); with a live context its stream is drained first so that no copy still targets the pages
    if (!hptr) return GINSIM_OK;
    if (c) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    HIP_TRY(hipHostFree(hptr));
    return GINSIM_OK;
}

int ginsim_free(ginsim_ctx* c, void* dptr) {
    REQUIRE(c, "free: NULL context");
    if (!dptr) return GINSIM_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (ginsim::placed_owns(c->device, dptr)) return ginsim::placed_free(c->device, dptr);     // back to the arena's free list
    HIP_TRY(hipFree(dptr));
    return GINSIM_OK;
}

int ginsim_memcpy_h2d(ginsim_ctx* c, void* dst, const void* src, size_t bytes) {
    REQUIRE(c && (bytes == 0 || (dst && src)), "memcpy_h2d: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_memcpy_d2h(ginsim_ctx* c, void* dst, const void* src, size_t bytes) {
    REQUIRE(c && (bytes == 0 || (dst && src)), "memcpy_d2h: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_memset(ginsim_ctx* c, void* dptr, int value, size_t bytes) {
    REQUIRE(c && (bytes == 0 || dptr), "memset: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(dptr, value, bytes, c->stream));
    return GINSIM_OK;
}

int ginsim_sync(ginsim_ctx* c) {
    REQUIRE(c, "sync: NULL context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_timer_begin(ginsim_ctx* c) {
    REQUIRE(c, "timer_begin: NULL context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return GINSIM_OK;
}

int ginsim_timer_end(ginsim_ctx* c, float* ms) {
    REQUIRE(c && ms, "timer_end: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return GINSIM_OK;
}

int ginsim_event_record(ginsim_ctx* c, int32_t slot) {
    REQUIRE(c && slot >= 0 && slot < GINSIM_MAX_EVENTS, "event_record: slot out of range");
    HIP_TRY(hipSetDevice(c->device));
    if ((size_t)slot >= c->pool.size()) c->pool.resize(slot + 1, nullptr);
    if (!c->pool[slot]) HIP_TRY(hipEventCreate(&c->pool[slot]));
    HIP_TRY(hipEventRecord(c->pool[slot], c->stream));
    return GINSIM_OK;
}

int ginsim_event_elapsed(ginsim_ctx* c, int32_t a, int32_t b, float* ms) {
    REQUIRE(c && ms && a >= 0 && b >= 0 && (size_t)a < c->pool.size() && (size_t)b < c->pool.size() && c->pool[a] &&
                c->pool[b], "event_elapsed: slots were not recorded");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->pool[b]));
    HIP_TRY(hipEventElapsedTime(ms, c->pool[a], c->pool[b]));
    return GINSIM_OK;
}

static int check_sensor(const ginsim_sensor_model& m, const char* what) {
    for (int i = 0; i < 3; ++i) {
        const double v[4] = {m.bias[i], m.gm_a[i], m.gm_b[i], m.white[i]};
        for (double x : v) REQUIRE(x == x && x - x == 0.0, "mc_run: %s model has a non-finite coefficient", what);
    }
    return GINSIM_OK;
}

int ginsim_mc_variant(const ginsim_mc_params* p, int32_t* variant) {
    REQUIRE(p && variant, "mc_variant: NULL argument");
    *variant = p->precision == 1 ? mc_variant_f32(*p) : (series_path_applies(*p) ? 2 : mc_variant(*p));
    return GINSIM_OK;
}

static int check_mc_params(const ginsim_mc_params* p);

int ginsim_mc_kernel_name(const ginsim_mc_params* p, char* buf, size_t cap) {
    REQUIRE(p && buf && cap > 0, "mc_kernel_name: bad arguments");
    const int rc = check_mc_params(p);
    if (rc) return rc;
    buf[0] = 0;
    if (p->precision == 1) (void)launch_mc_f32(*p, nullptr, nullptr, buf, cap);
    else if (series_path_applies(*p))       // the dominant one of series_kernel<0>, series_scan_kernel, series_kernel<1 | 2>
        snprintf(buf, cap, "ginsim::series_kernel<%d>", series_pass_b(*p));
    else (void)launch_mc(*p, nullptr, buf, cap, nullptr);
    REQUIRE(buf[0], "mc_kernel_name: no kernel serves these parameters");
    return GINSIM_OK;
}

int ginsim_mc_run(ginsim_ctx* c, const ginsim_mc_params* p) {
    REQUIRE(c, "mc_run: NULL argument");
    const int rc0 = check_mc_params(p);
    if (rc0) return rc0;
    HIP_TRY(hipSetDevice(c->device));
    if (p->precision == 1) {
        void* truth32 = nullptr;       // the wave-specialised fp32 kernel reads its truth as floats (converted by a pre-launch)
        const size_t tb = mc_f32_truth_bytes(*p);
        if (tb) HIP_TRY(scratch(c, 3, tb, &truth32));
        HIP_TRY(launch_mc_f32(*p, reinterpret_cast<float*>(truth32), c->stream, nullptr, 0));
    } else if (series_path_applies(*p)) {       // sensors only, few runs, long series: parallel along time
        int32_t L = 0;
        const int64_t nchunks = series_chunks(*p, &L);
        void* carry = nullptr;
        HIP_TRY(scratch(c, 3, sizeof(double) * 6 * (size_t)nchunks * (size_t)p->runs, &carry));
        HIP_TRY(launch_series(*p, reinterpret_cast<double*>(carry), c->stream));
    } else {
        void* about = nullptr;          // the launch-wide shift of the online process statistics (mc_kernel.hip, Proc): nine doubles
        if (p->out_proc[0] || p->out_proc[1]) HIP_TRY(scratch(c, 3, 9 * sizeof(double), &about));
        HIP_TRY(launch_mc(*p, c->stream, nullptr, 0, reinterpret_cast<double*>(about)));
    }
    return GINSIM_OK;
}

}  // extern "C"

static int check_mc_params(const ginsim_mc_params* p) {
    REQUIRE(p, "mc_run: NULL argument");
    REQUIRE(p->n >= 1 && p->runs >= 1, "mc_run: n=%lld runs=%lld must be >= 1", (long long)p->n, (long long)p->runs);
    REQUIRE(p->n <= 0xFFFFFFFFll, "mc_run: n exceeds the 32-bit sample counter of the RNG");
    REQUIRE(p->runs <= (int64_t)0x7FFFFFFF * 64, "mc_run: too many runs for one launch");
    REQUIRE(p->fs > 0.0, "mc_run: fs must be positive");
    REQUIRE(p->ref_frame == 0 || p->ref_frame == 1, "mc_run: ref_frame must be 0 or 1");
    REQUIRE(p->proc_plain_sums == 0 || p->proc_plain_sums == 1, "mc_run: proc_plain_sums must be 0 or 1");
    REQUIRE(p->algo_mask >= 0 && p->algo_mask <= 3, "mc_run: algo_mask must be a combination of GINSIM_ALGO_*");
    REQUIRE(p->algo_mask != 0 || (!p->given_sensors && (p->out_accel || p->out_gyro || p->out_odo)),
            "mc_run: algo_mask 0 (sensors only) needs sensor outputs");
    REQUIRE(p->algo_mask == 0 || (p->n_ini >= 1 && p->ini), "mc_run: initial-state table missing");
    REQUIRE(p->block_threads == 0 || p->block_threads == 64 || p->block_threads == 128 || p->block_threads == 256,
            "mc_run: block_threads must be 0, 64, 128 or 256");
    const bool odo = (p->algo_mask & GINSIM_ALGO_ODO) != 0, fre = (p->algo_mask & GINSIM_ALGO_FREE) != 0;
    if (p->given_sensors) {
        REQUIRE(p->in_gyro, "mc_run: given_sensors needs in_gyro");
        REQUIRE(!fre || p->in_accel, "mc_run: given_sensors free integration needs in_accel");
        REQUIRE(!odo || p->in_odo, "mc_run: given_sensors odometer integration needs in_odo");
    } else {
        REQUIRE(p->ref_gyro && p->ref_accel, "mc_run: ref_accel/ref_gyro missing");
        REQUIRE((!odo && !p->out_odo) || p->ref_odo, "mc_run: ref_odo missing");
        int rc = check_sensor(p->accel, "accel");
        if (rc) return rc;
        rc = check_sensor(p->gyro, "gyro");
        if (rc) return rc;
    }
    for (const ginsim_vibration* v : {&p->vib_accel, &p->vib_gyro}) {
        REQUIRE(v->type == GINSIM_VIB_NONE || v->type == GINSIM_VIB_RANDOM || v->type == GINSIM_VIB_SINUSOIDAL || v->type == GINSIM_VIB_PSD,
                "mc_run: vibration type must be 0 (none), 1 (random), 2 (sinusoidal) or 3 (psd)");
        if (v->type == GINSIM_VIB_NONE) continue;
        REQUIRE(!p->given_sensors, "mc_run: a vibration term cannot be added to given sensors");
        if (v->type == GINSIM_VIB_PSD) {
            REQUIRE(v->series && v->period >= 2 && v->period <= 16384, "mc_run: a psd vibration needs its series (ginsim_vib_psd_series) and their period (2 .. 16384)");
            REQUIRE(p->precision == 0, "mc_run: the psd vibration runs on the fp64 kernels only");
            REQUIRE(p->sensor_layout == 0, "mc_run: the psd vibration runs on the lane-per-run kernels only (sensor_layout 0)");
        }
        REQUIRE(std::isfinite(v->amp[0]) && std::isfinite(v->amp[1]) && std::isfinite(v->amp[2]) && std::isfinite(v->omega_dt),
                "mc_run: vibration amplitudes / frequency must be finite");
    }
    REQUIRE(p->precision == 0 || p->precision == 1, "mc_run: precision must be 0 (fp64) or 1 (fp32)");
    REQUIRE(p->sensor_layout == 0 || p->sensor_layout == 1, "mc_run: sensor_layout must be 0 ([axis][sample][run]) or 1 ([run][axis][sample])");
    REQUIRE(p->sensor_layout == 0 || series_path_applies(*p),
            "mc_run: sensor_layout 1 is written by the time-parallel series kernels only (sensors only, fp64, <= 1024 runs, >= 2048 samples)");
    if (p->out_proc[0] || p->out_proc[1]) {
        REQUIRE(!p->given_sensors && p->precision == 0, "mc_run: online process statistics need generate mode and fp64");
        REQUIRE((p->algo_mask == GINSIM_ALGO_FREE && p->out_proc[0] && !p->out_proc[1]) ||
                (p->algo_mask == GINSIM_ALGO_ODO && p->out_proc[1] && !p->out_proc[0]),
                "mc_run: online process statistics take ONE algorithm per launch (out_proc of that algorithm only)");
        REQUIRE(p->ref_nav, "mc_run: online process statistics need ref_nav");
        REQUIRE(p->proc_first >= 0 && p->proc_first < p->n, "mc_run: proc_first out of range");
        REQUIRE(!p->proc_pos_ned || p->ref_frame == 0, "mc_run: NED position errors exist in ref_frame 0 only");
    }
    REQUIRE(!(p->out_end_ned[0] || p->out_end_ned[1]) || (p->ref_frame == 0 && p->precision == 0 && !p->given_sensors),
            "mc_run: out_end_ned needs ref_frame 0, fp64, generate mode");
    if (p->precision == 1) {
        REQUIRE(p->algo_mask != 0, "mc_run: the fp32 kernel needs an algorithm");
        REQUIRE(!(p->out_proc[0] || p->out_proc[1] || p->out_end_ned[0] || p->out_end_ned[1] || p->wave_trace),
                "mc_run: the fp32 kernel has no online process statistics, NED record or wave trace");
        REQUIRE(p->block_threads == 0 || p->block_threads == 256, "mc_run: the fp32 kernel takes block_threads 0 or 256 (256 = the plain kernel)");
    }
    return GINSIM_OK;
}

extern "C" {

int ginsim_aux_sensors(ginsim_ctx* c, const ginsim_aux_params* p) {
    REQUIRE(c && p, "aux_sensors: NULL argument");
    REQUIRE(p->runs >= 1 && p->n >= 0 && p->m >= 0, "aux_sensors: bad sizes");
    REQUIRE(!p->out_gps || p->ref_gps, "aux_sensors: ref_gps missing");
    REQUIRE(!p->out_mag || p->ref_mag, "aux_sensors: ref_mag missing");
    REQUIRE(p->n <= 0xFFFFFFFFll && p->m <= 0xFFFFFFFFll, "aux_sensors: sample index exceeds the RNG counter");
    REQUIRE((double)p->n * (double)p->runs < 5.0e11 && (double)p->m * (double)p->runs < 5.0e11, "aux_sensors: too many elements");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_aux(*p, c->stream));
    return GINSIM_OK;
}

int ginsim_end_stats(ginsim_ctx* c, const double* end_err, int64_t runs, ginsim_stats* host_out) {
    REQUIRE(c && end_err && host_out && runs >= 1, "end_stats: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    void* ws = nullptr;
    HIP_TRY(scratch(c, 0, stats_scratch_bytes(runs), &ws));
    HIP_TRY(launch_end_stats(end_err, runs, ws, c->stream));
    const char* res = reinterpret_cast<char*>(ws) + stats_scratch_bytes(runs) - sizeof(ginsim_stats);
    HIP_TRY(hipMemcpyAsync(host_out, res, sizeof(ginsim_stats), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_end_stats_begin(ginsim_ctx* c, const double* end_err, int64_t runs, int32_t slot) {
    REQUIRE(c && end_err && runs >= 1 && slot >= 0 && slot < 8, "end_stats_begin: bad arguments");
    REQUIRE(!c->stat_pending[slot], "end_stats_begin: slot %d is still pending (call ginsim_end_stats_finish first)", slot);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->stat_slots) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->stat_slots), 8 * sizeof(ginsim_stats), hipHostMallocDefault));
    if (!c->stat_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&c->stat_ev[slot], hipEventDisableTiming));
    void* ws = nullptr;
    HIP_TRY(scratch(c, 0, stats_scratch_bytes(runs), &ws));
    HIP_TRY(launch_end_stats(end_err, runs, ws, c->stream));
    const char* res = reinterpret_cast<char*>(ws) + stats_scratch_bytes(runs) - sizeof(ginsim_stats);
    HIP_TRY(hipMemcpyAsync(&c->stat_slots[slot], res, sizeof(ginsim_stats), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->stat_ev[slot], c->stream));
    c->stat_pending[slot] = true;
    return GINSIM_OK;
}

int ginsim_end_stats_finish(ginsim_ctx* c, int32_t slot, ginsim_stats* host_out) {
    REQUIRE(c && host_out && slot >= 0 && slot < 8, "end_stats_finish: bad arguments");
    REQUIRE(c->stat_pending[slot], "end_stats_finish: nothing was begun in slot %d", slot);
    HIP_TRY(hipEventSynchronize(c->stat_ev[slot]));
    *host_out = c->stat_slots[slot];
    c->stat_pending[slot] = false;
    return GINSIM_OK;
}

// ---- multi-GPU exchange behind the ABI (RCCL on the context's stream; csrc/comm.cpp)
int ginsim_comm_unique_id(unsigned char* id) {
    REQUIRE(id, "comm_unique_id: NULL output");
    const char* err = comm_unique_id(id);
    if (err) { set_error("comm_unique_id: %s", err); return GINSIM_ERR_HIP; }
    return GINSIM_OK;
}

int ginsim_comm_init(ginsim_ctx* c, int32_t nranks, int32_t rank, const unsigned char* id) {
    REQUIRE(c && id && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments");
    REQUIRE(!c->comm, "comm_init: this context already has a communicator");
    HIP_TRY(hipSetDevice(c->device));
    // the buffers first: a failure here must not leave a communicator behind (the other ranks would already be inside the
    // collective ncclCommInitRank, and a retry on this context would be refused)
    const size_t bytes = sizeof(ginsim_stats) * 8 * (size_t)nranks;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->comm_recv), bytes);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->comm_host), bytes, hipHostMallocDefault);
    const char* err = e == hipSuccess ? comm_create(nranks, rank, id, &c->comm) : nullptr;
    if (e != hipSuccess || err) {
        if (c->comm_recv) { (void)hipFree(c->comm_recv); c->comm_recv = nullptr; }
        if (c->comm_host) { (void)hipHostFree(c->comm_host); c->comm_host = nullptr; }
        c->comm = nullptr;
        if (err) set_error("comm_init: %s", err);
        else set_error("comm_init: %s", hipGetErrorString(e));
        return GINSIM_ERR_HIP;
    }
    return GINSIM_OK;
}

int ginsim_comm_probe(void) {
    const char* err = comm_probe();
    if (err) { set_error("comm_probe: %s", err); return GINSIM_ERR_HIP; }
    return GINSIM_OK;
}

// where the FIRST workgroup of a launch on this context's stream lands: the accelerator complex die (XCD) of an MI300 / MI355X
__global__ void first_xcc_kernel(uint32_t* out) {
    if (threadIdx.x == 0) out[0] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;      // HW_REG_XCC_ID, bits 3:0
}

int ginsim_stream_first_xcc(ginsim_ctx* c, int32_t* xcc) {
    REQUIRE(c && xcc, "stream_first_xcc: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    void* ws = nullptr;
    HIP_TRY(scratch(c, 2, 64, &ws));
    hipLaunchKernelGGL(first_xcc_kernel, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<uint32_t*>(ws));
    HIP_TRY(hipGetLastError());
    uint32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, ws, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *xcc = (int32_t)v;
    return GINSIM_OK;
}

int ginsim_vib_psd_series(ginsim_ctx* c, const double* amp, int64_t period, int64_t runs, uint64_t run_offset, uint64_t seed,
                          int32_t sensor, int32_t halve_per_run, double* out) {
    REQUIRE(c && amp && out, "vib_psd_series: NULL argument");
    REQUIRE(period >= 2 && period <= 16384 && period % 2 == 0, "vib_psd_series: period %lld must be even, 2 .. 16384 (time_series_from_psd.py:36-43)",
            (long long)period);
    REQUIRE(runs >= 1 && runs <= (int64_t)0x7FFFFFFF * 64, "vib_psd_series: runs=%lld out of range", (long long)runs);
    REQUIRE(sensor == 0 || sensor == 1, "vib_psd_series: sensor must be 0 (accelerometer) or 1 (gyroscope)");
    REQUIRE(halve_per_run == 0 || halve_per_run == 1, "vib_psd_series: halve_per_run must be 0 or 1");
    for (int64_t k = 0; k < 3 * (period / 2 + 1); ++k)
        REQUIRE(std::isfinite(amp[k]) && amp[k] >= 0.0, "vib_psd_series: amplitude %lld is negative or not finite", (long long)k);
    HIP_TRY(hipSetDevice(c->device));
    void* ws = nullptr;
    HIP_TRY(scratch(c, 3, vib_psd_scratch_bytes(period, runs), &ws));
    return launch_vib_psd(c->device, c->stream, amp, period, runs, run_offset, seed, sensor, halve_per_run, ws, out);
}

int ginsim_comm_query(ginsim_ctx* c, int32_t* nranks, int32_t* rank, int32_t* device) {
    REQUIRE(c && nranks && rank && device, "comm_query: bad arguments");
    REQUIRE(c->comm, "comm_query: no communicator (ginsim_comm_init)");
    int n = -1, r = -1, d = -1;
    const char* err = comm_query(c->comm, &n, &r, &d);
    if (err) { set_error("comm_query: %s", err); return GINSIM_ERR_HIP; }
    *nranks = n; *rank = r; *device = d;
    return GINSIM_OK;
}

int ginsim_comm_destroy(ginsim_ctx* c) {
    REQUIRE(c, "comm_destroy: NULL context");
    if (!c->comm) return GINSIM_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    comm_destroy(c->comm);
    c->comm = nullptr;
    if (c->comm_recv) { (void)hipFree(c->comm_recv); c->comm_recv = nullptr; }
    if (c->comm_host) { (void)hipHostFree(c->comm_host); c->comm_host = nullptr; }
    for (bool& p : c->comm_pending) p = false;
    return GINSIM_OK;
}

int ginsim_end_stats_all_begin(ginsim_ctx* c, const double* end_err, int64_t runs, int32_t slot) {
    REQUIRE(c && runs >= 0 && (runs == 0 || end_err) && slot >= 0 && slot < 8, "end_stats_all_begin: bad arguments");
    REQUIRE(c->comm && c->comm_recv && c->comm_host, "end_stats_all_begin: no communicator (ginsim_comm_init)");
    REQUIRE(!c->comm_pending[slot], "end_stats_all_begin: slot %d is still pending (call ginsim_end_stats_all_finish first)", slot);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->comm_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&c->comm_ev[slot], hipEventDisableTiming));
    const int nranks = comm_nranks(c->comm);
    void* ws = nullptr;
    const size_t wb = stats_scratch_bytes(runs > 0 ? runs : 1);
    HIP_TRY(scratch(c, 0, wb, &ws));
    double* rec = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + wb - sizeof(ginsim_stats));
    if (runs > 0) HIP_TRY(launch_end_stats(end_err, runs, ws, c->stream));
    else HIP_TRY(hipMemsetAsync(rec, 0, sizeof(ginsim_stats), c->stream));      // a rank without runs: the empty record
    double* recv = c->comm_recv + (size_t)slot * nranks * (sizeof(ginsim_stats) / sizeof(double));
    const char* err = comm_allgather_f64(c->comm, rec, recv, sizeof(ginsim_stats) / sizeof(double), c->stream);
    if (err) { set_error("end_stats_all_begin: %s", err); return GINSIM_ERR_HIP; }
    HIP_TRY(hipMemcpyAsync(c->comm_host + (size_t)slot * nranks, recv, sizeof(ginsim_stats) * nranks, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->comm_ev[slot], c->stream));
    c->comm_pending[slot] = true;
    return GINSIM_OK;
}

int ginsim_end_stats_all_finish(ginsim_ctx* c, int32_t slot, ginsim_stats* merged) {
    REQUIRE(c && merged && slot >= 0 && slot < 8, "end_stats_all_finish: bad arguments");
    REQUIRE(c->comm && c->comm_pending[slot], "end_stats_all_finish: nothing was begun in slot %d", slot);
    HIP_TRY(hipEventSynchronize(c->comm_ev[slot]));
    c->comm_pending[slot] = false;
    const int nranks = comm_nranks(c->comm);
    std::vector<ginsim_stats> parts;
    for (int r = 0; r < nranks; ++r) {
        const ginsim_stats& p = c->comm_host[(size_t)slot * nranks + r];
        if (p.count > 0) parts.push_back(p);
    }
    memset(merged, 0, sizeof(*merged));
    if (!parts.empty()) stats_merge_host(parts.data(), (int)parts.size(), merged);     // fixed order: rank 0 .. nranks-1
    return GINSIM_OK;
}

int ginsim_process_stats(ginsim_ctx* c, const double* traj, const double* ref, int64_t n, int64_t runs, int64_t first_sample,
                         int32_t pos_ned, double* host_out) {
    REQUIRE(c && traj && ref && host_out, "process_stats: NULL argument");
    REQUIRE(n >= 1 && runs >= 1 && first_sample >= 0 && first_sample < n, "process_stats: bad sizes");
    HIP_TRY(hipSetDevice(c->device));
    void* ws = nullptr;
    const size_t bytes = sizeof(double) * 27 * (size_t)runs;
    HIP_TRY(scratch(c, 2, bytes, &ws));
    HIP_TRY(launch_process_stats(traj, ref, n, runs, first_sample, pos_ned, 1, reinterpret_cast<double*>(ws), c->stream));
    HIP_TRY(hipMemcpyAsync(host_out, ws, bytes, hipMemcpyDeviceToHost, c->stream));      // already [runs][3][9]
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_end_stats_from_traj(ginsim_ctx* c, const double* traj, const double* ref, int64_t n, int64_t runs, int32_t pos_ned,
                               ginsim_stats* host_out) {
    REQUIRE(c && traj && ref && host_out && n >= 1 && runs >= 1, "end_stats_from_traj: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    void* ws = nullptr;
    HIP_TRY(scratch(c, 2, sizeof(double) * 27 * (size_t)runs, &ws));
    // a one-sample window: the "mean" plane [9][runs] of the process kernel IS the end-point error
    HIP_TRY(launch_process_stats(traj, ref, n, runs, n - 1, pos_ned, 0, reinterpret_cast<double*>(ws), c->stream));
    return ginsim_end_stats(c, reinterpret_cast<double*>(ws) + (size_t)9 * runs, runs, host_out);
}

int ginsim_process_stats_f32(ginsim_ctx* c, const float* traj, const double* ref, int64_t n, int64_t runs, int64_t first_sample,
                             int32_t pos_ned, const double* origin, int32_t n_ini, uint64_t ini_first, double* host_out) {
    REQUIRE(c && traj && ref && origin && host_out, "process_stats_f32: NULL argument");
    REQUIRE(n >= 1 && runs >= 1 && first_sample >= 0 && first_sample < n && n_ini >= 1, "process_stats_f32: bad sizes");
    HIP_TRY(hipSetDevice(c->device));
    void* ws = nullptr;
    const size_t bytes = sizeof(double) * 27 * (size_t)runs;
    HIP_TRY(scratch(c, 2, bytes, &ws));
    HIP_TRY(launch_process_stats_f32(traj, ref, n, runs, first_sample, pos_ned, 1, reinterpret_cast<double*>(ws), origin, n_ini,
                                     ini_first, c->stream));
    HIP_TRY(hipMemcpyAsync(host_out, ws, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_end_stats_from_traj_f32(ginsim_ctx* c, const float* traj, const double* ref, int64_t n, int64_t runs, int32_t pos_ned,
                                   const double* origin, int32_t n_ini, uint64_t ini_first, ginsim_stats* host_out) {
    REQUIRE(c && traj && ref && origin && host_out && n >= 1 && runs >= 1 && n_ini >= 1, "end_stats_from_traj_f32: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    void* ws = nullptr;
    HIP_TRY(scratch(c, 2, sizeof(double) * 27 * (size_t)runs, &ws));
    HIP_TRY(launch_process_stats_f32(traj, ref, n, runs, n - 1, pos_ned, 0, reinterpret_cast<double*>(ws), origin, n_ini, ini_first,
                                     c->stream));
    return ginsim_end_stats(c, reinterpret_cast<double*>(ws) + (size_t)9 * runs, runs, host_out);
}

int ginsim_stats_merge(const ginsim_stats* parts, int32_t nparts, ginsim_stats* out) {
    REQUIRE(parts && out && nparts >= 1, "stats_merge: bad arguments");
    stats_merge_host(parts, nparts, out);
    return GINSIM_OK;
}

int ginsim_gather_runs(ginsim_ctx* c, const double* series, int32_t ncomp, int64_t n, int64_t runs,
                       const int64_t* run_ids, int32_t nsel, double* host_out) {
    REQUIRE(c && series && run_ids && host_out, "gather_runs: NULL argument");
    REQUIRE(ncomp >= 1 && n >= 1 && runs >= 1 && nsel >= 1, "gather_runs: bad sizes");
    for (int i = 0; i < nsel; ++i)
        REQUIRE(run_ids[i] >= 0 && run_ids[i] < runs, "gather_runs: run id %lld out of range", (long long)run_ids[i]);
    HIP_TRY(hipSetDevice(c->device));
    DevBuf ids, out;
    const size_t out_bytes = sizeof(double) * (size_t)nsel * n * ncomp;
    HIP_TRY(ids.alloc(sizeof(int64_t) * nsel));
    HIP_TRY(out.alloc(out_bytes));
    HIP_TRY(hipMemcpyAsync(ids.p, run_ids, sizeof(int64_t) * nsel, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_gather_runs(series, ncomp, n, runs, ids.as<int64_t>(), nsel, out.as<double>(), c->stream));
    HIP_TRY(hipMemcpyAsync(host_out, out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_gather_series(ginsim_ctx* c, const double* series, int32_t ncomp, int64_t n, int64_t runs, const int64_t* run_ids,
                         int32_t nsel, double* host_out) {
    REQUIRE(c && series && run_ids && host_out, "gather_series: NULL argument");
    REQUIRE(ncomp >= 1 && n >= 1 && runs >= 1 && nsel >= 1, "gather_series: bad sizes");
    for (int i = 0; i < nsel; ++i)
        REQUIRE(run_ids[i] >= 0 && run_ids[i] < runs, "gather_series: run id %lld out of range", (long long)run_ids[i]);
    HIP_TRY(hipSetDevice(c->device));
    DevBuf ids, out;
    const size_t out_bytes = sizeof(double) * (size_t)nsel * n * ncomp;
    HIP_TRY(ids.alloc(sizeof(int64_t) * nsel));
    HIP_TRY(out.alloc(out_bytes));
    HIP_TRY(hipMemcpyAsync(ids.p, run_ids, sizeof(int64_t) * nsel, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_gather_series(series, ncomp, n, ids.as<int64_t>(), nsel, out.as<double>(), c->stream));
    HIP_TRY(hipMemcpyAsync(host_out, out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_gather_runs_f32(ginsim_ctx* c, const float* series, int32_t ncomp, int64_t n, int64_t runs,
                           const int64_t* run_ids, int32_t nsel, double* host_out) {
    REQUIRE(c && series && run_ids && host_out, "gather_runs_f32: NULL argument");
    REQUIRE(ncomp >= 1 && n >= 1 && runs >= 1 && nsel >= 1, "gather_runs_f32: bad sizes");
    for (int i = 0; i < nsel; ++i)
        REQUIRE(run_ids[i] >= 0 && run_ids[i] < runs, "gather_runs_f32: run id %lld out of range", (long long)run_ids[i]);
    HIP_TRY(hipSetDevice(c->device));
    DevBuf ids, out;
    const size_t out_bytes = sizeof(double) * (size_t)nsel * n * ncomp;
    HIP_TRY(ids.alloc(sizeof(int64_t) * nsel));
    HIP_TRY(out.alloc(out_bytes));
    HIP_TRY(hipMemcpyAsync(ids.p, run_ids, sizeof(int64_t) * nsel, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_gather_runs_f32(series, ncomp, n, runs, ids.as<int64_t>(), nsel, out.as<double>(), c->stream));
    HIP_TRY(hipMemcpyAsync(host_out, out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_free_integration(ginsim_ctx* c, int32_t algo, int32_t ref_frame, double fs, int32_t earth_rot,
                            const double* gyro, const double* accel, const double* odo, int64_t R, int64_t n,
                            const double* ini, int32_t n_ini, int32_t ini_has_g, uint64_t ini_first, double* att,
                            double* pos, double* vel) {
    REQUIRE(c && gyro && ini && att && pos && vel, "free_integration: NULL argument");
    REQUIRE(algo == GINSIM_ALGO_FREE || algo == GINSIM_ALGO_ODO, "free_integration: algo must be one GINSIM_ALGO_* bit");
    REQUIRE(algo != GINSIM_ALGO_FREE || accel, "free_integration: accel missing");
    REQUIRE(algo != GINSIM_ALGO_ODO || odo, "free_integration: odo missing");
    REQUIRE(R >= 1 && n >= 1 && n_ini >= 1, "free_integration: bad sizes");
    HIP_TRY(hipSetDevice(c->device));
    const size_t plane = (size_t)R * n;
    DevBuf d_aos, d_gyro, d_accel, d_odo, d_ini, d_traj;
    HIP_TRY(d_aos.alloc(sizeof(double) * plane * 3));
    HIP_TRY(d_gyro.alloc(sizeof(double) * plane * 3));
    HIP_TRY(d_accel.alloc(sizeof(double) * plane * 3));
    HIP_TRY(d_odo.alloc(sizeof(double) * plane));
    HIP_TRY(d_ini.alloc(sizeof(double) * 10 * n_ini));
    HIP_TRY(d_traj.alloc(sizeof(double) * plane * 9));
    HIP_TRY(hipMemcpyAsync(d_ini.p, ini, sizeof(double) * 10 * n_ini, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_aos.p, gyro, sizeof(double) * plane * 3, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_aos_to_soa(d_aos.as<double>(), d_gyro.as<double>(), R, n, 3, c->stream));
    if (algo == GINSIM_ALGO_FREE) {
        HIP_TRY(hipMemcpyAsync(d_aos.p, accel, sizeof(double) * plane * 3, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_aos_to_soa(d_aos.as<double>(), d_accel.as<double>(), R, n, 3, c->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(d_aos.p, odo, sizeof(double) * plane, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_aos_to_soa(d_aos.as<double>(), d_odo.as<double>(), R, n, 1, c->stream));
    }
    ginsim_mc_params p;
    memset(&p, 0, sizeof(p));
    p.n = n; p.runs = R; p.fs = fs; p.ref_frame = ref_frame; p.algo_mask = algo; p.earth_rot = earth_rot;
    p.n_ini = n_ini; p.ini_first = ini_first; p.ini_has_g = ini_has_g; p.given_sensors = 1;
    p.ini = d_ini.as<double>();
    p.in_gyro = d_gyro.as<double>(); p.in_accel = d_accel.as<double>(); p.in_odo = d_odo.as<double>();
    p.out_traj[algo == GINSIM_ALGO_FREE ? 0 : 1] = d_traj.as<double>();
    const int rc = ginsim_mc_run(c, &p);
    if (rc) return rc;
    // [9][n][R] -> three host arrays [R][n][3]
    std::vector<int64_t> all(R);
    for (int64_t i = 0; i < R; ++i) all[i] = i;
    DevBuf d_ids, d_out;
    HIP_TRY(d_ids.alloc(sizeof(int64_t) * R));
    HIP_TRY(d_out.alloc(sizeof(double) * plane * 3));
    HIP_TRY(hipMemcpyAsync(d_ids.p, all.data(), sizeof(int64_t) * R, hipMemcpyHostToDevice, c->stream));
    double* host[3] = {att, pos, vel};
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(launch_gather_runs(d_traj.as<double>() + (size_t)3 * k * plane, 3, n, R, d_ids.as<int64_t>(), (int)R,
                                   d_out.as<double>(), c->stream));
        HIP_TRY(hipMemcpyAsync(host[k], d_out.p, sizeof(double) * plane * 3, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return GINSIM_OK;
}

int ginsim_allan(ginsim_ctx* c, const double* x, int64_t n, int32_t nseries, int64_t series_stride, double fs, double* tau,
                 double* avar, int32_t* ntau, int32_t cap) {
    REQUIRE(c && x && tau && avar && ntau, "allan: NULL argument");
    REQUIRE(n >= 1 && nseries >= 1 && series_stride >= n && fs > 0, "allan: bad sizes");
    // averaging factors exactly as allan.py:29-43
    const double ts = 1.0 / fs;
    const int64_t mmax = (int64_t)floor((double)n / 9.0);
    *ntau = 0;
    if ((double)mmax * ts < 1.0) return GINSIM_OK;
    std::vector<int64_t> mult;
    const int decades = (int)ceil(log10((double)mmax));
    double scale = 0.1;
    for (int i = 0; i < decades; ++i) {
        scale *= 10;
        for (int j = 1; j < 10; ++j) {
            const int64_t m = (int64_t)(j * scale);
            if (m > mmax) break;
            mult.push_back(m);
        }
    }
    const int nt = (int)mult.size();
    if (nt > cap) { set_error("allan: %d averaging factors but capacity %d", nt, cap); return GINSIM_ERR_RANGE; }
    HIP_TRY(hipSetDevice(c->device));
    const int levels = decades;
    const int64_t n1 = n / 10;
    // levels of more than one chunk: per-wavefront / per-workgroup partial sums; ONE launch at the end folds them and runs
    // the levels of at most one chunk (the last three or four).  Where the level's rows are 16-byte aligned the
    // LDS-DMA wave-pair kernel takes the level, otherwise the register-staged one.
    std::vector<AllanLevel> lvs(levels);
    AllanFold fold;
    fold.nlevels = 0;
    fold.fused_level = -1;
    fold.pad = 0;
    for (int j = 0; j < 9; ++j) fold.fused_nb[j] = 0;
    int64_t records = 0;
    struct Step { int k; int mode; };      // mode 0: allan_level_kernel, 1: wave-pair LDS-DMA kernel, 2: levels k and k+1 fused
    std::vector<Step> steps;
    {
        int64_t n_in = n, stride_in = series_stride, pow10 = 1;
        for (int k = 0; k < levels; ++k) {
            AllanLevel& lv = lvs[k];
            lv.n_in = n_in;
            lv.n_out = (k + 1 < levels) ? n_in / 10 : 0;
            lv.in_stride = stride_in;
            lv.out_stride = lv.n_out;
            for (int j = 1; j <= 9; ++j) lv.nb[j - 1] = (j * pow10 <= mmax) ? n / (j * pow10) : 0;
            lv.nchunks = allan_chunks(n_in);
            lv.chunks_per_block = allan_chunks_per_block((int64_t)lv.nchunks * nseries);
            stride_in = lv.n_out;
            n_in = lv.n_out;
            pow10 *= 10;
        }
        int k = 0;
        if (levels >= 2 && lvs[0].n_in > allan_chunk_entries() && allan_fuse_applies(x, lvs[0], lvs[1])) {
            // levels 0 and 1 in one launch: the entries of level 1 never leave the chip (round 5; csrc/allan.hip)
            const int parts = allan_fuse_parts(lvs[0]);
            fold.nparts[0] = parts;
            fold.offset[0] = records;
            records += (int64_t)parts * nseries;
            fold.nparts[1] = parts;
            fold.offset[1] = records;
            records += (int64_t)parts * nseries * (allan_fuse_record() / 9);
            fold.fused_level = 1;
            for (int j = 0; j < 9; ++j) fold.fused_nb[j] = lvs[1].nb[j];
            steps.push_back(Step{0, 2});
            k = 2;
        }
        while (k < levels && lvs[k].n_in > allan_chunk_entries()) {
            REQUIRE(k < 8, "allan: series too long");
            AllanLevel& lv = lvs[k];
            // intermediate levels live in this call's scratch region, whose rows start 256-byte aligned
            const bool dma = allan_dma_applies(k == 0 ? x : reinterpret_cast<const double*>(uintptr_t(256)), lv);
            int parts;
            if (dma) {      // four workgroups per CU: ~1024 in flight; up to 8 chunks each so that the first, exposed load is amortised
                // a level that fits ONE round of resident workgroups (1024) with at most 16 chunks each runs as one (a second,
                // partly filled round costs a whole workgroup time: 144 000 entries x 192 series 62 -> 51 us); longer levels
                // in runs of 8 chunks
                static const int cap = [] { const char* e = getenv("GINSIM_ALLAN_CPB"); return e && atoi(e) > 0 ? atoi(e) : 8; }();
                const int64_t total = (int64_t)lv.nchunks * nseries;
                const int64_t max_parts = nseries <= 1024 ? 1024 / nseries : 1;         // workgroups per series in one round
                const int64_t fit = (lv.nchunks + max_parts - 1) / max_parts;
                const bool single = fit <= 16;
                const int64_t per = single ? fit : total / 4096;
                const int64_t lim = single ? 16 : cap;
                lv.chunks_per_block = (int32_t)(per < 1 ? 1 : (per > lim ? lim : per));
                parts = allan_pair_parts(lv);
            } else {
                parts = allan_parts(lv);
            }
            fold.nparts[k] = parts;
            fold.offset[k] = records;
            records += (int64_t)parts * nseries;
            steps.push_back(Step{k, dma ? 1 : 0});
            ++k;
        }
        fold.nlevels = k;
    }
    REQUIRE(levels - fold.nlevels <= 4, "allan: internal level plan");
    struct Region { void* p; double* d() const { return reinterpret_cast<double*>(p); } } ping, pong, partial;
    const size_t b_ping = sizeof(double) * (size_t)nseries * (n1 + 1), b_pong = sizeof(double) * (size_t)nseries * (n1 / 10 + 1);
    const size_t b_part = sizeof(double) * 9 * (size_t)(records + 1), b_sums = 0;
    void* region = nullptr;
    HIP_TRY(scratch(c, 1, b_ping + b_pong + b_part + b_sums + 1024, &region));
    ping.p = region;
    pong.p = reinterpret_cast<char*>(region) + ((b_ping + 255) & ~(size_t)255);
    partial.p = reinterpret_cast<char*>(pong.p) + ((b_pong + 255) & ~(size_t)255);
    const double* in = x;
    int flip = 0;
    for (const Step& st : steps) {
        // level k+1 (<= n/10 entries per series) goes to ping, k+2 to pong, ...
        const int k = st.k;
        if (st.mode == 2) {             // level k+1 -> ping (its last workgroup per series only), level k+2 -> pong
            HIP_TRY(launch_allan_fused(in, ping.d(), pong.d(), partial.d() + 9 * fold.offset[k], partial.d() + 9 * fold.offset[k + 1],
                                       lvs[k], lvs[k + 1], nseries, c->stream));
            in = pong.d();
            flip = 2;
            continue;
        }
        double* out = (flip++ % 2 == 0) ? ping.d() : pong.d();
        if (st.mode == 1)
            HIP_TRY(launch_allan_pair(in, out, partial.d() + 9 * fold.offset[k], lvs[k], nseries, c->stream));
        else
            HIP_TRY(launch_allan_level(in, out, partial.d() + 9 * fold.offset[k], lvs[k], nseries, c->stream));
        in = out;
    }
    AllanTail t;
    t.first = fold.nlevels;
    t.nlevels = levels - fold.nlevels;
    t.in_stride = t.nlevels > 0 ? lvs[t.first].in_stride : 0;
    t.nseries = nseries;
    for (int l = 0; l < t.nlevels; ++l) {
        t.n_in[l] = lvs[t.first + l].n_in;
        for (int j = 0; j < 9; ++j) t.nb[l][j] = lvs[t.first + l].nb[j];
    }
    // the sums go straight into pinned host memory (83 KB for 192 series x 6 levels): no copy, one synchronisation
    const size_t nsums = (size_t)9 * nseries * levels;
    if (c->allan_host_doubles < nsums) {
        if (c->allan_host) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipHostFree(c->allan_host)); c->allan_host = nullptr; c->allan_host_doubles = 0; }
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->allan_host), sizeof(double) * (nsums + nsums / 4 + 64), hipHostMallocDefault));
        c->allan_host_doubles = nsums + nsums / 4 + 64;
    }
    HIP_TRY(launch_allan_finish(in, partial.d(), c->allan_host, t, fold, nseries, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const double* h = c->allan_host;
    for (int i = 0; i < nt; ++i) {
        const int64_t m = mult[i];
        int k = 0;
        int64_t p = 1;
        while (m / p >= 10) { p *= 10; ++k; }
        const int j = (int)(m / p);
        const int64_t nb = n / m;
        tau[i] = (double)m * ts;
        for (int s = 0; s < nseries; ++s) {
            const double sum = h[((size_t)k * nseries + s) * 9 + (j - 1)];
            avar[(size_t)s * cap + i] = 0.5 / (double)(nb - 1) * sum / ((double)m * (double)m);
        }
    }
    *ntau = nt;
    return GINSIM_OK;
}

int ginsim_rng_normals(ginsim_ctx* c, uint64_t seed, uint64_t run, uint32_t stream, int64_t count, double* host_z0,
                       double* host_z1, uint32_t* host_words) {
    REQUIRE(c && host_z0 && host_z1 && count >= 1, "rng_normals: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf z0, z1, w;
    HIP_TRY(z0.alloc(sizeof(double) * count));
    HIP_TRY(z1.alloc(sizeof(double) * count));
    HIP_TRY(w.alloc(sizeof(uint32_t) * 4 * count));
    HIP_TRY(launch_rng_probe(seed, run, stream, count, z0.as<double>(), z1.as<double>(),
                             host_words ? w.as<uint32_t>() : nullptr, c->stream));
    HIP_TRY(hipMemcpyAsync(host_z0, z0.p, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(host_z1, z1.p, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    if (host_words)
        HIP_TRY(hipMemcpyAsync(host_words, w.p, sizeof(uint32_t) * 4 * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

int ginsim_runs_to_series(ginsim_ctx* c, const double* series, int32_t ncomp, int64_t n, int64_t runs, double* out) {
    REQUIRE(c && series && out && series != out, "runs_to_series: bad pointers");
    REQUIRE(ncomp >= 1 && ncomp <= 65535 && n >= 1 && runs >= 1 && (runs + 63) / 64 <= 65535, "runs_to_series: bad sizes");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_runs_to_series(series, out, ncomp, n, runs, c->stream));
    return GINSIM_OK;
}

int ginsim_normal_transform(ginsim_ctx* c, const uint32_t* host_words, int64_t count, double* host_z0, double* host_z1) {
    REQUIRE(c && host_words && host_z0 && host_z1 && count >= 1, "normal_transform: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf z0, z1, w;
    HIP_TRY(z0.alloc(sizeof(double) * count));
    HIP_TRY(z1.alloc(sizeof(double) * count));
    HIP_TRY(w.alloc(sizeof(uint32_t) * 4 * count));
    HIP_TRY(hipMemcpyAsync(w.p, host_words, sizeof(uint32_t) * 4 * count, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_normal_transform(w.as<uint32_t>(), count, z0.as<double>(), z1.as<double>(), c->stream));
    HIP_TRY(hipMemcpyAsync(host_z0, z0.p, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(host_z1, z1.p, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GINSIM_OK;
}

}  // extern "C"
