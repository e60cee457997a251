// PSD vibration (ABI 8, include/ginsim.h "ginsim_vib_psd_series"): the vibration series of one 3-axis sensor for all the runs of
// a batch, made on the device before the Monte-Carlo launch that adds them.
//
// The reference (pathgen.py:479-484 accel, :541-546 gyro) calls time_series_from_psd(sxx, freq, fs, n) once per run and axis:
//   L = N/2 + 1 bins, N = n (n + 1 when n is odd), at most 16384                      (time_series_from_psd.py:36-43)
//   a = sqrt(sxx' N fs), sxx' = the PSD on the grid linspace(0, fs/2, L), interior bins halved     (:44-50; the caller's part here)
//   phi = pi randn(L);  X = a exp(i phi), extended Hermitian to N bins;  x = real(ifft(X))        (:51-57)
//   the N samples tiled to n                                                                      (:58-63; the launch reads j mod N)
// Here: one lane per (run, bin) draws the three phases of the bin -- the normals the 'random' vibration of the same sensor would draw
// at SAMPLE k (philox.hpp streams S_*_VIB_XY / _Z), so a run's phases depend on (seed, global run id) only -- and writes the three
// half-spectra; ONE batched complex-to-real transform per block of runs (hipFFT: the inverse, unnormalised; it ignores the
// imaginary parts of bins 0 and N/2 exactly as real(ifft) of the Hermitian extension does); a tiled transposition scales by 1/N and
// lays the series out [axis][sample][run], run fastest -- what a lane-per-run kernel reads coalesced.
// hipFFT is loaded at run time (dlopen): a host without it loses this entry point, nothing else.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <utility>

#include "ginsim.h"
#include "philox.hpp"

namespace ginsim {

void set_error(const char* fmt, ...);

namespace {

constexpr double kPiD = 3.14159265358979323846;

// X[(run * 3 + axis) * L + k] = amp[axis][k] (cos, sin)(pi z), z = the k-th normal of the vibration streams of that run and axis
template <uint32_t STREAM>
__global__ void __launch_bounds__(256) psd_spectrum_kernel(const double* __restrict__ amp, int64_t L, int64_t runs, uint64_t first_run,
                                                            uint64_t seed, int halve, double2* __restrict__ X) {
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (k >= L || r >= runs) return;
    const uint64_t grun = first_run + (uint64_t)r;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    double z0[2], z1[2];
    normal_pairs<STREAM, 2>(key, (uint32_t)k, z0, z1, tab);          // x = z0[0], y = z1[0], z = z0[1]: as add_vibration takes them
    const double z[3] = {z0[0], z1[0], z0[1]};
    // the reference halved the caller's own array once per earlier call (time_series_from_psd.py:49 on a PSD given on the grid):
    // run g sees interior bins of the PSD scaled by 0.5^(g + 1), the amplitude by its square root
    const double scale = (halve && k >= 1 && k <= L - 2) ? exp2(-0.5 * (double)(grun + 1)) : 1.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double s, co;
        sincos(kPiD * z[c], &s, &co);
        const double a = amp[c * L + k] * scale;
        X[((int64_t)r * 3 + c) * L + k] = double2{a * co, a * s};
    }
}

// x [(run * 3 + axis)][N] -> out [(axis * N + j) * runs_total + run0 + run] * inv_n; 64 x 64 tiles through LDS
__global__ void __launch_bounds__(256) psd_transpose_kernel(const double* __restrict__ x, int64_t N, int64_t runs, int64_t run0,
                                                             int64_t runs_total, double inv_n, double* __restrict__ out) {
    __shared__ double tile[64][65];
    const int axis = blockIdx.z;
    const int64_t j0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
    for (int i = ty; i < 64; i += 4) {                               // row i = run r0 + i, lanes along the samples
        const int64_t r = r0 + i, j = j0 + tx;
        if (r < runs && j < N) tile[i][tx] = x[((int64_t)r * 3 + axis) * N + j];
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {                               // row i = sample j0 + i, lanes along the runs
        const int64_t j = j0 + i, r = r0 + tx;
        if (r < runs && j < N) out[((int64_t)axis * N + j) * runs_total + run0 + r] = tile[tx][i] * inv_n;
    }
}

// ---- hipFFT through dlopen: the four calls this file makes
typedef struct hipfftHandle_t* fft_handle;
constexpr int kFftZ2D = 0x6c;           // HIPFFT_Z2D
struct Fft {
    void* lib = nullptr;
    int (*plan_many)(fft_handle*, int, int*, int*, int, int, int*, int, int, int, int) = nullptr;
    int (*set_stream)(fft_handle, hipStream_t) = nullptr;
    int (*exec_z2d)(fft_handle, double2*, double*) = nullptr;
    int (*destroy)(fft_handle) = nullptr;
    bool tried = false;
};
Fft g_fft;
std::mutex g_mu;
// (stream, (N, batch)) -> plan.  Per STREAM: a plan owns its work buffer, and two contexts of a process (the two launches of a Sim
// that keeps a few runs) make their series at the same time
std::map<std::pair<hipStream_t, std::pair<int64_t, int64_t>>, fft_handle> g_plans;

bool load_fft() {
    if (g_fft.tried) return g_fft.lib != nullptr;
    g_fft.tried = true;
    for (const char* name : {"libhipfft.so", "libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"}) {
        g_fft.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_fft.lib) break;
    }
    if (!g_fft.lib) return false;
    g_fft.plan_many = reinterpret_cast<decltype(g_fft.plan_many)>(dlsym(g_fft.lib, "hipfftPlanMany"));
    g_fft.set_stream = reinterpret_cast<decltype(g_fft.set_stream)>(dlsym(g_fft.lib, "hipfftSetStream"));
    g_fft.exec_z2d = reinterpret_cast<decltype(g_fft.exec_z2d)>(dlsym(g_fft.lib, "hipfftExecZ2D"));
    g_fft.destroy = reinterpret_cast<decltype(g_fft.destroy)>(dlsym(g_fft.lib, "hipfftDestroy"));
    if (!g_fft.plan_many || !g_fft.set_stream || !g_fft.exec_z2d || !g_fft.destroy) {
        dlclose(g_fft.lib);
        g_fft.lib = nullptr;
    }
    return g_fft.lib != nullptr;
}

}  // namespace

// runs per block of the batched transform: spectra and series of a block take about 2 x 256 MiB
int64_t vib_psd_block_runs(int64_t period, int64_t runs) {
    int64_t b = ((int64_t)256 << 20) / (3 * period * 8);
    b = b / 64 * 64;
    if (b < 64) b = 64;
    if (b > 32768) b = 32768;           // grid.y of the spectrum kernel
    return b < runs ? b : runs;
}
size_t vib_psd_scratch_bytes(int64_t period, int64_t runs) {
    const int64_t b = vib_psd_block_runs(period, runs), L = period / 2 + 1;
    return (size_t)(3 * L * 8) + (size_t)(b * 3 * L * 16) + (size_t)(b * 3 * period * 8) + 512;
}

// scratch: device, vib_psd_scratch_bytes(); amp: host [3][L]
int launch_vib_psd(int device, hipStream_t stream, const double* amp, int64_t period, int64_t runs, uint64_t run_offset, uint64_t seed,
                   int sensor, int halve, void* scratch, double* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!load_fft()) {
        const char* why = dlerror();            // (reading it clears it)
        set_error("vib_psd_series: libhipfft.so could not be loaded (%s): the PSD vibration needs hipFFT", why ? why : "symbols missing");
        return GINSIM_ERR_HIP;
    }
    const int64_t L = period / 2 + 1, block = vib_psd_block_runs(period, runs);
    char* ws = reinterpret_cast<char*>(scratch);
    double* amp_d = reinterpret_cast<double*>(ws);
    double2* X = reinterpret_cast<double2*>(ws + ((3 * L * 8 + 255) / 256) * 256);
    double* x = reinterpret_cast<double*>(reinterpret_cast<char*>(X) + (size_t)block * 3 * L * 16);
    hipError_t e = hipMemcpyAsync(amp_d, amp, sizeof(double) * 3 * L, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { set_error("vib_psd_series: %s", hipGetErrorString(e)); return GINSIM_ERR_HIP; }
    e = hipStreamSynchronize(stream);           // amp is the caller's pageable memory: it may go once this returns
    if (e != hipSuccess) { set_error("vib_psd_series: %s", hipGetErrorString(e)); return GINSIM_ERR_HIP; }
    for (int64_t r0 = 0; r0 < runs; r0 += block) {
        const int64_t nb = runs - r0 < block ? runs - r0 : block;
        const dim3 grid((unsigned)((L + 255) / 256), (unsigned)nb);
        if (sensor == 0)
            hipLaunchKernelGGL((psd_spectrum_kernel<S_ACC_VIB_XY>), grid, dim3(256), 0, stream, amp_d, L, nb, run_offset + (uint64_t)r0, seed, halve, X);
        else
            hipLaunchKernelGGL((psd_spectrum_kernel<S_GYR_VIB_XY>), grid, dim3(256), 0, stream, amp_d, L, nb, run_offset + (uint64_t)r0, seed, halve, X);
        e = hipGetLastError();
        if (e != hipSuccess) { set_error("vib_psd_series: %s", hipGetErrorString(e)); return GINSIM_ERR_HIP; }
        fft_handle& plan = g_plans[{stream, {period, nb}}];
        if (!plan) {
            int n1 = (int)period;
            const int rc = g_fft.plan_many(&plan, 1, &n1, nullptr, 1, 0, nullptr, 1, 0, kFftZ2D, (int)(nb * 3));
            if (rc != 0) {
                plan = nullptr;
                g_plans.erase({stream, {period, nb}});
                set_error("vib_psd_series: hipfftPlanMany(%lld points, %lld series) failed with %d", (long long)period, (long long)(nb * 3), rc);
                return rc == 2 ? GINSIM_ERR_NOMEM : GINSIM_ERR_HIP;            // HIPFFT_ALLOC_FAILED
            }
        }
        int rc = g_fft.set_stream(plan, stream);
        if (rc == 0) rc = g_fft.exec_z2d(plan, X, x);
        if (rc != 0) { set_error("vib_psd_series: hipfftExecZ2D failed with %d", rc); return GINSIM_ERR_HIP; }
        const dim3 tgrid((unsigned)((period + 63) / 64), (unsigned)((nb + 63) / 64), 3);
        hipLaunchKernelGGL(psd_transpose_kernel, tgrid, dim3(256), 0, stream, x, period, nb, r0, runs, 1.0 / (double)period, out);
        e = hipGetLastError();
        if (e != hipSuccess) { set_error("vib_psd_series: %s", hipGetErrorString(e)); return GINSIM_ERR_HIP; }
    }
    return GINSIM_OK;
}

// the plans of a context's stream go with the context (its stream is idle by then)
void vib_psd_drop_plans(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_plans.begin(); it != g_plans.end();) {
        if (it->first.first == stream) {
            if (it->second && g_fft.destroy) (void)g_fft.destroy(it->second);
            it = g_plans.erase(it);
        } else {
            ++it;
        }
    }
}

}  // namespace ginsim
