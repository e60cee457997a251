// End-point error statistics over Monte-Carlo runs: count, mean, M2 (sum of squared deviations) and
// max|e| for the 9 error components, reduced with wavefront shuffles (Chan/Welford merges).
//
// Restates InsDataMgr.__end_point_error_stats + __array_stats
// (gnss_ins_sim/sim/ins_data_manager.py:717-759, 797-808): {'max': max|e|, 'avg': mean, 'std': std(ddof=0)}.
// (count, mean, M2, max) partials are mergeable, so per-device results combine exactly across GPUs.
#include <hip/hip_runtime.h>
#include "ginsim.h"
#include "fastmath.hpp"
#include "ins_math.hpp"

namespace ginsim {

struct Mom { double n, mean, m2, mx; };

__host__ __device__ inline Mom merge(const Mom& a, const Mom& b) {
    const double n = a.n + b.n;
    if (n == 0.0) return Mom{0.0, 0.0, 0.0, 0.0};
    const double d = b.mean - a.mean;
    Mom o;
    o.n = n;
    o.mean = a.mean + d * (b.n / n);
    o.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / n);
    o.mx = a.mx > b.mx ? a.mx : b.mx;
    return o;
}

__device__ inline Mom shfl_xor(const Mom& m, int mask) {
    return Mom{__shfl_xor(m.n, mask, 64), __shfl_xor(m.mean, mask, 64), __shfl_xor(m.m2, mask, 64),
               __shfl_xor(m.mx, mask, 64)};
}

constexpr int kStatBlock = 256;

// grid = (blocks, 9); partial[(comp*blocks + block)] = moments of that block's strided slice
__global__ void __launch_bounds__(kStatBlock) stats_partial_kernel(const double* __restrict__ e, int64_t runs,
                                                                   Mom* __restrict__ partial) {
    const int comp = blockIdx.y;
    const double* x = e + (int64_t)comp * runs;
    Mom m{0.0, 0.0, 0.0, 0.0};
    for (int64_t r = (int64_t)blockIdx.x * kStatBlock + threadIdx.x; r < runs; r += (int64_t)gridDim.x * kStatBlock) {
        const double v = x[r];
        m.n += 1.0;
        const double d = v - m.mean;
        m.mean += d / m.n;
        m.m2 += d * (v - m.mean);
        const double av = fabs(v);
        m.mx = av > m.mx ? av : m.mx;
    }
#pragma unroll
    for (int mask = 32; mask >= 1; mask >>= 1) m = merge(m, shfl_xor(m, mask));
    __shared__ Mom wave_part[kStatBlock / 64];
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        Mom t = wave_part[0];
        for (int w = 1; w < kStatBlock / 64; ++w) t = merge(t, wave_part[w]);
        partial[comp * gridDim.x + blockIdx.x] = t;
    }
}

// one 64-lane block per component: lanes fold the block partials they own (stride 64, fixed order), then the same
// butterfly of Chan merges as above -- deterministic, and ~10x shorter than one lane walking all partials
__global__ void stats_final_kernel(const Mom* __restrict__ partial, int blocks, ginsim_stats* __restrict__ out) {
    const int c = blockIdx.x;
    Mom t{0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < blocks; b += 64) t = merge(t, partial[c * blocks + b]);
#pragma unroll
    for (int mask = 32; mask >= 1; mask >>= 1) t = merge(t, shfl_xor(t, mask));
    if (threadIdx.x == 0) {
        if (c == 0) out->count = t.n;
        out->mean[c] = t.mean;
        out->m2[c] = t.m2;
        out->maxabs[c] = t.mx;
    }
}

// Process-error statistics: the time axis of 64 runs is cut into kSeg segments, one wavefront each (coalesced across
// lanes, the truth row is wave-uniform -> scalar loads), a Welford accumulator per component; the segment records
// are Chan-merged in a fixed order through LDS.  HBM-bound: 72 B per sample*run read once; four wavefronts per SIMD
// at 65 536 runs hide the load latency a lone wavefront per run group would wait out every step.
// out = max|e|, mean, std (ddof = 0) as [runs][3][9] (run_major, what the host API returns) or [3][9][runs].
typedef const double __attribute__((address_space(4))) * uniform_ref;
constexpr int kSeg = 4;

// attitude.angle_range_pi with the division by 2 pi replaced by a multiplication (same result: the two range
// fix-ups after the floor absorb a quotient that lands one ulp on the other side of an integer)
__device__ __forceinline__ double angle_range_pi_mul(double x) {
    double m = x - kTwoPi * floor(x * (1.0 / kTwoPi));
    if (m >= kTwoPi) m -= kTwoPi;
    if (m < 0.0) m += kTwoPi;
    return m > kPi ? m - kTwoPi : m;
}

// T = float: the series of the fp32 kernel -- the position planes hold the DISPLACEMENT from the run's initial position
// (ginsim_mc_params.precision), so the position of a sample is origin[run's initial state] + displacement, formed in fp64;
// everything after that (errors, moments) is the fp64 arithmetic of the double version.
struct ProcOrigin {
    const double* table;    // device [n_ini][3]: ECEF (ref_frame 1) or LLA (ref_frame 0) of every initial state, or nullptr (T = double)
    int64_t n_ini;
    uint64_t ini_first;     // the run's initial state is row (ini_first + run < n_ini ? ini_first + run : 0), as in the MC kernels
};

template <typename T>
__global__ void __launch_bounds__(64 * kSeg) process_stats_kernel(const T* __restrict__ traj, const double* __restrict__ ref,
                                                                 int64_t n, int64_t runs, int64_t j0, int pos_ned,
                                                                 int run_major, double* __restrict__ out, const ProcOrigin org) {
    __shared__ Mom part[kSeg][9][64];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 64 + lane;
    const bool active = r < runs;
    const int64_t plane = n * runs;
    const uniform_ref truth = (uniform_ref)(uintptr_t)ref;
    const int64_t len = n - j0, per = (len + kSeg - 1) / kSeg;
    const int64_t jb = j0 + seg * per, je = (jb + per < n) ? jb + per : n;
    double mean[9], m2[9], mx[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) { mean[c] = 0.0; m2[c] = 0.0; mx[c] = 0.0; }
    double cnt = 0.0;
    double o3[3] = {0.0, 0.0, 0.0};
    if (org.table && active) {
        const uint64_t call = org.ini_first + (uint64_t)r;
        const double* row = org.table + 3 * (call < (uint64_t)org.n_ini ? call : 0);
        o3[0] = row[0]; o3[1] = row[1]; o3[2] = row[2];
    }
    if (active) {
        for (int64_t j = jb; j < je; ++j) {
            double x[9], t[9], e[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) { x[c] = (double)traj[c * plane + j * runs + r]; t[c] = truth[9 * j + c]; }
            x[3] += o3[0]; x[4] += o3[1]; x[5] += o3[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {       // wrapped only when outside [-pi, pi], exactly as the online accumulator does (mc_kernel.hip wrap_pi3)
                const double d = x[c] - t[c];
                e[c] = fabs(d) <= kPi ? d : angle_range_pi_mul(d);
            }
            if (pos_ned) {
                const Vec3 d = lla_error_ned(Vec3{x[3], x[4], x[5]}, Vec3{t[3], t[4], t[5]});
                e[3] = d.x; e[4] = d.y; e[5] = d.z;
            } else {
#pragma unroll
                for (int c = 3; c < 6; ++c) e[c] = x[c] - t[c];
            }
#pragma unroll
            for (int c = 6; c < 9; ++c) e[c] = x[c] - t[c];
            cnt += 1.0;
            const double icnt = rcp_nr(cnt);
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const double d = e[c] - mean[c];
                mean[c] = __builtin_fma(d, icnt, mean[c]);
                m2[c] = __builtin_fma(d, e[c] - mean[c], m2[c]);
                const double a = fabs(e[c]);
                mx[c] = a > mx[c] ? a : mx[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) part[seg][c][lane] = Mom{cnt, mean[c], m2[c], mx[c]};
    __syncthreads();
    if (seg == 0 && active) {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            Mom t = part[0][c][lane];
#pragma unroll
            for (int k = 1; k < kSeg; ++k) t = merge(t, part[k][c][lane]);
            const double sd = t.n > 0.0 ? sqrt(t.m2 / t.n) : 0.0;
            if (run_major) {
                out[(r * 3 + 0) * 9 + c] = t.mx;
                out[(r * 3 + 1) * 9 + c] = t.mean;
                out[(r * 3 + 2) * 9 + c] = sd;
            } else {
                out[(0 * 9 + c) * runs + r] = t.mx;
                out[(1 * 9 + c) * runs + r] = t.mean;
                out[(2 * 9 + c) * runs + r] = sd;
            }
        }
    }
}

hipError_t launch_process_stats(const double* traj, const double* ref, int64_t n, int64_t runs, int64_t j0, int pos_ned,
                                int run_major, double* out, hipStream_t s) {
    hipLaunchKernelGGL((process_stats_kernel<double>), dim3((unsigned)((runs + 63) / 64)), dim3(64 * kSeg), 0, s, traj, ref, n, runs, j0,
                       pos_ned, run_major, out, ProcOrigin{nullptr, 0, 0});
    return hipGetLastError();
}

hipError_t launch_process_stats_f32(const float* traj, const double* ref, int64_t n, int64_t runs, int64_t j0, int pos_ned,
                                    int run_major, double* out, const double* origin, int64_t n_ini, uint64_t ini_first, hipStream_t s) {
    hipLaunchKernelGGL((process_stats_kernel<float>), dim3((unsigned)((runs + 63) / 64)), dim3(64 * kSeg), 0, s, traj, ref, n, runs, j0,
                       pos_ned, run_major, out, ProcOrigin{origin, n_ini, ini_first});
    return hipGetLastError();
}

int stats_blocks(int64_t runs) {
    int64_t b = (runs + kStatBlock - 1) / kStatBlock;
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

size_t stats_scratch_bytes(int64_t runs) { return sizeof(Mom) * 9 * (size_t)stats_blocks(runs) + sizeof(ginsim_stats); }

hipError_t launch_end_stats(const double* end_err, int64_t runs, void* scratch, hipStream_t s) {
    const int blocks = stats_blocks(runs);
    Mom* partial = reinterpret_cast<Mom*>(scratch);
    ginsim_stats* out = reinterpret_cast<ginsim_stats*>(partial + 9 * blocks);
    hipLaunchKernelGGL(stats_partial_kernel, dim3(blocks, 9), dim3(kStatBlock), 0, s, end_err, runs, partial);
    hipLaunchKernelGGL(stats_final_kernel, dim3(9), dim3(64), 0, s, partial, blocks, out);
    return hipGetLastError();
}

void stats_merge_host(const ginsim_stats* parts, int nparts, ginsim_stats* out) {
    for (int c = 0; c < 9; ++c) {
        Mom t{0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < nparts; ++k)
            t = merge(t, Mom{parts[k].count, parts[k].mean[c], parts[k].m2[c], parts[k].maxabs[c]});
        out->count = t.n;
        out->mean[c] = t.mean;
        out->m2[c] = t.m2;
        out->maxabs[c] = t.mx;
    }
}

}  // namespace ginsim
