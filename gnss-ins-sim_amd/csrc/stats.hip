// End-point error statistics over Monte-Carlo runs: count, mean, M2 (sum of squared deviations) and
// max|e| for the 9 error components, reduced with wavefront shuffles (Chan/Welford merges).
//
// Restates InsDataMgr.__end_point_error_stats + __array_stats
// (gnss_ins_sim/sim/ins_data_manager.py:717-759, 797-808): {'max': max|e|, 'avg': mean, 'std': std(ddof=0)}.
// (count, mean, M2, max) partials are mergeable, so per-device results combine exactly across GPUs.
#include <hip/hip_runtime.h>
#include "ginsim.h"

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

// one block of 64 lanes, lane c < 9 folds the partials of component c in block order (deterministic)
__global__ void stats_final_kernel(const Mom* __restrict__ partial, int blocks, ginsim_stats* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= 9) return;
    Mom t = partial[c * blocks];
    for (int b = 1; b < blocks; ++b) t = merge(t, partial[c * blocks + b]);
    if (c == 0) out->count = t.n;
    out->mean[c] = t.mean;
    out->m2[c] = t.m2;
    out->maxabs[c] = t.mx;
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
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(64), 0, s, partial, blocks, out);
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
