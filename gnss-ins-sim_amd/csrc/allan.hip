// Non-overlapping Allan variance of a batch of series, one pass per decade.
//
// Restates allan.allan_var (gnss_ins_sim/allan/allan.py:18-59) as used by the Allan plugin
// (demo_algorithms/allan_analysis.py:33-49): averaging factors m = j*10^k (j = 1..9), nb = floor(n/m) bins,
// avar(m) = 0.5/(nb-1) * sum_b (mean_{b+1} - mean_b)^2, tau = m/fs.  The reference re-reads the whole series once
// per averaging factor (46 passes for 3600 s @ 400 Hz); here level k (entries = sums of 10^k samples) is read
// ONCE: a 256-thread block stages a chunk of 2520 = lcm(1..9) entries in LDS, so the bins of every j are aligned
// to the chunk, accumulates sum (S_{b+1} - S_b)^2 of the bin sums for all nine j, and writes the sums of 10 that
// form level k+1.  HBM traffic: 8 B per sample at level 0, a tenth of that per further level (1.11 x 8 B total).
// Allan variance is shift invariant; level 0 subtracts the first sample of the series so that bin-sum differences
// do not cancel against a large mean (bias / Earth rate).  Partials are folded in a fixed order (no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "ginsim.h"

namespace ginsim {

constexpr int kChunk = 2520;        // lcm(1..9)
constexpr int kHalo = 16;           // >= 9 entries of the next chunk (first bin of every j), padded
constexpr int kAllanBlock = 256;

struct AllanLevel {
    int64_t n_in;           // entries of this level per series
    int64_t n_out;          // entries of the next level per series (n_in / 10), 0 = do not emit
    int64_t in_stride;      // series stride of the input (entries)
    int64_t out_stride;     // series stride of the output
    int64_t nb[9];          // valid bins for j = 1..9 at this level (0 = factor not evaluated)
    int32_t chunks_per_block;   // chunks folded into one set of accumulators before the block reduction
    int32_t nchunks;
};

#ifndef GINSIM_ALLAN_UNROLL
#define GINSIM_ALLAN_UNROLL 2
#endif
constexpr int kMaxChunksPerBlock = 8;
constexpr int kPer = 10;                // entries owned by a thread: 254 threads cover 2536 >= kChunk + kHalo

__device__ __forceinline__ int pad(int i) { return i + (i >> 5); }   // one spare slot per 32 entries (bank spread)

// sum over the pairs of adjacent bins of size J inside one chunk, from the padded exclusive prefix in LDS.
// Lane t handles bins t, t+256, ...: the padded index of bin b+256 is that of bin b plus the CONSTANT 264*J
// (256*J is a multiple of 32), so the three indices are formed once and every read uses an immediate offset.
template <int J, bool CHECK>
__device__ __forceinline__ double pair_sum(const double* __restrict__ pre, int tid, int64_t g0, int64_t nb) {
    constexpr int bins = kChunk / J;
    constexpr int step = kAllanBlock * J + (kAllanBlock * J) / 32;
    const int i0 = J * tid;
    const double* p0 = pre + pad(i0);
    const double* p1 = pre + pad(i0 + J);
    const double* p2 = pre + pad(i0 + 2 * J);
    double a = 0.0;
#pragma unroll GINSIM_ALLAN_UNROLL
    for (int it = 0; it * kAllanBlock < bins; ++it) {
        const int b = it * kAllanBlock + tid;
        const bool in_chunk = ((it + 1) * kAllanBlock <= bins) || (b < bins);
        if (in_chunk && (!CHECK || g0 + b + 1 < nb)) {
            const double m = p1[it * step];
            const double d = (p2[it * step] - m) - (m - p0[it * step]);       // S_{b+1} - S_b
            a = __builtin_fma(d, d, a);
        }
    }
    return a;
}

// factors whose bins do not line up with the 10-entry ownership of a thread: from the LDS prefix
template <bool CHECK>
__device__ __forceinline__ void lds_pairs(const double* __restrict__ pre, int tid, int64_t c, const AllanLevel& lv, double (&acc)[9]) {
    acc[2] += pair_sum<3, CHECK>(pre, tid, c * (kChunk / 3), lv.nb[2]);
    acc[3] += pair_sum<4, CHECK>(pre, tid, c * (kChunk / 4), lv.nb[3]);
    acc[5] += pair_sum<6, CHECK>(pre, tid, c * (kChunk / 6), lv.nb[5]);
    acc[6] += pair_sum<7, CHECK>(pre, tid, c * (kChunk / 7), lv.nb[6]);
    acc[7] += pair_sum<8, CHECK>(pre, tid, c * (kChunk / 8), lv.nb[7]);
    acc[8] += pair_sum<9, CHECK>(pre, tid, c * (kChunk / 9), lv.nb[8]);
}

// factors 1, 2 and 5 divide 10: their bins are aligned with the thread's own entries v[0..9]; v[10..14] are the
// first five entries of the next thread (the halo for the last owner).  60 % of all bin pairs, no LDS traffic.
template <bool CHECK>
__device__ __forceinline__ void reg_pairs(const double (&v)[15], int tid, int64_t c, const AllanLevel& lv, double (&acc)[9]) {
    const int64_t g1 = c * kChunk + 10 * tid, g2 = c * (kChunk / 2) + 5 * tid, g5 = c * (kChunk / 5) + 2 * tid;
    double a1 = 0.0, a2 = 0.0, a5 = 0.0;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        double d = v[q + 1] - v[q];
        if (CHECK) d = (g1 + q + 1 < lv.nb[0]) ? d : 0.0;
        a1 = __builtin_fma(d, d, a1);
    }
    double s2[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s2[k] = v[2 * k] + v[2 * k + 1];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double d = s2[k + 1] - s2[k];
        if (CHECK) d = (g2 + k + 1 < lv.nb[1]) ? d : 0.0;
        a2 = __builtin_fma(d, d, a2);
    }
    double s5[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) s5[k] = ((v[5 * k] + v[5 * k + 1]) + (v[5 * k + 2] + v[5 * k + 3])) + v[5 * k + 4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double d = s5[k + 1] - s5[k];
        if (CHECK) d = (g5 + k + 1 < lv.nb[4]) ? d : 0.0;
        a5 = __builtin_fma(d, d, a5);
    }
    acc[0] += a1;
    acc[1] += a2;
    acc[4] += a5;
}

#ifndef GINSIM_ALLAN_WAVES
#define GINSIM_ALLAN_WAVES 3
#endif

// coalesced read of one chunk into registers: lane l takes entries q*256 + l (512 contiguous bytes per wave-load)
__device__ __forceinline__ void load_chunk(const double* __restrict__ x, int64_t base, int64_t n_in, int tid,
                                           double (&w)[kPer], double& first_entry) {
    const bool full = base + kAllanBlock * kPer <= n_in;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int64_t g = base + q * kAllanBlock + tid;
        w[q] = (full || g < n_in) ? x[g] : 0.0;
    }
    first_entry = x[base];
}

__global__ void __launch_bounds__(kAllanBlock, GINSIM_ALLAN_WAVES)
allan_level_kernel(const double* __restrict__ in, double* __restrict__ out, double* __restrict__ partial, const AllanLevel lv) {
    // raw chunk (2560 entries as loaded), then its padded exclusive prefix (2537 + 80 pad slots)
    __shared__ double pre[kAllanBlock * kPer + (kAllanBlock * kPer) / 32 + 8];
    __shared__ double wave_tot[kAllanBlock / 64];
    __shared__ double red[kAllanBlock / 64][9];
    const int tid = threadIdx.x;
    const int64_t s = blockIdx.y;
    const double* x = in + s * lv.in_stride;
    const int first = tid * kPer;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t c_begin = (int64_t)blockIdx.x * lv.chunks_per_block;
    int64_t c_end = c_begin + lv.chunks_per_block;
    if (c_end > lv.nchunks) c_end = lv.nchunks;
    // software pipeline: the global loads of chunk c+1 are in flight while chunk c is scanned and differenced
    double nxt[kPer], nxt_first;
    load_chunk(x, c_begin * kChunk, lv.n_in, tid, nxt, nxt_first);
    for (int64_t c = c_begin; c < c_end; ++c) {
        // Shift by the first entry of the chunk: differences of bin sums are shift invariant, and a local origin
        // keeps the prefix sums small so that P[a] - P[b] does not cancel against a large level (bias, drift).
        const double shift = nxt_first;
        const int64_t base = c * kChunk;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int i = q * kAllanBlock + tid;
            pre[i] = (base + i < lv.n_in) ? nxt[q] - shift : 0.0;
        }
        if (c + 1 < c_end) load_chunk(x, (c + 1) * kChunk, lv.n_in, tid, nxt, nxt_first);
        __syncthreads();
        double v[15];       // own ten entries and the first five of the next owner
#pragma unroll
        for (int q = 0; q < 15; ++q) v[q] = (first + q < kAllanBlock * kPer) ? pre[first + q] : 0.0;
        if (lv.n_out > 0 && first < kChunk) {   // level k+1: sums of 10 (aligned: 2520 = 252 * 10), unshifted
            const int64_t g = c * (kChunk / 10) + tid;
            if (g < lv.n_out) {
                const double t10 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) + (v[8] + v[9]);
                out[s * lv.out_stride + g] = __builtin_fma(10.0, shift, t10);
            }
        }
        // block-wide exclusive prefix: serial in the thread, shuffle scan across the wave, LDS across the 4 waves
        double run = 0.0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) run += v[q];
        double scan = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = __shfl_up(scan, d, 64);
            if ((tid & 63) >= d) scan += up;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = scan;
        __syncthreads();                        // every thread has picked up its raw entries: pre[] may be overwritten
        double offset = scan - run;
        for (int w = 0; w < (tid >> 6); ++w) offset += wave_tot[w];
        if (first < kChunk + kHalo + 8 - kPer) {
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                pre[pad(first + q)] = offset;
                offset += v[q];
            }
            if (first + kPer == ((kChunk + kHalo) / kPer + 1) * kPer) pre[pad(first + kPer)] = offset;
        }
        __syncthreads();
        // interior chunk: the pair (last bin of this chunk, first bin of the next) exists for every evaluated j
        bool interior = true;
#pragma unroll
        for (int j = 1; j <= 9; ++j) interior = interior && ((c + 1) * (kChunk / j) + 1 <= lv.nb[j - 1]);
        if (first < kChunk) {                   // owners of chunk entries (threads 252..255 only hold the halo)
            if (interior) reg_pairs<false>(v, tid, c, lv, acc);
            else reg_pairs<true>(v, tid, c, lv, acc);
        }
        if (interior) lds_pairs<false>(pre, tid, c, lv, acc);
        else lds_pairs<true>(pre, tid, c, lv, acc);
        __syncthreads();                        // pre[] and wave_tot[] are rewritten by the next chunk
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double a = acc[j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if ((tid & 63) == 0) red[tid >> 6][j] = a;
    }
    __syncthreads();
    if (tid < 9) {
        double a = 0.0;
        for (int w = 0; w < kAllanBlock / 64; ++w) a += red[w][tid];
        partial[(s * gridDim.x + blockIdx.x) * 9 + tid] = a;
    }
}

// one 64-lane block per series: lanes stride over the block partials of each factor, then a fixed butterfly
__global__ void allan_fold_kernel(const double* __restrict__ partial, int nparts, double* __restrict__ sums) {
    const int64_t s = blockIdx.x;
    for (int j = 0; j < 9; ++j) {
        double a = 0.0;
        for (int c = threadIdx.x; c < nparts; c += 64) a += partial[(s * nparts + c) * 9 + j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (threadIdx.x == 0) sums[s * 9 + j] = a;
    }
}

hipError_t launch_allan_level(const double* in, double* out, double* partial, const AllanLevel& lv, int64_t nseries,
                              double* sums, hipStream_t st) {
    const int nblocks = (lv.nchunks + lv.chunks_per_block - 1) / lv.chunks_per_block;
    hipLaunchKernelGGL(allan_level_kernel, dim3((unsigned)nblocks, (unsigned)nseries), dim3(kAllanBlock), 0, st, in, out,
                       partial, lv);
    hipLaunchKernelGGL(allan_fold_kernel, dim3((unsigned)nseries), dim3(64), 0, st, partial, nblocks, sums);
    return hipGetLastError();
}

int allan_chunks(int64_t n_in) { return (int)((n_in + kChunk - 1) / kChunk); }

// enough blocks to fill the chip (>= ~16 per CU) before chunks are serialised inside a block
int allan_chunks_per_block(int64_t total_chunks) {
    const int64_t c = total_chunks / 4096;
    return (int)(c < 1 ? 1 : (c > kMaxChunksPerBlock ? kMaxChunksPerBlock : c));
}

}  // namespace ginsim
