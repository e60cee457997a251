// Non-overlapping Allan variance of a batch of series, one pass per decade.
//
// Restates allan.allan_var (gnss_ins_sim/allan/allan.py:18-59) as used by the Allan plugin
// (demo_algorithms/allan_analysis.py:33-49): averaging factors m = j*10^k (j = 1..9), nb = floor(n/m) bins,
// avar(m) = 0.5/(nb-1) * sum_b (mean_{b+1} - mean_b)^2, tau = m/fs.  The reference re-reads the whole series once
// per averaging factor (46 passes for 3600 s @ 400 Hz); here level k (entries = sums of 10^k samples) is read
// ONCE: a wavefront stages a chunk of 2520 = lcm(1..9) entries in LDS, so the bins of every j are aligned to the
// chunk, accumulates sum (S_{b+1} - S_b)^2 of the bin sums for all nine j, and writes the sums of 10 that form
// level k+1.  HBM traffic: 8 B per sample at level 0, a tenth of that per further level (1.11 x 8 B total).
// Allan variance is shift invariant; every chunk subtracts its first entry so that bin-sum differences do not
// cancel against a large mean (bias / Earth rate).  Partials are folded in a fixed order (no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <stdlib.h>
#include "ginsim.h"
#include "allan.hpp"

namespace ginsim {

constexpr int kChunk = 2520;        // lcm(1..9): the bins of every factor start on a chunk boundary
constexpr int kStage = 2530;        // chunk + the first 9 entries of the next one (first bin of every factor), even
constexpr int kWavesPerBlock = 2;   // 2 x 20 240 B of LDS per block -> four blocks (eight wavefronts) per CU
constexpr int kLoads = 40;          // 40 x 64 entries >= kStage

constexpr int kMaxChunksPerBlock = 4;

// sum over the adjacent-bin pairs (k, k+1), k < OWN, of (S_{k+1} - S_k)^2; g0 = global index of bin 0
template <int OWN, bool CHECK, int N>
__device__ __forceinline__ double pair_sq(const double (&S)[N], int64_t g0, int64_t nb) {
    static_assert(N >= OWN + 1, "needs the first bin of the next owner");
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
        double d = S[k + 1] - S[k];
        if (CHECK) d = (g0 + k + 1 < nb) ? d : 0.0;
        a = __builtin_fma(d, d, a);
    }
    return a;
}

// One WAVEFRONT owns a chunk.  It loads the chunk coalesced (lane l takes entries q*64 + l), drops it into its
// private LDS stage, and then every lane reads back a CONTIGUOUS segment, three times with different segment lengths,
// so that the bins of each factor are aligned with some segmentation and all bin sums are register arithmetic:
//   pass A: 63 lanes x 40 entries (+8 halo)  -> j = 1, 2, 4, 8, 5 and the sums of 10 that form the next level
//   pass B: 40 lanes x 63 entries (+9 halo)  -> j = 3, 9, 7
//   pass C: 60 lanes x 42 entries (+6 halo)  -> j = 6
// No __syncthreads anywhere (wavefronts never share data), no prefix sums.  LDS bank behaviour of the lane-strided
// reads: 63 (odd) and 42 (16-byte reads, 20-bank stride) are conflict free; 40 is read with 16-byte accesses, which
// halves its 4-way conflict.  Eight wavefronts per CU keep ~160 KB of loads in flight per CU.
template <bool CHECK>
__device__ __forceinline__ void chunk_passes(const double* __restrict__ w, int lane, int64_t c, const AllanLevel& lv,
                                             double shift, double* __restrict__ out_series, double (&acc)[9]) {
    if (lane < 63) {                                            // ---- pass A
        const double2* p = reinterpret_cast<const double2*>(w + 40 * lane);
        double e[48];
#pragma unroll
        for (int q = 0; q < 24; ++q) { const double2 t = p[q]; e[2 * q] = t.x; e[2 * q + 1] = t.y; }
        {
            double a = 0.0;
            const int64_t g0 = c * kChunk + 40 * lane;
#pragma unroll
            for (int q = 0; q < 40; ++q) {
                double d = e[q + 1] - e[q];
                if (CHECK) d = (g0 + q + 1 < lv.nb[0]) ? d : 0.0;
                a = __builtin_fma(d, d, a);
            }
            acc[0] += a;
        }
        double s2[24], s4[12], s8[6], s5[9];
#pragma unroll
        for (int k = 0; k < 24; ++k) s2[k] = e[2 * k] + e[2 * k + 1];
#pragma unroll
        for (int k = 0; k < 12; ++k) s4[k] = s2[2 * k] + s2[2 * k + 1];
#pragma unroll
        for (int k = 0; k < 6; ++k) s8[k] = s4[2 * k] + s4[2 * k + 1];
#pragma unroll
        for (int k = 0; k < 9; ++k) s5[k] = (s2[(5 * k) / 2 + (k & 1)] + s2[(5 * k) / 2 + 1 + (k & 1)]) + e[(k & 1) ? 5 * k : 5 * k + 4];
        acc[1] += pair_sq<20, CHECK>(s2, c * (kChunk / 2) + 20 * lane, lv.nb[1]);
        acc[3] += pair_sq<10, CHECK>(s4, c * (kChunk / 4) + 10 * lane, lv.nb[3]);
        acc[7] += pair_sq<5, CHECK>(s8, c * (kChunk / 8) + 5 * lane, lv.nb[7]);
        acc[4] += pair_sq<8, CHECK>(s5, c * (kChunk / 5) + 8 * lane, lv.nb[4]);
        if (out_series) {                                       // level k+1: sums of 10, unshifted
            const int64_t g = c * (kChunk / 10) + 4 * lane;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!CHECK || g + k < lv.n_out) out_series[g + k] = __builtin_fma(10.0, shift, s5[2 * k] + s5[2 * k + 1]);
        }
    }
    if (lane < 40) {                                            // ---- pass B
        const double* p = w + 63 * lane;
        double s3[24], s7[10];
        {
            double e[72];
#pragma unroll
            for (int q = 0; q < 72; ++q) e[q] = p[q];
#pragma unroll
            for (int k = 0; k < 24; ++k) s3[k] = (e[3 * k] + e[3 * k + 1]) + e[3 * k + 2];
#pragma unroll
            for (int k = 0; k < 10; ++k)
                s7[k] = ((e[7 * k] + e[7 * k + 1]) + (e[7 * k + 2] + e[7 * k + 3])) + ((e[7 * k + 4] + e[7 * k + 5]) + e[7 * k + 6]);
        }
        double s9[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s9[k] = (s3[3 * k] + s3[3 * k + 1]) + s3[3 * k + 2];
        acc[2] += pair_sq<21, CHECK>(s3, c * (kChunk / 3) + 21 * lane, lv.nb[2]);
        acc[8] += pair_sq<7, CHECK>(s9, c * (kChunk / 9) + 7 * lane, lv.nb[8]);
        acc[6] += pair_sq<9, CHECK>(s7, c * (kChunk / 7) + 9 * lane, lv.nb[6]);
    }
    if (lane < 60) {                                            // ---- pass C
        const double2* p = reinterpret_cast<const double2*>(w + 42 * lane);
        double s6[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double2 t0 = p[3 * k], t1 = p[3 * k + 1], t2 = p[3 * k + 2];
            s6[k] = ((t0.x + t0.y) + (t1.x + t1.y)) + (t2.x + t2.y);
        }
        acc[5] += pair_sq<7, CHECK>(s6, c * (kChunk / 6) + 7 * lane, lv.nb[5]);
    }
}

__global__ void __launch_bounds__(64 * kWavesPerBlock)
allan_level_kernel(const double* __restrict__ in, double* __restrict__ out, double* __restrict__ partial, const AllanLevel lv) {
    __shared__ __attribute__((aligned(16))) double stage[kWavesPerBlock][kStage];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* w = stage[wave];
    const int64_t s = blockIdx.y;
    const double* x = in + s * lv.in_stride;
    double* out_series = lv.n_out > 0 ? out + s * lv.out_stride : nullptr;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t part = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const int64_t c_begin = part * lv.chunks_per_block;
    int64_t c_end = c_begin + lv.chunks_per_block;
    if (c_end > lv.nchunks) c_end = lv.nchunks;
    for (int64_t c = c_begin; c < c_end; ++c) {
        const int64_t base = c * kChunk;
        // Shift by the first entry of the chunk: differences of bin sums are shift invariant, and a local origin
        // keeps the bin sums small (no cancellation against a large bias / Earth rate / drifted level).
        const double shift = x[base];
        const bool full = base + 64 * kLoads <= lv.n_in;
        double v[kLoads];
        if (full) {                 // forty independent 512-byte wave loads in flight
#pragma unroll
            for (int q = 0; q < kLoads; ++q) v[q] = __builtin_nontemporal_load(&x[base + q * 64 + lane]);
        } else {                    // tail of the series: clamp the address, zero what lies beyond the end
#pragma unroll
            for (int q = 0; q < kLoads; ++q) {
                const int64_t g = base + q * 64 + lane;
                const double t = x[g < lv.n_in ? g : lv.n_in - 1];
                v[q] = g < lv.n_in ? t : shift;
            }
        }
#pragma unroll
        for (int q = 0; q < kLoads; ++q) v[q] -= shift;
#pragma unroll
        for (int q = 0; q < kLoads; ++q) {
            const int i = q * 64 + lane;
            if (i < kStage) w[i] = v[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // interior chunk: the pair (last bin of this chunk, first bin of the next) exists for every factor
        bool interior = lv.n_out == 0 || (c + 1) * (kChunk / 10) <= lv.n_out;
#pragma unroll
        for (int j = 1; j <= 9; ++j) interior = interior && ((c + 1) * (kChunk / j) + 1 <= lv.nb[j - 1]);
        if (interior) chunk_passes<false>(w, lane, c, lv, shift, out_series, acc);
        else chunk_passes<true>(w, lane, c, lv, shift, out_series, acc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();        // the stage is rewritten by the next chunk
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int64_t nparts = (int64_t)gridDim.x * kWavesPerBlock;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double a = acc[j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (lane == 0) partial[(s * nparts + part) * 9 + j] = a;
    }
}

// Levels that fit one chunk (n_in <= 2520; for 3600 s @ 400 Hz the levels of 1440, 144 and 14 entries) are finished by
// ONE wavefront per series in a single launch: the sums of 10 go to a second LDS stage instead of HBM and become
// the next level in place.  Sums are final (one wavefront saw the whole level): written straight to sums[].
__global__ void __launch_bounds__(64) allan_tail_kernel(const double* __restrict__ in, double* __restrict__ sums, const AllanTail t) {
    __shared__ __attribute__((aligned(16))) double stage[2][kStage];
    const int lane = threadIdx.x;
    const int64_t s = blockIdx.x;
    const double* x = in + s * t.in_stride;
    int cur = 0;
    {
        const int64_t n0 = t.n_in[0];
        const double shift = x[0];
        for (int i = lane; i < kStage; i += 64) stage[0][i] = i < n0 ? x[i] - shift : 0.0;
    }
    double shift = x[0];
    for (int l = 0; l < t.nlevels; ++l) {
        AllanLevel lv;
        lv.n_in = t.n_in[l];
        lv.n_out = (l + 1 < t.nlevels) ? t.n_in[l + 1] : 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) lv.nb[j] = t.nb[l][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double* nxt = stage[cur ^ 1];
        chunk_passes<true>(stage[cur], lane, 0, lv, shift, lv.n_out > 0 ? nxt : nullptr, acc);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            double a = acc[j];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            if (lane == 0) sums[((int64_t)(t.first + l) * t.nseries + s) * 9 + j] = a;
        }
        if (lv.n_out > 0) {      // the sums of 10 (unshifted) become the next level: shift by its first entry, zero the rest
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            shift = nxt[0];
            double v[kLoads];
#pragma unroll
            for (int q = 0; q < kLoads; ++q) {
                const int i = q * 64 + lane;
                v[q] = (i < lv.n_out) ? nxt[i < kStage ? i : 0] - shift : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int q = 0; q < kLoads; ++q) {
                const int i = q * 64 + lane;
                if (i < kStage) nxt[i] = v[q];
            }
            cur ^= 1;
        }
    }
}

// one 64-lane block per (series, level): lanes stride over the wavefront partials of each factor, then a fixed butterfly
__global__ void allan_fold_kernel(const double* __restrict__ partial, double* __restrict__ sums, const AllanFold f) {
    const int64_t s = blockIdx.x, nseries = gridDim.x;
    const int k = blockIdx.y, nparts = f.nparts[k];
    const double* p = partial + (f.offset[k] + s * nparts) * 9;
    for (int j = 0; j < 9; ++j) {
        double a = 0.0;
        for (int c = threadIdx.x; c < nparts; c += 64) a += p[c * 9 + j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (threadIdx.x == 0) sums[((int64_t)k * nseries + s) * 9 + j] = a;
    }
}

int allan_parts(const AllanLevel& lv) {
    const int per_block = lv.chunks_per_block * kWavesPerBlock;
    return (lv.nchunks + per_block - 1) / per_block * kWavesPerBlock;
}

// one level that spans more than one chunk: partial records of 9 doubles, allan_parts(lv) per series
hipError_t launch_allan_level(const double* in, double* out, double* partial, const AllanLevel& lv, int64_t nseries, hipStream_t st) {
    hipLaunchKernelGGL(allan_level_kernel, dim3((unsigned)(allan_parts(lv) / kWavesPerBlock), (unsigned)nseries),
                       dim3(64 * kWavesPerBlock), 0, st, in, out, partial, lv);
    return hipGetLastError();
}

hipError_t launch_allan_fold(const double* partial, double* sums, const AllanFold& f, int64_t nseries, hipStream_t st) {
    if (f.nlevels > 0)
        hipLaunchKernelGGL(allan_fold_kernel, dim3((unsigned)nseries, (unsigned)f.nlevels), dim3(64), 0, st, partial, sums, f);
    return hipGetLastError();
}

hipError_t launch_allan_tail(const double* in, double* sums, const AllanTail& t, hipStream_t st) {
    hipLaunchKernelGGL(allan_tail_kernel, dim3((unsigned)t.nseries), dim3(64), 0, st, in, sums, t);
    return hipGetLastError();
}

int allan_chunk_entries() { return kChunk; }
int allan_chunks(int64_t n_in) { return (int)((n_in + kChunk - 1) / kChunk); }

// enough wavefronts to fill the chip several times over before chunks are serialised inside a wavefront
// (measured at 192 x 1 440 000: 1 / 2 / 4 / 8 / 16 chunks per wavefront -> 0.73 / 0.67 / 0.69 / 0.72 / 0.74 ms)
int allan_chunks_per_block(int64_t total_chunks) {
    static const int forced = [] { const char* e = getenv("GINSIM_ALLAN_CPW"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    const int64_t c = total_chunks / 32768;
    return (int)(c < 1 ? 1 : (c > kMaxChunksPerBlock ? kMaxChunksPerBlock : c));
}

}  // namespace ginsim
