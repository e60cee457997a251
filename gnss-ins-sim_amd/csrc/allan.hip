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

// sum over the adjacent-bin pairs (k, k+1), k < OWN, of (S_{k+1} - S_k)^2; g0 = global index of bin 0.  Four interleaved
// partial sums: a single chain of dependent fp64 FMAs runs at the FMA latency, which a wavefront that is alone on its
// SIMD (the LDS-DMA kernel) cannot hide.
template <int OWN, bool CHECK, int N>
__device__ __forceinline__ double pair_sq(const double (&S)[N], int64_t g0, int64_t nb) {
    static_assert(N >= OWN + 1, "needs the first bin of the next owner");
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
        double d = S[k + 1] - S[k];
        if (CHECK) d = (g0 + k + 1 < nb) ? d : 0.0;
        a[k & 3] = __builtin_fma(d, d, a[k & 3]);
    }
    return (a[0] + a[1]) + (a[2] + a[3]);
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
// RAW: the stage holds the entries as they are in memory (an LDS-DMA put them there) and the chunk's origin `shift` is
// subtracted as they are read; otherwise the stage was filled with shifted entries.  Same values either way.
// Every pass is "read my segment of the stage into registers" followed by arithmetic on registers only; the wave-pair
// kernel puts a workgroup barrier between the two so that the stage can be refilled while the arithmetic runs.
__device__ __forceinline__ void read_a(const double* __restrict__ w, int lane, double (&e)[48]) {
    const double2* p = reinterpret_cast<const double2*>(w + 40 * (lane < 63 ? lane : 62));
#pragma unroll
    for (int q = 0; q < 24; ++q) {
        const double2 t = p[q];
        e[2 * q] = t.x;
        e[2 * q + 1] = t.y;
    }
}

// `between(q)`, q = 0..9, is called at ten points spread over the arithmetic (the wave-pair kernel issues one LDS-DMA piece
// at each: ten in a row queue up behind the other wavefronts' pieces at the CU's address unit and the wavefront stands still)
struct NothingBetween { __device__ __forceinline__ void operator()(int) const {} };

// o4 != nullptr: the lane's four sums of 10 (entries 4 lane .. 4 lane + 3 of the next level's share of this chunk) are ALSO
// handed back in registers -- the fused kernel below keeps them on the chip instead of writing them out.
template <bool CHECK, bool RAW, class Between = NothingBetween>
__device__ __forceinline__ void compute_a(double (&e)[48], int lane, int64_t c, const AllanLevel& lv, double shift,
                                          double* __restrict__ out_series, double (&acc)[9], const Between& between = Between(),
                                          double* __restrict__ o4 = nullptr) {
    {
        if (RAW) {
#pragma unroll
            for (int q = 0; q < 48; ++q) e[q] -= shift;
        }
        between(0);
        {
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            const int64_t g0 = c * kChunk + 40 * lane;
#pragma unroll
            for (int q = 0; q < 40; ++q) {
                double d = e[q + 1] - e[q];
                if (CHECK) d = (g0 + q + 1 < lv.nb[0]) ? d : 0.0;
                a[q & 3] = __builtin_fma(d, d, a[q & 3]);
                if (q == 12) between(1);
                if (q == 25) between(2);
            }
            acc[0] += lane < 63 ? (a[0] + a[1]) + (a[2] + a[3]) : 0.0;
        }
        between(3);
        // order: the entries die after the bins of 2 and of 5, the bins of 2 after those of 4, ...
        double s2[24], s4[12], s8[6], s5[9];
        const bool mine = lane < 63;
#pragma unroll
        for (int k = 0; k < 24; ++k) s2[k] = e[2 * k] + e[2 * k + 1];
        between(4);
#pragma unroll
        for (int k = 0; k < 9; ++k) s5[k] = (s2[(5 * k) / 2 + (k & 1)] + s2[(5 * k) / 2 + 1 + (k & 1)]) + e[(k & 1) ? 5 * k : 5 * k + 4];
#pragma unroll
        for (int k = 0; k < 12; ++k) s4[k] = s2[2 * k] + s2[2 * k + 1];
        between(5);
        const double r1 = pair_sq<20, CHECK>(s2, c * (kChunk / 2) + 20 * lane, lv.nb[1]);
        acc[1] += mine ? r1 : 0.0;
        between(6);
#pragma unroll
        for (int k = 0; k < 6; ++k) s8[k] = s4[2 * k] + s4[2 * k + 1];
        const double r3 = pair_sq<10, CHECK>(s4, c * (kChunk / 4) + 10 * lane, lv.nb[3]);
        acc[3] += mine ? r3 : 0.0;
        between(7);
        const double r7 = pair_sq<5, CHECK>(s8, c * (kChunk / 8) + 5 * lane, lv.nb[7]);
        acc[7] += mine ? r7 : 0.0;
        between(8);
        const double r4 = pair_sq<8, CHECK>(s5, c * (kChunk / 5) + 8 * lane, lv.nb[4]);
        acc[4] += mine ? r4 : 0.0;
        between(9);
        if (o4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o4[k] = __builtin_fma(10.0, shift, s5[2 * k] + s5[2 * k + 1]);
        }
        if (lane < 63 && out_series) {                                       // level k+1: sums of 10, unshifted
            const int64_t g = c * (kChunk / 10) + 4 * lane;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!CHECK || g + k < lv.n_out) out_series[g + k] = __builtin_fma(10.0, shift, s5[2 * k] + s5[2 * k + 1]);
        }
    }
}

template <bool CHECK, bool RAW = false>
__device__ __forceinline__ void pass_a(const double* __restrict__ w, int lane, int64_t c, const AllanLevel& lv, double shift,
                                       double* __restrict__ out_series, double (&acc)[9], double* __restrict__ o4 = nullptr) {
    double e[48];
    read_a(w, lane, e);
    compute_a<CHECK, RAW>(e, lane, c, lv, shift, out_series, acc, NothingBetween(), o4);
}

__device__ __forceinline__ void read_b(const double* __restrict__ w, int lane, double (&e)[72]) {
    const double* p = w + 63 * (lane < 40 ? lane : 39);
#pragma unroll
    for (int q = 0; q < 72; ++q) e[q] = p[q];
}

// The same as 72 separate ds_read_b64: the compiler pairs 8-byte LDS reads into ds_read2_b64, which the LDS serves as two
// accesses of 4 x 16 lanes (8 cycles), while a lone ds_read_b64 goes in 2 x 32 lanes (2 cycles) -- and a 63-entry lane
// stride is conflict free either way.  The results are not tracked by the compiler: the caller waits (lgkmcnt) before use.
__device__ __forceinline__ void read_b_single(const double* __restrict__ w, int lane, double (&e)[72]) {
    const unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) double*)(w + 63 * (lane < 40 ? lane : 39));
#pragma unroll
    for (int q = 0; q < 72; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(e[q]) : "v"(a), "n"(8 * q));
}

// SIX: factor 6 from the same registers.  63 l is 0 or 3 modulo 6, so an even lane holds the bins of 6 that start at its
// entries 0, 6, .., 66 (it owns the 11 pairs that begin in its 63 entries) and an odd lane those at 3, 9, .., 63 (10 pairs);
// both are sums of two adjacent bins of 3.
template <bool CHECK, bool RAW, bool SIX, class Between = NothingBetween>
__device__ __forceinline__ void compute_b(double (&e)[72], int lane, int64_t c, const AllanLevel& lv, double shift, double (&acc)[9],
                                          const Between& between = Between()) {
    // every lane does the arithmetic (lanes >= 40 on a copy of lane 39's entries): `between` must run with all lanes on
    if (RAW) {
#pragma unroll
        for (int q = 0; q < 72; ++q) e[q] -= shift;
    }
    between(0);
    double s3[24], s7[10], s9[8];
    // in blocks of 21 = lcm(3, 7) entries, so that the entries die as the bins of 3 and of 7 appear
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int k = 7 * b; k < 7 * b + 7; ++k) s3[k] = (e[3 * k] + e[3 * k + 1]) + e[3 * k + 2];
#pragma unroll
        for (int k = 3 * b; k < 3 * b + 3; ++k)
            s7[k] = ((e[7 * k] + e[7 * k + 1]) + (e[7 * k + 2] + e[7 * k + 3])) + ((e[7 * k + 4] + e[7 * k + 5]) + e[7 * k + 6]);
        between(1 + b);
    }
#pragma unroll
    for (int k = 21; k < 24; ++k) s3[k] = (e[3 * k] + e[3 * k + 1]) + e[3 * k + 2];
    s7[9] = ((e[63] + e[64]) + (e[65] + e[66])) + ((e[67] + e[68]) + e[69]);
    between(4);
    const bool mine = lane < 40;
    // order: the bins of 7 and of 9 die first; the bins of 3 also make those of 6
    const double r6 = pair_sq<9, CHECK>(s7, c * (kChunk / 7) + 9 * lane, lv.nb[6]);
    acc[6] += mine ? r6 : 0.0;
    between(5);
#pragma unroll
    for (int k = 0; k < 8; ++k) s9[k] = (s3[3 * k] + s3[3 * k + 1]) + s3[3 * k + 2];
    const double r8 = pair_sq<7, CHECK>(s9, c * (kChunk / 9) + 7 * lane, lv.nb[8]);
    acc[8] += mine ? r8 : 0.0;
    between(6);
    const double r2 = pair_sq<21, CHECK>(s3, c * (kChunk / 3) + 21 * lane, lv.nb[2]);
    acc[2] += mine ? r2 : 0.0;
    between(7);
    if (SIX) {
        const bool odd = (lane & 1) != 0;
        const int64_t g0 = c * (kChunk / 6) + 21 * (lane >> 1) + (odd ? 11 : 0);
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        double prev = s3[1] + (odd ? s3[2] : s3[0]);       // bin m of 6: bins 2m, 2m+1 of 3 (even lane) or 2m+1, 2m+2 (odd lane)
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const double next = s3[2 * k + 3] + ((k < 10 && odd) ? s3[k < 10 ? 2 * k + 4 : 0] : s3[2 * k + 2]);
            double d = next - prev;
            prev = next;
            bool own = k < 10 || !odd;
            if (CHECK) own = own && (g0 + k + 1 < lv.nb[5]);
            if (CHECK || k == 10) d = own ? d : 0.0;
            a[k & 3] = __builtin_fma(d, d, a[k & 3]);
            if (k == 5) between(8);
        }
        acc[5] += mine ? (a[0] + a[1]) + (a[2] + a[3]) : 0.0;
    } else {
        between(8);
    }
    between(9);
}

template <bool CHECK, bool RAW = false>
__device__ __forceinline__ void pass_bc(const double* __restrict__ w, int lane, int64_t c, const AllanLevel& lv, double shift,
                                        double (&acc)[9]) {
    {                                                           // ---- pass B
        double e[72];
        read_b(w, lane, e);
        compute_b<CHECK, RAW, false>(e, lane, c, lv, shift, acc);
    }
    const double sh = RAW ? shift : 0.0;
    if (lane < 60) {                                            // ---- pass C
        const double2* p = reinterpret_cast<const double2*>(w + 42 * lane);
        double s6[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double2 t0 = p[3 * k], t1 = p[3 * k + 1], t2 = p[3 * k + 2];
            if (RAW) { t0.x -= sh; t0.y -= sh; t1.x -= sh; t1.y -= sh; t2.x -= sh; t2.y -= sh; }
            s6[k] = ((t0.x + t0.y) + (t1.x + t1.y)) + (t2.x + t2.y);
        }
        acc[5] += pair_sq<7, CHECK>(s6, c * (kChunk / 6) + 7 * lane, lv.nb[5]);
    }
}

template <bool CHECK, bool RAW = false>
__device__ __forceinline__ void chunk_passes(const double* __restrict__ w, int lane, int64_t c, const AllanLevel& lv,
                                             double shift, double* __restrict__ out_series, double (&acc)[9]) {
    pass_a<CHECK, RAW>(w, lane, c, lv, shift, out_series, acc);
    pass_bc<CHECK, RAW>(w, lane, c, lv, shift, acc);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// interior chunk: the pair (last bin of this chunk, first bin of the next) exists for every factor
__device__ __forceinline__ bool chunk_is_interior(int64_t c, const AllanLevel& lv) {
    bool interior = lv.n_out == 0 || (c + 1) * (kChunk / 10) <= lv.n_out;
#pragma unroll
    for (int j = 1; j <= 9; ++j) interior = interior && ((c + 1) * (kChunk / j) + 1 <= lv.nb[j - 1]);
    return interior;
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

// ---------------------------------------------------------------------------------------------------------------------
// The same level pass with the chunk brought in by LDS-DMA (global_load_lds_dwordx4: memory -> LDS without passing
// through registers) into one of TWO stages per workgroup, so that chunk c+1 is in flight while the passes run on chunk c.
// A plain streaming read reaches 7.0-7.5 TB/s on MI355X (tools/experiments/rbench.hip), also double-buffered through LDS-DMA with four
// requesting wavefronts per CU; the register-staged kernel above gets 4.6-4.9 TB/s out of eight because each wavefront
// alternates between waiting for its loads and computing.  The entries land unshifted; the passes subtract the chunk's
// origin as they read (RAW).  Chunks that do not lie wholly inside the series (the last one or two) are staged through
// registers, unshifted as well, with the origin beyond the end.
// Measured on 192 x 1 440 000 (level 0, 2.21 GB): requests only (no arithmetic) 327 us = 6.75 TB/s; arithmetic only 327 us;
// both 400-460 us depending on the box and the minute (the same binary moves by 12 % between runs).  What the two
// wavefronts of a workgroup spend per chunk, from s_memtime stamps: reading the segments ~30 %, arithmetic with the
// requests in between ~50 % (a third of that standing still on the ten requests), barriers and waiting ~20 %.
// A third wavefront that only requests (168 registers each, the 63-entry segment read in two halves) was slower: 500 us.
constexpr int kDmaPieces = 20;                      // 20 x 1 KiB (64 lanes x 16 B) >= kStage entries
constexpr int kDmaStage = kDmaPieces * 128;         // 2560 doubles
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;

// A PAIR of wavefronts per workgroup shares the two stages and splits the work on a staged chunk by PASS, not by data:
// wavefront 0 takes the 40-entry segmentation (factors 1, 2, 4, 8, 5 and the sums of 10), wavefront 1 the 63-entry one
// (3, 9, 7 and, from the same registers, 6) -- about the same number of instructions -- and each requests half of every
// chunk.  Four workgroups per CU = two wavefronts per SIMD.  Per chunk c, in stage c mod 2:
//     wait until my pieces of chunk c have landed; barrier            (chunk c is complete for both wavefronts)
//     read my segment of the stage into registers; barrier            (nobody needs the stage any more)
//     arithmetic on registers, and in between, piece by piece, the request for chunk c+2 into this stage
//                                                                     (chunk c+1 is still in flight to the other one)
// so a request has the arithmetic of two chunks to land in, and up to 40 KB per workgroup are in flight.  (Requesting
// chunk c+1 only once the arithmetic on c-1 is over -- one request in flight -- ran at the latency of the request:
// 4.3 us per chunk, of which the arithmetic was 2 us.)  An LDS-DMA is ordered for a reader by the issuing wavefront's
// vmcnt followed by a barrier the reader has passed; vector memory operations of one wavefront complete in order, so
// "all but the youngest K" is exact when K counts what was issued after the request: the next request (10 pieces) and the
// stores of the sums of 10.  A K that is too small only waits longer.
__device__ __forceinline__ void dma_request_half(const double* __restrict__ chunk, double* __restrict__ dst, int wave, int lane) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < kDmaPieces / 2; ++q) {
        const int piece = wave * (kDmaPieces / 2) + q;
        __builtin_amdgcn_global_load_lds((global_void_ptr)(chunk + piece * 128 + 2 * lane), (lds_void_ptr)(dst + piece * 128), 16, 0, 2 /* nt */);
    }
    asm volatile("" ::: "memory");
}

// one piece of my half at a time, pinned where it is called
struct DmaPieceIssue {
    const double* chunk;
    double* dst;
    int wave, lane;
    bool on;
    __device__ __forceinline__ void operator()(int q) const {
        __builtin_amdgcn_sched_barrier(0);
        if (on) {
            const int piece = wave * (kDmaPieces / 2) + q;
            __builtin_amdgcn_global_load_lds((global_void_ptr)(chunk + piece * 128 + 2 * lane), (lds_void_ptr)(dst + piece * 128), 16, 0, 2 /* nt */);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};

// the same with the decision taken at compile time: with a run-time flag that is the same for all ten pieces the compiler merges
// the ten conditional blocks and sinks the arithmetic behind them -- the pieces then go out in one burst after the barrier,
// which is exactly what the placement is there to avoid
template <bool ISSUE>
struct DmaPieceIssueT {
    const double* chunk;
    double* dst;
    int wave, lane;
    __device__ __forceinline__ void operator()(int q) const {
        __builtin_amdgcn_sched_barrier(0);
        if (ISSUE) {
            const int piece = wave * (kDmaPieces / 2) + q;
            __builtin_amdgcn_global_load_lds((global_void_ptr)(chunk + piece * 128 + 2 * lane), (lds_void_ptr)(dst + piece * 128), 16, 0, 2 /* nt */);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};

// a chunk the DMA cannot take, half per wavefront: unshifted, `shift` beyond the end (RAW passes subtract it again)
__device__ __forceinline__ void stage_ragged_half(const double* __restrict__ x, double* __restrict__ w, int wave, int lane, int64_t c,
                                                  int64_t n_in, double shift) {
    const int64_t base = c * kChunk;
#pragma unroll 1
    for (int q0 = 0; q0 < kLoads / 2; q0 += 10) {
        double v[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const int64_t g = base + (wave * (kLoads / 2) + q0 + q) * 64 + lane;
            v[q] = x[g < n_in ? g : n_in - 1];
        }
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const int i = (wave * (kLoads / 2) + q0 + q) * 64 + lane;
            w[i] = (base + i < n_in) ? v[q] : shift;
        }
    }
}

// wait until at most `younger` of my vector memory operations are outstanding (s_waitcnt takes an immediate)
__device__ __forceinline__ void wait_all_but(int younger) {
    if (younger >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if (younger >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (younger >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my LDS reads / writes are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


__global__ void __launch_bounds__(128, 2)
allan_pair_kernel(const double* __restrict__ in, double* __restrict__ out, double* __restrict__ partial, const AllanLevel lv) {
    __shared__ __attribute__((aligned(1024))) double stage[2][kDmaStage];       // 40 960 B: four workgroups per CU
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t s = blockIdx.y, part = blockIdx.x, nparts = gridDim.x;
    const double* x = in + s * lv.in_stride;
    double* out_series = lv.n_out > 0 ? out + s * lv.out_stride : nullptr;
    const int64_t c_begin = part * lv.chunks_per_block;
    int64_t c_end = c_begin + lv.chunks_per_block;
    if (c_end > lv.nchunks) c_end = lv.nchunks;
    const int64_t c_dma = (lv.n_in - kDmaStage) / kChunk;       // chunks 0 .. c_dma lie wholly inside the series
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c_begin < c_end && c_begin <= c_dma) dma_request_half(x + c_begin * kChunk, stage[0], wave, lane);
    if (c_begin + 1 < c_end && c_begin + 1 <= c_dma) dma_request_half(x + (c_begin + 1) * kChunk, stage[1], wave, lane);
    int stores1 = 0, stores2 = 0;   // store instructions this wavefront is KNOWN to have issued in the last two iterations
#pragma unroll 1
    for (int64_t c = c_begin; c < c_end; ++c) {
        double* w = stage[(c - c_begin) & 1];
        if (c > c_dma) {            // staged through registers; nothing else of mine is in flight
            stage_ragged_half(x, w, wave, lane, c, lv.n_in, x[c * kChunk]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            wait_all_but(((c + 1 < c_end && c + 1 <= c_dma) ? kDmaPieces / 2 : 0) + stores1 + stores2);
        }
        block_barrier();                                        // chunk c is in its stage, for both wavefronts
        const bool interior = chunk_is_interior(c, lv);
        const bool again = c + 2 < c_end && c + 2 <= c_dma;
        const double shift = w[0];                              // the chunk's origin, from the stage
        if (!interior) {
            // the last chunk or two of a series: bounds-checked passes straight from the stage (they keep fewer entries
            // in registers than the split form, which would not fit next to the 64-bit index compares)
            if (wave == 0) pass_a<true, true>(w, lane, c, lv, shift, out_series, acc);
            else pass_bc<true, true>(w, lane, c, lv, shift, acc);
            block_barrier();
            if (again) dma_request_half(x + (c + 2) * kChunk, w, wave, lane);
            stores2 = stores1;
            stores1 = 0;            // not known: the checked stores may be skipped
        } else if (wave == 0) {
            double e[48];
            read_a(w, lane, e);
            block_barrier();                                    // both wavefronts hold their segments: the stage is free
            const DmaPieceIssue piece{x + (c + 2) * kChunk, w, wave, lane, again};
            compute_a<false, true>(e, lane, c, lv, shift, out_series, acc, piece);
            asm volatile("" ::: "memory");
            stores2 = stores1;
            stores1 = out_series ? 4 : 0;
        } else {
            double e[72];
            read_b_single(w, lane, e);
            block_barrier();
            const DmaPieceIssue piece{x + (c + 2) * kChunk, w, wave, lane, again};
            compute_b<false, true, true>(e, lane, c, lv, shift, acc, piece);
        }
    }
    // wavefront 0 owns the factors of its segmentation (j = 1, 2, 4, 5, 8 -> records 0, 1, 3, 4, 7), wavefront 1 the others
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const bool mine = (j == 0 || j == 1 || j == 3 || j == 4 || j == 7) == (wave == 0);
        double a = acc[j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (lane == 0 && mine) partial[(s * nparts + part) * 9 + j] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Levels k and k+1 in ONE launch (round 5).  What bounds the level kernel above at config 5's size is not its pipeline but the
// HBM WRITES of the next level's entries -- 0.22 GB between 2.21 GB of reads cost 75 of level 0's 450 us (DESIGN_EXPERIMENTS
// E4) -- and the next level then costs another launch that reads them back.  Here a workgroup takes exactly TEN chunks of level
// k = 25 200 entries = ONE chunk of level k+1, bin boundaries of every factor included (2520 = lcm(1..9)), so level k+1 needs no
// alignment-free formulation (E6.1's prefix-sum form put ~200 instructions and two LDS round trips per chunk on wavefront 0's
// critical path and lost):
//   * per chunk wavefront 0 does what it did, except that its four sums of 10 per lane stay in registers (40 doubles over the
//     ten chunks, selected by a wave-uniform switch: nothing is indexed dynamically);
//   * after the last chunk both LDS stages are idle: wavefront 0 drops the 2520 entries of the level-k+1 chunk into stage 0 and
//     the wave pair runs the SAME two passes on it -- one chunk's worth of arithmetic more per ten, no request to wait for -- and
//     writes the 252 entries of level k+2 (1 % of the input).
// The pair (last bin of this level-k+1 chunk, first bin of the next) belongs to two workgroups.  Neither computes it: the
// halo of the staged chunk is its own MIRROR image (entry 2520 + i = entry 2519 - i), so that the first bin of "the next
// chunk" of every factor equals the last bin of this one and the pair contributes (rounding)^2; each workgroup records the
// sums of its first and last j entries (relative to its origin) and the finishing launch adds the boundary pairs.
// Wavefront 0's pass in TWO phases around the barrier that frees the stage, for the fused kernel: it carries 40 doubles of the
// next level across the chunk loop, and the one-phase form (48 entries live across the barrier) would spill next to them --
// and a scratch reload is a vector-memory operation: its s_waitcnt waits for every LDS-DMA piece issued before it.
//   phase 1 (stage still needed): read the 48 entries, subtract the origin, factor 1, the bins of 2; what survives the barrier
//            are 24 bins of 2 and the nine entries the bins of 5 need
//   phase 2 (requests in between): bins of 5, 4, 8 and their pair sums, the sums of 10
// Same operations in the same association as compute_a<false, true>: the level's sums do not change.
struct PassA2 { double s2[24], eo[9]; };

__device__ __forceinline__ void pass_a2_phase1(const double* __restrict__ w, int lane, double shift, PassA2& st, double& acc0) {
    double e[48];
    read_a(w, lane, e);
#pragma unroll
    for (int q = 0; q < 48; ++q) e[q] -= shift;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 40; ++q) {
        const double d = e[q + 1] - e[q];
        a[q & 3] = __builtin_fma(d, d, a[q & 3]);
    }
    acc0 += lane < 63 ? (a[0] + a[1]) + (a[2] + a[3]) : 0.0;
#pragma unroll
    for (int k = 0; k < 24; ++k) st.s2[k] = e[2 * k] + e[2 * k + 1];
#pragma unroll
    for (int k = 0; k < 9; ++k) st.eo[k] = e[(k & 1) ? 5 * k : 5 * k + 4];
}

// The empty asm statements pin the arithmetic of a stretch BEFORE the request that follows it: the pieces are volatile, the
// arithmetic is not, and without the pins the compiler sinks every stretch towards its use at the end -- all ten pieces then go
// out in one burst right after the barrier (seen in the ISA: a piece every 4 instructions), which is what the placement avoids.
template <class Between>
__device__ __forceinline__ void pass_a2_phase2(const PassA2& st, int lane, double shift, double (&acc)[9], const Between& between,
                                               double (&o4)[4]) {
    const double (&s2)[24] = st.s2;
    double s4[12], s8[6], s5[9];
    const bool mine = lane < 63;
#pragma unroll
    for (int k = 0; k < 5; ++k) s5[k] = (s2[(5 * k) / 2 + (k & 1)] + s2[(5 * k) / 2 + 1 + (k & 1)]) + st.eo[k];
    asm volatile("" : "+v"(s5[0]), "+v"(s5[1]), "+v"(s5[2]), "+v"(s5[3]), "+v"(s5[4]));
    between(0);
#pragma unroll
    for (int k = 5; k < 9; ++k) s5[k] = (s2[(5 * k) / 2 + (k & 1)] + s2[(5 * k) / 2 + 1 + (k & 1)]) + st.eo[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) s4[k] = s2[2 * k] + s2[2 * k + 1];
    asm volatile("" : "+v"(s5[5]), "+v"(s5[6]), "+v"(s5[7]), "+v"(s5[8]), "+v"(s4[0]), "+v"(s4[1]), "+v"(s4[2]), "+v"(s4[3]));
    between(1);
#pragma unroll
    for (int k = 4; k < 12; ++k) s4[k] = s2[2 * k] + s2[2 * k + 1];
    asm volatile("" : "+v"(s4[4]), "+v"(s4[5]), "+v"(s4[6]), "+v"(s4[7]), "+v"(s4[8]), "+v"(s4[9]), "+v"(s4[10]), "+v"(s4[11]));
    between(2);
    {
        // pair_sq<20> with the requests in between
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            const double d = s2[k + 1] - s2[k];
            a[k & 3] = __builtin_fma(d, d, a[k & 3]);
            if (k == 5 || k == 11 || k == 17) {
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
                between(k == 5 ? 3 : (k == 11 ? 4 : 5));
            }
        }
        const double r1 = (a[0] + a[1]) + (a[2] + a[3]);
        acc[1] += mine ? r1 : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s8[k] = s4[2 * k] + s4[2 * k + 1];
    asm volatile("" : "+v"(acc[1]), "+v"(s8[0]), "+v"(s8[1]), "+v"(s8[2]), "+v"(s8[3]), "+v"(s8[4]), "+v"(s8[5]));
    between(6);
    const double r3 = pair_sq<10, false>(s4, 0, 0);
    acc[3] += mine ? r3 : 0.0;
    asm volatile("" : "+v"(acc[3]));
    between(7);
    const double r7 = pair_sq<5, false>(s8, 0, 0);
    acc[7] += mine ? r7 : 0.0;
    asm volatile("" : "+v"(acc[7]));
    between(8);
    const double r4 = pair_sq<8, false>(s5, 0, 0);
    acc[4] += mine ? r4 : 0.0;
    asm volatile("" : "+v"(acc[4]));
    between(9);
#pragma unroll
    for (int k = 0; k < 4; ++k) o4[k] = __builtin_fma(10.0, shift, s5[2 * k] + s5[2 * k + 1]);
}

constexpr int kFuseChunks = 10;
constexpr int kFuseRecord = 36;                     // doubles per level-k+1 record: sums[9], first[9], last[9], origin, pad

// One chunk of a workgroup whose ten chunks are whole and interior.  ISSUE: request chunk q + 2 (`next`) into the stage that is
// being read, piece by piece inside the arithmetic; last: nothing younger than this chunk's pieces is in flight.
template <bool ISSUE>
__device__ __forceinline__ void fused_step_a(const double* __restrict__ next, double* __restrict__ w, int lane, double (&acc)[9],
                                             double (&o4)[4], bool last = false) {
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");         // all but the next chunk's ten pieces
    block_barrier();
    const double shift = w[0];
    PassA2 st;
    pass_a2_phase1(w, lane, shift, st, acc[0]);
    block_barrier();
    const DmaPieceIssueT<ISSUE> piece{next, w, 0, lane};
    pass_a2_phase2(st, lane, shift, acc, piece, o4);
    asm volatile("" ::: "memory");
}

template <bool ISSUE>
__device__ __forceinline__ void fused_step_b(const double* __restrict__ next, double* __restrict__ w, int lane, int64_t c,
                                             const AllanLevel& lv, double (&acc)[9], bool last = false) {
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    block_barrier();
    const double shift = w[0];
    double e[72];
    read_b_single(w, lane, e);
    block_barrier();
    const DmaPieceIssueT<ISSUE> piece{next, w, 1, lane};
    compute_b<false, true, true>(e, lane, c, lv, shift, acc, piece);
}

__global__ void __launch_bounds__(128, 2)
allan_fused_kernel(const double* __restrict__ in, double* __restrict__ out1, double* __restrict__ out2, double* __restrict__ partial0,
                   double* __restrict__ partial1, const AllanLevel lv, const AllanLevel lv1) {
    __shared__ __attribute__((aligned(1024))) double stage[2][kDmaStage];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t s = blockIdx.y, part = blockIdx.x, nparts = gridDim.x;
    const double* x = in + s * lv.in_stride;
    const int64_t c_begin = part * kFuseChunks;
    int64_t c_end = c_begin + kFuseChunks;
    if (c_end > lv.nchunks) c_end = lv.nchunks;
    const int64_t c_dma = (lv.n_in - kDmaStage) / kChunk;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c_begin < c_end && c_begin <= c_dma) dma_request_half(x + c_begin * kChunk, stage[0], wave, lane);
    if (c_begin + 1 < c_end && c_begin + 1 <= c_dma) dma_request_half(x + (c_begin + 1) * kChunk, stage[1], wave, lane);
    // ten whole chunks, every pair of bins in them exists (interior is monotone in c): all workgroups but the last of a series
    const bool plain = c_end - c_begin == kFuseChunks && c_end - 1 <= c_dma && chunk_is_interior(c_end - 1, lv);
    const int64_t e_first = part * kChunk;                      // first level-k+1 entry of this workgroup
    if (plain && wave == 0) {
        double l1[kFuseChunks][4];
#pragma unroll
        for (int q = 0; q < kFuseChunks; ++q) { l1[q][0] = 0.0; l1[q][1] = 0.0; l1[q][2] = 0.0; l1[q][3] = 0.0; }
        // The chunk loop stays ROLLED and a wave-uniform switch names the registers that keep chunk q's sums.  (Unrolled, the
        // scheduler stretches live ranges over the ten bodies and spills 3 KB per lane; with plain assignments in the arms the
        // compiler turns the switch into 80 conditional moves per chunk -- the empty asm keeps every arm a real block of moves.)
        // Two loops: chunks 0..7 request chunk q + 2 piece by piece inside their arithmetic, the last two request nothing.
#pragma unroll 1
        for (int q = 0; q < kFuseChunks - 2; ++q) {
            double o4[4];
            fused_step_a<true>(x + (c_begin + q + 2) * kChunk, stage[q & 1], lane, acc, o4);
            switch (q) {
                case 0: l1[0][0] = o4[0]; l1[0][1] = o4[1]; l1[0][2] = o4[2]; l1[0][3] = o4[3];
                    asm volatile("" : "+v"(l1[0][0]), "+v"(l1[0][1]), "+v"(l1[0][2]), "+v"(l1[0][3])); break;
                case 1: l1[1][0] = o4[0]; l1[1][1] = o4[1]; l1[1][2] = o4[2]; l1[1][3] = o4[3];
                    asm volatile("" : "+v"(l1[1][0]), "+v"(l1[1][1]), "+v"(l1[1][2]), "+v"(l1[1][3])); break;
                case 2: l1[2][0] = o4[0]; l1[2][1] = o4[1]; l1[2][2] = o4[2]; l1[2][3] = o4[3];
                    asm volatile("" : "+v"(l1[2][0]), "+v"(l1[2][1]), "+v"(l1[2][2]), "+v"(l1[2][3])); break;
                case 3: l1[3][0] = o4[0]; l1[3][1] = o4[1]; l1[3][2] = o4[2]; l1[3][3] = o4[3];
                    asm volatile("" : "+v"(l1[3][0]), "+v"(l1[3][1]), "+v"(l1[3][2]), "+v"(l1[3][3])); break;
                case 4: l1[4][0] = o4[0]; l1[4][1] = o4[1]; l1[4][2] = o4[2]; l1[4][3] = o4[3];
                    asm volatile("" : "+v"(l1[4][0]), "+v"(l1[4][1]), "+v"(l1[4][2]), "+v"(l1[4][3])); break;
                case 5: l1[5][0] = o4[0]; l1[5][1] = o4[1]; l1[5][2] = o4[2]; l1[5][3] = o4[3];
                    asm volatile("" : "+v"(l1[5][0]), "+v"(l1[5][1]), "+v"(l1[5][2]), "+v"(l1[5][3])); break;
                case 6: l1[6][0] = o4[0]; l1[6][1] = o4[1]; l1[6][2] = o4[2]; l1[6][3] = o4[3];
                    asm volatile("" : "+v"(l1[6][0]), "+v"(l1[6][1]), "+v"(l1[6][2]), "+v"(l1[6][3])); break;
                default: l1[7][0] = o4[0]; l1[7][1] = o4[1]; l1[7][2] = o4[2]; l1[7][3] = o4[3];
                    asm volatile("" : "+v"(l1[7][0]), "+v"(l1[7][1]), "+v"(l1[7][2]), "+v"(l1[7][3])); break;
            }
        }
#pragma unroll 1
        for (int q = kFuseChunks - 2; q < kFuseChunks; ++q) {       // rolled as well: side by side the two bodies spill
            double o4[4];
            fused_step_a<false>(x, stage[q & 1], lane, acc, o4, q == kFuseChunks - 1);
            switch (q) {
                case 8: l1[8][0] = o4[0]; l1[8][1] = o4[1]; l1[8][2] = o4[2]; l1[8][3] = o4[3];
                    asm volatile("" : "+v"(l1[8][0]), "+v"(l1[8][1]), "+v"(l1[8][2]), "+v"(l1[8][3])); break;
                default: l1[9][0] = o4[0]; l1[9][1] = o4[1]; l1[9][2] = o4[2]; l1[9][3] = o4[3];
                    asm volatile("" : "+v"(l1[9][0]), "+v"(l1[9][1]), "+v"(l1[9][2]), "+v"(l1[9][3])); break;
            }
        }
        // both stages are idle now (the last two chunks requested nothing; everybody passed the last chunk's second barrier)
        double2* w1 = reinterpret_cast<double2*>(stage[0]);
        if (lane < 63) {
#pragma unroll
            for (int q = 0; q < kFuseChunks; ++q) {
                w1[(252 * q + 4 * lane) / 2] = double2{l1[q][0], l1[q][1]};
                w1[(252 * q + 4 * lane) / 2 + 1] = double2{l1[q][2], l1[q][3]};
            }
        }
    } else if (plain) {
#pragma unroll 1
        for (int q = 0; q < kFuseChunks - 2; ++q)
            fused_step_b<true>(x + (c_begin + q + 2) * kChunk, stage[q & 1], lane, c_begin + q, lv, acc);
#pragma unroll 1
        for (int q = kFuseChunks - 2; q < kFuseChunks; ++q)
            fused_step_b<false>(x, stage[q & 1], lane, c_begin + q, lv, acc, q == kFuseChunks - 1);
    } else {
        // the last workgroup of a series (a ragged chunk, pairs of bins that do not exist, fewer than ten chunks): the level
        // kernel's own loop, the sums of 10 through memory (out1) like there -- one workgroup in 58 at config 5's size
        double* out_series = out1 + s * lv.out_stride;
        int stores1 = 0, stores2 = 0;
#pragma unroll 1
        for (int64_t c = c_begin; c < c_end; ++c) {
            double* w = stage[(c - c_begin) & 1];
            if (c > c_dma) {
                stage_ragged_half(x, w, wave, lane, c, lv.n_in, x[c * kChunk]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                wait_all_but(((c + 1 < c_end && c + 1 <= c_dma) ? kDmaPieces / 2 : 0) + stores1 + stores2);
            }
            block_barrier();
            const bool interior = chunk_is_interior(c, lv);
            const bool again = c + 2 < c_end && c + 2 <= c_dma;
            const double shift = w[0];
            if (wave == 0) pass_a<true, true>(w, lane, c, lv, shift, out_series, acc);
            else pass_bc<true, true>(w, lane, c, lv, shift, acc);
            (void)interior;
            block_barrier();
            if (again) dma_request_half(x + (c + 2) * kChunk, w, wave, lane);
            stores2 = stores1;
            stores1 = 0;
        }
        // make the stores visible to the whole workgroup, then stage them like the others
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        block_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int64_t have = lv1.n_in - e_first < kChunk ? lv1.n_in - e_first : kChunk;
        const double* src = out_series + e_first;
        for (int i = threadIdx.x; i < kChunk; i += 128)
            if (i < have) stage[0][i] = src[i];
    }
    // ---- the level-k+1 chunk of this workgroup: entries [2520 part, 2520 part + 2520) of the series
    block_barrier();                                            // the entries are in stage 0
    double* w1 = stage[0];
    const bool full1 = e_first + kChunk <= lv1.n_in;
    if (wave == 0 && lane < 10) w1[kChunk + lane] = w1[kChunk - 1 - lane];      // the mirror halo (see above)
    if (!full1 && wave == 1) {                                  // the series' last workgroup: nothing beyond its end
        const double fill = w1[0];
        for (int i = (int)(lv1.n_in - e_first) + lane; i < kStage; i += 64) w1[i] = fill;
    }
    block_barrier();
    const double shift1 = w1[0];
    double acc1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double* out2_series = lv1.n_out > 0 ? out2 + s * lv1.out_stride : nullptr;
    if (full1 && chunk_is_interior(part, lv1)) {
        // every pair INSIDE the chunk exists; the cross pair vanishes against the mirror halo
        if (wave == 0) {
            double e[48];
            read_a(w1, lane, e);
            compute_a<false, true>(e, lane, part, lv1, shift1, out2_series, acc1);
        } else {
            double e[72];
            read_b(w1, lane, e);        // compiler-visible reads: nothing orders a use of read_b_single's results behind a wait
            compute_b<false, true, true>(e, lane, part, lv1, shift1, acc1);
        }
    } else {
        // the last chunk or two of the series: bounds-checked passes; a cross pair that exists is the finishing launch's, so
        // the bins of "the next chunk" are cut off here
        AllanLevel lvE = lv1;
#pragma unroll
        for (int j = 1; j <= 9; ++j) {
            const int64_t lim = (part + 1) * (kChunk / j);
            if (lvE.nb[j - 1] > lim) lvE.nb[j - 1] = lim;
        }
        if (wave == 0) pass_a<true, true>(w1, lane, part, lvE, shift1, out2_series, acc1);
        else pass_bc<true, true>(w1, lane, part, lvE, shift1, acc1);
    }
    double* rec = partial1 + (s * nparts + part) * kFuseRecord;
    if (wave == 0 && lane < 9) {                               // sums of the first / last j entries, j = lane + 1, about the origin
        double a = 0.0, b = 0.0;
        for (int i = 0; i <= lane; ++i) {
            a += w1[i] - shift1;
            b += w1[kChunk - 1 - lane + i] - shift1;
        }
        rec[9 + lane] = a;
        rec[18 + lane] = b;
        if (lane == 0) rec[27] = shift1;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const bool mine = (j == 0 || j == 1 || j == 3 || j == 4 || j == 7) == (wave == 0);
        double a = acc[j], b = acc1[j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
        if (lane == 0 && mine) {
            partial0[(s * nparts + part) * 9 + j] = a;
            rec[j] = b;
        }
    }
}

// Levels that fit one chunk (n_in <= 2520; for 3600 s @ 400 Hz the levels of 1440, 144 and 14 entries) are finished by
// ONE wavefront per series in a single launch: the sums of 10 go to a second LDS stage instead of HBM and become
// the next level in place.  Sums are final (one wavefront saw the whole level): written straight to sums[].
// Workgroups beyond the first t.nseries * (t.nlevels > 0) fold the per-workgroup partial records of the levels before: one
// 64-lane workgroup per (series, level), lanes stride over the records of each factor, then a fixed butterfly.
__device__ __forceinline__ void fold_partials(const double* __restrict__ partial, double* __restrict__ sums, const AllanFold& f,
                                              int64_t s, int k, int64_t nseries) {
    const int nparts = f.nparts[k];
    const double* p = partial + (f.offset[k] + s * nparts) * 9;
    for (int j = 0; j < 9; ++j) {
        double a = 0.0;
        for (int c = threadIdx.x; c < nparts; c += 64) a += p[c * 9 + j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (threadIdx.x == 0) sums[((int64_t)k * nseries + s) * 9 + j] = a;
    }
}

// The fused kernel's level: sums inside the workgroups' chunks + the pairs (last bin of workgroup c, first bin of workgroup c+1)
// from the recorded first / last sums, d = (first' - last) + j (origin' - origin).  A pair exists when its second bin does.
__device__ __forceinline__ void fold_partials_fused(const double* __restrict__ partial, double* __restrict__ sums, const AllanFold& f,
                                                    int64_t s, int k, int64_t nseries) {
    const int nparts = f.nparts[k];
    const double* p = partial + f.offset[k] * 9 + s * nparts * kFuseRecord;
    for (int j = 0; j < 9; ++j) {
        double a = 0.0;
        for (int c = threadIdx.x; c < nparts; c += 64) {
            a += p[(int64_t)c * kFuseRecord + j];
            if (c + 1 < nparts && (int64_t)(c + 1) * (kChunk / (j + 1)) < f.fused_nb[j]) {
                const double* q0 = p + (int64_t)c * kFuseRecord;
                const double* q1 = q0 + kFuseRecord;
                const double d = (q1[9 + j] - q0[18 + j]) + (double)(j + 1) * (q1[27] - q0[27]);
                a = __builtin_fma(d, d, a);
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (threadIdx.x == 0) sums[((int64_t)k * nseries + s) * 9 + j] = a;
    }
}

// The levels that fit a chunk, for one series per workgroup of three wavefronts: wavefront 0 first builds the entries of the
// levels after the first (sums of 10 of the level before: 252, 25, 2 entries at most) in LDS, then every wavefront runs the
// passes of its own level(s) at the same time -- wavefront w level w, wavefront 2 also the tiny fourth one.  (One wavefront
// doing the levels one after the other, each producing the next, took 30 us for 1440 / 144 / 14 entries.)
constexpr int kTailWaves = 3;
__global__ void __launch_bounds__(64 * kTailWaves) allan_tail_kernel(const double* __restrict__ in, const double* __restrict__ partial,
                                                                     double* __restrict__ sums, const AllanTail t, const AllanFold f) {
    __shared__ __attribute__((aligned(16))) double stage[kTailWaves][kStage];
    __shared__ double lev_store[256 + 32 + 8];          // entries of the second, third and fourth tail level (<= 252, 25, 2)
    auto lev = [&](int l) -> double* { return lev_store + (l == 0 ? 0 : (l == 1 ? 256 : 288)); };
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ntail = t.nlevels > 0 ? t.nseries : 0;
    if ((int64_t)blockIdx.x >= ntail) {
        if (wave != 0) return;
        const int64_t b = (int64_t)blockIdx.x - ntail;
        const int k = (int)(b / t.nseries);
        if (k == f.fused_level) fold_partials_fused(partial, sums, f, b % t.nseries, k, t.nseries);
        else fold_partials(partial, sums, f, b % t.nseries, k, t.nseries);
        return;
    }
    const int64_t s = blockIdx.x;
    const double* x = in + s * t.in_stride;
    if (wave == 0) {
        // level l+1 entry i = 10 sh + sum_t (E_l[10 i + t] - sh), sh = E_l[0]: the same value the level passes hand on
        const double* src = x;
        for (int l = 0; l + 1 < t.nlevels; ++l) {
            const int64_t n_next = t.n_in[l + 1];
            const double sh = src[0];
            for (int i = lane; i < n_next; i += 64) {
                double acc10 = 0.0;
#pragma unroll
                for (int q = 0; q < 10; ++q) acc10 += src[10 * i + q] - sh;
                lev(l)[i] = __builtin_fma(10.0, sh, acc10);
            }
            wave_sync();
            src = lev(l);
        }
    }
    __syncthreads();
    for (int l = wave; l < t.nlevels; l += kTailWaves) {
        const double* src = l == 0 ? x : lev(l - 1);
        AllanLevel lv;
        lv.n_in = t.n_in[l];
        lv.n_out = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) lv.nb[j] = t.nb[l][j];
        const double shift = src[0];
        double* w = stage[wave];
        wave_sync();                                    // my previous level's passes are done with the stage
        for (int i = lane; i < kStage; i += 64) w[i] = i < lv.n_in ? src[i] - shift : 0.0;
        wave_sync();
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        chunk_passes<true>(w, lane, 0, lv, shift, nullptr, acc);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            double a = acc[j];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            if (lane == 0) sums[((int64_t)(t.first + l) * t.nseries + s) * 9 + j] = a;
        }
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

bool allan_dma_applies(const double* in, const AllanLevel& lv) {
    static const int forced = [] { const char* e = getenv("GINSIM_ALLAN_DMA"); return e ? atoi(e) : 1; }();
    return forced != 0 && ((uintptr_t)in % 16) == 0 && (lv.in_stride % 2) == 0 && lv.n_in >= kDmaStage;
}
// partial records per series: one per workgroup
int allan_pair_parts(const AllanLevel& lv) { return (lv.nchunks + lv.chunks_per_block - 1) / lv.chunks_per_block; }

hipError_t launch_allan_pair(const double* in, double* out, double* partial, const AllanLevel& lv, int64_t nseries, hipStream_t st) {
    hipLaunchKernelGGL(allan_pair_kernel, dim3((unsigned)allan_pair_parts(lv), (unsigned)nseries), dim3(128), 0, st, in, out, partial, lv);
    return hipGetLastError();
}

bool allan_fuse_applies(const double* in, const AllanLevel& lv, const AllanLevel& lv1) {
    const char* e = getenv("GINSIM_ALLAN_FUSE");        // read per call: the tests run both forms in one process
    const int on = e ? atoi(e) : 1;
    return on != 0 && allan_dma_applies(in, lv) && lv.n_out == lv1.n_in && lv1.n_in > kChunk;
}
int allan_fuse_parts(const AllanLevel& lv) { return (lv.nchunks + kFuseChunks - 1) / kFuseChunks; }
int allan_fuse_record() { return kFuseRecord; }

hipError_t launch_allan_fused(const double* in, double* out1, double* out2, double* partial0, double* partial1, const AllanLevel& lv,
                              const AllanLevel& lv1, int64_t nseries, hipStream_t st) {
    hipLaunchKernelGGL(allan_fused_kernel, dim3((unsigned)allan_fuse_parts(lv), (unsigned)nseries), dim3(128), 0, st, in, out1, out2,
                       partial0, partial1, lv, lv1);
    return hipGetLastError();
}

hipError_t launch_allan_finish(const double* in, const double* partial, double* sums, const AllanTail& t, const AllanFold& f,
                               int64_t nseries, hipStream_t st) {
    const int64_t blocks = (t.nlevels > 0 ? nseries : 0) + nseries * f.nlevels;
    if (blocks > 0) hipLaunchKernelGGL(allan_tail_kernel, dim3((unsigned)blocks), dim3(64 * kTailWaves), 0, st, in, partial, sums, t, f);
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
