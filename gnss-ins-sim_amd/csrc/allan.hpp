// Launch interface of the Allan-variance kernels (allan.hip) for the C ABI glue (ginsim_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ginsim {

struct AllanLevel {
    int64_t n_in;           // entries of this level per series
    int64_t n_out;          // entries of the next level per series (n_in / 10), 0 = do not emit
    int64_t in_stride;      // series stride of the input (entries)
    int64_t out_stride;     // series stride of the output
    int64_t nb[9];          // valid bins for j = 1..9 at this level (0 = factor not evaluated)
    int32_t chunks_per_block;   // chunks folded into one wavefront's accumulators before they are written out
    int32_t nchunks;
};

// levels that fit one chunk, finished by one wavefront per series in a single launch
struct AllanTail {
    int32_t nlevels, first;     // tail levels first .. first + nlevels - 1
    int64_t in_stride;          // series stride of the first tail level's input
    int64_t nseries;
    int64_t n_in[4];
    int64_t nb[4][9];
};

struct AllanFold {
    int32_t nlevels;
    int32_t nparts[8];
    int64_t offset[8];          // first record of the level in partial[] (records of 9 doubles)
    int32_t fused_level;        // -1, or the level whose records the fused kernel wrote (allan_fuse_record doubles each: sums[9],
    int32_t pad;                // first[9], last[9], origin): the pairs of bins across its workgroups are added by the fold
    int64_t fused_nb[9];        // valid bins of that level for j = 1..9
};

int allan_chunk_entries();
int allan_chunks(int64_t n_in);
int allan_chunks_per_block(int64_t total_chunks);
int allan_parts(const AllanLevel& lv);
hipError_t launch_allan_level(const double* in, double* out, double* partial, const AllanLevel& lv, int64_t nseries, hipStream_t st);
bool allan_dma_applies(const double* in, const AllanLevel& lv);
int allan_pair_parts(const AllanLevel& lv);
hipError_t launch_allan_pair(const double* in, double* out, double* partial, const AllanLevel& lv, int64_t nseries, hipStream_t st);
// levels k and k+1 in one launch (a workgroup = ten chunks of level k = one chunk of level k+1): out1 = level k+1 (written by the
// last workgroup of a series only), out2 = level k+2, partial0 / partial1 = the records of the two levels
bool allan_fuse_applies(const double* in, const AllanLevel& lv, const AllanLevel& lv1);
int allan_fuse_parts(const AllanLevel& lv);
int allan_fuse_record();        // doubles per level-k+1 record (a multiple of 9)
hipError_t launch_allan_fused(const double* in, double* out1, double* out2, double* partial0, double* partial1, const AllanLevel& lv,
                              const AllanLevel& lv1, int64_t nseries, hipStream_t st);
// ONE launch finishes the call: workgroups 0 .. nseries-1 run the levels that fit a chunk, the others fold the partial records
// of the levels before (either part may be empty)
hipError_t launch_allan_finish(const double* in, const double* partial, double* sums, const AllanTail& t, const AllanFold& f,
                               int64_t nseries, hipStream_t st);

}  // namespace ginsim
