// Multi-GPU exchange of the statistics records behind the C ABI: RCCL (librccl, resolved at run time) on the context's stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ginsim {

struct Comm;     // one RCCL communicator (one rank) bound to a context's device

// nullptr + message on failure; the library is dlopen()ed on first use, so single-GPU users never load RCCL
const char* comm_probe();        // can librccl be reached at all? (dlopen + dlsym only: no bootstrap socket, no thread)
const char* comm_unique_id(unsigned char* id128);
const char* comm_create(int nranks, int rank, const unsigned char* id128, Comm** out);
void comm_destroy(Comm* c);
int comm_nranks(const Comm* c);
const char* comm_query(const Comm* c, int* nranks, int* rank, int* device);      // read back from the communicator; -1 = query missing
// all-gather `count` doubles per rank: send [count] -> recv [nranks][count], enqueued on `stream`
const char* comm_allgather_f64(Comm* c, const double* send, double* recv, size_t count, hipStream_t stream);

}  // namespace ginsim
