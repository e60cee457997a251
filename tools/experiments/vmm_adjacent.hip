// Round 6 experiment: does mapping a chunk NEXT TO an already mapped one (the arena's growth) disturb the neighbour's contents?
// (tests/test_gpu_placement.py: a region's tail in stripe 2 read back as zeros after stripes 3.. were mapped)
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/vmm_adjacent tools/experiments/vmm_adjacent.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)

__global__ void fill(uint64_t* p, size_t n, uint64_t tag) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = tag + i; }
__global__ void check(const uint64_t* p, size_t n, uint64_t tag, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) if (p[i] != tag + i) atomicAdd(bad, 1ull);
}
static unsigned long long* d_bad;
static size_t S;
static unsigned long long verify(char* base, int k) {
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, (const uint64_t*)(base + (size_t)k * S), S / 8, (uint64_t)(k + 1) << 40, d_bad);
    unsigned long long b; CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost)); return b;
}
int main(int argc, char** argv) {
    S = (size_t)(argc > 1 ? atoi(argv[1]) : 512) << 20;
    const int K = 8;
    CK(hipSetDevice(0));
    CK(hipMalloc(&d_bad, 8));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: map + set access stripe by stripe; mode 1: the same, but every chunk first visits a staging slot (mapped, written, unmapped);
        // mode 2: as 1 with a hipDeviceSynchronize before every unmap / map
        void* v; CK(hipMemAddressReserve(&v, (size_t)K * S, 0, nullptr, 0));
        void* st; CK(hipMemAddressReserve(&st, S, 0, nullptr, 0));
        char* base = (char*)v;
        printf("mode %d: range %p (1 GiB aligned: %d), staging %p\n", mode, v, (int)(((uintptr_t)v & ((1ull << 30) - 1)) == 0), st);
        std::vector<hipMemGenericAllocationHandle_t> h(K);
        for (int k = 0; k < K; ++k) {
            CK(hipMemCreate(&h[k], S, &prop, 0));
            if (mode >= 1) {
                CK(hipMemMap(st, S, 0, h[k], 0)); CK(hipMemSetAccess(st, S, &acc, 1));
                hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t*)st, S / 8, 0ull);
                if (mode == 2) CK(hipDeviceSynchronize());
                CK(hipMemUnmap(st, S));
            }
            CK(hipMemMap(base + (size_t)k * S, S, 0, h[k], 0));
            CK(hipMemSetAccess(base + (size_t)k * S, S, &acc, 1));
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t*)(base + (size_t)k * S), S / 8, (uint64_t)(k + 1) << 40);
            CK(hipDeviceSynchronize());
            // a host copy into the tail of this stripe and back
            uint64_t host[512], back[512];
            for (int i = 0; i < 512; ++i) host[i] = ((uint64_t)(k + 1) << 40) + (S / 8 - 512 + i);
            CK(hipMemcpy(base + (size_t)(k + 1) * S - 4096, host, 4096, hipMemcpyHostToDevice));
            CK(hipMemcpy(back, base + (size_t)(k + 1) * S - 4096, 4096, hipMemcpyDeviceToHost));
            int hb = 0; for (int i = 0; i < 512; ++i) hb += back[i] != host[i];
            printf("  stripe %d mapped; bad words per stripe so far:", k);
            for (int j = 0; j <= k; ++j) printf(" %llu", verify(base, j));
            printf("  (host copy round trip: %d bad)\n", hb);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(v, (size_t)K * S));
        for (int k = 0; k < K; ++k) CK(hipMemRelease(h[k]));
    }
    return 0;
}
