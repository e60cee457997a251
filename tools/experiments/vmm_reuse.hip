// Round 6 experiment: which re-uses of a virtual address are safe with the HIP virtual-memory API (ROCm 7.2, gfx950)?
//  A  free a reservation, reserve again (same address?), map a NEW chunk there: does a kernel see the new chunk?
//  B  inside one reservation: unmap a slot, map another chunk at it (vmm_adjacent: the old chunk is written) -- does a
//     hipMalloc + hipFree in between (a TLB flush in the driver) cure it?
//  C  does hipMemAddressReserve honour an address hint?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)
__global__ void fill(uint64_t* p, size_t n, uint64_t tag) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = tag + i; }
__global__ void check(const uint64_t* p, size_t n, uint64_t tag, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) if (p[i] != tag + i) atomicAdd(bad, 1ull);
}
static unsigned long long* d_bad;
static const size_t S = (size_t)512 << 20;
static unsigned long long verify(void* p, uint64_t tag) {
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, (const uint64_t*)p, S / 8, tag, d_bad);
    unsigned long long b; CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost)); return b;
}
static double freeg() { size_t f, t; CK(hipMemGetInfo(&f, &t)); return f / 1073741824.0; }
int main() {
    CK(hipSetDevice(0));
    CK(hipMalloc(&d_bad, 8));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    // ---- A
    printf("A: reserve / map / fill / unmap / release / address free, 24 times; a keeper chunk mapped elsewhere must keep its contents\n");
    void* keepv; CK(hipMemAddressReserve(&keepv, S, 0, nullptr, 0));
    hipMemGenericAllocationHandle_t keep; CK(hipMemCreate(&keep, S, &prop, 0));
    CK(hipMemMap(keepv, S, 0, keep, 0)); CK(hipMemSetAccess(keepv, S, &acc, 1));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t*)keepv, S / 8, 77ull << 40); CK(hipDeviceSynchronize());
    void* last = nullptr; int same = 0; unsigned long long badA = 0, badKeep = 0;
    for (int k = 0; k < 24; ++k) {
        void* v; CK(hipMemAddressReserve(&v, S, 0, nullptr, 0));
        same += v == last; last = v;
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, S, &prop, 0));
        CK(hipMemMap(v, S, 0, h, 0)); CK(hipMemSetAccess(v, S, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t*)v, S / 8, (uint64_t)(k + 1) << 40);
        badA += verify(v, (uint64_t)(k + 1) << 40);
        badKeep += verify(keepv, 77ull << 40);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(v, S)); CK(hipMemRelease(h)); CK(hipMemAddressFree(v, S));
    }
    printf("   same address as the previous reservation: %d of 23; bad words in the re-mapped range %llu, in the keeper %llu; free %.2f GiB\n", same, badA, badKeep, freeg());
    // ---- B
    for (int cure = 0; cure < 3; ++cure) {
        void* v; CK(hipMemAddressReserve(&v, 2 * S, 0, nullptr, 0));
        char* slot = (char*)v; char* home = slot + S;
        hipMemGenericAllocationHandle_t c0, c1;
        CK(hipMemCreate(&c0, S, &prop, 0)); CK(hipMemCreate(&c1, S, &prop, 0));
        CK(hipMemMap(slot, S, 0, c0, 0)); CK(hipMemSetAccess(slot, S, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t*)slot, S / 8, 1ull << 40); CK(hipDeviceSynchronize());
        CK(hipMemUnmap(slot, S));
        CK(hipMemMap(home, S, 0, c0, 0)); CK(hipMemSetAccess(home, S, &acc, 1));      // c0 moves to its home (as a stripe of the arena)
        if (cure == 1) { void* d; CK(hipMalloc(&d, 64 << 20)); CK(hipFree(d)); }
        if (cure == 2) { void* d; CK(hipMalloc(&d, 64 << 20)); hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, (uint64_t*)d, (64 << 20) / 8, 0ull); CK(hipDeviceSynchronize()); CK(hipFree(d)); }
        CK(hipMemMap(slot, S, 0, c1, 0)); CK(hipMemSetAccess(slot, S, &acc, 1));      // the slot is used again for c1
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint64_t*)slot, S / 8, 2ull << 40); CK(hipDeviceSynchronize());
        printf("B (%s): bad words in c0 at its home %llu (the fill of c1 through the re-used slot), in c1 through the slot %llu\n",
               cure == 0 ? "nothing in between" : cure == 1 ? "hipMalloc + hipFree in between" : "hipMalloc + kernel + hipFree in between",
               verify(home, 1ull << 40), verify(slot, 2ull << 40));
    }
    // ---- C
    void* a; CK(hipMemAddressReserve(&a, S, 0, nullptr, 0));
    void* hint = (char*)a + ((size_t)64 << 30);
    void* b; hipError_t e = hipMemAddressReserve(&b, S, 0, hint, 0);
    printf("C: reserve with a hint 64 GiB above %p: %s, got %p (%s)\n", a, hipGetErrorString(e), b, b == hint ? "honoured" : "not honoured");
    return 0;
}
