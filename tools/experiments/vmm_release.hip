// Round 6 experiment: when does the physical memory of a hipMemCreate'd chunk come back?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)
static double freeg() { size_t f, t; CK(hipMemGetInfo(&f, &t)); return f / 1073741824.0; }
__global__ void touch(double* p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.0; }
int main() {
    const size_t S = (size_t)4 << 30;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    printf("start                                   free %.2f GiB\n", freeg());
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, S, &prop, 0));            printf("create                                  free %.2f\n", freeg());
    CK(hipMemRelease(h));                          printf("release (never mapped)                  free %.2f\n", freeg());
    void* v; CK(hipMemAddressReserve(&v, 4 * S, 0, nullptr, 0));
    CK(hipMemCreate(&h, S, &prop, 0));
    CK(hipMemMap(v, S, 0, h, 0));                  printf("create + map                            free %.2f\n", freeg());
    CK(hipMemUnmap(v, S));                         printf("unmap (no access set)                   free %.2f\n", freeg());
    CK(hipMemRelease(h));                          printf("release                                 free %.2f\n", freeg());
    CK(hipMemCreate(&h, S, &prop, 0));
    CK(hipMemMap((char*)v + S, S, 0, h, 0)); CK(hipMemSetAccess((char*)v + S, S, &acc, 1));
    printf("create + map + set access               free %.2f\n", freeg());
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, (double*)((char*)v + S), S / 8); CK(hipDeviceSynchronize());
    CK(hipMemUnmap((char*)v + S, S));              printf("touched, unmap                          free %.2f\n", freeg());
    CK(hipMemRelease(h));                          printf("release                                 free %.2f\n", freeg());
    usleep(500000);                                printf("0.5 s later                             free %.2f\n", freeg());
    // release BEFORE unmap
    CK(hipMemCreate(&h, S, &prop, 0));
    CK(hipMemMap((char*)v + 2 * S, S, 0, h, 0)); CK(hipMemSetAccess((char*)v + 2 * S, S, &acc, 1));
    CK(hipMemRelease(h));                          printf("create + map + access, release first    free %.2f\n", freeg());
    CK(hipMemUnmap((char*)v + 2 * S, S));          printf("then unmap                              free %.2f\n", freeg());
    // retain count?
    CK(hipMemCreate(&h, S, &prop, 0));
    CK(hipMemMap((char*)v + 3 * S, S, 0, h, 0)); CK(hipMemSetAccess((char*)v + 3 * S, S, &acc, 1));
    hipMemGenericAllocationHandle_t h2;
    hipError_t e = hipMemRetainAllocationHandle(&h2, (char*)v + 3 * S);
    printf("retain handle from address: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) CK(hipMemRelease(h2));
    CK(hipMemUnmap((char*)v + 3 * S, S)); CK(hipMemRelease(h)); printf("map + access + unmap + release          free %.2f\n", freeg());
    e = hipMemRelease(h);                          printf("a second release: %s                     free %.2f\n", hipGetErrorString(e), freeg());
    CK(hipMemAddressFree(v, 4 * S));               printf("address free                            free %.2f\n", freeg());
    return 0;
}
