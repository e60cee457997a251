// Is v_sqrt_f32 the correctly rounded square root on this chip for the arguments the Box-Muller radius produces (0 and
// [2^-24, 64))?  Exhaustive over every float in the range; prints the number of mismatches against the residual-corrected root
// (fastmath.hpp sqrt_rn_f32).      hipcc --offload-arch=gfx950 -O3 -o sqrt_exhaustive tools/sqrt_exhaustive.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ float sqrt_rn(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vdn = __builtin_fmaf(-dn, s, x), vup = __builtin_fmaf(-up, s, x);
    s = vdn <= 0.0f ? dn : s;
    s = vup > 0.0f ? up : s;
    return s;
}
__global__ void check(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + lo;
    if (i >= hi) return;
    const float x = __uint_as_float((uint32_t)i);
    if (__builtin_amdgcn_sqrtf(x) != sqrt_rn(x)) {
        if (atomicAdd(bad, 1ull) == 0) *first = (uint32_t)i;
    }
}
int main() {
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    const uint32_t lo = 0x33800000u /* 2^-24 */, hi = 0x42800000u /* 64 */;
    const uint64_t count = hi - lo;
    check<<<(unsigned)((count + 255) / 256), 256>>>(lo, hi, bad, first);
    unsigned long long h = 0; uint32_t f = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 vs correctly rounded over %llu floats in [2^-24, 64): %llu mismatches (first bits 0x%08x)\n", (unsigned long long)count, h, f);
    return 0;
}
