// Round 6 experiment: can the three 96 GB thirds of an MI355X's memory be told apart WITHOUT physical addresses, chunk by chunk,
// so that a placed pair of regions (sensors | trajectories) can be stitched from physical chunks with the HIP virtual-memory API?
//   1. hipMemCreate K chunks of 1 GiB, map them into one reserved range (time per chunk, granularity, mapping with an offset)
//   2. pair probe: a fill that streams into the first half of chunk 0 and the first half of chunk i at the same time
//      (i = 0: both halves of chunk 0 = certainly one third); a chunk in another third should be faster
//   3. the 15-plane store pattern of the headline kernel (65 536 runs x 1000 samples, fp64) with its planes in a range stitched
//      from chunks of ONE class, and with sensors | trajectories in different classes; the same on a plain hipMalloc region
//   4. a streaming read of 2.2 GB from one class, and from a range whose halves lie in different classes
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vmm_thirds tools/experiments/vmm_thirds.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// every workgroup streams into BOTH regions: block b writes rows b, b + grid, ... of 4 KiB in each
__global__ void __launch_bounds__(256) pair_fill(d2* a, d2* b, size_t rows) {
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const size_t i = r * 256 + threadIdx.x;
        __builtin_nontemporal_store(d2{(double)i, 1.0}, a + i);
        __builtin_nontemporal_store(d2{(double)i, 2.0}, b + i);
    }
}

// the MC kernel's store pattern: one lane = one run, time loop inside, planes [n][runs]; sensors (6 planes) and trajectory (9)
__global__ void __launch_bounds__(256) fill15(double* sens, double* traj, int64_t n, int64_t runs) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = n * runs;
    double v = (double)r;
    for (int64_t j = 0; j < n; ++j) {
        v = v * 1.0000001 + 0.5;
#pragma unroll
        for (int c = 0; c < 6; ++c) __builtin_nontemporal_store(v + c, sens + c * plane + j * runs + r);
#pragma unroll
        for (int c = 0; c < 9; ++c) __builtin_nontemporal_store(v - c, traj + c * plane + j * runs + r);
    }
}

// streaming read of two regions at the same time (sum kept alive through a rarely taken store)
__global__ void __launch_bounds__(256) pair_read(const d2* a, const d2* b, size_t rows, double* sink) {
    d2 acc = {0.0, 0.0};
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const size_t i = r * 256 + threadIdx.x;
        acc += __builtin_nontemporal_load(a + i);
        acc += __builtin_nontemporal_load(b + i);
    }
    if (acc.x == 1.2345e-300) sink[0] = acc.y;
}

static hipEvent_t ev0, ev1;
template <typename F> static float time_min(F f, int warm, int reps, float* avg = nullptr) {
    for (int i = 0; i < warm; ++i) f();
    float best = 1e30f, sum = 0.f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(ev0)); f(); CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
        float ms; CK(hipEventElapsedTime(&ms, ev0, ev1)); best = std::min(best, ms); sum += ms;
    }
    if (avg) *avg = sum / reps;
    return best;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t G = (size_t)1 << 30;
    int K = argc > 1 ? atoi(argv[1]) : 200;
    const size_t CH = (argc > 2 ? (size_t)atoi(argv[2]) : 1024) << 20;
    CK(hipSetDevice(0));
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GiB of %.1f GiB\n", fr / (double)G, tot / (double)G);
    int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    K = (int)std::min<size_t>(K, (fr - 24 * G) / CH);
    hipDeviceptr_t va; CK(hipMemAddressReserve(&va, (size_t)K * CH, 0, 0, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(K);
    std::vector<double> t_create(K), t_map(K);
    for (int i = 0; i < K; ++i) {
        double t0 = now_ms();
        hipError_t e = hipMemCreate(&h[i], CH, &prop, 0);
        if (e != hipSuccess) { printf("hipMemCreate %d failed: %s\n", i, hipGetErrorString(e)); K = i; break; }
        double t1 = now_ms();
        CK(hipMemMap((char*)va + (size_t)i * CH, CH, 0, h[i], 0));
        t_create[i] = t1 - t0; t_map[i] = now_ms() - t1;
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    double t0 = now_ms();
    CK(hipMemSetAccess(va, (size_t)K * CH, &acc, 1));
    printf("%d chunks of %zu MiB: create avg %.2f ms (max %.2f), map avg %.3f ms, set access (all) %.1f ms\n", K, CH >> 20,
           [&] { double s = 0; for (int i = 0; i < K; ++i) s += t_create[i]; return s / K; }(), *std::max_element(t_create.begin(), t_create.begin() + K),
           [&] { double s = 0; for (int i = 0; i < K; ++i) s += t_map[i]; return s / K; }(), now_ms() - t0);
    {   // can a handle be mapped from an offset?
        hipDeviceptr_t v2; CK(hipMemAddressReserve(&v2, CH, 0, 0, 0));
        hipError_t e = hipMemMap(v2, CH / 2, CH / 2, h[0], 0);
        printf("hipMemMap with offset CH/2: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) hipMemUnmap(v2, CH / 2);
        hipMemAddressFree(v2, CH);
    }
    char* base = (char*)va;
    const size_t half = CH / 2, rows = half / 4096;
    // ---- 2. pair probe
    std::vector<float> probe(K);
    auto pf = [&](char* a, char* b) { hipLaunchKernelGGL(pair_fill, dim3(4096), dim3(256), 0, 0, (d2*)a, (d2*)b, rows); };
    time_min([&] { pf(base, base + half); }, 30, 1);
    for (int i = 0; i < K; ++i) {
        char* other = i == 0 ? base + half : base + (size_t)i * CH;
        probe[i] = time_min([&] { pf(base, other); }, 2, 5);
    }
    printf("pair fill (2 x %zu MiB), first half of chunk 0 + first half of chunk i, min of 5 [ms]; i = 0: the two halves of chunk 0\n", half >> 20);
    for (int i = 0; i < K; ++i) printf("%s%.4f", i % 16 ? " " : "\n  ", probe[i]);
    printf("\n");
    std::vector<float> sorted(probe.begin(), probe.begin() + K); std::sort(sorted.begin(), sorted.end());
    printf("probe: min %.4f  p10 %.4f  median %.4f  p90 %.4f  max %.4f\n", sorted[0], sorted[K / 10], sorted[K / 2], sorted[K * 9 / 10], sorted[K - 1]);
    const float same = sorted[K * 9 / 10] < probe[0] ? probe[1] : probe[0];
    std::vector<int> clsA, clsB;
    for (int i = 0; i < K; ++i) (probe[i] < 0.95f * same ? clsB : clsA).push_back(i);
    printf("class A (as chunk 0; probe >= 0.95 x %.4f): %zu chunks; class B: %zu chunks\n  B:", same, clsA.size(), clsB.size());
    for (int i : clsB) printf(" %d", i);
    printf("\n");
    // a second reference: the first B chunk -> are the B chunks one class or two?
    if (!clsB.empty()) {
        char* rb = base + (size_t)clsB[0] * CH;
        printf("pair fill against the first B chunk (%d):\n", clsB[0]);
        for (int i = 0; i < K; ++i) {
            char* other = i == clsB[0] ? rb + half : base + (size_t)i * CH;
            float ms = time_min([&] { pf(rb, other); }, 2, 5);
            printf("%s%.4f", i % 16 ? " " : "\n  ", ms);
        }
        printf("\n");
    }
    // repeat of the probe (is it stable?)
    int flips = 0;
    for (int i = 1; i < K; ++i) {
        float ms = time_min([&] { pf(base, base + (size_t)i * CH); }, 2, 5);
        if ((ms < 0.95f * same) != (probe[i] < 0.95f * same)) ++flips;
    }
    printf("probe repeated: %d of %d chunks changed class\n", flips, K - 1);
    // ---- 3. the 15-plane pattern: sensors fixed in chunks S0.., trajectories in a sliding window of chunks
    const int64_t n = 1000, runs = 65536;
    const size_t plane = (size_t)n * runs * 8, sens_b = 6 * plane, traj_b = 9 * plane;
    const int sens_ch = (int)((sens_b + CH - 1) / CH), traj_ch = (int)((traj_b + CH - 1) / CH);
    auto f15 = [&](char* s, char* t) { hipLaunchKernelGGL(fill15, dim3(runs / 256), dim3(256), 0, 0, (double*)s, (double*)t, n, runs); };
    {
        char* p; CK(hipMalloc(&p, 15 * plane));
        float avg, mn = time_min([&] { f15(p, p + sens_b); }, 30, 20, &avg);
        printf("fill15 on ONE hipMalloc region: min %.4f avg %.4f ms = %.0f GB/s\n", mn, avg, 15.0 * plane / avg / 1e6);
        CK(hipFree(p));
    }
    CK(hipMemUnmap(va, (size_t)K * CH));
    hipDeviceptr_t v3; CK(hipMemAddressReserve(&v3, (size_t)(sens_ch + traj_ch) * CH, 0, 0, 0));
    char* S = (char*)v3; char* T = S + (size_t)sens_ch * CH;
    const int step = argc > 3 ? atoi(argv[3]) : 4;
    for (int s0 : {0, 56, 120}) {
        if (s0 + sens_ch > K) continue;
        for (int k = 0; k < sens_ch; ++k) CK(hipMemMap(S + (size_t)k * CH, CH, 0, h[s0 + k], 0));
        CK(hipMemSetAccess(S, (size_t)sens_ch * CH, &acc, 1));
        printf("fill15, sensors in chunks %d..%d, trajectories in chunks w..w+%d: [w: avg ms]", s0, s0 + sens_ch - 1, traj_ch - 1);
        int col = 0;
        for (int w = 0; w + traj_ch <= K; w += step) {
            if (w < s0 + sens_ch && w + traj_ch > s0) continue;      // overlaps the sensors' chunks
            for (int k = 0; k < traj_ch; ++k) CK(hipMemMap(T + (size_t)k * CH, CH, 0, h[w + k], 0));
            CK(hipMemSetAccess(T, (size_t)traj_ch * CH, &acc, 1));
            float avg; time_min([&] { f15(S, T); }, 12, 8, &avg);
            printf("%s%d:%.3f", col++ % 12 ? " " : "\n  ", w, avg);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(T, (size_t)traj_ch * CH));
        }
        printf("\n");
        CK(hipMemUnmap(S, (size_t)sens_ch * CH));
    }
    // release: does the memory come back?
    t0 = now_ms();
    for (int i = 0; i < K; ++i) hipMemRelease(h[i]);
    CK(hipDeviceSynchronize());
    size_t fr2; CK(hipMemGetInfo(&fr2, &tot));
    printf("released %d handles in %.1f ms (ranges still mapped keep their chunks alive); free now %.1f GiB\n", K, now_ms() - t0, fr2 / (double)G);
    return 0;
}
