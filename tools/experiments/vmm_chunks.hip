// Round 6 experiment 2: WHY is the 15-plane store pattern fast (1.21 ms) on ranges stitched from 1 GiB hipMemCreate chunks and slow
// (1.34-1.40 ms) on a hipMalloc'ed region of the same process?  One layout throughout (15 planes of 500 MiB, contiguous in the
// virtual range), the backing varies:
//   A  hipMalloc
//   B  ONE hipMemCreate handle for the whole range
//   C  chunks of 2 GiB ... 2 MiB, created in address order
//   D  chunks of 1 GiB created in address order but mapped in REVERSED / interleaved order
//   E  is VMM memory cached like hipMalloc memory?  a 16 MiB buffer read 64 times (L2 / MALL resident) from both
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/vmm_chunks tools/experiments/vmm_chunks.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

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
__global__ void __launch_bounds__(256) reread(const d2* a, size_t elems, int passes, double* sink) {
    d2 acc = {0.0, 0.0};
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < elems; i += (size_t)gridDim.x * 256) acc += a[i];
    if (acc.x == 1.2345e-300) sink[0] = acc.y;
}
__global__ void __launch_bounds__(256) stream_read(const d2* a, size_t elems, double* sink) {
    d2 acc = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < elems; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(a + i);
    if (acc.x == 1.2345e-300) sink[0] = acc.y;
}

static hipEvent_t ev0, ev1;
template <typename F> static float time_avg(F f, int warm, int reps, float* mn = nullptr) {
    for (int i = 0; i < warm; ++i) f();
    float best = 1e30f, sum = 0.f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(ev0)); f(); CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
        float ms; CK(hipEventElapsedTime(&ms, ev0, ev1)); best = std::min(best, ms); sum += ms;
    }
    if (mn) *mn = best;
    return sum / reps;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static const int64_t n = 1000, runs = 65536;
static const size_t plane = (size_t)n * runs * 8, sens_b = 6 * plane, total = 15 * plane;
static double* sink;

static void f15(char* p) { hipLaunchKernelGGL(fill15, dim3(runs / 256), dim3(256), 0, 0, (double*)p, (double*)(p + sens_b), n, runs); }

struct Range {      // a virtual range backed by chunks
    char* va = nullptr; size_t bytes = 0, ch = 0; std::vector<hipMemGenericAllocationHandle_t> h;
    double create_ms = 0;
    // order: 0 = as created, 1 = reversed, 2 = even chunks first then odd ones
    void make(size_t chunk, int order = 0) {
        ch = chunk; const size_t k = (total + ch - 1) / ch; bytes = k * ch;
        hipDeviceptr_t v; CK(hipMemAddressReserve(&v, bytes, 0, 0, 0)); va = (char*)v;
        h.resize(k);
        double t0 = now_ms();
        for (size_t i = 0; i < k; ++i) CK(hipMemCreate(&h[i], ch, &prop, 0));
        create_ms = now_ms() - t0;
        for (size_t i = 0; i < k; ++i) {
            size_t src = order == 0 ? i : order == 1 ? k - 1 - i : (i < (k + 1) / 2 ? 2 * i : 2 * (i - (k + 1) / 2) + 1);
            CK(hipMemMap(va + i * ch, ch, 0, h[src], 0));
        }
        CK(hipMemSetAccess(va, bytes, &acc, 1));
    }
    void drop() {
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, bytes));
        for (auto x : h) CK(hipMemRelease(x));
        // the reservation is kept: a range reserved again at the same address faulted ("write access to a read-only page")
        h.clear(); va = nullptr;
    }
};

static void report(const char* tag, char* p, double extra_ms = -1) {
    float mn, avg = time_avg([&] { f15(p); }, 30, 20, &mn);
    float rmn, ravg = time_avg([&] { hipLaunchKernelGGL(stream_read, dim3(8192), dim3(256), 0, 0, (const d2*)p, (size_t)(2200u << 20) / 16, sink); }, 10, 10, &rmn);
    printf("%-58s fill15 avg %.4f min %.4f ms = %.0f GB/s | read 2.2 GiB avg %.4f ms = %.0f GB/s", tag, avg, mn, total / avg / 1e6, ravg,
           (double)(2200u << 20) / ravg / 1e6);
    if (extra_ms >= 0) printf(" | create %.1f ms", extra_ms);
    printf("\n"); fflush(stdout);
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMalloc(&sink, 64));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GiB of %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);
    const int rounds = argc > 1 ? atoi(argv[1]) : 2;
    for (int round = 0; round < rounds; ++round) {
        printf("--- round %d\n", round);
        {
            char* p; double t0 = now_ms(); CK(hipMalloc(&p, total)); double ms = now_ms() - t0;
            report("A  hipMalloc", p, ms);
            // a second hipMalloc while the first is held
            char* q; t0 = now_ms(); CK(hipMalloc(&q, total)); ms = now_ms() - t0;
            report("A' second hipMalloc (first still held)", q, ms);
            CK(hipFree(q)); CK(hipFree(p));
        }
        {
            Range r; r.make((total + (2u << 20) - 1) / (2u << 20) * (2u << 20));
            report("B  ONE hipMemCreate handle", r.va, r.create_ms); r.drop();
        }
        for (size_t mb : {4096, 2048, 1024, 512, 256, 64, 16, 2}) {
            Range r; r.make(mb << 20);
            char tag[96]; snprintf(tag, sizeof tag, "C  chunks of %zu MiB, mapped as created", mb);
            report(tag, r.va, r.create_ms); r.drop();
        }
        { Range r; r.make((size_t)1 << 30, 1); report("D  1 GiB chunks, mapped in reversed order", r.va, r.create_ms); r.drop(); }
        { Range r; r.make((size_t)1 << 30, 2); report("D  1 GiB chunks, even ones first then odd ones", r.va, r.create_ms); r.drop(); }
        { Range r; r.make((size_t)500 << 20, 0); report("D  chunks of one plane (500 MiB)", r.va, r.create_ms); r.drop(); }
        {   // E: cached?
            char* p; CK(hipMalloc(&p, 16 << 20));
            Range r; r.make((size_t)1 << 30);
            float a = time_avg([&] { hipLaunchKernelGGL(reread, dim3(2048), dim3(256), 0, 0, (const d2*)p, (size_t)(16 << 20) / 16, 64, sink); }, 3, 5);
            float b = time_avg([&] { hipLaunchKernelGGL(reread, dim3(2048), dim3(256), 0, 0, (const d2*)r.va, (size_t)(16 << 20) / 16, 64, sink); }, 3, 5);
            printf("E  16 MiB read 64 times: hipMalloc %.4f ms (%.0f GB/s), hipMemCreate %.4f ms (%.0f GB/s)\n", a, 64.0 * (16 << 20) / a / 1e6, b,
                   64.0 * (16 << 20) / b / 1e6);
            r.drop(); CK(hipFree(p));
        }
    }
    return 0;
}
