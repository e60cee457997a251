// Placed device memory (ABI 7, include/ginsim.h "placed device memory"): one arena per GPU, a reserved virtual range whose
// 512 MiB stripes are physical chunks (hipMemCreate) dealt so that consecutive stripes cycle through the classes of physical
// memory the device has -- three on MI355X, the three 96 GB thirds at the top level of the physical address.
//
// Why: the fused Monte-Carlo kernels stream 15 (fp64, fp32) or 6 + 9 (given sensors) planes at once.  ONE class of the memory
// takes writes at 4.3-4.9 TB/s however many streams feed it; with all planes in one class the launch of BASELINE config 2 takes
// 1.33-1.40 ms, across two or three classes 1.17-1.25 ms, and a process's first hipMalloc'ed tens of GB come from one class.
// Round 5 looked for a better place by re-allocating and timing the launch itself (a heuristic that missed on two of four
// boxes); here the PLACE is constructed: the class of a chunk is measured, the range is stitched from chunks of known classes,
// and every region carved from it spans the classes whatever its planes' sizes.
//
// How a chunk's class is found: `pair_fill` streams into a reference chunk and the candidate at once.  In the same class the
// pair takes as long as twice a single fill of one chunk (the "same-class level" s = 2 w, calibrated warm on single-window
// fills), in different classes 0.65-0.89 s (tools/experiments/vmm_classes.hip, profiles/r06_placed_memory.json).  The first
// chunk is reference A, the first chunk that stays below 0.82 s against A is reference B, the first that stays below it against
// both is C; every other chunk belongs to the reference against which it is slowest (>= 0.88 s, and 8 % above the runner-up), or
// to none (ambiguous: given back).
//
// Three facts about the virtual-memory API of ROCm 7.2 on gfx950 shape the code (tools/experiments/vmm_adjacent.hip,
// vmm_release.hip, vmm_reuse.hip; profiles/r06_placed_memory.json):
//  1. an address that was unmapped and mapped to ANOTHER chunk keeps translating to the old chunk for the kernels that follow (the
//     fill of the new chunk lands in the old one -- in a stripe of the arena, if that is where the old chunk went); a
//     hipMalloc + hipFree between the unmap and the next use cures it (the driver's ordinary unmap path flushes the translations);
//  2. the physical memory of a chunk that was ever mapped comes back only when its RESERVATION is freed (hipMemAddressFree), not
//     with hipMemUnmap + hipMemRelease;
//  3. a freed reservation's address is handed out again by the next hipMemAddressReserve, stale translations included; an address
//     hint is honoured.
// Hence: a search maps its chunks into a staging reservation of its own, one slot per chunk; when the classes are known EVERYTHING
// is unmapped, the driver is made to flush (a hipMalloc + hipFree), the staging reservation is freed (the chunks that were not
// taken go back to the driver here) and its addresses are reserved again at once as a tombstone that is never mapped; only then are
// the taken chunks mapped into the arena's range and the new references into reservations of their own.  No address is mapped twice.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "placed.hpp"
#include "placed_logic.hpp"

namespace ginsim {

void set_error(const char* fmt, ...);

typedef double d2 __attribute__((ext_vector_type(2)));

// every workgroup streams rows of 4 KiB into BOTH regions: two windows of (resident workgroups x 4 KiB) sweep the two chunks
__global__ void __launch_bounds__(256) placed_pair_fill(d2* a, d2* b, size_t rows) {
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const size_t i = r * 256 + threadIdx.x;
        __builtin_nontemporal_store(d2{0.0, 0.0}, a + i);
        __builtin_nontemporal_store(d2{0.0, 0.0}, b + i);
    }
}
// one window sweeping one region
__global__ void __launch_bounds__(256) placed_window_fill(d2* a, size_t rows) {
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) __builtin_nontemporal_store(d2{0.0, 0.0}, a + r * 256 + threadIdx.x);
}

namespace {

constexpr size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
constexpr size_t GRAIN = 2 * MiB;                 // regions are carved in multiples of this
constexpr double SAME = 0.88, OTHER = 0.82, MARGIN = 1.08;      // against the same-class level s: a same-class pair is >= SAME s,
                                                                // a different-class pair <= OTHER s; MARGIN over the runner-up
constexpr int PROBE_BLOCKS = 4096;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Chunk {
    hipMemGenericAllocationHandle_t h{};
    int cls = -1;
    char* at = nullptr;         // where it is mapped now (a staging slot, its stripe, a reference's own range), nullptr = nowhere
    bool taken = false;         // a search's chunk that goes into the arena
    bool is_ref = false;        // a search's chunk that became a reference
};

struct Arena {
    int device = 0;
    std::mutex mu;
    ginsim_placed_options opt{};
    bool configured = false, vmm_checked = false, vmm_ok = false, failed = false, no_growth = false;
    hipStream_t stream = nullptr;                       // the stream of the context that asked (probes run on it: a stream of the arena's own
                                                        // would shift every later stream of the process to another hardware queue)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipMemAllocationProp prop{};
    hipMemAccessDesc access{};
    char* va = nullptr; size_t va_bytes = 0;            // the arena's range; stripes.size() * stripe bytes of it are mapped
    std::vector<Chunk> stripes;
    Chunk ref[3]; int nref = 0;                         // each in a reservation of its own (stripe bytes)
    double anchor_ms = 0.0;                             // s: the time of a same-class pair fill (2 x the fastest single-window fill)
    placed::FreeList regions;                           // what is carved from the mapped part of the range, and by which context
    int contexts = 0;
    // report
    int searches = 0; int64_t created = 0, ambiguous = 0, probes = 0, peak_held = 0; double search_s = 0.0, last_s = 0.0;

    size_t stripe() const { return (size_t)opt.stripe_bytes; }
    size_t mapped() const { return stripes.size() * stripe(); }
};

std::mutex g_mu;
std::map<int, Arena*> g_arenas;

Arena* arena_of(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    Arena*& a = g_arenas[device];
    if (!a) { a = new Arena(); a->device = device; }
    return a;
}

#define P_TRY(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            set_error("placed memory: %s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                                  \
            return e_ == hipErrorOutOfMemory ? GINSIM_ERR_NOMEM : GINSIM_ERR_HIP;                     \
        }                                                                                             \
    } while (0)

// A reservation (addresses only).  It is fresh -- no range this code freed before -- because every freed range is taken
// again at once as a tombstone (retire_range).
int reserve_fresh(size_t bytes, char** out) {
    void* p = nullptr;
    P_TRY(hipMemAddressReserve(&p, bytes, 0, nullptr, 0));
    *out = (char*)p;
    return GINSIM_OK;
}

// Free a reservation whose chunks are unmapped (their memory returns to the driver here, fact 2) and take the same addresses
// again, never to be mapped: the next reservation of anybody in this process then cannot be handed them with translations that
// may still be stale (fact 3; the flush before the free is the first line of defence, this the second).  Addresses only.
void retire_range(char* p, size_t bytes) {
    if (!p) return;
    if (hipMemAddressFree(p, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
    void* again = nullptr;
    if (hipMemAddressReserve(&again, bytes, 0, p, 0) != hipSuccess) { (void)hipGetLastError(); return; }
    if (again != (void*)p) (void)hipMemAddressFree(again, bytes);       // the hint was not honoured: nothing to hold on to
}

// make the driver flush the translations of what was just unmapped (fact 1): its ordinary allocation path does
int driver_flush() {
    void* d = nullptr;
    P_TRY(hipMalloc(&d, 64 * MiB));
    P_TRY(hipFree(d));
    P_TRY(hipDeviceSynchronize());
    return GINSIM_OK;
}

int resolve_options(Arena& a) {
    if (a.configured) return GINSIM_OK;
    size_t fr = 0, tot = 0;
    P_TRY(hipMemGetInfo(&fr, &tot));
    if (a.opt.stripe_bytes <= 0) a.opt.stripe_bytes = (int64_t)(512 * MiB);
    if (a.opt.budget_bytes <= 0) a.opt.budget_bytes = (int64_t)(200 * GiB);
    if (a.opt.limit_bytes <= 0) a.opt.limit_bytes = (int64_t)(tot / 3);
    if (a.opt.search_seconds <= 0.0) a.opt.search_seconds = 3.0;
    a.opt.limit_bytes = std::max<int64_t>(a.opt.limit_bytes / a.opt.stripe_bytes, 1) * a.opt.stripe_bytes;
    a.configured = true;
    return GINSIM_OK;
}

int ensure_ready(Arena& a) {
    P_TRY(hipSetDevice(a.device));
    if (!a.vmm_checked) {
        int v = 0;
        a.vmm_checked = true;
        a.vmm_ok = hipDeviceGetAttribute(&v, hipDeviceAttributeVirtualMemoryManagementSupported, a.device) == hipSuccess && v != 0;
    }
    if (!a.vmm_ok) { set_error("placed memory: device %d has no virtual-memory management", a.device); return GINSIM_ERR_PLACED; }
    if (a.failed) { set_error("placed memory: device %d: an earlier search found fewer than two classes of physical memory", a.device); return GINSIM_ERR_PLACED; }
    int rc = resolve_options(a);
    if (rc) return rc;
    if (!a.e0) {
        P_TRY(hipEventCreate(&a.e0));
        P_TRY(hipEventCreate(&a.e1));
        a.prop = {};
        a.prop.type = hipMemAllocationTypePinned; a.prop.location.type = hipMemLocationTypeDevice; a.prop.location.id = a.device;
        a.access = {};
        a.access.location = a.prop.location; a.access.flags = hipMemAccessFlagsProtReadWrite;
    }
    if (!a.va) {
        rc = reserve_fresh((size_t)a.opt.limit_bytes, &a.va);
        if (rc) return rc;
        a.va_bytes = (size_t)a.opt.limit_bytes;
    }
    return GINSIM_OK;
}

int map_chunk(Arena& a, Chunk& c, char* at) {
    P_TRY(hipMemMap(at, a.stripe(), 0, c.h, 0));
    c.at = at;
    return GINSIM_OK;
}
int unmap_chunk(Arena& a, Chunk& c) {
    if (c.at) { P_TRY(hipMemUnmap(c.at, a.stripe())); c.at = nullptr; }
    return GINSIM_OK;
}

// min over `reps` timed launches (after one untimed) of f on the arena's stream, in ms
template <class F> int timed(Arena& a, F f, int reps, double* ms) {
    f();
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        P_TRY(hipEventRecord(a.e0, a.stream));
        f();
        P_TRY(hipEventRecord(a.e1, a.stream));
        P_TRY(hipEventSynchronize(a.e1));
        float t = 0.f;
        P_TRY(hipEventElapsedTime(&t, a.e0, a.e1));
        best = std::min(best, t);
    }
    P_TRY(hipGetLastError());
    a.probes += reps + 1;
    *ms = best;
    return GINSIM_OK;
}
int pair_ms(Arena& a, char* x, char* y, double* ms) {
    const size_t rows = a.stripe() / 4096;
    return timed(a, [&] { hipLaunchKernelGGL(placed_pair_fill, dim3(PROBE_BLOCKS), dim3(256), 0, a.stream, (d2*)x, (d2*)y, rows); }, 2, ms);
}
int window_ms(Arena& a, char* x, double* ms) {
    const size_t rows = a.stripe() / 4096;
    return timed(a, [&] { hipLaunchKernelGGL(placed_window_fill, dim3(PROBE_BLOCKS), dim3(256), 0, a.stream, (d2*)x, rows); }, 3, ms);
}

// Add `add` stripes to the arena: create chunks, find their classes, deal them to the range, give the rest back.
int grow(Arena& a, size_t add) {
    const size_t S = a.stripe();
    if (a.no_growth) {
        set_error("placed memory: the arena of device %d stopped growing after a mapping failure", a.device);
        return GINSIM_ERR_PLACED;
    }
    if ((a.stripes.size() + add) * S > a.va_bytes) {
        set_error("placed memory: the arena of device %d would exceed its limit of %.1f GiB", a.device, a.va_bytes / (double)GiB);
        return GINSIM_ERR_PLACED;
    }
    size_t fr = 0, tot = 0;
    P_TRY(hipMemGetInfo(&fr, &tot));
    const size_t keep_free = 2 * GiB;
    const size_t budget = std::min((size_t)a.opt.budget_bytes, fr > keep_free ? fr - keep_free : 0) / S;
    if (budget < add + (size_t)(3 - a.nref)) {
        if ((size_t)a.opt.budget_bytes / S < add + (size_t)(3 - a.nref)) {         // the option, not the device, is what is short
            set_error("placed memory: a search budget of %.1f GiB cannot hold a request of %.1f GiB and the references", a.opt.budget_bytes / (double)GiB, add * S / (double)GiB);
            return GINSIM_ERR_PLACED;
        }
        set_error("placed memory: %.1f GiB free on device %d, %.1f GiB wanted", fr / (double)GiB, a.device, add * S / (double)GiB);
        return GINSIM_ERR_NOMEM;
    }
    const double t0 = now_s();
    const bool trace = getenv("GINSIM_PLACED_TRACE") != nullptr;        // development aid: every chunk's probe times on stderr
    ++a.searches;
    char* stage = nullptr;                      // this search's staging reservation: one slot per chunk (facts 1-3 of the header)
    int rc = reserve_fresh(budget * S, &stage);
    if (rc) return rc;
    std::vector<Chunk> pool;                    // this search's chunks, new references included
    pool.reserve(budget);
    std::vector<size_t> of[3];                  // pool indices by class (references excluded)
    const int nref0 = a.nref;
    auto usable = [&](int k) {                  // stripes available when no class gives more than its share of `add`
        const size_t cap = (add + k - 1) / k;
        size_t s = 0;
        for (int c = 0; c < 3; ++c) s += std::min(of[c].size(), cap);
        return s;
    };
    for (size_t i = 0; i < budget; ++i) {
        if (a.nref == 3 && usable(3) >= add) break;                         // every class can give its third
        if (now_s() - t0 > a.opt.search_seconds / 3.0) {
            // A third of the time is gone and the driver keeps handing out chunks of two classes: two classes IN BALANCE (none gives
            // more than three fifths) will do.  The fused kernels are as fast on them as on three (C2's launch: 0.804 / 0.807 on
            // 8 + 7 stripes of two classes, 0.799-0.805 on 5 + 5 + 5 in the other processes of the same box,
            // profiles/r06_placement_reliability.json box 4) -- only the bare store pattern gains from the third class.
            size_t tk[3];
            const size_t have[3] = {of[0].size(), of[1].size(), of[2].size()};
            if (placed::plan(add, have, tk) && 5 * std::max({tk[0], tk[1], tk[2]}) <= 3 * add) break;
        }
        if (now_s() - t0 > a.opt.search_seconds) {                          // out of time: settle for an uneven deal if there is one
            size_t tk[3];
            const size_t have[3] = {of[0].size(), of[1].size(), of[2].size()};
            if (placed::plan(add, have, tk) || now_s() - t0 > 3.0 * a.opt.search_seconds) break;
        }
        Chunk c;
        if (hipMemCreate(&c.h, S, &a.prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }      // out of memory: settle
        ++a.created;
        pool.push_back(c);
        Chunk& x = pool.back();
        a.peak_held = std::max<int64_t>(a.peak_held, (int64_t)((pool.size() + a.stripes.size() + nref0) * S));
        char* slot = stage + i * S;
        if ((rc = map_chunk(a, x, slot)) != GINSIM_OK) break;
        if (hipMemSetAccess(slot, S, &a.access, 1) != hipSuccess) { set_error("placed memory: hipMemSetAccess failed"); rc = GINSIM_ERR_HIP; break; }
        // The same-class level s: a pair of streams into ONE class takes as long as twice a single-window fill of one chunk.
        // Calibrated warm (the first 20-30 ms of back-to-back launches of a process run at a lower clock) and kept as the
        // smallest value the first chunks of a search give (noise only ever makes a fill slower).
        if (a.anchor_ms <= 0.0) {
            const double w0 = now_s();
            while (now_s() - w0 < 0.05) {
                for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(placed_window_fill, dim3(PROBE_BLOCKS), dim3(256), 0, a.stream, (d2*)slot, S / 4096);
                (void)hipStreamSynchronize(a.stream);
            }
        }
        if (a.anchor_ms <= 0.0 || i < 8) {
            double w = 0;
            if ((rc = window_ms(a, slot, &w)) != GINSIM_OK) break;
            if (a.anchor_ms <= 0.0 || 2.0 * w < a.anchor_ms) a.anchor_ms = 2.0 * w;
        }
        double t[3] = {0, 0, 0};
        for (int r = 0; r < a.nref && rc == GINSIM_OK; ++r) rc = pair_ms(a, a.ref[r].at, slot, &t[r]);
        if (rc != GINSIM_OK) break;
        int best = -1; double tmax = 0, second = 0;
        bool clear_of_all = true;
        for (int r = 0; r < a.nref; ++r) {
            if (t[r] > tmax) { second = tmax; tmax = t[r]; best = r; } else if (t[r] > second) second = t[r];
            if (t[r] > OTHER * a.anchor_ms) clear_of_all = false;
        }
        if (a.nref < 3 && clear_of_all) {       // below the different-class mark against every reference so far: a new class
            x.cls = a.nref;                     // (probed against where it is until the search ends, then moved to a range of its own)
            x.is_ref = true;
            if (trace) fprintf(stderr, "placed: chunk %zu  pair ms %.4f %.4f %.4f  same-class level %.4f -> reference %c\n", i, t[0], t[1], t[2], a.anchor_ms, 'A' + x.cls);
            a.ref[a.nref++] = x;
            continue;
        }
        if (best >= 0 && tmax >= SAME * a.anchor_ms && (a.nref == 1 || tmax >= MARGIN * second)) x.cls = best;
        else ++a.ambiguous;
        if (trace) fprintf(stderr, "placed: chunk %zu  pair ms %.4f %.4f %.4f  same-class level %.4f -> %c\n", i, t[0], t[1], t[2], a.anchor_ms, x.cls < 0 ? '?' : 'A' + x.cls);
        if (x.cls >= 0) of[x.cls].push_back(pool.size() - 1);
    }
    // settle: every class gives an equal share of `add`; what a short class cannot give the others make up (water-filling).
    // A request of four stripes or more must come from at least two classes with no class giving more than three quarters.
    size_t takes[3] = {0, 0, 0};
    const size_t have[3] = {of[0].size(), of[1].size(), of[2].size()};
    if (rc == GINSIM_OK && !placed::plan(add, have, takes)) {
        set_error("placed memory: device %d: %d class(es) of physical memory told apart within %.0f GiB / %.1f s (chunks by class: %zu %zu %zu, wanted %zu)",
                  a.device, a.nref, pool.size() * S / (double)GiB, now_s() - t0, of[0].size(), of[1].size(), of[2].size(), add);
        rc = GINSIM_ERR_PLACED;
        a.failed = a.stripes.empty();           // an arena that exists keeps serving what it has
    }
    if (rc == GINSIM_OK)
        for (int c = 0; c < 3; ++c)
            for (size_t q = 0; q < takes[c]; ++q) pool[of[c][q]].taken = true;
    // Everything leaves the staging range together; what is neither taken nor a reference goes back to the driver when the
    // reservation is freed (fact 2).  A failure in here leaves mappings nobody can vouch for: it ends the search as a HIP error.
    (void)hipStreamSynchronize(a.stream);
    int down = GINSIM_OK;
    for (Chunk& c : pool) {
        const int u = unmap_chunk(a, c);
        if (u != GINSIM_OK) down = u;
        if (!c.taken && !c.is_ref && hipMemRelease(c.h) != hipSuccess) down = GINSIM_ERR_HIP;
    }
    { const int f = driver_flush(); if (f != GINSIM_OK) down = f; }
    retire_range(stage, budget * S);
    // the references found in this search: a range of their own each
    for (int r = nref0; r < a.nref; ++r) {
        a.ref[r].at = nullptr;
        char* home = nullptr;
        int m = down == GINSIM_OK ? reserve_fresh(S, &home) : down;
        if (m == GINSIM_OK) m = map_chunk(a, a.ref[r], home);
        if (m == GINSIM_OK && hipMemSetAccess(home, S, &a.access, 1) != hipSuccess) { set_error("placed memory: hipMemSetAccess failed"); m = GINSIM_ERR_HIP; }
        if (m != GINSIM_OK) down = m;
    }
    if (down != GINSIM_OK) {                    // the new references are unusable: forget them (their chunks with them) and stop
        for (int r = nref0; r < a.nref; ++r) { (void)unmap_chunk(a, a.ref[r]); (void)hipMemRelease(a.ref[r].h); a.ref[r] = Chunk(); }
        a.nref = nref0;
        for (Chunk& c : pool) if (c.taken) (void)hipMemRelease(c.h);
        a.last_s = now_s() - t0; a.search_s += a.last_s;
        return rc != GINSIM_OK ? rc : down;
    }
    if (rc == GINSIM_OK) {
        const size_t first = a.stripes.size();
        size_t next[3] = {0, 0, 0};
        for (const int c : placed::deal(first, add, takes)) {       // the classes of the new stripes, in address order
            Chunk& src = pool[of[c][next[c]++]];
            Chunk ch = src;
            ch.taken = ch.is_ref = false;
            if ((rc = map_chunk(a, ch, a.va + a.stripes.size() * S)) != GINSIM_OK) break;
            src.taken = false;                              // it is the arena's now
            a.stripes.push_back(ch);
        }
        for (Chunk& c : pool) if (c.taken) (void)hipMemRelease(c.h);       // (a map failed: what was not reached)
        if (rc == GINSIM_OK && a.stripes.size() > first &&
            hipMemSetAccess(a.va + first * S, (a.stripes.size() - first) * S, &a.access, 1) != hipSuccess) {
            set_error("placed memory: hipMemSetAccess on the arena failed");
            rc = GINSIM_ERR_HIP;
        }
        if (rc != GINSIM_OK) {                  // nothing half-mapped stays behind: the new stripes go back, the arena is what it was
            while (a.stripes.size() > first) {
                (void)unmap_chunk(a, a.stripes.back());
                (void)hipMemRelease(a.stripes.back().h);
                a.stripes.pop_back();
            }
            (void)driver_flush();
            a.no_growth = true;                 // the addresses above `first` were mapped once: this arena does not grow again
        } else if (a.stripes.size() > first) {
            a.regions.extend((a.stripes.size() - first) * S);       // the new stripes are free space
        }
    }
    a.last_s = now_s() - t0;
    a.search_s += a.last_s;
    return rc;
}

size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// Give everything back: unmap, release, flush, free the reservations (the memory returns with them; the next arena reserves anew).
void drop_all(Arena& a) {
    if (!a.va && !a.nref && a.stripes.empty()) return;
    (void)hipSetDevice(a.device);
    (void)hipDeviceSynchronize();
    if (getenv("GINSIM_PLACED_TRACE")) fprintf(stderr, "placed: dropping %zu stripes and %d references\n", a.stripes.size(), a.nref);
    for (Chunk& c : a.stripes) { (void)unmap_chunk(a, c); (void)hipMemRelease(c.h); }
    a.stripes.clear();
    char* homes[3] = {nullptr, nullptr, nullptr};
    for (int r = 0; r < a.nref; ++r) { homes[r] = a.ref[r].at; (void)unmap_chunk(a, a.ref[r]); (void)hipMemRelease(a.ref[r].h); a.ref[r] = Chunk(); }
    (void)driver_flush();
    for (int r = 0; r < a.nref; ++r) retire_range(homes[r], a.stripe());
    if (a.va) { retire_range(a.va, a.va_bytes); a.va = nullptr; a.va_bytes = 0; }
    (void)hipGetLastError();
    a.nref = 0;
    a.anchor_ms = 0.0;
    a.regions.clear();
    a.failed = a.no_growth = false;
}

}  // namespace

int placed_configure(int device, const ginsim_placed_options& o) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    if (!a.stripes.empty() || a.nref) { set_error("placed memory: the arena of device %d exists already (configure before the first placed request, or release it)", device); return GINSIM_ERR_ARG; }
    if (o.stripe_bytes < 0 || (o.stripe_bytes && (o.stripe_bytes < (int64_t)(64 * MiB) || (o.stripe_bytes & (o.stripe_bytes - 1))))) {
        set_error("placed memory: stripe_bytes must be 0 or a power of two >= 64 MiB"); return GINSIM_ERR_ARG;
    }
    if (o.budget_bytes < 0 || o.limit_bytes < 0 || !(o.search_seconds >= 0.0)) { set_error("placed memory: negative option"); return GINSIM_ERR_ARG; }
    drop_all(a);                                // an empty range reserved under the old options
    a.opt = o;
    a.configured = false;
    return GINSIM_OK;
}

int placed_reserve(int device, hipStream_t stream, size_t bytes) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    a.stream = stream;
    int rc = ensure_ready(a);
    if (rc) return rc;
    // what a first-fit carve of `bytes` (in up to a few regions) can count on: the free space
    const size_t free_total = a.regions.free_total();
    bytes = round_up(bytes, GRAIN) + 4 * GRAIN;
    if (free_total >= bytes) return GINSIM_OK;
    return grow(a, (bytes - free_total + a.stripe() - 1) / a.stripe());
}

int placed_malloc(int device, hipStream_t stream, size_t bytes, const void* owner, void** out) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    a.stream = stream;
    int rc = ensure_ready(a);
    if (rc) return rc;
    const size_t size = round_up(bytes ? bytes : 8, GRAIN);
    for (int attempt = 0; attempt < 2; ++attempt) {
        size_t off = 0;
        if (a.regions.carve(size, owner, &off)) {
            *out = a.va + off;
            return GINSIM_OK;
        }
        if (attempt) break;
        const size_t tail = a.regions.free_tail();          // a free block at the end of the mapped range is extended by the growth
        rc = grow(a, (size - tail + a.stripe() - 1) / a.stripe());
        if (rc) return rc;
    }
    set_error("placed memory: no room for %.1f MiB after growing the arena", size / (double)MiB);
    return GINSIM_ERR_PLACED;
}

bool placed_owns(int device, const void* p) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    return a.va && (const char*)p >= a.va && (const char*)p < a.va + a.va_bytes;
}

int placed_free(int device, void* p) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    if (!a.regions.give_back((size_t)((char*)p - a.va))) {
        set_error("placed memory: %p is not the start of a region carved from the arena of device %d", p, device);
        return GINSIM_ERR_ARG;
    }
    return GINSIM_OK;
}

// a context is going away: what it carved and never freed (buffers that outlive their context, garbage collected in any order)
// returns to the free list
void placed_free_owner(int device, const void* owner) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    (void)a.regions.give_back_all_of(owner);
}

int placed_release(int device, bool force) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    if (!a.regions.nothing_carved() && !force) return GINSIM_OK;
    drop_all(a);
    return GINSIM_OK;
}

void placed_info(int device, ginsim_placed_info* out) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    std::memset(out, 0, sizeof(*out));
    int seen[3] = {0, 0, 0};
    for (size_t i = 0; i < a.stripes.size(); ++i) {
        const int c = a.stripes[i].cls;
        if (c >= 0 && c < 3) { ++out->stripes_of_class[c]; seen[c] = 1; }
        if (i < sizeof(out->stripe_classes) - 1) out->stripe_classes[i] = c >= 0 && c < 3 ? (char)('A' + c) : '?';
    }
    out->available = !a.stripes.empty() && seen[0] + seen[1] + seen[2] >= 2;
    out->classes = a.nref;
    out->searches = a.searches;
    out->failed = a.failed;
    out->stripe_bytes = a.opt.stripe_bytes;
    out->mapped_bytes = (int64_t)a.mapped();
    out->used_bytes = (int64_t)a.regions.used_bytes();
    out->limit_bytes = a.opt.limit_bytes;
    out->chunks_created = a.created;
    out->chunks_ambiguous = a.ambiguous;
    out->probes = a.probes;
    out->peak_held_bytes = a.peak_held;
    out->search_seconds = a.search_s;
    out->last_search_seconds = a.last_s;
    out->anchor_ms = a.anchor_ms;
}

void placed_context_created(int device) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    ++a.contexts;
}

void placed_context_destroyed(int device) {
    Arena& a = *arena_of(device);
    std::lock_guard<std::mutex> lk(a.mu);
    if (--a.contexts <= 0) { a.contexts = 0; drop_all(a); }
}

}  // namespace ginsim
