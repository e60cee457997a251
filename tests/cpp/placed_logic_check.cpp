// CPU check of the device-free parts of the placed arena (gnss-ins-sim_amd/csrc/placed_logic.hpp): compiled with g++ and run by
// tests/test_host_cpu.py::test_placed_arena_logic_on_the_host.  Exit code 0 and "ok" on success; the first failed check otherwise.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "placed_logic.hpp"

using namespace ginsim::placed;

#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) { printf("FAILED line %d: %s\n", __LINE__, #cond); return 1; } \
    } while (0)

int main() {
    const size_t M = (size_t)1 << 20;
    // ---- the free list: first fit, holes, merging, growth, owners
    {
        FreeList f;
        size_t a = 0, b = 0, c = 0, d = 0;
        int ctx1 = 0, ctx2 = 0;
        CHECK(!f.carve(M, &ctx1, &a) && f.free_total() == 0 && f.nothing_carved());
        f.extend(512 * M);
        CHECK(f.end() == 512 * M && f.free_tail() == 512 * M && f.free_blocks() == 1);
        CHECK(f.carve(300 * M, &ctx1, &a) && a == 0);
        CHECK(!f.carve(300 * M, &ctx1, &b) && f.free_tail() == 212 * M);          // does not fit: the caller grows by (300 - 212) rounded up
        f.extend(512 * M);                                                        // merged with the free tail
        CHECK(f.free_blocks() == 1 && f.free_tail() == 724 * M);
        CHECK(f.carve(700 * M, &ctx2, &b) && b == 300 * M);
        CHECK(f.carve(24 * M, &ctx1, &c) && c == 1000 * M && f.free_total() == 0 && f.free_tail() == 0);
        CHECK(f.used_bytes() == 1024 * M);
        CHECK(!f.give_back(b + 2 * M));                                           // not the start of a region
        CHECK(f.give_back(b) && f.free_total() == 700 * M && f.free_blocks() == 1);
        CHECK(f.carve(100 * M, &ctx1, &d) && d == b);                             // first fit: the hole
        CHECK(f.carve(600 * M, &ctx2, &b) && b == 400 * M);
        CHECK(f.give_back(d) && f.give_back(a));                                  // two neighbours merge: [0, 400 M)
        CHECK(f.free_blocks() == 1 && f.free_total() == 400 * M && f.free_tail() == 0);
        CHECK(f.give_back_all_of(&ctx1) == 1 && f.free_blocks() == 2);            // c at the end; b (ctx2) still between
        CHECK(f.give_back_all_of(&ctx2) == 1 && f.nothing_carved() && f.free_blocks() == 1 && f.free_total() == 1024 * M && f.used_bytes() == 0);
        CHECK(!f.give_back(0));
        // random traffic: never two regions overlapping, the bytes add up, everything merges back into one block
        std::mt19937 rng(7);
        std::vector<std::pair<size_t, size_t>> live;
        for (int it = 0; it < 20000; ++it) {
            if (live.empty() || (rng() % 3 != 0 && live.size() < 200)) {
                const size_t sz = ((rng() % 64) + 1) * 2 * M;
                size_t off = 0;
                if (!f.carve(sz, &ctx1, &off)) { f.extend(512 * M); CHECK(f.carve(sz, &ctx1, &off)); }
                CHECK(off + sz <= f.end());
                for (auto& l : live) CHECK(off + sz <= l.first || l.first + l.second <= off);
                live.emplace_back(off, sz);
            } else {
                const size_t k = rng() % live.size();
                CHECK(f.give_back(live[k].first));
                live.erase(live.begin() + k);
            }
            size_t used = 0;
            for (auto& l : live) used += l.second;
            CHECK(used == f.used_bytes() && used + f.free_total() == f.end());
        }
        for (auto& l : live) CHECK(f.give_back(l.first));
        CHECK(f.nothing_carved() && f.free_blocks() == 1 && f.free_total() == f.end());
    }
    // ---- the plan: equal shares, water-filling, the three-quarters rule
    {
        size_t t[3];
        { const size_t h[3] = {10, 10, 10}; CHECK(plan(15, h, t) && t[0] == 5 && t[1] == 5 && t[2] == 5); }
        { const size_t h[3] = {20, 3, 20}; CHECK(plan(15, h, t) && t[1] == 3 && t[0] + t[2] == 12 && t[0] <= 6 + 1 && t[2] <= 6 + 1); }
        { const size_t h[3] = {20, 2, 1}; CHECK(!plan(15, h, t)); }                     // 12 of 15 from one class: more than three quarters
        { const size_t h[3] = {20, 4, 0}; CHECK(plan(15, h, t) && t[0] == 11 && t[1] == 4 && t[2] == 0); }   // two classes, 11 <= 11.25
        { const size_t h[3] = {20, 0, 0}; CHECK(!plan(4, h, t) && plan(3, h, t) && t[0] == 3); }            // small requests: any chunk
        { const size_t h[3] = {2, 2, 2}; CHECK(!plan(7, h, t) && plan(6, h, t)); }                          // not enough chunks
        for (size_t add = 1; add < 60; ++add)
            for (size_t x = 0; x < 40; x += 3) {
                const size_t h[3] = {x, 40 - x, 13};
                if (plan(add, h, t)) {
                    CHECK(t[0] + t[1] + t[2] == add && t[0] <= h[0] && t[1] <= h[1] && t[2] <= h[2]);
                    if (add >= 4) CHECK(4 * std::max({t[0], t[1], t[2]}) <= 3 * add);
                }
            }
    }
    // ---- the deal: the right number of every class, every group a permutation, no plane stride stuck in one class
    {
        const size_t takes[3] = {10, 10, 10};
        std::vector<int> d = deal(0, 30, takes);
        CHECK(d.size() == 30);
        size_t cnt[3] = {0, 0, 0};
        for (int c : d) { CHECK(c >= 0 && c < 3); ++cnt[c]; }
        CHECK(cnt[0] == 10 && cnt[1] == 10 && cnt[2] == 10);
        std::set<int> orders;
        for (size_t g = 0; g < 10; ++g) {
            std::set<int> grp{d[3 * g], d[3 * g + 1], d[3 * g + 2]};
            CHECK(grp.size() == 3);
            orders.insert(d[3 * g] * 9 + d[3 * g + 1] * 3 + d[3 * g + 2]);
        }
        CHECK(orders.size() >= 3);                                                  // not one fixed order
        // planes of exactly three stripes (the stride a fixed A B C order would defeat): the 15 fronts of a launch, at every phase
        std::vector<int> big = deal(0, 192, (const size_t[3]){64, 64, 64});
        for (size_t phase = 0; phase < 3; ++phase) {
            size_t load[3] = {0, 0, 0};
            for (size_t p = 0; p < 15; ++p) ++load[big[3 * p + phase]];
            CHECK(std::max({load[0], load[1], load[2]}) <= 9);                      // a fixed order would put all 15 into one class
        }
        // a growth continues the sequence of groups; uneven takes: the short class is passed over
        const size_t uneven[3] = {6, 1, 5};
        std::vector<int> u = deal(30, 12, uneven);
        size_t cu[3] = {0, 0, 0};
        for (int c : u) ++cu[c];
        CHECK(u.size() == 12 && cu[0] == 6 && cu[1] == 1 && cu[2] == 5);
    }
    printf("ok\n");
    return 0;
}
