// The parts of the placed arena (csrc/placed.hip) that touch no device: the first-fit free list over the arena's range, the plan of
// how many chunks each class gives to a growth, and the order of the classes inside a group of stripes.  Plain C++, so that the CPU
// test suite exercises them (tests/test_host_cpu.py compiles tests/cpp/placed_logic_check.cpp against this header).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <map>
#include <utility>
#include <vector>

namespace ginsim {
namespace placed {

// Regions of a range [0, end) handed out first-fit, merged with their neighbours when they come back.  Offsets and sizes in bytes;
// every region remembers who carved it (a context), so that what a context leaves behind can be returned when it goes away.
class FreeList {
public:
    // `bytes` more at the end of the range (the arena grew): free space, merged with a free block that ends where it begins
    void extend(size_t bytes) {
        size_t off = end_, len = bytes;
        if (!free_.empty()) {
            auto last = std::prev(free_.end());
            if (last->first + last->second == off) { off = last->first; len += last->second; free_.erase(last); }
        }
        free_[off] = len;
        end_ += bytes;
    }
    // first fit; false when no free block is large enough
    bool carve(size_t size, const void* owner, size_t* offset) {
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second < size) continue;
            const size_t off = it->first, len = it->second;
            free_.erase(it);
            if (len > size) free_[off + size] = len - size;
            used_[off] = {size, owner};
            used_bytes_ += size;
            *offset = off;
            return true;
        }
        return false;
    }
    // false when `offset` is not the start of a carved region
    bool give_back(size_t offset) {
        auto it = used_.find(offset);
        if (it == used_.end()) return false;
        release(it);
        return true;
    }
    // everything `owner` carved and did not give back; returns how many regions
    size_t give_back_all_of(const void* owner) {
        size_t k = 0;
        for (auto it = used_.begin(); it != used_.end();) {
            auto cur = it++;
            if (cur->second.second == owner) { release(cur); ++k; }
        }
        return k;
    }
    size_t free_total() const { size_t s = 0; for (auto& f : free_) s += f.second; return s; }
    // the free block that ends where the range ends (a growth extends it), 0 if there is none
    size_t free_tail() const {
        if (free_.empty()) return 0;
        auto last = std::prev(free_.end());
        return last->first + last->second == end_ ? last->second : 0;
    }
    size_t used_bytes() const { return used_bytes_; }
    size_t end() const { return end_; }
    bool nothing_carved() const { return used_.empty(); }
    size_t free_blocks() const { return free_.size(); }
    void clear() { free_.clear(); used_.clear(); used_bytes_ = 0; end_ = 0; }

private:
    void release(std::map<size_t, std::pair<size_t, const void*>>::iterator it) {
        size_t o = it->first, len = it->second.first;
        used_bytes_ -= len;
        used_.erase(it);
        auto nx = free_.lower_bound(o);
        if (nx != free_.end() && o + len == nx->first) { len += nx->second; nx = free_.erase(nx); }
        if (nx != free_.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == o) { o = pv->first; len += pv->second; free_.erase(pv); } }
        free_[o] = len;
    }
    std::map<size_t, size_t> free_;                                  // offset -> bytes
    std::map<size_t, std::pair<size_t, const void*>> used_;           // offset -> (bytes, who carved it)
    size_t used_bytes_ = 0, end_ = 0;
};

// How many chunks each of the three classes gives to a growth of `add` stripes when `have[c]` chunks of class c were found: equal
// shares, the others making up what a short class lacks (water-filling).  false: not enough chunks, or (four stripes and more) one
// class would give more than three quarters of them -- a region carved from such stripes would mostly lie in one class.
inline bool plan(size_t add, const size_t (&have)[3], size_t (&takes)[3]) {
    takes[0] = takes[1] = takes[2] = 0;
    size_t need = add;
    while (need > 0) {
        int active = 0;
        for (int c = 0; c < 3; ++c) active += takes[c] < have[c];
        if (!active) return false;
        const size_t share = (need + active - 1) / active;
        for (int c = 0; c < 3 && need > 0; ++c) {
            const size_t g = std::min({share, have[c] - takes[c], need});
            takes[c] += g;
            need -= g;
        }
    }
    if (add < 4) return true;
    return 4 * std::max({takes[0], takes[1], takes[2]}) <= 3 * add;
}

// The order of the classes inside the t-th group of three stripes: one of the six permutations, chosen by a hash of t, so that no
// plane stride meets the same class at every one of its planes (a strict A B C A B C ... would, for planes of three stripes).
inline void group_order(size_t t, int (&order)[3]) {
    static const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {2, 1, 0}, {1, 0, 2}};
    const uint32_t h = ((uint32_t)t * 2654435761u) >> 13;
    for (int i = 0; i < 3; ++i) order[i] = perm[h % 6][i];
}

// The classes of `add` new stripes in address order, the first of them being stripe number `first` of the arena, when class c gives
// takes[c] of them: group after group in group_order, a class that has given its share is passed over.
inline std::vector<int> deal(size_t first, size_t add, const size_t (&takes)[3]) {
    std::vector<int> out;
    size_t given[3] = {0, 0, 0};
    for (size_t g = 0; out.size() < add; ++g) {
        int order[3];
        group_order(first / 3 + g, order);
        bool any = false;
        for (int j = 0; j < 3 && out.size() < add; ++j) {
            const int c = order[j];
            if (given[c] >= takes[c]) continue;
            ++given[c];
            out.push_back(c);
            any = true;
        }
        if (!any && given[0] >= takes[0] && given[1] >= takes[1] && given[2] >= takes[2]) break;      // (takes do not add up to `add`)
    }
    return out;
}

}  // namespace placed
}  // namespace ginsim
