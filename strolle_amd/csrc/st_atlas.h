// st_atlas.h — rectangle allocator of the image atlas (strolle/src/images.rs:54-127).
//
// The reference keeps every image in one 8192 x 8192 RGBA8 texture and places rectangles with the `guillotiere` crate
// (allocate on insert, deallocate on remove or when an image comes back with another size). guillotiere is not part of
// the reference tree, and where a rectangle lands is invisible to rendering (materials carry the rectangle, sampling
// clamps to it), so the placement policy here is this project's own: shelves, with freed spans kept per shelf and reused.
//   * a shelf is a horizontal band [y, y + h) holding rectangles left to right; shelves stack bottom to top;
//   * allocate: the lowest-height closed shelf that has a free span wide enough (leftmost span), else the top shelf
//     (whose height may still grow), else a new shelf on top, else "no more space" (images.rs:71-79 warns and drops);
//   * release: the span goes back to its shelf and merges with free neighbours; top shelves that became empty are popped.
// Deterministic: placement depends only on the sequence of calls.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace st {

class AtlasShelves {
  public:
    AtlasShelves(uint32_t width, uint32_t max_height) : width_(width), max_height_(max_height) {}

    bool allocate(uint32_t w, uint32_t h, uint32_t* x, uint32_t* y) {
        if (w == 0 || h == 0 || w > width_ || h > max_height_) return false;
        // closed shelves: every shelf but the top one has a fixed height
        int best = -1;
        for (size_t i = 0; i + 1 < shelves_.size(); i++) {
            const Shelf& s = shelves_[i];
            if (s.h < h || (best >= 0 && shelves_[(size_t)best].h <= s.h)) continue;
            if (find_span(s, w) >= 0) best = (int)i;
        }
        if (best >= 0) return take(shelves_[(size_t)best], w, x, y);
        if (!shelves_.empty()) {
            Shelf& top = shelves_.back();
            const uint32_t grown = top.h > h ? top.h : h;
            if (find_span(top, w) >= 0 && top.y + grown <= max_height_) {
                top.h = grown;
                return take(top, w, x, y);
            }
        }
        const uint32_t y0 = shelves_.empty() ? 0u : shelves_.back().y + shelves_.back().h;
        if (y0 + h > max_height_ || y0 + h < y0) return false;
        shelves_.push_back(Shelf{y0, h, {{0u, width_}}});
        return take(shelves_.back(), w, x, y);
    }

    void release(uint32_t x, uint32_t y, uint32_t w) {
        for (Shelf& s : shelves_) {
            if (s.y != y) continue;
            size_t at = 0;
            while (at < s.free.size() && s.free[at].first < x) at++;
            s.free.insert(s.free.begin() + (ptrdiff_t)at, std::make_pair(x, x + w));
            if (at + 1 < s.free.size() && s.free[at].second == s.free[at + 1].first) {
                s.free[at].second = s.free[at + 1].second;
                s.free.erase(s.free.begin() + (ptrdiff_t)at + 1);
            }
            if (at > 0 && s.free[at - 1].second == s.free[at].first) {
                s.free[at - 1].second = s.free[at].second;
                s.free.erase(s.free.begin() + (ptrdiff_t)at);
            }
            break;
        }
        while (!shelves_.empty() && shelves_.back().free.size() == 1 && shelves_.back().free[0].first == 0 && shelves_.back().free[0].second == width_)
            shelves_.pop_back();
    }

    uint32_t used_height() const { return shelves_.empty() ? 0u : shelves_.back().y + shelves_.back().h; }

  private:
    struct Shelf {
        uint32_t y, h;
        std::vector<std::pair<uint32_t, uint32_t>> free;  // sorted, disjoint, non-adjacent [x0, x1) spans
    };
    uint32_t width_, max_height_;
    std::vector<Shelf> shelves_;

    static int find_span(const Shelf& s, uint32_t w) {
        for (size_t i = 0; i < s.free.size(); i++)
            if (s.free[i].second - s.free[i].first >= w) return (int)i;
        return -1;
    }
    static bool take(Shelf& s, uint32_t w, uint32_t* x, uint32_t* y) {
        const int i = find_span(s, w);
        if (i < 0) return false;
        *x = s.free[(size_t)i].first;
        *y = s.y;
        s.free[(size_t)i].first += w;
        if (s.free[(size_t)i].first == s.free[(size_t)i].second) s.free.erase(s.free.begin() + i);
        return true;
    }
};

}  // namespace st
