#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace mipt {
namespace {

struct Aabb {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::max();
            hi[k] = std::numeric_limits<float>::lowest();
        }
    }
    void grow(const float* l, const float* h) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], l[k]);
            hi[k] = std::max(hi[k], h[k]);
        }
    }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 16;

struct Builder {
    std::vector<BuildBox>& b;
    std::vector<BvhNode> nodes;
    int max_depth;
    int depth_seen = 0;

    static int ceil_log2(size_t n) {
        int l = 0;
        while ((size_t(1) << l) < n) ++l;
        return l;
    }

    // Splits [begin, end) and returns the split position; level = depth of the node being made (root = 1).
    size_t split(size_t begin, size_t end, int level) {
        const size_t n = end - begin;
        Aabb cb;   // bounds of the box centres
        cb.reset();
        for (size_t i = begin; i < end; ++i) {
            const float c[3] = {0.5f * (b[i].lo[0] + b[i].hi[0]), 0.5f * (b[i].lo[1] + b[i].hi[1]), 0.5f * (b[i].lo[2] + b[i].hi[2])};
            cb.grow(c, c);
        }
        int axis = 0;
        float ext = cb.hi[0] - cb.lo[0];
        for (int k = 1; k < 3; ++k)
            if (cb.hi[k] - cb.lo[k] > ext) {
                ext = cb.hi[k] - cb.lo[k];
                axis = k;
            }
        auto centre = [&](const BuildBox& x, int k) { return 0.5f * (x.lo[k] + x.hi[k]); };
        auto median = [&]() {
            const size_t mid = begin + n / 2;
            std::nth_element(b.begin() + begin, b.begin() + mid, b.begin() + end,
                             [&](const BuildBox& x, const BuildBox& y) { return centre(x, axis) < centre(y, axis); });
            return mid;
        };
        // A subtree of n leaves needs ceil(log2 n) more levels when split at the median.
        const bool must_balance = level + ceil_log2(n) >= max_depth;
        if (n <= 2 || !(ext > 0.0f) || must_balance) return median();

        // binned SAH on the three axes
        float best_cost = std::numeric_limits<float>::max();
        int best_axis = -1, best_bin = -1;
        for (int k = 0; k < 3; ++k) {
            const float e = cb.hi[k] - cb.lo[k];
            if (!(e > 0.0f)) continue;
            Aabb bin_box[kBins];
            size_t bin_n[kBins] = {};
            for (auto& bb : bin_box) bb.reset();
            const float scale = kBins / e;
            for (size_t i = begin; i < end; ++i) {
                int bi = static_cast<int>((centre(b[i], k) - cb.lo[k]) * scale);
                bi = std::min(std::max(bi, 0), kBins - 1);
                bin_box[bi].grow(b[i].lo, b[i].hi);
                ++bin_n[bi];
            }
            float right_area[kBins];
            size_t right_n[kBins];
            Aabb acc;
            acc.reset();
            size_t cnt = 0;
            for (int i = kBins - 1; i > 0; --i) {
                if (bin_n[i]) acc.grow(bin_box[i].lo, bin_box[i].hi);
                cnt += bin_n[i];
                right_area[i] = cnt ? acc.half_area() : 0.0f;
                right_n[i] = cnt;
            }
            acc.reset();
            cnt = 0;
            for (int i = 0; i < kBins - 1; ++i) {
                if (bin_n[i]) acc.grow(bin_box[i].lo, bin_box[i].hi);
                cnt += bin_n[i];
                if (cnt == 0 || right_n[i + 1] == 0) continue;
                const float cost = acc.half_area() * static_cast<float>(cnt) + right_area[i + 1] * static_cast<float>(right_n[i + 1]);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = k;
                    best_bin = i;
                }
            }
        }
        if (best_axis < 0) return median();
        const float e = cb.hi[best_axis] - cb.lo[best_axis];
        const float scale = kBins / e;
        const float lo = cb.lo[best_axis];
        auto it = std::partition(b.begin() + begin, b.begin() + end, [&](const BuildBox& x) {
            int bi = static_cast<int>((centre(x, best_axis) - lo) * scale);
            bi = std::min(std::max(bi, 0), kBins - 1);
            return bi <= best_bin;
        });
        const size_t mid = static_cast<size_t>(it - b.begin());
        if (mid == begin || mid == end) return median();
        return mid;
    }

    static void set_child(BvhNode& nd, int side, const Aabb& box, int32_t link) {
        float* lo = side == 0 ? nd.lo0 : nd.lo1;
        float* hi = side == 0 ? nd.hi0 : nd.hi1;
        std::memcpy(lo, box.lo, sizeof(float) * 3);
        std::memcpy(hi, box.hi, sizeof(float) * 3);
        (side == 0 ? nd.child0 : nd.child1) = link;
    }

    // Builds the subtree over [begin, end), n >= 2, into node `self`; returns its bounds.
    Aabb build(size_t begin, size_t end, int32_t self, int level) {
        depth_seen = std::max(depth_seen, level);
        const size_t mid = split(begin, end, level);
        Aabb total;
        total.reset();
        const size_t range[2][2] = {{begin, mid}, {mid, end}};
        for (int side = 0; side < 2; ++side) {
            const size_t lo = range[side][0], hi = range[side][1];
            Aabb box;
            if (hi - lo == 1) {
                box.reset();
                box.grow(b[lo].lo, b[lo].hi);
                set_child(nodes[self], side, box, ~static_cast<int32_t>(b[lo].prim));
            } else {
                const int32_t child = static_cast<int32_t>(nodes.size());
                nodes.emplace_back();
                box = build(lo, hi, child, level + 1);
                set_child(nodes[self], side, box, child);
            }
            total.grow(box.lo, box.hi);
        }
        return total;
    }
};

}   // namespace

BvhResult build_bvh(std::vector<BuildBox> boxes, int max_depth) {
    BvhResult r;
    if (boxes.empty()) return r;
    Builder bl{boxes, {}, max_depth};
    bl.nodes.reserve(boxes.size());
    bl.nodes.emplace_back();
    if (boxes.size() == 1) {
        Aabb box, none;
        box.reset();
        box.grow(boxes[0].lo, boxes[0].hi);
        // The absent second child: a box at +infinity on every axis.  Both plane distances of an axis are then
        // infinite with the sign of the ray's direction there, so the slab test's entry distance is +inf (beyond any
        // limit) or its exit distance is -inf (before the entry): no ray ever enters it.  (An inverted finite box
        // would not do: the test orders the two plane distances itself, and every ray would pass.)
        const float inf = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) none.lo[a] = none.hi[a] = inf;
        Builder::set_child(bl.nodes[0], 0, box, ~static_cast<int32_t>(boxes[0].prim));
        Builder::set_child(bl.nodes[0], 1, none, ~static_cast<int32_t>(boxes[0].prim));
        bl.depth_seen = 1;
    } else {
        bl.build(0, boxes.size(), 0, 1);
    }
    r.nodes = std::move(bl.nodes);
    r.depth = bl.depth_seen;
    return r;
}

}   // namespace mipt
