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

namespace {

struct Collapser {
    const BvhNode* in;
    Bvh4Result& out;

    struct Entry {
        float lo[3], hi[3];
        int32_t link;
    };
    static Entry entry_of(const BvhNode& n, int side) {
        Entry e;
        std::memcpy(e.lo, side == 0 ? n.lo0 : n.lo1, sizeof(e.lo));
        std::memcpy(e.hi, side == 0 ? n.hi0 : n.hi1, sizeof(e.hi));
        e.link = side == 0 ? n.child0 : n.child1;
        return e;
    }
    static bool absent(const Entry& e) { return e.lo[0] == std::numeric_limits<float>::infinity(); }
    static float half_area(const Entry& e) {
        const float x = e.hi[0] - e.lo[0], y = e.hi[1] - e.lo[1], z = e.hi[2] - e.lo[2];
        return x * y + y * z + z * x;
    }

    // Emits the four-child node for two-child node `n2` at index `self` (already allocated); returns the stack
    // entries a walk of its subtree can hold at once.
    int emit(int32_t n2, int32_t self, int level) {
        out.depth = std::max(out.depth, level);
        Entry kids[4];
        int n_kids = 0;
        for (int side = 0; side < 2; ++side) {
            const Entry e = entry_of(in[n2], side);
            if (!absent(e)) kids[n_kids++] = e;
        }
        while (n_kids < 4) {
            int open = -1;
            for (int k = 0; k < n_kids; ++k)
                if (kids[k].link >= 0 && (open < 0 || half_area(kids[k]) > half_area(kids[open]))) open = k;
            if (open < 0) break;
            const BvhNode& g = in[kids[open].link];
            kids[open] = entry_of(g, 0);   // the tree builders give every inner node two real children
            kids[n_kids++] = entry_of(g, 1);
        }
        const float inf = std::numeric_limits<float>::infinity();
        // a node's inner children get consecutive indices: they are what is visited next
        int32_t child_nodes[4];
        for (int k = 0; k < 4; ++k) {
            child_nodes[k] = -1;
            if (k < n_kids && kids[k].link >= 0) {
                child_nodes[k] = static_cast<int32_t>(out.nodes.size());
                out.nodes.emplace_back();
            }
        }
        Bvh4Node& nd = out.nodes[self];
        for (int k = 0; k < 4; ++k) {
            const bool have = k < n_kids;
            nd.lox[k] = have ? kids[k].lo[0] : inf;
            nd.loy[k] = have ? kids[k].lo[1] : inf;
            nd.loz[k] = have ? kids[k].lo[2] : inf;
            nd.hix[k] = have ? kids[k].hi[0] : inf;
            nd.hiy[k] = have ? kids[k].hi[1] : inf;
            nd.hiz[k] = have ? kids[k].hi[2] : inf;
            nd.pad[k] = 0;
            if (!have) {
                nd.child[k] = kBvh4Absent;
            } else if (kids[k].link < 0) {
                nd.child[k] = kids[k].link;
                float* lb = out.leaf_boxes.data() + 8 * static_cast<size_t>(~kids[k].link);
                std::memcpy(lb, kids[k].lo, 12);
                std::memcpy(lb + 4, kids[k].hi, 12);
            } else {
                nd.child[k] = child_nodes[k];
            }
        }
        int need = n_kids - 1;
        for (int k = 0; k < n_kids; ++k)
            if (child_nodes[k] >= 0) need = std::max(need, n_kids - 1 + emit(kids[k].link, child_nodes[k], level + 1));
        return need;
    }
};

}   // namespace

Bvh4Result collapse_bvh4(const BvhNode* nodes2, size_t n_nodes2, uint32_t n_prims) {
    Bvh4Result r;
    if (n_nodes2 == 0) return r;
    r.leaf_boxes.assign(8 * static_cast<size_t>(n_prims), 0.0f);
    r.nodes.reserve(n_nodes2 / 2 + 1);
    r.nodes.emplace_back();
    Collapser c{nodes2, r};
    r.stack_need = std::max(1, c.emit(0, 0, 1));
    return r;
}

float bvh4q_plane(const Bvh4QNode& n, int axis, uint32_t q) { return std::fmaf(static_cast<float>(q), n.scale[axis], n.org[axis]); }

void quantize_bvh4_node(const Bvh4Node& n, Bvh4QNode& q) {
    const float inf = std::numeric_limits<float>::infinity();
    {
        const float* lo[3] = {n.lox, n.loy, n.loz};
        const float* hi[3] = {n.hix, n.hiy, n.hiz};
        for (int a = 0; a < 3; ++a) {
            float mn = inf, mx = -inf;
            for (int k = 0; k < 4; ++k)
                if (n.child[k] != kBvh4Absent) {
                    mn = std::min(mn, lo[a][k]);
                    mx = std::max(mx, hi[a][k]);
                }
            if (!(mn <= mx)) mn = mx = 0.0f;   // a node without children (the root of an empty tree)
            // 253 steps span the node, so that one step can be given away below every low plane and above every high one
            float scale = std::nextafter((mx - mn) / 253.0f, inf);
            if (!(scale >= 1e-30f)) scale = 1e-30f;
            q.org[a] = mn;
            q.scale[a] = scale;
            uint32_t wlo = 0, whi = 0;
            for (int k = 0; k < 4; ++k) {
                uint32_t ql = 255u, qh = 0u;
                if (n.child[k] != kBvh4Absent) {
                    const double sl = (static_cast<double>(lo[a][k]) - mn) / scale, sh = (static_cast<double>(hi[a][k]) - mn) / scale;
                    long l = static_cast<long>(std::floor(sl)) - 1, h = static_cast<long>(std::ceil(sh)) + 1;
                    l = std::min(255L, std::max(0L, l));
                    h = std::min(255L, std::max(0L, h));
                    ql = static_cast<uint32_t>(l);
                    qh = static_cast<uint32_t>(h);
                    while (ql > 0 && !(bvh4q_plane(q, a, ql) <= lo[a][k])) --ql;
                    while (qh < 255 && !(bvh4q_plane(q, a, qh) >= hi[a][k])) ++qh;
                }
                wlo |= ql << (8 * k);
                whi |= qh << (8 * k);
            }
            q.qlo[a] = wlo;
            q.qhi[a] = whi;
        }
        for (int k = 0; k < 4; ++k) q.child[k] = n.child[k];
    }
}

std::vector<Bvh4QNode> quantize_bvh4(const std::vector<Bvh4Node>& nodes) {
    std::vector<Bvh4QNode> out(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) quantize_bvh4_node(nodes[i], out[i]);
    return out;
}

}   // namespace mipt
