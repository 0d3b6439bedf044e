// Invariants of the four-child tree that the traversal kernels walk (pathtracer_amd/csrc/bvh.cpp, collapse_bvh4), checked
// on the host: built by tests/test_bvh4_collapse.py with g++ from this file and bvh.cpp, no GPU involved.
//   usage: bvh4_check <n_prims> <seed> <layout>   (layout 0: scattered boxes, 1: all coincident, 2: a long diagonal chain)
// Prints "ok ..." and exits 0 when every check holds.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>

#include "bvh.h"

using namespace mipt;

static int fail(const char* what, long a = 0, long b = 0) {
    std::printf("FAILED: %s (%ld, %ld)\n", what, a, b);
    return 1;
}

struct Walk {
    const Bvh4Result& t;
    std::vector<int> seen;          // per primitive: times met as a leaf
    std::vector<char> node_seen;
    int max_stack = 0;
    int bad = 0;
    // Depth-first walk that defers every child but one, as a ray that hits all four boxes does: `held` entries wait on the
    // stack while node n is walked.
    void go(int n, int held, const float* lo, const float* hi) {
        if (node_seen[n]++) { ++bad; return; }
        const Bvh4Node& nd = t.nodes[n];
        int kids = 0;
        for (int k = 0; k < 4; ++k) kids += nd.child[k] != kBvh4Absent;
        if (kids < 1) ++bad;
        if (held + kids - 1 > max_stack) max_stack = held + kids - 1;
        for (int k = 0; k < 4; ++k) {
            const float clo[3] = {nd.lox[k], nd.loy[k], nd.loz[k]}, chi[3] = {nd.hix[k], nd.hiy[k], nd.hiz[k]};
            if (nd.child[k] == kBvh4Absent) {
                // an absent child's box lies at +infinity on every axis
                for (int a = 0; a < 3; ++a)
                    if (!(clo[a] == std::numeric_limits<float>::infinity() && chi[a] == clo[a])) ++bad;
                continue;
            }
            for (int a = 0; a < 3; ++a) {
                if (!(clo[a] <= chi[a])) ++bad;
                if (lo && (clo[a] < lo[a] || chi[a] > hi[a])) ++bad;   // inside the box its parent holds for this node
            }
            if (nd.child[k] < 0) {
                const uint32_t prim = static_cast<uint32_t>(~nd.child[k]);
                if (prim >= seen.size()) { ++bad; continue; }
                ++seen[prim];
                const float* lb = t.leaf_boxes.data() + 8 * static_cast<size_t>(prim);
                for (int a = 0; a < 3; ++a)
                    if (lb[a] != clo[a] || lb[4 + a] != chi[a]) ++bad;   // the per-primitive copy of the leaf's box
            } else {
                if (nd.child[k] <= n || static_cast<size_t>(nd.child[k]) >= t.nodes.size()) { ++bad; continue; }
                go(nd.child[k], held + kids - 1, clo, chi);
            }
        }
    }
};

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1000;
    const unsigned seed = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 1;
    const int layout = argc > 3 ? std::atoi(argv[3]) : 0;
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> u(-8.0f, 8.0f), r(0.01f, 0.4f);
    std::vector<BuildBox> boxes(n);
    for (size_t i = 0; i < n; ++i) {
        float c[3] = {u(g), u(g), u(g)};
        float h[3] = {r(g), r(g), r(g)};
        if (layout == 1) c[0] = c[1] = c[2] = 1.0f, h[0] = h[1] = h[2] = 0.5f;
        if (layout == 2) c[0] = c[1] = c[2] = 0.001f * static_cast<float>(i), h[0] = h[1] = h[2] = 0.3f;
        for (int a = 0; a < 3; ++a) {
            boxes[i].lo[a] = c[a] - h[a];
            boxes[i].hi[a] = c[a] + h[a];
        }
        boxes[i].prim = static_cast<uint32_t>(2 * i + 1);   // primitive indices need not be dense (SDF objects take none)
    }
    const uint32_t n_prims = static_cast<uint32_t>(2 * n + 2);
    const std::vector<BuildBox> kept = boxes;
    const BvhResult two = build_bvh(std::move(boxes));
    if (n >= 2 && two.nodes.size() != n - 1) return fail("two-child node count", two.nodes.size(), n - 1);
    if (two.depth > 60) return fail("two-child depth", two.depth);
    const Bvh4Result four = collapse_bvh4(two.nodes.data(), two.nodes.size(), n_prims);
    if (n == 0) {
        if (!four.nodes.empty()) return fail("empty scene has nodes");
        std::printf("ok empty\n");
        return 0;
    }
    if (four.leaf_boxes.size() != 8 * static_cast<size_t>(n_prims)) return fail("leaf box table size");
    if (four.nodes.size() > two.nodes.size() + 1) return fail("more four-child than two-child nodes", four.nodes.size(), two.nodes.size());
    Walk w{four, std::vector<int>(n_prims, 0), std::vector<char>(four.nodes.size(), 0)};
    w.go(0, 0, nullptr, nullptr);
    if (w.bad) return fail("structure", w.bad);
    for (size_t i = 0; i < four.nodes.size(); ++i)
        if (w.node_seen[i] != 1) return fail("node not reached exactly once", static_cast<long>(i), w.node_seen[i]);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t prim = kept[i].prim;
        if (w.seen[prim] != 1) return fail("primitive not a leaf exactly once", prim, w.seen[prim]);
        const float* lb = four.leaf_boxes.data() + 8 * static_cast<size_t>(prim);
        for (int a = 0; a < 3; ++a)
            if (lb[a] != kept[i].lo[a] || lb[4 + a] != kept[i].hi[a]) return fail("leaf box differs from the primitive's box", prim, a);
    }
    for (uint32_t p = 0; p < n_prims; p += 2)
        if (w.seen[p]) return fail("a primitive that was never given is a leaf", p);
    if (w.max_stack > four.stack_need) return fail("walk needs more stack than reported", w.max_stack, four.stack_need);
    if (four.stack_need > 3 * four.depth) return fail("stack need beyond three per level", four.stack_need, four.depth);
    // the 64-byte form the kernels read: same links, every decoded box contains the child's and stays within two steps of it
    const std::vector<Bvh4QNode> q = quantize_bvh4(four.nodes);
    if (q.size() != four.nodes.size()) return fail("quantised node count");
    for (size_t i = 0; i < q.size(); ++i) {
        const Bvh4Node& nd = four.nodes[i];
        const float* lo[3] = {nd.lox, nd.loy, nd.loz};
        const float* hi[3] = {nd.hix, nd.hiy, nd.hiz};
        for (int k = 0; k < 4; ++k) {
            if (q[i].child[k] != nd.child[k]) return fail("quantised link differs", static_cast<long>(i), k);
            for (int a = 0; a < 3; ++a) {
                const uint32_t ql = (q[i].qlo[a] >> (8 * k)) & 0xffu, qh = (q[i].qhi[a] >> (8 * k)) & 0xffu;
                if (nd.child[k] == kBvh4Absent) {
                    if (ql != 255u || qh != 0u) return fail("absent child's offsets", static_cast<long>(i), k);
                    continue;
                }
                if (!(q[i].scale[a] > 0.0f)) return fail("step not positive", static_cast<long>(i), a);
                const float dl = bvh4q_plane(q[i], a, ql), dh = bvh4q_plane(q[i], a, qh);
                if (!(dl <= lo[a][k] && dh >= hi[a][k])) return fail("decoded box does not contain the child's", static_cast<long>(i), k);
                if (lo[a][k] - dl > 2.001f * q[i].scale[a] || dh - hi[a][k] > 2.001f * q[i].scale[a]) return fail("decoded box looser than two steps", static_cast<long>(i), k);
            }
        }
    }
    size_t kids = 0;
    for (const Bvh4Node& nd : four.nodes)
        for (int k = 0; k < 4; ++k) kids += nd.child[k] != kBvh4Absent;
    std::printf("ok n %zu nodes2 %zu nodes4 %zu depth2 %d depth4 %d stack_need %d walk_stack %d fanout %.2f\n", n, two.nodes.size(), four.nodes.size(),
                two.depth, four.depth, four.stack_need, w.max_stack, double(kids) / double(four.nodes.size()));
    return 0;
}
