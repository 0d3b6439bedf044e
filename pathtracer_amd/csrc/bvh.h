// Host BVH builder: binned surface-area heuristic over primitive world boxes, emitted as a linear
// two-child node array, and the collapse of such an array (this builder's or the device builder's)
// into the four-child nodes the closest-hit kernels walk.  Together they replace
// ObjectTree::Tree::makeTree (src/ObjectTree.cpp:90-184), whose cost grows like N^2.6; because the
// reference's tree never culls, every topology returns the same closest hit.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "device_types.h"

namespace mipt {

struct BuildBox {
    float lo[3], hi[3];
    uint32_t prim;
};

struct BvhResult {
    std::vector<BvhNode> nodes;
    int depth = 0;   // number of node levels: bounds the traversal stack
};

// max_depth bounds the tree height (and with it the traversal stack); deeper subtrees fall back to
// median splits, which always fit.
BvhResult build_bvh(std::vector<BuildBox> boxes, int max_depth = 60);

// The four-child tree the kernels walk, made from a two-child tree by pulling grandchildren up: a node's child
// list starts as the two children and, while it has fewer than four entries, the inner entry with the largest
// surface area is replaced by its own two children.
struct Bvh4Result {
    std::vector<Bvh4Node> nodes;
    std::vector<float> leaf_boxes;   // 8 floats per primitive index below n_prims (zeros for primitives not in the tree)
    int depth = 0;                   // node levels
    int stack_need = 0;              // entries a depth-first walk that defers up to three children per node can hold at once
};
Bvh4Result collapse_bvh4(const BvhNode* nodes2, size_t n_nodes2, uint32_t n_prims);

// The 64-byte form of the four-child nodes (same indices, same links): every decoded child box contains the original.
std::vector<Bvh4QNode> quantize_bvh4(const std::vector<Bvh4Node>& nodes);
void quantize_bvh4_node(const Bvh4Node& node, Bvh4QNode& out);   // one node (nodes are independent: callers may spread them over threads)
// plane = org + scale * q as the quantiser guarantees it (one fused multiply-add)
float bvh4q_plane(const Bvh4QNode& n, int axis, uint32_t q);

}   // namespace mipt
