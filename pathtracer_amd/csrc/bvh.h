// Host BVH builder: binned surface-area heuristic over primitive world boxes, emitted as the
// linear two-child node array the extend kernel walks.  It replaces ObjectTree::Tree::makeTree
// (src/ObjectTree.cpp:90-184), whose cost grows like N^2.6; because the reference's tree never
// culls, every topology returns the same closest hit.
#pragma once
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

}   // namespace mipt
