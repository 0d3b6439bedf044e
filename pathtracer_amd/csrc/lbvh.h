// Device BVH builder for scenes that change every frame: Morton order + radix tree (Karras 2012) + bottom-up
// box fitting, emitted as the same two-child node array the host builder (bvh.h) produces.  The reference has
// one builder, ObjectTree::Tree::makeTree (src/ObjectTree.cpp:90-184); because its tree never culls, every
// topology returns the same closest hit, so either builder serves the same frames.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh.h"

namespace mipt {

// Builds over `n` >= 2 boxes (host memory; every box inside [lo, hi]).  On success *d_nodes holds n - 1 nodes in
// memory from hipMalloc (node 0 is the root) and *depth the number of node levels.  Work is enqueued on `stream`;
// the call returns after the stream has drained.  Temporary device memory is released before returning.
hipError_t build_lbvh_device(hipStream_t stream, const BuildBox* boxes, uint32_t n, const float lo[3], const float hi[3], void** d_nodes, int* depth);

}   // namespace mipt
