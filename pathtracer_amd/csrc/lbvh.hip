// Device BVH builder (see lbvh.h).  Five small launches over n boxes:
//   codes     30-bit Morton code of each box centre within the scene box
//   sort      rocPRIM radix sort of (code, box index) pairs; stable, so equal codes keep their order
//   topology  one thread per internal node of the radix tree (n - 1 of them): range, split, child links, parents
//   fit       one thread per leaf climbs towards the root; the second thread to reach a node joins the two child
//             boxes and goes on, the first one stops (a per-node arrival counter decides)
//   depth     one thread per leaf counts the nodes above it; maximum over all leaves
// All of it is bandwidth-trivial next to a frame (100,000 boxes: 2.8 MB in, 6.4 MB of nodes out).
#include "lbvh.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "lbvh_core.h"

namespace mipt {
namespace {

constexpr uint32_t kNoParent = 0xffffffffu;
constexpr int kMaxClimb = 128;   // no radix tree over 62-bit keys is deeper than 62; the cap only ends a climb over corrupt links

__global__ void k_lbvh_codes(const BuildBox* __restrict__ boxes, uint32_t n, float lox, float loy, float loz, float sx, float sy, float sz,
                             uint32_t* __restrict__ codes, uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BuildBox b = boxes[i];
    const float cx = 0.5f * (b.lo[0] + b.hi[0]), cy = 0.5f * (b.lo[1] + b.hi[1]), cz = 0.5f * (b.lo[2] + b.hi[2]);
    codes[i] = lbvh_morton30((cx - lox) * sx, (cy - loy) * sy, (cz - loz) * sz);
    order[i] = i;
}

// parents are stored as (node << 1) | side
__global__ void k_lbvh_topology(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ order, const BuildBox* __restrict__ boxes, uint32_t n,
                                BvhNode* __restrict__ nodes, uint32_t* __restrict__ parent_node, uint32_t* __restrict__ parent_leaf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1u) return;
    const LbvhRange r = lbvh_node_range(codes, n, i);
    if (i == 0u) parent_node[0] = kNoParent;
    // a range outside [0, n) cannot come out of lbvh_node_range; if it ever did, leave the node unlinked rather than write out of bounds
    if (r.first < 0 || r.last >= static_cast<int64_t>(n) || r.split < r.first || r.split >= r.last) return;
    int32_t link[2];
    for (int side = 0; side < 2; ++side) {
        const int64_t at = r.split + side;
        const bool leaf = side == 0 ? at == r.first : at == r.last;
        if (leaf) {
            link[side] = ~static_cast<int32_t>(boxes[order[at]].prim);
            parent_leaf[at] = (i << 1) | static_cast<uint32_t>(side);
        } else {
            link[side] = static_cast<int32_t>(at);
            parent_node[at] = (i << 1) | static_cast<uint32_t>(side);
        }
    }
    nodes[i].child0 = link[0];
    nodes[i].child1 = link[1];
    nodes[i].pad[0] = nodes[i].pad[1] = 0u;
}

__global__ void k_lbvh_fit(const uint32_t* __restrict__ order, const BuildBox* __restrict__ boxes, uint32_t n, const uint32_t* __restrict__ parent_node,
                           const uint32_t* __restrict__ parent_leaf, BvhNode* nodes, uint32_t* arrivals) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const BuildBox b = boxes[order[j]];
    float lo[3] = {b.lo[0], b.lo[1], b.lo[2]}, hi[3] = {b.hi[0], b.hi[1], b.hi[2]};
    uint32_t up = parent_leaf[j];
    for (int climb = 0; climb < kMaxClimb && up != kNoParent; ++climb) {
        const uint32_t node = up >> 1, side = up & 1u;
        if (node >= n - 1u) return;
        BvhNode* nd = nodes + node;
        float* wlo = side == 0u ? nd->lo0 : nd->lo1;
        float* whi = side == 0u ? nd->hi0 : nd->hi1;
        for (int a = 0; a < 3; ++a) {
            wlo[a] = lo[a];
            whi[a] = hi[a];
        }
        // release: this side's box is visible to whoever arrives second; acquire: the second arrival sees the other side's box
        const uint32_t before = __hip_atomic_fetch_add(arrivals + node, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == 0u) return;
        const float* olo = side == 0u ? nd->lo1 : nd->lo0;
        const float* ohi = side == 0u ? nd->hi1 : nd->hi0;
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], olo[a]);
            hi[a] = fmaxf(hi[a], ohi[a]);
        }
        up = parent_node[node];
    }
}

__global__ void k_lbvh_depth(uint32_t n, const uint32_t* __restrict__ parent_node, const uint32_t* __restrict__ parent_leaf, uint32_t* depth) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t levels = 0;
    if (j < n) {
        uint32_t up = parent_leaf[j];
        while (up != kNoParent && levels < static_cast<uint32_t>(kMaxClimb)) {
            ++levels;
            const uint32_t node = up >> 1;
            if (node >= n - 1u) break;
            up = parent_node[node];
        }
    }
    for (int off = 32; off > 0; off >>= 1) levels = max(levels, static_cast<uint32_t>(__shfl_xor(static_cast<int>(levels), off)));
    if ((threadIdx.x & 63u) == 0u) atomicMax(depth, levels);
}

}   // namespace

hipError_t build_lbvh_device(hipStream_t st, const BuildBox* boxes, uint32_t n, const float lo[3], const float hi[3], void** d_nodes_out, int* depth_out) {
    *d_nodes_out = nullptr;
    *depth_out = 0;
    if (n < 2u) return hipErrorInvalidValue;
    const size_t n_nodes = n - 1u;
    // one allocation for everything temporary: boxes | codes x2 | order x2 | parents | arrivals + depth | sort storage
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                             static_cast<uint32_t*>(nullptr), static_cast<size_t>(n), 0u, 30u, st);
    if (e != hipSuccess) return e;
    auto round_up = [](size_t v) { return (v + 255u) & ~static_cast<size_t>(255u); };
    const size_t box_bytes = round_up(static_cast<size_t>(n) * sizeof(BuildBox)), word_bytes = round_up(static_cast<size_t>(n) * sizeof(uint32_t));
    const size_t zero_bytes = round_up((n_nodes + 1u) * sizeof(uint32_t));   // arrival counters, then the depth word
    const size_t total = box_bytes + 6u * word_bytes + zero_bytes + round_up(sort_bytes);
    char* tmp = nullptr;
    void* nodes = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&tmp), total);
    if (e != hipSuccess) return e;
    e = hipMalloc(&nodes, n_nodes * sizeof(BvhNode));
    if (e != hipSuccess) {
        (void)hipFree(tmp);
        return e;
    }
    char* at = tmp;
    auto take = [&](size_t bytes) {
        char* p = at;
        at += bytes;
        return p;
    };
    BuildBox* d_boxes = reinterpret_cast<BuildBox*>(take(box_bytes));
    uint32_t* codes_in = reinterpret_cast<uint32_t*>(take(word_bytes));
    uint32_t* codes = reinterpret_cast<uint32_t*>(take(word_bytes));
    uint32_t* order_in = reinterpret_cast<uint32_t*>(take(word_bytes));
    uint32_t* order = reinterpret_cast<uint32_t*>(take(word_bytes));
    uint32_t* parent_node = reinterpret_cast<uint32_t*>(take(word_bytes));
    uint32_t* parent_leaf = reinterpret_cast<uint32_t*>(take(word_bytes));
    uint32_t* arrivals = reinterpret_cast<uint32_t*>(take(zero_bytes));
    uint32_t* d_depth = arrivals + n_nodes;
    void* sort_tmp = take(round_up(sort_bytes));

    float scale[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = hi[a] - lo[a];
        scale[a] = ext > 0.0f ? 1.0f / ext : 0.0f;
    }
    const dim3 block(256), grid_leaf((n + 255u) / 256u), grid_node((n - 1u + 255u) / 256u);
    uint32_t depth = 0;
    e = hipMemcpyAsync(d_boxes, boxes, static_cast<size_t>(n) * sizeof(BuildBox), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(arrivals, 0, zero_bytes, st);
    if (e == hipSuccess) e = hipMemsetAsync(parent_node, 0xff, 2u * word_bytes, st);   // parent_node and parent_leaf: kNoParent until linked
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_lbvh_codes, grid_leaf, block, 0, st, d_boxes, n, lo[0], lo[1], lo[2], scale[0], scale[1], scale[2], codes_in, order_in);
        e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, codes_in, codes, order_in, order, static_cast<size_t>(n), 0u, 30u, st);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_lbvh_topology, grid_node, block, 0, st, codes, order, d_boxes, n, static_cast<BvhNode*>(nodes), parent_node, parent_leaf);
        hipLaunchKernelGGL(k_lbvh_fit, grid_leaf, block, 0, st, order, d_boxes, n, parent_node, parent_leaf, static_cast<BvhNode*>(nodes), arrivals);
        hipLaunchKernelGGL(k_lbvh_depth, grid_leaf, block, 0, st, n, parent_node, parent_leaf, d_depth);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&depth, d_depth, sizeof(depth), hipMemcpyDeviceToHost, st);
    const hipError_t se = hipStreamSynchronize(st);   // also before freeing what the kernels use
    if (e == hipSuccess) e = se;
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        (void)hipFree(nodes);
        return e;
    }
    *d_nodes_out = nodes;
    *depth_out = static_cast<int>(depth);
    return hipSuccess;
}

}   // namespace mipt
