// Records as they lie in HBM, shared by the host packer and the kernels.
#pragma once
#include <stdint.h>

namespace mipt {

// One primitive = one 128-byte line in two halves.  The first 64 bytes are all that Object::hitRay's test of a sphere or
// a box reads (four 16-byte pieces per lane: the traversal is bound by the pieces it fetches); the second half is
// read when the test hits (the way back to world space) and by the stages after it.  Affines are Linear row-major
// (m[3*r+c]) followed by the translation.
struct alignas(128) PrimRec {
    float inv[12];      // world -> local  (Object::m_inverseTransformation)
    float p[3];         // parameters 0-2: sphere {r}; box {hx,hy,hz}; crater ball {offset}; rough box {halfSize}; fractal {world box centre}; triangle {a}
    uint32_t kind;
    float fwd[12];      // local -> world  (Object::m_transformation)
    float q[3];         // parameters 3-5: crater ball {first crater, craters}; fractal {world box half size}; triangle {b} (c in tri_c)
    uint32_t material;
#if defined(__HIPCC__)
    __host__ __device__
#endif
    float& param(int i) { return i < 3 ? p[i] : q[i - 3]; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    const float& param(int i) const { return i < 3 ? p[i] : q[i - 3]; }
};
static_assert(sizeof(PrimRec) == 128, "PrimRec must be one 128-byte line");

// BasicMaterial padded to 64 bytes.
struct alignas(64) MaterialRec {
    float diffuse, specular, sharpness;
    float albedo[3];
    float emitted[3];
    float transmittance, ior;
    float internal[3];
    float pad[2];
};
static_assert(sizeof(MaterialRec) == 64, "MaterialRec must be 64 bytes");

// BVH2 node carrying both children's boxes (64 bytes = 4 x dwordx4).  child >= 0: node index;
// child < 0: leaf holding primitive ~child.  An absent child has an inverted (never hit) box.
struct alignas(64) BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// Four-child node, the form the traversal kernels walk (128 bytes, one cache line): one coordinate of all four
// children's boxes per dwordx4, then the links.  child >= 0: node index; kBvh4Absent: no child (its box lies at
// +infinity on every axis, which no ray reaches); any other negative value: leaf holding primitive ~child.
struct alignas(128) Bvh4Node {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    int32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");
constexpr int32_t kBvh4Absent = INT32_MIN;

// The same node in 64 bytes, the form the kernels read: the children's boxes as 8-bit offsets from the node's own low
// corner, in steps of `scale` per axis -- plane = org + scale * q, the low planes rounded down and the high planes up
// (one step to spare each), so a decoded box always contains the child's.  Byte c of a word belongs to child c; an
// absent child has low offset 255 and high offset 0.  A visit reads four 16-byte pieces instead of seven: the
// traversal is bound by the vector-memory pipeline (address and tag processing per load instruction and lane), not by
// arithmetic, and the looser boxes cost a few per cent more visits.
struct alignas(64) Bvh4QNode {
    float org[3];
    float scale[3];
    uint32_t qlo[3];   // x, y, z
    uint32_t qhi[3];
    int32_t child[4];
};
static_assert(sizeof(Bvh4QNode) == 64, "Bvh4QNode must be 64 bytes");

// World box of one SDF primitive (tested by every ray after the analytic traversal).
struct SdfBox {
    float lo[3];
    uint32_t prim;
    float hi[3];
    uint32_t pad;
};

struct CameraRec {
    float lin[9];
    float tr[3];
    float aspect, fov_scale /* tan(fov * pi/180), hoisted */, inv_focal /* 1/focalDistance */, blur_rad /* blur * max(aspect, 1/aspect) */;
};

struct SceneDev {
    const PrimRec* prims;
    const float* tri_c;            // 3 floats per primitive, only read for triangles (may be null)
    const MaterialRec* materials;
    const Bvh4QNode* nodes;        // null when the scene has no analytic primitive
    // world box of every analytic primitive as its BVH leaf holds it, 8 floats {lo.xyz, -, hi.xyz, -} at 8 * prim: lets
    // a leaf that waited on a traversal stack be tested against the limit the ray has reached since
    const float* leaf_boxes;
    const SdfBox* sdf_boxes;
    uint32_t n_sdf;
    uint32_t n_prims;
    uint32_t n_nodes;              // four-child nodes of the tree
    // noise() (PathTracerDemo.cpp:56-70) unrolled: octave i samples A_i p + b_i with weight k_i; 13 floats per octave (A row-major,
    // b + 0.25, k), 8 octaves (null when the scene has no crater ball)
    const float* noise_oct;
    // big craters of every crater ball: 12 floats per crater {cell.xyz, size, on_sphere.xyz, t, age bits, 1/(5 size), -, -};
    // a ball's records start at record PrimRec::q[0] and number PrimRec::q[1] (both uint32 bit patterns)
    const float* craters;
    // per crater ball, in the order of their tables: for each of the 9^3 lattice cells about floor(offset) - 4 that a
    // query can fall into, the mask of table entries within the 4^3 cells the reference visits from there; the ball's
    // ordinal sits above the low 8 bits of PrimRec::q[1]
    const unsigned long long* crater_masks;
    // per crater ball, in the same order: the small and tiny craters' lattice cells about the ball (craterball.h:
    // k_minor_cells), 16 bytes per cell; null: the evaluators compute the cells as they go
    const void* minor_cells;
    // the raw noise transform (Linear row-major, then the translation), read by the SDF kinds that run noise() octave by octave
    const float* noise_raw;
    uint32_t plain_sdf;            // the scene holds SDF kinds evaluated one point per lane (RoughBoxObject)
    // Objects that only the host can evaluate (kind 8: a user's Object subclass behind callbacks): their padded world
    // boxes.  Rays that reach one are handed to the host between the device stages.
    const SdfBox* host_boxes;
    uint32_t n_host;
};

#ifndef MIPT_MARCH_CURSORS
#define MIPT_MARCH_CURSORS 16
#endif
constexpr uint32_t kMarchCursors = MIPT_MARCH_CURSORS;

constexpr uint32_t kSortKeys = 1024;   // sort keys of the SDF queue: the 9^3 query cells of a crater ball and a few for everything else

// Per-bounce queue counters of one sample batch, zeroed once per batch.
struct BounceCounters {
    uint32_t n_live;        // entries in the path queue feeding this bounce (bounce > 0)
    uint32_t n_sdf;         // rays queued for the SDF march
    uint32_t sdf_head;      // (unused)
    uint32_t n_sdf_shade;   // hits queued for SDF shading
    uint32_t unused_heads[8];
    uint32_t n_host_rays;   // rays gathered for one host object (reset per object)
    uint32_t n_host_shade;  // hits on host objects queued for shading on the host
    uint32_t n_carry;       // SDF marches this round's launch left unfinished for the next round's (may exceed the buffer: readers clamp)
    uint32_t n_late;        // rays of earlier rounds whose handed-on march this round's launch finished (they joined n_live)
    uint32_t pad2[12];
    // work-fetch cursors of the SDF march, each on a cache line of its own: atomics on one line are served one at a time
    uint32_t sdf_heads[kMarchCursors][32];
    uint32_t extend_heads[8][32];   // work-fetch cursors of extend, one per XCD, likewise
    // the SDF queue's rays sorted by the lattice cell of their first query (k_sort_*): rays per key, and where each
    // key's run begins / has been filled to in the sorted queue
    uint32_t sort_hist[kSortKeys];
    uint32_t sort_cursor[kSortKeys];
};

// An SDF march that its launch handed on to the next round's launch instead of finishing it (k_sdf_march): everything
// of the lane's state that cannot be read back from the path's slot.  64 bytes.
struct alignas(16) CarryRec {
    uint32_t slot;
    uint32_t k;                 // index into SceneDev::sdf_boxes of the primitive being marched
    uint32_t best_lo, best_hi;  // the ray's best hit key so far (the slot's own key holds kCarriedKey meanwhile)
    float p[3];                 // March::p
    float d;                    // March::d
    float seen_p[3];            // March::seen_p
    float seen_d;
    uint32_t iter_flags;        // March::iter | first << 16 | sign << 17
    uint32_t since, span;
    uint32_t pad;
};
static_assert(sizeof(CarryRec) == 64, "CarryRec must be 64 bytes");

struct Totals {
    unsigned long long rays, sdf_rays, sdf_shades;
    unsigned long long sdf_far_evals, sdf_near_evals, sdf_noise_calls;   // sdf_march: SDF evaluations by kind, noise() calls
    unsigned long long shade_near_evals, shade_noise_calls;              // shade_sdf: the same for the normals
    unsigned long long primary_traced;                                   // primary rays that were traced rather than culled on screen
    unsigned long long node_visits, prim_tests;                          // counting build of the traversal only
    unsigned long long host_rays, host_shades;                           // work handed to the host for objects without a device form
    unsigned long long trav_rounds, trav_leaf_steps, trav_restarts;     // counting build of k_extend_pair: wave rounds, leaf steps, second rays started over
    unsigned long long samples_done;                                     // samples per pixel that entered the frame (batches accumulated before a cancel)
    unsigned long long marches_carried;                                  // SDF marches handed from one round's launch to the next
};

constexpr int kMaxBounces = 64;
// A path whose march was handed on falls one round behind the others; it may do so this many times, and a batch
// runs that many rounds after the last bounce for the paths that did.
constexpr int kMaxCarryLag = 8;
constexpr int kCounterRows = kMaxBounces + kMaxCarryLag + 1;

}   // namespace mipt
