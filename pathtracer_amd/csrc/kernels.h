// Launchers of the wavefront stages (defined in kernels.hip).  Every launcher only enqueues work
// on `stream`; none of them synchronises or allocates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

namespace mipt {

// Per-batch working set, all in HBM, indexed by path slot unless noted.
struct PathBuffers {
    float4* ray_o;        // xyz origin
    float4* ray_d;        // xyz direction
    float4* att;          // xyz attenuation (throughput)
    float4* rad;          // xyz radiance gathered so far
    unsigned long long* hit_key;   // (float bits of t) << 32 | primitive, ~0 = miss
    float4* hit_pos;      // xyz world hit position as Object::hitRay returns it
    uint32_t* queue[2];   // live path slots, ping-pong per bounce
    uint32_t* sdf_queue;  // slots whose ray entered an SDF primitive's box
    // per slot that is in sdf_queue, written by the kernel that put it there: where the ray's first march starts
    // {entry point in the object's frame, index of the SDF box | 1 << 31 if a march starts there}
    float4* march_start;
    uint32_t* sdf_shade_queue;   // slots whose closest hit is an SDF primitive
    uint32_t* host_queue;        // slots handed to the host (rays that reach a host object's box; hits on host objects); null without host objects
    BounceCounters* counters;    // kCounterRows entries, one per round of the batch
    // SDF marches handed from one round's march launch to the next one's, ping-pong by round (null: never hand on)
    CarryRec* carry[2];
    Totals* totals;
    // one bit per slot, set for the paths whose primary ray hit something or reached an SDF box (only
    // those ever carry radiance); sample s of pixel p is bit p & 63 of word s * ceil(n_pix / 64) + p / 64
    unsigned long long* live_bits;
    // cancellation: `cancel` points at a word in pinned host memory that mipt_cancel sets.  A one-thread
    // kernel copies it into `cancel_latch` (device memory) before every bounce and before the accumulation;
    // every kernel returns at once when the latch is set, so a cancelled frame drains within one bounce, and
    // all threads of the accumulation agree on whether the batch enters the frame.
    const volatile uint32_t* cancel;
    uint32_t* cancel_latch;
};

struct BatchParams {
    uint32_t n_slots;       // path slots in this batch = n_pix * samples_in_batch
    uint32_t n_pix;         // pixel ids per sample: width * height, or whole 8 x 8 tiles (see tiles_x)
    uint32_t width, height;
    // 0: pixel id = y * width + x.  Otherwise the image is cut into 8 x 8 tiles, tiles_x per row, and
    // pixel id = tile * 64 + (y & 7) * 8 + (x & 7): a wave's 64 paths are a square patch of the image (their
    // rays meet the same few objects) and still occupy 8 full cache lines of every per-slot array.
    uint32_t tiles_x;
    // With tiles: which image tile each group of 64 pixel ids stands for (tile index ty * tiles_x + tx), or null for
    // all tiles in order.  A frame whose objects cover a small part of the screen enumerates only the tiles that
    // touch their screen bounds, so path slots (and the memory behind them) exist only for pixels that can see
    // anything; n_pix is then 64 times the number of listed tiles.
    const uint32_t* vis_tiles;
    uint32_t first_sample;  // global sample index of sample 0 of this batch
    uint32_t flags;
    uint32_t lane_threshold;   // SDF near field: waiting lanes from which a wave evaluates one point per lane
    uint32_t trav_quorum;      // BVH traversal: lanes with a pending leaf from which a wave runs a leaf step (20)
    uint32_t trav_quorum_primary;   // the same for camera rays, whose waves hold neighbouring pixels and reach their leaves together (1)
    uint32_t trav_keep;        // BVH traversal: a wave refills its idle lanes when fewer than trav_keep / 64 of its busy lanes are unfinished (40)
    uint32_t trav_quorum_pair;   // two rays per lane (0: one ray per lane): lanes with a pending leaf from which a wave runs a leaf step
    uint32_t trav_keep_pair;     // ... and a wave turns its finished rays over when fewer than trav_keep_pair / 128 of its rays are unfinished
    uint32_t pair_plain_entries; // test aid: stack entries without entry distances, as trees beyond 2^20 nodes or primitives get them
    uint32_t pair_rows;          // test aid: rows of the shared stack column in use (0: all; at least the tree's need + 3)
    // SDF marches still running when a wave has no more rays to fetch are handed to the next round's launch once the
    // wave has gone carry_after more rounds, if their path has fallen behind fewer than carry_max_lag times (0: every
    // launch finishes its marches) and the buffer of carry_cap records has room.
    uint32_t carry_max_lag, carry_after, carry_cap;
    uint32_t sort_marches;       // rounds 1 .. sort_marches sort their SDF queue by the rays' first query cell before the march (0: never)
    uint32_t no_few;             // test aid: scenes of a handful of primitives walk their tree like any other (1) instead of testing every primitive per ray
    uint64_t seed;
};

// Pixel rectangles (inclusive) outside of which no primary ray can reach anything: the screen bounds of
// the scene's boxes.  Pixels outside all of them are not traced at all.
constexpr uint32_t kFlagFullMarch = 0x80000000u;   // BatchParams::flags: SDF marches run out their 256 iterations (no shortcut for marches that cannot end)

struct CullRects {
    int n;
    int r[8][4];   // x0, y0, x1, y1
};

// generate + first extend fused; needs counters zeroed; fills queue[0] and the SDF queue of bounce 0
// `count`: the counting build of the traversal (same steps, same order; adds node visits and primitive tests to Totals)
// stack_need: the traversal stack entries the scene's tree can need (at most kMaxTraversalStack)
constexpr int kMaxTraversalStack = 60;
void launch_generate_extend(hipStream_t, const CameraRec&, const SceneDev&, const BatchParams&, const PathBuffers&, const CullRects&, int stack_need,
                            bool count = false);
// generate alone: every visible path's camera ray into the arena, queue[0] and the bitmap (n_pix must be a multiple of 64); needs counters zeroed
void launch_generate(hipStream_t, const CameraRec&, const BatchParams&, const PathBuffers&, const CullRects&);
void launch_extend(hipStream_t, const SceneDev&, const BatchParams&, const PathBuffers&, uint32_t bounce, int stack_need, bool count = false);
// big-crater candidates of a crater ball: 12 floats per cell of the n^3 lattice cube at `base` (x-major)
void launch_crater_cells(hipStream_t, const float off[3], const int base[3], int n, float* out);
// The small and tiny craters' cell tables of one crater ball (craterball.h: k_minor_cells) into `out`.
void launch_minor_cells(hipStream_t, const float off[3], void* out);
size_t minor_cells_bytes_per_ball();
void launch_sdf_march(hipStream_t, const SceneDev&, const BatchParams&, const PathBuffers&, uint32_t round);
// signed distance of SDF primitive `prim` at n points of its local frame (device arrays)
void launch_sdf_eval(hipStream_t, const SceneDev&, uint32_t prim, uint32_t lane_threshold, const float* pos3, uint32_t n, float* out);
// the material a hit at each of n local points of SDF primitive `prim` is shaded with (9 floats per point; device arrays)
void launch_sdf_material(hipStream_t, const SceneDev&, uint32_t prim, const float* pos3, uint32_t n, float* out9);
void launch_shade(hipStream_t, const SceneDev&, const BatchParams&, const PathBuffers&, uint32_t bounce, uint32_t max_bounces);
void launch_shade_sdf(hipStream_t, const SceneDev&, const BatchParams&, const PathBuffers&, uint32_t bounce, uint32_t max_bounces);
// ---- objects evaluated on the host (kind 8) ---------------------------------------------------------------
// live rays of `bounce` that reach host_boxes[which] before their best hit -> host_queue, counters[bounce].n_host_rays
void launch_host_gather(hipStream_t, const SceneDev&, const PathBuffers&, uint32_t bounce, uint32_t which);
// origin and direction (6 floats per entry) of the first n slots of host_queue
void launch_host_export_rays(hipStream_t, const PathBuffers&, uint32_t n, float* out6);
// the host's hits: entry i (slot of host_queue[idx[i]]) offers key[i] / pos[i]; the smaller key wins (two launches: keys, then positions)
void launch_host_merge(hipStream_t, const PathBuffers&, uint32_t n, const uint32_t* idx, const unsigned long long* key, const float* pos3);
// hit position, incoming direction (6 floats) and primitive of the first n slots of host_queue (hits queued by shade)
void launch_host_export_hits(hipStream_t, const PathBuffers&, uint32_t n, float* out6, uint32_t* out_prim);
// applies the host's bounces (9 floats per entry: emitted, attenuation, world direction) to the first n slots of host_queue
void launch_host_apply(hipStream_t, const PathBuffers&, uint32_t n, const float* bounce9, uint32_t bounce, uint32_t max_bounces);
void launch_finish_batch(hipStream_t, const PathBuffers&, uint32_t rounds, uint32_t n_slots, uint32_t carry_cap);
// frame[pix] += sum over the batch's samples, in sample order (the `acc +=` of src/Renderer.cpp:97-99)
void launch_latch_cancel(hipStream_t, const PathBuffers&);
void launch_accumulate(hipStream_t, const BatchParams&, const PathBuffers&, float* frame_sum);
void launch_scale(hipStream_t, const float* in, float k, size_t n, float* out);
void launch_add(hipStream_t, float* dst, const float* src, size_t n);   // dst[i] += src[i]
void launch_pack_rgba8(hipStream_t, const float* in_rgb, size_t n_pix, uint32_t* out_rgba);
void launch_load_rays(hipStream_t, const float* pos3, const float* dir3, uint32_t n, const PathBuffers&);
void launch_export_rad(hipStream_t, const PathBuffers&, uint32_t n, float* out3);
void launch_tonemap_filmic(hipStream_t, const float* in, float* out, size_t n_floats);
// Reinhard: scratch holds 3 floats per block of the reduction plus 4 result floats
void launch_tonemap_reinhard(hipStream_t, const float* in, float* out, size_t width, size_t height, float* scratch, size_t scratch_floats);
size_t reinhard_scratch_floats();
// Address, on the CURRENT device, of a file-scope device array of reinhard_scratch_floats() floats (every GPU
// holds its own copy at its own address): the Reinhard scratch of the entry point that allocates nothing.
hipError_t reinhard_static_scratch(float** out);

}   // namespace mipt
