// The demo's signed-distance objects that are evaluated one point per lane, in the reference's own
// order of operations (PathTracerDemo.cpp:26-349).  Included by kernels.hip after craterball.h (whose
// hashed lattice cell it shares).  RoughBoxObject is +, -, *, sqrt, floor, min and max with contraction
// off, so its values are the reference's bit for bit; RoughSphereObject and WeirdWallObject also call
// exp2, log2, exp, hypot and atan2, whose device versions differ from the host library's in the last
// bits, so they agree with the reference to about 1e-6.
#pragma once

#pragma clang fp contract(off)

__device__ __forceinline__ float tri_wave(float x) {   // triangle() :26-28
    return -1.0f + 4.0f * fabsf(x - 0.25f - floorf(x + 0.25f));
}

// noise() :56-70, octave after octave through the raw transform T (Linear row-major, then the translation).
__device__ __forceinline__ float noise_seq(const float* T, f3 p) {
    float v = 0.0f, k = 0.02f;
    for (int i = 0; i < 8; ++i) {
        p = aff_pos(T, p);
        v += k * (tri_wave(p.x) + tri_wave(p.y) + tri_wave(p.z));
        p = mk3(p.x * 1.1f, p.y * 1.1f, p.z * 1.1f);
        k *= 0.9f;
    }
    return v;
}

// voronoi() :118-154 with its default arguments (scatter 1, identity transform): the two smallest squared
// distances to the jittered points of the 4 x 4 x 4 lattice cells about the location, and the hash of the
// nearest.  Lattice coordinates must fit 24 signed bits (the upload refuses boxes that large).
struct VoronoiHit {
    float sqr, next_sqr;
    uint32_t hash;
};
// SHEAR: the brick pattern's metric, Linear(1, 0.5, 0; 0, 0.5, 0; 0, 0, 1) (PathTracerDemo.cpp:309-313).  Its
// inverse by Linear::inverse (src/LinearAlgebra.cpp:207-233) is (1, -1, 0; 0, 2, 0; 0, 0, 1), exactly.
// Of the 4 x 4 x 4 cells the reference visits, 20 can never hold one of the two nearest points of the brick pattern
// (scatter 0.3 under the sheared metric): with the location anywhere in its cell and every point anywhere in its jitter
// range, the second-nearest point is never farther than 1.006 (squared, sheared; maximum over a 160^3 grid of the second
// smallest worst case among the 8 cells about the location, plus the grid's Lipschitz allowance), and the nearest
// that a point of these 20 cells can come is 1.0225 or more (minimum of the convex distance over the cell's box, in
// closed form).  Skipping them leaves `sqr`, `next_sqr` and `hash` exactly what the full loop returns.
//   kept for i = -1: j in {0, 1, 2}, k in {0, 1};   for i = 2: j in {-1, 0, 1}, k in {0, 1};   all 16 for i in {0, 1}
__device__ __forceinline__ bool sheared_cell_matters(int i, int j, int k) {
    if (i == 0 || i == 1) return true;
    if (k < 0 || k > 1) return false;
    return i == -1 ? j >= 0 : j <= 1;
}

template <bool SHEAR>
__device__ __forceinline__ VoronoiHit voronoi_general(f3 location, float half_scatter) {
    if (SHEAR) location = mk3((1.0f * location.x + -1.0f * location.y) + 0.0f * location.z, (0.0f * location.x + 2.0f * location.y) + 0.0f * location.z,
                              (0.0f * location.x + 0.0f * location.y) + 1.0f * location.z);
    VoronoiHit v{3.402823466e38f, 3.402823466e38f, 0u};
    const f3 origin = mk3(floorf(location.x), floorf(location.y), floorf(location.z));
    // the hash of a cell starts from a sum that is linear in its lattice coordinates (mod 2^32): one seed for the
    // origin's cell, constant steps to its neighbours
    const uint32_t seed0 = cell_hash_seed(origin);
    for (int i = -1; i <= 2; ++i)
        for (int j = -1; j <= 2; ++j)
#pragma unroll
            for (int k = -1; k <= 2; ++k) {
                if (SHEAR && !sheared_cell_matters(i, j, k)) continue;
                const uint32_t h = cell_hash_mix(seed0 + static_cast<uint32_t>(i * kHashX + j * kHashY + k * kHashZ));
                const f3 g = origin + mk3(static_cast<float>(i), static_cast<float>(j), static_cast<float>(k));
                f3 c;   // scattered_cell's point
                if (SHEAR) {
                    const float mx = fmaf(static_cast<float>(h & 0x3ffu), 1.0f / 512.0f, -1.0f);
                    const float my = fmaf(static_cast<float>((h >> 10) & 0x3ffu), 1.0f / 512.0f, -1.0f);
                    const float mz = fmaf(static_cast<float>((h >> 20) & 0x3ffu), 1.0f / 512.0f, -1.0f);
                    c = mk3(fmaf(half_scatter, mx, g.x), fmaf(half_scatter, my, g.y), fmaf(half_scatter, mz, g.z));
                } else {
                    c = jitter_unit(mk3(g.x - 0.5f, g.y - 0.5f, g.z - 0.5f), h);   // scatter 1: the same value with one rounding, see jitter_unit
                }
                f3 q = location - c;
                if (SHEAR) q = mk3((1.0f * q.x + 0.5f * q.y) + 0.0f * q.z, (0.0f * q.x + 0.5f * q.y) + 0.0f * q.z, (0.0f * q.x + 0.0f * q.y) + 1.0f * q.z);
                const float d = norm_sq(q);
                if (d < v.sqr) {
                    v.next_sqr = fminf(v.sqr, v.next_sqr);
                    v.sqr = d;
                    v.hash = h;
                } else {
                    v.next_sqr = fminf(d, v.next_sqr);
                }
            }
    return v;
}
// (the plain form is only ever used with scatter 1: its cells' points come from jitter_unit)
__device__ __forceinline__ VoronoiHit voronoi_cells(f3 location) { return voronoi_general<false>(location, 0.5f); }

__device__ __forceinline__ float rect_sdf(f3 half, f3 p) {   // Rectangle::signedDistance src/Geometry.cpp:66-75
    const float dx = fabsf(p.x) - half.x, dy = fabsf(p.y) - half.y, dz = fabsf(p.z) - half.z;
    return norm(mk3(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f), fmaxf(dz, 0.0f))) + fminf(fmaxf(fmaxf(dx, dy), dz), 0.0f);
}

// RoughBoxObject::signedDistance :237-246: a box eroded by noise, cut by the walls of a voronoi foam.
__device__ __forceinline__ float roughbox_sdf(f3 half, const float* T, f3 pos) {
    const float voronoi_scale = 0.03723f;
    const VoronoiHit v = voronoi_cells(pos * (1.0f / voronoi_scale));   // Vec / float multiplies by the reciprocal (src/LinearAlgebra.cpp:278-281)
    return fmaxf(rect_sdf(half, pos) - 0.3f + 2.0f * noise_seq(T, mk3(1.0f * pos.x, 1.0f * pos.y, 1.0f * pos.z)),
                 (sqrtf(v.sqr) - sqrtf(v.next_sqr) + 0.1f) * voronoi_scale);
}

// RoughBoxObject::deflectLocalRay :248-259 and WeirdWallObject::deflectLocalRay :330-341: the cell of a
// foam colours the surface.  The reference forms each channel through double (`/ 256.0`); the float sum
// rounds the same exact value once.
__device__ __forceinline__ f3 foam_albedo(f3 pos, float voronoi_scale) {
    const VoronoiHit v = voronoi_cells(pos * (1.0f / voronoi_scale));
    return mk3(0.6f + 0.4f * static_cast<float>(v.hash & 0xffu) / 256.0f,
               0.3f + 0.5f * static_cast<float>((v.hash >> 8) & 0xffu) / 256.0f,
               0.1f + 0.4f * static_cast<float>((v.hash >> 16) & 0xffu) / 256.0f);
}

__device__ __forceinline__ float smin(float a, float b, float k) {   // :72-76
    return -log2f(exp2f(-k * a) + exp2f(-k * b)) / k;
}

// RoughSphereObject::signedDistance :165-182: a sphere blended with a box, cratered and bumped by noise.
__device__ __forceinline__ float roughsphere_sdf(float radius, const float* T, f3 pos) {
    const float r = 0.8f * radius;
    const float sd_box = rect_sdf(mk3(r, r, r), pos);
    const float sd_sphere = norm(pos) - radius;
    const float sd = smin(sd_box, sd_sphere, 8.0f);
    const float craters = fmaxf(0.0f, -1.0f + 3.0f * noise_seq(T, 1.0f * pos));
    const float bumps = -0.15f + 0.5f * noise_seq(T, 4.0f * pos);
    return fminf(sd + craters, sd + craters + bumps);
}

// WeirdWallObject::signedDistance :281-328: a ground slab, a brick wall with an arch, eroded by noise.
__device__ __forceinline__ float weirdwall_sdf(const float* T, f3 pos) {
    const float pi = 3.141592654f;
    const float sdf_ground_clean = rect_sdf(mk3(50.0f, 0.2f, 10.0f), pos - mk3(0.0f, 3.0f, 0.0f));
    const float sdf_wall_proxy = rect_sdf(mk3(50.0f, 4.0f, 0.4f), pos);
    const float sdf_proxy = smin(sdf_ground_clean - 0.2f, sdf_wall_proxy, 2.0f);
    if (sdf_proxy > 0.1f) return sdf_proxy;

    const float sdf_ground = sdf_ground_clean + noise_seq(T, pos);
    const float dist_xy = hypotf(pos.x, pos.y);
    const float angle_xy = atan2f(pos.y, pos.x);
    const float ring_band = fabsf(dist_xy - 1.5f) - (fabsf(angle_xy + 0.5f * pi) < 0.18f ? 0.5f : 0.3f);
    const float spoke_spacing = 18.0f;
    const float scaled_spoke_angle = angle_xy * spoke_spacing * 0.5f / pi;
    const float spoke_angle = (scaled_spoke_angle - roundf(scaled_spoke_angle)) / spoke_spacing;
    const float ring_spokes = fabsf(spoke_angle * dist_xy) * (2.0f * pi) - 0.06f;
    const float arch_ring = fmaxf(ring_band, -ring_spokes);
    const float pillar_columns = fabsf(fabsf(pos.x) - 1.5f) - 0.3f;
    const float slice_spacing = 1.5f;
    const float pillar_slices = fabsf(pos.y * slice_spacing - roundf(pos.y * slice_spacing)) / slice_spacing - 0.06f;
    const float arch_pillars = fmaxf(pillar_columns, -pillar_slices);
    const float sdf_arch = fmaxf(pos.y < 0.0f ? arch_ring : arch_pillars, fabsf(pos.z) - 0.3f);
    const float arch_center = pos.y < 0.0f ? dist_xy : fabsf(pos.x);

    const VoronoiHit bricks = voronoi_general<true>(pos + 1.5f * mk3(noise_seq(T, pos), noise_seq(T, pos + mk3(10.0f, 10.0f, 10.0f)), 0.0f), 0.3f * 0.5f);
    const float bricks_everywhere = fmaxf(-fabsf(sqrtf(bricks.sqr) - sqrtf(bricks.next_sqr)) + 0.05f, fabsf(pos.z) - 0.2f);
    const float brick_wall = -smin(-fmaxf(bricks_everywhere, -arch_center + 1.85f), 3.0f + pos.y + expf(-0.125f * (pos.x * pos.x)), 2.0f);

    return smin(fminf(brick_wall, sdf_arch), sdf_ground, 3.0f) + 0.23f * noise_seq(T, pos * 5.1712f) + 0.06f * noise_seq(T, pos * 14.81552f);
}

// The lane-evaluated kinds behind one switch: kind 5 RoughBoxObject {halfSize}, 6 RoughSphereObject {radius},
// 7 WeirdWallObject {}.  `prm` holds the first three parameters of the primitive.
__device__ __forceinline__ float plain_sdf(uint32_t kind, f3 prm, const float* T, f3 pos) {
    if (kind == 5u) return roughbox_sdf(prm, T, pos);
    if (kind == 6u) return roughsphere_sdf(prm.x, T, pos);
    return weirdwall_sdf(T, pos);
}
// Half extent of *::localBoundingBox (:216-221, :261-266, :343-348).
__device__ __forceinline__ f3 plain_box(uint32_t kind, f3 prm) {
    if (kind == 5u) return mk3(2.0f * prm.x, 2.0f * prm.y, 2.0f * prm.z);
    if (kind == 6u) {
        const float h = 1.2f * prm.x;
        return mk3(h, h, h);
    }
    return mk3(50.0f, 20.0f, 20.0f);
}
