// The demo's signed-distance objects that are evaluated one point per lane, in the reference's own
// order of operations (PathTracerDemo.cpp:26-154, 228-270).  Included by kernels.hip after
// craterball.h (whose hashed lattice cell it shares).  Everything here is +, -, *, sqrt, floor,
// min and max with contraction off, so the values are the reference's bit for bit.
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
__device__ __forceinline__ VoronoiHit voronoi_cells(f3 location) {
    VoronoiHit v{3.402823466e38f, 3.402823466e38f, 0u};
    const f3 origin = mk3(floorf(location.x), floorf(location.y), floorf(location.z));
    for (int i = -1; i <= 2; ++i)
        for (int j = -1; j <= 2; ++j)
#pragma unroll
            for (int k = -1; k <= 2; ++k) {
                uint32_t h;
                const f3 c = scattered_cell(origin + mk3(static_cast<float>(i), static_cast<float>(j), static_cast<float>(k)), 0.5f, h);
                const float d = norm_sq(location - c);
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

// RoughBoxObject::deflectLocalRay :248-259: the cell of a coarser foam colours the surface.  The reference
// forms each channel through double (`/ 256.0`); the float sum rounds the same exact value once.
__device__ __forceinline__ f3 roughbox_albedo(f3 pos) {
    const float voronoi_scale = 0.2f;
    const VoronoiHit v = voronoi_cells(pos * (1.0f / voronoi_scale));
    return mk3(0.6f + 0.4f * static_cast<float>(v.hash & 0xffu) / 256.0f,
               0.3f + 0.5f * static_cast<float>((v.hash >> 8) & 0xffu) / 256.0f,
               0.1f + 0.4f * static_cast<float>((v.hash >> 16) & 0xffu) / 256.0f);
}
