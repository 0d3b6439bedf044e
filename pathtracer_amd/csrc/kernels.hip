// Wavefront path-tracing stages for gfx950 (MI355X): generate -> [extend -> sdf_march -> shade ->
// shade_sdf] per bounce -> accumulate, plus the tone mappers.
//
// Numerics.  This file is compiled with -ffp-contract=off and HIP's correctly rounded fp32 divide and
// square root, so everything in the "exact" part evaluates the reference's expressions in the
// reference's order and returns the same bits as a baseline x86-64 build of it: camera rays, the
// affine wrapping of Object::hitRay / deflectRay, the analytic intersections and normals, the material.
// The crater-ball SDF (tanh, pow, an 8-octave noise: a few thousand flops per evaluation, the
// VALU-bound part of the demo scene) lives in the "fast" part, where contraction to FMA and the
// hardware rcp / rsq / exp2 / log2 are allowed; its results agree with the reference to ~1e-6.
//
// Citations are file:line of the reference checkout (timstr/pathtracer).
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace mipt {

#pragma clang fp contract(off)

// =============================================================================================
// exact part
// =============================================================================================
// Two floats that packed instructions (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32) work on at once: the same
// IEEE operations as on single floats, two per issue slot.
typedef float fx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fx2 fma2(fx2 a, fx2 b, fx2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ fx2 splat2(float v) { return fx2{v, v}; }

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float t) { return {a.x * t, a.y * t, a.z * t}; }
__device__ __forceinline__ f3 operator*(float t, f3 a) { return {a.x * t, a.y * t, a.z * t}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 u, f3 v) {
    return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
__device__ __forceinline__ float norm_sq(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float norm(f3 a) { return sqrtf(norm_sq(a)); }
__device__ __forceinline__ f3 unit(f3 a) {   // Vec::unit src/LinearAlgebra.cpp:25-29
    const float l = norm(a);
    return {a.x / l, a.y / l, a.z / l};
}
// Linear * Vec = column(0)*x + column(1)*y + column(2)*z  (src/LinearAlgebra.cpp:380-382)
__device__ __forceinline__ f3 lin_mul(const float* m, f3 v) {
    return {(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
            (m[6] * v.x + m[7] * v.y) + m[8] * v.z};
}
__device__ __forceinline__ f3 aff_pos(const float* a, f3 p) {   // Affine * Pos :401-403
    const f3 l = lin_mul(a, p);
    return {l.x + a[9], l.y + a[10], l.z + a[11]};
}
__device__ __forceinline__ f3 ld3(float4 v) { return {v.x, v.y, v.z}; }
__device__ __forceinline__ float4 st4(f3 v, float w = 0.0f) { return make_float4(v.x, v.y, v.z, w); }

// ---- counter-based random numbers: value = f(seed, pixel, sample, stage, k) ------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
struct Rng {
    uint64_t key;
    uint32_t ctr;   // stage * 64 + k + 1
    __device__ __forceinline__ Rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stage)
        : key(mix64(((static_cast<uint64_t>(pixel) << 32) | sample) + seed * 0x9e3779b97f4a7c15ull)), ctr(stage * 64u + 1u) {}
    __device__ __forceinline__ float u01() {
        const uint64_t r = mix64(key + static_cast<uint64_t>(ctr++) * 0x9e3779b97f4a7c15ull);
        return static_cast<float>(static_cast<uint32_t>(r >> 40)) * (1.0f / 16777216.0f);
    }
    // std::uniform_real_distribution<float>{a, b}: (b - a) * u + a
    __device__ __forceinline__ float uniform(float a, float b) { return (b - a) * u01() + a; }
};

// ---- hit keys: closest = smallest 64-bit key; ties go to the lower primitive index -----------
__device__ __forceinline__ uint32_t ordered_bits(float t) {
    const uint32_t b = __float_as_uint(t);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ unsigned long long make_key(float t, uint32_t prim) {
    return (static_cast<unsigned long long>(ordered_bits(t)) << 32) | prim;
}
constexpr unsigned long long kMissKey = ~0ull;
// In a slot's hit key while the path's SDF march waits for the next round's launch (k_sdf_march): the closest hit is
// not known yet, shading passes the path by.  (No hit has this key: its distance field is a NaN's.)
constexpr unsigned long long kCarriedKey = ~0ull - 1ull;

// ---- analytic intersections (src/Geometry.cpp) -------------------------------------------------
constexpr float kGeoEps = 1e-3f;   // :8

__device__ __forceinline__ bool hit_sphere(f3 o, f3 d, float radius, f3& out) {   // :173-201
    const float a = norm_sq(d);
    const float b = 2.0f * dot(d, o);
    const float c = norm_sq(o) - (radius * radius);
    const float disc = b * b - 4.0f * a * c;
    if (disc <= 0.0f) return false;
    const float sq = sqrtf(disc);
    const float t0 = (-b - sq) / (2.0f * a);
    const float t1 = (-b + sq) / (2.0f * a);
    float t;
    if (t0 > kGeoEps) {
        t = (t1 > kGeoEps) ? fminf(t0, t1) : t0;
    } else {
        if (t1 > kGeoEps) t = t1;
        else return false;
    }
    out = o + t * d;
    return true;
}

// intersect(Ray, Rectangle) :203-252.  Faces whose direction component is below 1e-3 in magnitude are
// skipped, t > 0 carries no epsilon, and from inside the exit face is returned.
__device__ __forceinline__ bool hit_rect(f3 o, f3 d, f3 half, f3& out) {
    const f3 lo = -o - half;
    const f3 hi = -o + half;
    float best = 0.0f;
    bool have = false;
#define MIPT_FACE(DK, CK, DA, LA, HA, DB, LB, HB)                                  \
    if (fabsf(DK) >= kGeoEps) {                                                    \
        const float t = (CK) / (DK);                                               \
        if (t > 0.0f) {                                                            \
            const float va = (DA) * t, vb = (DB) * t;                              \
            if (va >= (LA) && va <= (HA) && vb >= (LB) && vb <= (HB)) {            \
                best = have ? fminf(best, t) : t;                                  \
                have = true;                                                       \
            }                                                                      \
        }                                                                          \
    }
    MIPT_FACE(d.x, lo.x, d.y, lo.y, hi.y, d.z, lo.z, hi.z)
    MIPT_FACE(d.x, hi.x, d.y, lo.y, hi.y, d.z, lo.z, hi.z)
    MIPT_FACE(d.y, lo.y, d.x, lo.x, hi.x, d.z, lo.z, hi.z)
    MIPT_FACE(d.y, hi.y, d.x, lo.x, hi.x, d.z, lo.z, hi.z)
    MIPT_FACE(d.z, lo.z, d.x, lo.x, hi.x, d.y, lo.y, hi.y)
    MIPT_FACE(d.z, hi.z, d.x, lo.x, hi.x, d.y, lo.y, hi.y)
#undef MIPT_FACE
    if (!have) return false;
    out = o + (best * d);
    return true;
}

__device__ __forceinline__ bool hit_triangle(f3 o, f3 d, f3 a, f3 b, f3 c, f3& out) {   // :146-171
    const f3 e1 = b - a, e2 = c - a;
    const f3 h = cross(d, e2);
    const float det = dot(e1, h);
    if (fabsf(det) < kGeoEps) return false;
    const float f = 1.0f / det;
    const f3 s = o - a;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const f3 q = cross(s, e1);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = f * dot(e2, q);
    if (!(t > kGeoEps)) return false;
    out = o + t * d;
    return true;
}

__device__ __forceinline__ bool inside_rect(f3 p, f3 half) {   // :267-272
    return (fabsf(p.x) <= half.x) && (fabsf(p.y) <= half.y) && (fabsf(p.z) <= half.z);
}

__device__ __forceinline__ f3 rect_normal(f3 p, f3 half) {   // Rectangle::normal :77-103
    const float ax = fabsf(p.x / half.x), ay = fabsf(p.y / half.y), az = fabsf(p.z / half.z);
    const float sx = p.x > 0.0f ? 1.0f : -1.0f, sy = p.y > 0.0f ? 1.0f : -1.0f, sz = p.z > 0.0f ? 1.0f : -1.0f;
    if (ax > ay) {
        if (ax > az) return mk3(sx, 0.0f, 0.0f);
        return mk3(0.0f, 0.0f, sz);
    }
    if (ay > az) return mk3(0.0f, sy, 0.0f);
    return mk3(0.0f, 0.0f, sz);
}

// ---- material (src/BasicMaterial.cpp:97-159) and samplers (src/Geometry.cpp:290-341) ------------
__device__ __forceinline__ f3 mirror(f3 in, f3 n) { return in - (2.0f * (dot(in, n) * n)); }   // bounce()

__device__ __forceinline__ f3 cosine_hemisphere(f3 n, Rng& rng) {
    n = unit(n);
    float x, y, l2;
    do {
        x = rng.uniform(-1.0f, 1.0f);
        y = rng.uniform(-1.0f, 1.0f);
        l2 = (x * x) + (y * y);
    } while (l2 > 1.0f);
    const float z = sqrtf(1.0f - l2);
    // orthogonalPair :290-299 (normalises its argument once more; the 0.1 is a double literal)
    const f3 v = unit(n);
    const f3 c = cross(v, mk3(1.0f, 0.0f, 0.0f));
    const bool aligned_to_x = static_cast<double>(norm_sq(c)) < 0.1;
    const f3 w = aligned_to_x ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    const f3 v1 = cross(w, v);
    const f3 v2 = cross(v1, v);
    return (z * n) + (x * unit(v1)) + (y * unit(v2));
}

struct Bounce {
    f3 emitted, attenuation, dir;
};

// The reference's branches (struck from inside; diffuse, specular, transmitted) written so that lanes of one wave
// that take different branches share the two expensive pieces, the hemisphere sample and the final normalisation,
// instead of running a copy of each per branch one after the other.  Expressions and the order of the random
// draws are the reference's.
__device__ __forceinline__ Bounce deflect(const MaterialRec& m, f3 in, f3 n, Rng& rng) {
    Bounce b;
    const float s = dot(in, n);
    const bool inside = s >= 0.0f;   // struck from inside
    bool hemi = false, specular = false;
    if (!inside) {
        const float reflection = m.diffuse + m.specular;
        const float options = reflection + m.transmittance;
        const float which = rng.uniform(0.0f, options);
        hemi = which < reflection;
        specular = hemi && !(which < m.diffuse);
    }
    f3 h = mk3(0.0f, 0.0f, 0.0f);
    if (hemi) h = cosine_hemisphere(n, rng);
    const bool diffuse = hemi && !specular;
    f3 v = h;
    if (!diffuse) {
        f3 raw;
        if (specular) {
            const f3 sc = (1.0f - m.sharpness) * h;
            raw = mirror(in, n) + sc;
        } else {   // refraction, out of or into the object
            const float k = inside ? 1.0f - m.ior : 1.0f - 1.0f / m.ior;
            raw = in + ((s * k) * n);
        }
        v = unit(raw);
    }
    if (inside) {
        if (dot(v, n) >= 0.0f) {
            b.emitted = mk3(0.0f, 0.0f, 0.0f);
            b.attenuation = mk3(1.0f, 1.0f, 1.0f);
            b.dir = v;
        } else {
            b.emitted = mk3(0.0f, 1.0f, 1.0f);
            b.attenuation = mk3(0.0f, 0.0f, 0.0f);
            b.dir = mirror(in, -n);
        }
        return b;
    }
    b.emitted = mk3(m.emitted[0], m.emitted[1], m.emitted[2]);
    b.attenuation = mk3(m.albedo[0], m.albedo[1], m.albedo[2]);
    b.dir = v;
    return b;
}

// ---- loads -----------------------------------------------------------------------------------
struct PrimRegs {
    float inv[12], fwd[12], p[6];
    uint32_t kind, material;
};
// First half of a record: what the test of a sphere or a box reads.
__device__ __forceinline__ void load_prim_test(const PrimRec* prims, uint32_t i, PrimRegs& r) {
    const float4* src = reinterpret_cast<const float4*>(prims + i);
    const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    r.inv[0] = v0.x; r.inv[1] = v0.y; r.inv[2] = v0.z; r.inv[3] = v0.w;
    r.inv[4] = v1.x; r.inv[5] = v1.y; r.inv[6] = v1.z; r.inv[7] = v1.w;
    r.inv[8] = v2.x; r.inv[9] = v2.y; r.inv[10] = v2.z; r.inv[11] = v2.w;
    r.p[0] = v3.x; r.p[1] = v3.y; r.p[2] = v3.z;
    r.kind = __float_as_uint(v3.w);
}
// Second half: the way back to world space, parameters 3-5, the material.
__device__ __forceinline__ void load_prim_rest(const PrimRec* prims, uint32_t i, PrimRegs& r) {
    const float4* src = reinterpret_cast<const float4*>(prims + i) + 4;
    const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    r.fwd[0] = v0.x; r.fwd[1] = v0.y; r.fwd[2] = v0.z; r.fwd[3] = v0.w;
    r.fwd[4] = v1.x; r.fwd[5] = v1.y; r.fwd[6] = v1.z; r.fwd[7] = v1.w;
    r.fwd[8] = v2.x; r.fwd[9] = v2.y; r.fwd[10] = v2.z; r.fwd[11] = v2.w;
    r.p[3] = v3.x; r.p[4] = v3.y; r.p[5] = v3.z;
    r.material = __float_as_uint(v3.w);
}
__device__ __forceinline__ void load_prim(const PrimRec* prims, uint32_t i, PrimRegs& r) {
    load_prim_test(prims, i, r);
    load_prim_rest(prims, i, r);
}

// Object::hitRay (src/Object.cpp:22-33) for the analytic kinds: local ray (direction not
// renormalised) -> *::hitLocalRay -> world hit position; t = (hit - ray.pos) . ray.dir.
__device__ __forceinline__ bool hit_analytic(const PrimRegs& pr, const float* tri_c, uint32_t prim, f3 o, f3 d, f3& world, float& t) {
    const f3 lo = aff_pos(pr.inv, o);
    const f3 ld = lin_mul(pr.inv, d);
    f3 lh;
    bool hit = false;
    if (pr.kind == 0u) {
        hit = hit_sphere(lo, ld, pr.p[0], lh);
    } else if (pr.kind == 1u) {
        hit = hit_rect(lo, ld, mk3(pr.p[0], pr.p[1], pr.p[2]), lh);
    } else if (pr.kind == 2u) {
        const f3 c = mk3(tri_c[3 * prim], tri_c[3 * prim + 1], tri_c[3 * prim + 2]);
        hit = hit_triangle(lo, ld, mk3(pr.p[0], pr.p[1], pr.p[2]), mk3(pr.p[3], pr.p[4], pr.p[5]), c, lh);
    }
    if (!hit) return false;
    world = aff_pos(pr.fwd, lh);
    t = dot(world - o, d);
    return true;
}

// The same from the record in memory, in the two steps its layout is made for: the test reads the first half (four
// 16-byte pieces), and only a lane that hits fetches the way back to world space (three more).
__device__ __forceinline__ bool hit_analytic_from(const PrimRec* prims, const float* tri_c, uint32_t prim, f3 o, f3 d, f3& world, float& t) {
    PrimRegs pr;
    load_prim_test(prims, prim, pr);
    const f3 lo = aff_pos(pr.inv, o);
    const f3 ld = lin_mul(pr.inv, d);
    f3 lh;
    bool hit = false;
    if (pr.kind == 0u) {
        hit = hit_sphere(lo, ld, pr.p[0], lh);
    } else if (pr.kind == 1u) {
        hit = hit_rect(lo, ld, mk3(pr.p[0], pr.p[1], pr.p[2]), lh);
    } else if (pr.kind == 2u) {
        const float4 v = reinterpret_cast<const float4*>(prims + prim)[7];
        const f3 c = mk3(tri_c[3 * prim], tri_c[3 * prim + 1], tri_c[3 * prim + 2]);
        hit = hit_triangle(lo, ld, mk3(pr.p[0], pr.p[1], pr.p[2]), mk3(v.x, v.y, v.z), c, lh);
    }
    if (!hit) return false;
    const float4* src = reinterpret_cast<const float4*>(prims + prim) + 4;
    const float4 f0 = src[0], f1 = src[1], f2 = src[2];
    const float fwd[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
    world = aff_pos(fwd, lh);
    t = dot(world - o, d);
    return true;
}

// Conservative ray / box test in ray-parameter units.  NaNs from 0 * inf drop out of fmin/fmax, which
// only ever widens the interval.
__device__ __forceinline__ bool slab(f3 o, f3 inv_d, const float* lo, const float* hi, float t_limit, float& t_near) {
    const float x0 = (lo[0] - o.x) * inv_d.x, x1 = (hi[0] - o.x) * inv_d.x;
    const float y0 = (lo[1] - o.y) * inv_d.y, y1 = (hi[1] - o.y) * inv_d.y;
    const float z0 = (lo[2] - o.z) * inv_d.z, z1 = (hi[2] - o.z) * inv_d.z;
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * 1.0000005f;
    t_near = tn;
    return tn <= tf && tn <= t_limit;
}

// =============================================================================================
// FractalObject (src/Object.cpp:160-252): a 3 x 3 x 3 lattice of spheres of radius 0.2 as a signed
// distance, marched lane by lane (an evaluation is a dozen instructions).  Exact arithmetic in the
// reference's order, so its hits are bit-identical to the reference's.
// =============================================================================================
__device__ __forceinline__ float fractal_axis(float v) {   // the lambda `f` :231-242
    const float l = 0.5f, r = 2.0f;
    if (v < -l) return v + l;
    if (v > l) return v - l;
    v *= r;
    return (v - roundf(v)) / r;
}
__device__ __forceinline__ float fractal_sdf(f3 p) {   // :198-252
    return norm(mk3(fractal_axis(p.x), fractal_axis(p.y), fractal_axis(p.z))) - 0.2f;
}
// hitLocalRay :160-175.  The march starts where the LOCAL ray enters the object's WORLD bounding box
// (`intersect(ray, getBoundingBox())`; centre c, half h), takes up to 256 steps and never leaves early.
__device__ __forceinline__ bool fractal_march(f3 lo, f3 ld, f3 c, f3 h, f3& hit) {
    f3 p;
    if (!hit_rect(lo - c, ld, h, p)) return false;   // intersect(Ray, AxisAlignedBox) src/Geometry.cpp:254-261
    p = p + c;
    for (int i = 0; i < 256; ++i) {
        const float dist = fractal_sdf(p);
        if (dist < 1e-3f) {
            hit = p;
            return true;
        }
        p = p + dist * ld;
    }
    return false;
}
__device__ __forceinline__ f3 fractal_normal(f3 p) {   // deflectLocalRay :177-189, central differences
    const float delta = 1e-3f;
    const f3 dx = mk3(delta, 0.0f, 0.0f), dy = mk3(0.0f, delta, 0.0f), dz = mk3(0.0f, 0.0f, delta);
    const f3 g = mk3(fractal_sdf(p + dx) - fractal_sdf(p - dx), fractal_sdf(p + dy) - fractal_sdf(p - dy), fractal_sdf(p + dz) - fractal_sdf(p - dz));
    return unit(g * (1.0f / delta));   // Vec / float multiplies by the reciprocal (src/LinearAlgebra.cpp:278-281)
}

// Pixel id -> image coordinates (BatchParams::tiles_x); false for ids past the image (tile padding).
__device__ __forceinline__ bool pixel_of(const BatchParams& bp, uint32_t id, uint32_t& x, uint32_t& y) {
    if (bp.tiles_x == 0u) {
        y = id / bp.width;
        x = id - y * bp.width;
        return id < bp.n_pix;
    }
    if (id >= bp.n_pix) {
        x = y = 0u;
        return false;
    }
    const uint32_t tile = bp.vis_tiles ? bp.vis_tiles[id >> 6] : id >> 6;
    const uint32_t ty = tile / bp.tiles_x, tx = tile - ty * bp.tiles_x;
    x = tx * 8u + (id & 7u);
    y = ty * 8u + ((id >> 3) & 7u);
    return x < bp.width && y < bp.height;
}

// =============================================================================================
// Camera ray of sample `sample` of pixel (x, y): Renderer::doWork's sample set-up (src/Renderer.cpp:72-94)
// + Camera::getViewRay (src/Camera.cpp:65-81).  (sx0, sy0) = (x / (w - 1), y / (h - 1)) is the same for
// every sample of the pixel and is passed in.
// =============================================================================================
__device__ __forceinline__ void camera_ray(const CameraRec& cam, const BatchParams& bp, uint32_t pix, uint32_t sample, float sx0, float sy0, f3& pos,
                                           f3& dir) {
    const float max_w = static_cast<float>(bp.width - 1), max_h = static_cast<float>(bp.height - 1);
    const float dx = 1.0f / max_w, dy = 1.0f / max_h;
    Rng rng(bp.seed, pix, sample, 0u);
    float sx = sx0, sy = sy0;
    if (!(bp.flags & 1u)) {
        sx = sx + rng.uniform(-0.5f * dx, 0.5f * dx);
        sy = sy + rng.uniform(-0.5f * dy, 0.5f * dy);
    }
    const float vx = sx * 2.0f - 1.0f;
    const float vy = sy * 2.0f - 1.0f;
    const f3 sp = cam.aspect > 1.0f ? mk3(vx, vy / cam.aspect, 0.0f) : mk3(vx * cam.aspect, vy, 0.0f);
    float rx = 0.0f, ry = 0.0f;
    if (cam.blur_rad != 0.0f) {   // randomPointInCircle src/Geometry.cpp:278-288 (draws k = 2, 3, ...)
        rng.ctr = 3u;
        do {
            rx = rng.uniform(-1.0f, 1.0f);
            ry = rng.uniform(-1.0f, 1.0f);
        } while (rx * rx + ry * ry > 1.0f);
    }
    const f3 blur = mk3(rx * cam.blur_rad, ry * cam.blur_rad, 0.0f);
    const f3 view = mk3(cam.fov_scale * sp.x, cam.fov_scale * sp.y, 1.0f) + blur * cam.inv_focal;
    dir = unit(lin_mul(cam.lin, view));
    const f3 l = lin_mul(cam.lin, sp - blur);
    pos = mk3(l.x + cam.tr[0], l.y + cam.tr[1], l.z + cam.tr[2]);
}

// =============================================================================================
// Closest hit over the analytic primitives: ObjectTree::Tree::hit (src/ObjectTree.cpp:22-88) with real
// culling, as a per-lane state machine that a wave steps in two phases:
//   * inner step -- visit one four-child node: slab-test the four boxes against the current limit and order
//     the hits by entry distance; the nearest is taken next (a node to visit, or a leaf that becomes the
//     lane's pending leaf), the others go on the lane's stack, farthest first (a column of LDS dwords:
//     entry e of lane l at e * 256 + l, conflict-free);
//   * leaf step -- Object::hitRay on the pending leaf.  A wave runs it when enough lanes have a leaf
//     pending (or none can do anything else), so the long exact intersection code runs with most lanes
//     active instead of once per lane that happens to reach a leaf.
// A stack entry carries no distance: a node taken off the stack is tested against the limit when it is
// visited (its children's boxes lie inside its own), a leaf by its own box, kept per primitive.
// A lane whose ray is finished takes the next ray at once (the kernels below refill idle lanes), so a
// wave is not held up by its longest ray.  Hits are ranked by t = (hit - o).d as the reference ranks
// them; boxes are culled in ray-parameter units, t / |d|^2, with a relative and an absolute margin.
// =============================================================================================
struct Trav {
    f3 o, d, inv_d;
    float inv_len2, limit;
    unsigned long long best;
    f3 best_pos;
    int cur;     // >= 0: node to visit (once no leaf is pending); kTravNone: nothing chosen
    int leaf;    // pending leaf (primitive index) or kTravNone
    int sp;
    uint32_t slot;
    bool busy;   // holds a ray
};
constexpr int kTravNone = -1;

__device__ __forceinline__ void trav_begin(Trav& t, f3 o, f3 d, uint32_t slot, bool has_nodes) {
    t.o = o;
    t.d = d;
    t.inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    t.inv_len2 = 1.0f / norm_sq(d);
    t.limit = 3.0e38f;
    t.best = kMissKey;
    t.best_pos = mk3(0.0f, 0.0f, 0.0f);
    t.cur = has_nodes ? 0 : kTravNone;
    t.leaf = kTravNone;
    t.sp = 0;
    t.slot = slot;
    t.busy = true;
}
__device__ __forceinline__ bool trav_wants_inner(const Trav& t) { return t.busy && t.cur >= 0 && t.leaf == kTravNone; }
__device__ __forceinline__ bool trav_wants_leaf(const Trav& t) { return t.busy && t.leaf != kTravNone; }
__device__ __forceinline__ bool trav_finished(const Trav& t) { return t.busy && t.cur == kTravNone && t.leaf == kTravNone && t.sp == 0; }

// A deferred child on the stack with the distance at which the ray enters its box: the link as the tree stores it
// (node index, or ~primitive for a leaf) shifted up by 11 bits, the top 11 bits of the distance's float pattern below it
// -- sign (never set), exponent, three bits of mantissa, i.e. rounded down by at most an eighth.  An arithmetic shift
// gives the link back, sign and all.  An entry whose distance lies beyond the limit the ray has reached by the time it
// comes off the stack is dropped there and then: no node fetch, no leaf-box fetch.  For trees of up to 2^20 nodes and
// 2^20 primitives (DIST): links then fit 21 signed bits; larger ones keep plain links and the leaf boxes' re-test.
__device__ __forceinline__ int entry_pack(int c, float t) {
    return static_cast<int>((static_cast<uint32_t>(c) << 11) | (__float_as_uint(t) >> 20));
}

// Nothing chosen: take entries off the stack until one is a node, or a leaf whose box the limit still reaches.
__device__ __forceinline__ void trav_pop(const SceneDev& sc, Trav& t, const int* stack) {
    while (t.busy && t.cur == kTravNone && t.leaf == kTravNone && t.sp > 0) {
        --t.sp;
        const int e = stack[t.sp * 256];
        if (e >= 0) {
            t.cur = e;
        } else {
            const float4* lb = reinterpret_cast<const float4*>(sc.leaf_boxes) + 2 * static_cast<size_t>(~e);
            const float4 lo = lb[0], hi = lb[1];
            float tn;
            if (slab(t.o, t.inv_d, &lo.x, &hi.x, t.limit, tn)) t.leaf = ~e;
        }
    }
}

// Orders two (distance, link) pairs, the nearer first.
__device__ __forceinline__ void trav_order(float& ta, int& ca, float& tb, int& cb) {
    const bool sw = tb < ta;
    const float tl = sw ? tb : ta, th = sw ? ta : tb;
    const int cl = sw ? cb : ca, ch = sw ? ca : cb;
    ta = tl;
    tb = th;
    ca = cl;
    cb = ch;
}

// Entry distance of the ray into each of two boxes given the ray parameters at which it crosses the boxes' near (a)
// and far (b) planes per axis, or `far` for a box the ray misses or enters beyond `limit`.  The same decision as
// slab(), two boxes at a time.
__device__ __forceinline__ fx2 slab_pair(fx2 ax, fx2 ay, fx2 az, fx2 bx, fx2 by, fx2 bz, float limit, float far) {
    const fx2 grow = {1.000001f, 1.000001f};
    fx2 tn, tf;
    tn.x = fmaxf(fmaxf(ax.x, ay.x), fmaxf(az.x, 0.0f));
    tn.y = fmaxf(fmaxf(ax.y, ay.y), fmaxf(az.y, 0.0f));
    tf.x = fminf(fminf(bx.x, by.x), bz.x);
    tf.y = fminf(fminf(bx.y, by.y), bz.y);
    tf = tf * grow;
    fx2 r;
    r.x = (tn.x <= tf.x && tn.x <= limit) ? tn.x : far;
    r.y = (tn.y <= tf.y && tn.y <= limit) ? tn.y : far;
    return r;
}

__device__ __forceinline__ float ubyte0(uint32_t w) { return static_cast<float>(w & 0xffu); }
__device__ __forceinline__ float ubyte1(uint32_t w) { return static_cast<float>((w >> 8) & 0xffu); }
__device__ __forceinline__ float ubyte2(uint32_t w) { return static_cast<float>((w >> 16) & 0xffu); }
__device__ __forceinline__ float ubyte3(uint32_t w) { return static_cast<float>(w >> 24); }

// Visits four-child node `cur`: the links of the children whose boxes the ray enters within `limit`, nearest first
// (kBvh4Absent for the rest).  The node is the 64-byte form (Bvh4QNode): a plane lies at org + scale q, so the ray
// crosses it at q (scale / d) + (org - o) / d -- one multiply-add per plane once the two factors of an axis are made,
// the offsets q read as bytes.  Of each axis' two planes the ray crosses the one on the side it comes from first,
// which depends on the sign of the direction alone: the low or the high word is the near one for all four children.
// The decoded boxes contain the children's (quantize_bvh4), rounding in the three steps stays far below one step of
// the offsets, which the quantiser gives away on either side, and relative error in the far regime is covered by
// `grow` as in slab(): the test never drops a box the ray enters.
__device__ __forceinline__ void trav_visit(const SceneDev& sc, int cur, f3 o, f3 inv_d, float limit, int& c0, int& c1, int& c2, int& c3, float& e1, float& e2,
                                           float& e3) {
    const uint4* q = reinterpret_cast<const uint4*>(sc.nodes) + 4u * static_cast<uint32_t>(cur);
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
    const int4 ch = *reinterpret_cast<const int4*>(q + 3);
    // Everything below is selects rather than branches: the lanes of a wave take every combination of
    // hit / miss and leaf / node at once, and a branch per case costs more than computing all of them.
    const float far = 3.4e38f;
    const float Ax = __uint_as_float(q0.w) * inv_d.x, Ay = __uint_as_float(q1.x) * inv_d.y, Az = __uint_as_float(q1.y) * inv_d.z;
    const float Bx = (__uint_as_float(q0.x) - o.x) * inv_d.x, By = (__uint_as_float(q0.y) - o.y) * inv_d.y, Bz = (__uint_as_float(q0.z) - o.z) * inv_d.z;
    const bool nx = inv_d.x < 0.0f, ny = inv_d.y < 0.0f, nz = inv_d.z < 0.0f;
    const uint32_t near_x = nx ? q2.y : q1.z, far_x = nx ? q1.z : q2.y;
    const uint32_t near_y = ny ? q2.z : q1.w, far_y = ny ? q1.w : q2.z;
    const uint32_t near_z = nz ? q2.w : q2.x, far_z = nz ? q2.x : q2.w;
    const fx2 ax2 = splat2(Ax), ay2 = splat2(Ay), az2 = splat2(Az), bx2 = splat2(Bx), by2 = splat2(By), bz2 = splat2(Bz);
    const fx2 t01 = slab_pair(fma2(fx2{ubyte0(near_x), ubyte1(near_x)}, ax2, bx2), fma2(fx2{ubyte0(near_y), ubyte1(near_y)}, ay2, by2),
                              fma2(fx2{ubyte0(near_z), ubyte1(near_z)}, az2, bz2), fma2(fx2{ubyte0(far_x), ubyte1(far_x)}, ax2, bx2),
                              fma2(fx2{ubyte0(far_y), ubyte1(far_y)}, ay2, by2), fma2(fx2{ubyte0(far_z), ubyte1(far_z)}, az2, bz2), limit, far);
    const fx2 t23 = slab_pair(fma2(fx2{ubyte2(near_x), ubyte3(near_x)}, ax2, bx2), fma2(fx2{ubyte2(near_y), ubyte3(near_y)}, ay2, by2),
                              fma2(fx2{ubyte2(near_z), ubyte3(near_z)}, az2, bz2), fma2(fx2{ubyte2(far_x), ubyte3(far_x)}, ax2, bx2),
                              fma2(fx2{ubyte2(far_y), ubyte3(far_y)}, ay2, by2), fma2(fx2{ubyte2(far_z), ubyte3(far_z)}, az2, bz2), limit, far);
    float t0 = t01.x, t1 = t01.y, t2 = t23.x, t3 = t23.y;
    // a child that is not hit is ordered last and marked absent (an entry distance is far below 3.4e38:
    // limits start at 3e38)
    c0 = t0 < far ? ch.x : kBvh4Absent;
    c1 = t1 < far ? ch.y : kBvh4Absent;
    c2 = t2 < far ? ch.z : kBvh4Absent;
    c3 = t3 < far ? ch.w : kBvh4Absent;
    trav_order(t0, c0, t1, c1);
    trav_order(t2, c2, t3, c3);
    trav_order(t0, c0, t2, c2);
    trav_order(t1, c1, t3, c3);
    trav_order(t1, c1, t2, c2);
    e1 = t1;   // where the ray enters the deferred children
    e2 = t2;
    e3 = t3;
}

__device__ __forceinline__ void trav_inner_step(const SceneDev& sc, Trav& t, int* stack) {
    int c0, c1, c2, c3;
    float e1, e2, e3;
    trav_visit(sc, t.cur, t.o, t.inv_d, t.limit, c0, c1, c2, c3, e1, e2, e3);
    // the farther hits wait on the stack, the farthest at the bottom
    if (c3 != kBvh4Absent) {
        stack[t.sp * 256] = c3;
        ++t.sp;
    }
    if (c2 != kBvh4Absent) {
        stack[t.sp * 256] = c2;
        ++t.sp;
    }
    if (c1 != kBvh4Absent) {
        stack[t.sp * 256] = c1;
        ++t.sp;
    }
    // the nearest is next: a node to visit or a leaf to test
    t.cur = c0 >= 0 ? c0 : kTravNone;
    t.leaf = (c0 < 0 && c0 != kBvh4Absent) ? ~c0 : kTravNone;
}

__device__ __forceinline__ void trav_test_leaf(const SceneDev& sc, Trav& t, uint32_t prim) {
    PrimRegs pr;
    load_prim(sc.prims, prim, pr);
    f3 w;
    float th;
    if (hit_analytic(pr, sc.tri_c, prim, t.o, t.d, w, th)) {
        const unsigned long long key = make_key(th, prim);
        if (key < t.best) {
            t.best = key;
            t.best_pos = w;
            t.limit = fmaxf(th, 0.0f) * t.inv_len2 * 1.00001f + 1e-6f;
        }
    }
}

__device__ __forceinline__ void trav_leaf_step(const SceneDev& sc, Trav& t) {
    trav_test_leaf(sc, t, static_cast<uint32_t>(t.leaf));
    t.leaf = kTravNone;
}

// Steps the wave's rays until fewer than `keep_busy` lanes hold an unfinished ray (so the caller can
// refill) -- or, with keep_busy = 0, until all are finished.  Always takes at least one step.  All 64
// lanes call.
// COUNT: the counting build -- the same steps in the same order, and per lane the number of nodes visited and of
// primitives tested (what the traversal's memory traffic is priced by).
struct TravCounts {
    uint32_t nodes, prims;
    uint32_t rounds = 0, leaf_steps = 0, restarts = 0;   // wave-uniform (two rays per lane only)
};
template <bool COUNT>
__device__ __forceinline__ void trav_run(const SceneDev& sc, Trav& t, int* stack, uint32_t keep_busy, TravCounts& counts, uint32_t kLeafQuorum = 20u) {
    // kLeafQuorum: lanes with a pending leaf that make a leaf step worth its cost
    for (;;) {
        trav_pop(sc, t, stack);
        if (trav_wants_inner(t)) {
            trav_inner_step(sc, t, stack);
            if (COUNT) ++counts.nodes;
        }
        const uint32_t n_leaf = static_cast<uint32_t>(__popcll(__ballot(trav_wants_leaf(t))));
        if (n_leaf >= kLeafQuorum || (n_leaf > 0 && !__ballot(t.busy && t.leaf == kTravNone && (t.cur >= 0 || t.sp > 0)))) {
            if (trav_wants_leaf(t)) {
                trav_leaf_step(sc, t);
                if (COUNT) ++counts.prims;
            }
        }
        const uint32_t n_unfinished = static_cast<uint32_t>(__popcll(__ballot(t.busy && !trav_finished(t))));
        if (n_unfinished == 0 || n_unfinished < keep_busy) return;
    }
}

// =============================================================================================
// Two rays per lane.  A lane whose ray waits -- for the wave's next leaf step, or finished and not yet replaced --
// would idle through the wave's inner steps; with a second ray it visits that ray's nodes meanwhile, so inner steps
// run with most lanes and leaf steps can wait for a large quorum at no cost.  Both rays' traversal stacks share one
// LDS column of S entries: ray 0 fills it from the bottom, ray 1 from the top.  Ray 0 is never held up: when its next
// visit might not find three free entries, ray 1 starts over from the root (it keeps its best hit and limit, so the
// second pass is cheaper and the result the same -- the closest hit does not depend on the order of the tests); ray 1
// only advances while three entries are free.  S >= the tree's single-ray need + 3, so a ray alone always finds them.
// =============================================================================================
// A ray of a pair, pared down to what must live in registers: `next` is a node link as the tree stores it (>= 0: node to
// visit, kBvh4Absent: nothing chosen, any other negative value: leaf ~next waiting for the wave's next leaf step);
// slot ~0: the lane holds no ray here.  The best hit's position goes to the path's slot whenever a hit improves.
struct PairRay {
    f3 o, d, inv_d;
    float limit;
    unsigned long long best;
    int next, sp;
    uint32_t slot;
};
constexpr uint32_t kNoSlot = 0xffffffffu;
__device__ __forceinline__ bool pair_busy(const PairRay& t) { return t.slot != kNoSlot; }
__device__ __forceinline__ bool pair_wants_leaf(const PairRay& t) { return t.next < 0 && t.next != kBvh4Absent; }
__device__ __forceinline__ bool pair_can_advance(const PairRay& t) { return t.next >= 0 || (t.next == kBvh4Absent && t.sp > 0); }
__device__ __forceinline__ bool pair_finished(const PairRay& t) { return pair_busy(t) && t.next == kBvh4Absent && t.sp == 0; }
__device__ __forceinline__ bool pair_unfinished(const PairRay& t) { return pair_busy(t) && !(t.next == kBvh4Absent && t.sp == 0); }
__device__ __forceinline__ void pair_idle(PairRay& t) {
    t.slot = kNoSlot;
    t.next = kBvh4Absent;
    t.sp = 0;
}
__device__ __forceinline__ void pair_begin(PairRay& t, f3 o, f3 d, uint32_t slot, bool has_nodes) {
    t.o = o;
    t.d = d;
    t.inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    t.limit = 3.0e38f;
    t.best = kMissKey;
    t.next = has_nodes ? 0 : kBvh4Absent;
    t.sp = 0;
    t.slot = slot;
}

template <int S, bool COUNT, bool DIST>
__device__ __forceinline__ void trav_run_pair(const SceneDev& sc, const PathBuffers& pb, PairRay& r0, PairRay& r1, int* col, int rows, uint32_t keep_unfinished,
                                              TravCounts& counts, uint32_t leaf_quorum) {
    // rows: entries of the column in use (S, or fewer as a test aid; never below the tree's single-ray need + 3)
    for (;;) {
        // ---- one visit per lane: ray 0 if it can go on, else ray 1 ----
        const bool a0 = pair_can_advance(r0);
        const bool room = r0.sp + r1.sp + 3 <= rows;
        if (COUNT) ++counts.rounds;
        if (a0 && !room && pair_unfinished(r1)) {   // (rare) ray 0 needs the column: ray 1 starts over
            if (COUNT) ++counts.restarts;
            r1.next = 0;
            r1.sp = 0;
        }
        const bool a1 = !a0 && room && pair_can_advance(r1);
        if (a0 || a1) {
            f3 o = mk3(a1 ? r1.o.x : r0.o.x, a1 ? r1.o.y : r0.o.y, a1 ? r1.o.z : r0.o.z);
            f3 inv_d = mk3(a1 ? r1.inv_d.x : r0.inv_d.x, a1 ? r1.inv_d.y : r0.inv_d.y, a1 ? r1.inv_d.z : r0.inv_d.z);
            const float limit = a1 ? r1.limit : r0.limit;
            int next = a1 ? r1.next : r0.next, sp = a1 ? r1.sp : r0.sp;
            // entry k of ray 0 at row k, of ray 1 at row rows - 1 - k; `p` is the next free row, `step` the way it grows
            const int step = a1 ? -256 : 256;
            int* p = col + (a1 ? (rows - 1 - sp) * 256 : sp * 256);
            // nothing chosen: take entries off the stack until one is a node, or a leaf whose box the limit still reaches
            while (next == kBvh4Absent && sp > 0) {
                --sp;
                p -= step;
                const int e = *p;
                if (DIST) {
                    if (((static_cast<uint32_t>(e) & 0x7ffu) << 20) > __float_as_uint(limit)) continue;   // entered beyond the limit (both patterns are of non-negative floats)
                    next = e >> 11;   // arithmetic: a leaf's link stays negative
                } else if (e >= 0) {
                    next = e;
                } else {
                    const float4* lb = reinterpret_cast<const float4*>(sc.leaf_boxes) + 2 * static_cast<size_t>(~e);
                    const float4 lo = lb[0], hi = lb[1];
                    float tn;
                    if (slab(o, inv_d, &lo.x, &hi.x, limit, tn)) next = e;
                }
            }
            if (next >= 0) {
                int c0, c1, c2, c3;
                float e1, e2, e3;
                trav_visit(sc, next, o, inv_d, limit, c0, c1, c2, c3, e1, e2, e3);
                if (COUNT) ++counts.nodes;
                if (c3 != kBvh4Absent) {
                    *p = DIST ? entry_pack(c3, e3) : c3;
                    p += step;
                    ++sp;
                }
                if (c2 != kBvh4Absent) {
                    *p = DIST ? entry_pack(c2, e2) : c2;
                    p += step;
                    ++sp;
                }
                if (c1 != kBvh4Absent) {
                    *p = DIST ? entry_pack(c1, e1) : c1;
                    p += step;
                    ++sp;
                }
                next = c0;   // the nearest is next: a node to visit, a leaf to test, or nothing
            }
            r0.next = a1 ? r0.next : next;
            r0.sp = a1 ? r0.sp : sp;
            r1.next = a1 ? next : r1.next;
            r1.sp = a1 ? sp : r1.sp;
        }
        // ---- leaf step: when enough lanes hold a pending leaf, or no lane can do anything else ----
        const bool p0 = pair_wants_leaf(r0), p1 = pair_wants_leaf(r1);
        const uint32_t n_leaf = static_cast<uint32_t>(__popcll(__ballot(p0 || p1)));
        if (n_leaf >= leaf_quorum || (n_leaf > 0 && !__ballot(pair_can_advance(r0) || (pair_can_advance(r1) && r0.sp + r1.sp + 3 <= rows)))) {
            if (COUNT) ++counts.leaf_steps;
            if (p0 || p1) {
                const bool l1 = !p0;
                const f3 o = mk3(l1 ? r1.o.x : r0.o.x, l1 ? r1.o.y : r0.o.y, l1 ? r1.o.z : r0.o.z);
                const f3 d = mk3(l1 ? r1.d.x : r0.d.x, l1 ? r1.d.y : r0.d.y, l1 ? r1.d.z : r0.d.z);
                const uint32_t prim = static_cast<uint32_t>(~(l1 ? r1.next : r0.next));
                const unsigned long long best = l1 ? r1.best : r0.best;
                if (COUNT) ++counts.prims;
                f3 w;
                float th;
                if (hit_analytic_from(sc.prims, sc.tri_c, prim, o, d, w, th)) {
                    const unsigned long long key = make_key(th, prim);
                    if (key < best) {
                        const float lim = fmaxf(th, 0.0f) * (1.0f / norm_sq(d)) * 1.00001f + 1e-6f;
                        pb.hit_pos[l1 ? r1.slot : r0.slot] = st4(w);
                        r0.best = l1 ? r0.best : key;
                        r0.limit = l1 ? r0.limit : lim;
                        r1.best = l1 ? key : r1.best;
                        r1.limit = l1 ? lim : r1.limit;
                    }
                }
                r0.next = l1 ? r0.next : kBvh4Absent;
                r1.next = l1 ? kBvh4Absent : r1.next;
            }
        }
        const uint32_t n_unfinished = static_cast<uint32_t>(__popcll(__ballot(pair_unfinished(r0))) + __popcll(__ballot(pair_unfinished(r1))));
        if (n_unfinished == 0 || n_unfinished < keep_unfinished) return;
    }
}

// Can a ray that reaches the box of SDF primitive `prim` end in anything but a miss?  For a crater ball the
// answer is no when the ray never comes within 0.6 of the ball's centre (local units): out there
// CraterBallObject::signedDistance returns |p| - 0.5 > 0.1 (PathTracerDemo.cpp:362-366), so the march
// (include/Object.hpp:135-167) only ever sees values above its 1e-4 hit threshold and none that changes sign
// -- it walks out of the box, a miss.  The test allows 1e-3 for rounding in the ray's points.  Other kinds: yes.
__device__ __forceinline__ bool sdf_prim_can_hit(const SceneDev& sc, uint32_t prim, f3 o, f3 d) {
    const PrimRec* pr = sc.prims + prim;
    if (pr->kind != 3u) return true;
    const f3 lo = aff_pos(pr->inv, o), ld = lin_mul(pr->inv, d);
    const float tc = fmaxf(-dot(lo, ld) / norm_sq(ld), 0.0f);   // closest approach to the centre, not behind the origin
    const f3 c = mk3(fmaf(tc, ld.x, lo.x), fmaf(tc, ld.y, lo.y), fmaf(tc, ld.z, lo.z));
    return !(norm_sq(c) > 0.601f * 0.601f);
}

// Does the ray reach any SDF primitive (its box, and for a crater ball its near field) before its best analytic hit?
// (lo0, hi0 are the corners of sdf_boxes[0], which a kernel loads once for all its rays.)
// `shortcuts` off (MIPT_FULL_MARCH=1): the box alone decides, and the march finds the miss itself.
__device__ __forceinline__ bool reaches_sdf_box(const SceneDev& sc, f3 lo0, f3 hi0, f3 o, f3 d, f3 inv_d, float limit, bool shortcuts) {
    if (sc.n_sdf == 0) return false;
    float tn;
    const float lo[3] = {lo0.x, lo0.y, lo0.z}, hi[3] = {hi0.x, hi0.y, hi0.z};
    bool any = slab(o, inv_d, lo, hi, limit, tn) && (!shortcuts || sdf_prim_can_hit(sc, sc.sdf_boxes[0].prim, o, d));
    for (uint32_t k = 1; k < sc.n_sdf; ++k)
        any = any || (slab(o, inv_d, sc.sdf_boxes[k].lo, sc.sdf_boxes[k].hi, limit, tn) && (!shortcuts || sdf_prim_can_hit(sc, sc.sdf_boxes[k].prim, o, d)));
    return any;
}

// Does the ray reach the box of an object that only the host can evaluate, before its best hit?
__device__ __forceinline__ bool reaches_host_box(const SceneDev& sc, f3 o, f3 inv_d, float limit) {
    bool any = false;
    float tn;
    for (uint32_t k = 0; k < sc.n_host; ++k) any = any || slab(o, inv_d, sc.host_boxes[k].lo, sc.host_boxes[k].hi, limit, tn);
    return any;
}

// Where a ray that goes to the SDF queue starts its first march (SDFObjectCRTP::hitLocalRay include/Object.hpp:135-143:
// the ray in the object's frame, its entry into localBoundingBox unless it starts inside): {entry point, index of the
// SDF box (20 bits) | far-field steps already taken << 20 | 1 << 31 if a march starts there}.  The kernel that finds the ray's closest analytic hit computes it, with
// the ray in registers and -- for camera rays -- all lanes of a wave at it, and leaves it in PathBuffers::march_start
// for k_sdf_march, whose waves would otherwise run these few hundred instructions (three exact divisions for the box
// test, six for the entry point) every round for the handful of lanes that have just fetched a ray.  The boxes are
// tried in order as the march does; at a FractalObject the search stops unprepared, because its march, done where the
// ray is taken up, may bring the ray's limit in.
__device__ __forceinline__ f3 plain_box(uint32_t kind, f3 prm);   // demo_sdf.h
__device__ __forceinline__ float4 sdf_first_march(const SceneDev& sc, f3 o, f3 d, f3 inv_d, float limit) {
    uint32_t k = 0, prepared = 0, iter = 0;
    f3 p0 = mk3(0, 0, 0);
    for (; k < sc.n_sdf; ++k) {
        const SdfBox bx = sc.sdf_boxes[k];
        float tn;
        if (!slab(o, inv_d, bx.lo, bx.hi, limit, tn)) continue;
        const PrimRec* pr = sc.prims + bx.prim;
        if (pr->kind == 4u) break;
        const f3 lo = aff_pos(pr->inv, o);
        const f3 ld = lin_mul(pr->inv, d);
        const f3 box = pr->kind >= 5u ? plain_box(pr->kind, mk3(pr->p[0], pr->p[1], pr->p[2])) : mk3(1.0f, 1.0f, 1.0f);
        p0 = lo;
        if (inside_rect(lo, box) || hit_rect(lo, ld, box, p0)) {
            prepared = 0x80000000u;
            if (pr->kind == 3u) {
                // A crater ball's far-field steps (signedDistance = |p| - 0.5 where that exceeds 0.1,
                // PathTracerDemo.cpp:362-366) are taken here as well, all but the last: the state handed over is the
                // one whose next step is that last far-field step, which sets the march's step length anew -- so the
                // point and the number of steps taken are all there is to hand over.
                f3 p = p0;
                while (iter < 255u) {
                    const float s = norm(p) - 0.5f;
                    if (!(static_cast<double>(s) > 0.1)) break;
                    const f3 q = p + fabsf(s) * ld;
                    if (!inside_rect(q, box)) break;
                    p0 = p;
                    p = q;
                    ++iter;
                }
                if (iter) --iter;   // (p0 is the state before the last step taken)
            }
            break;
        }
    }
    return make_float4(p0.x, p0.y, p0.z, __uint_as_float(k | (iter << 20) | prepared));
}

// Adds a per-lane counter into a 64-bit total with one atomic per wave.
__device__ __forceinline__ void wave_add_total(unsigned long long* total, uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (__lane_id() == 0 && v) atomicAdd(total, static_cast<unsigned long long>(v));
}

// A wave's pending queue entries, gathered in LDS so that one atomic reserves room for hundreds of them:
// atomics on one counter from all over the chip are served one at a time (tens of nanoseconds each),
// which would otherwise bound a kernel whose waves append every iteration.
struct PendingQueue {
    uint32_t* buf;      // `cap` entries of this wave in LDS
    uint32_t cap;       // at least 64
    uint32_t count;     // wave-uniform
};
__device__ __forceinline__ void pending_flush(PendingQueue& q, uint32_t* counter, uint32_t* queue) {
    if (q.count == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t lane = __lane_id();
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(counter, q.count);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < q.count; i += 64u) queue[base + i] = q.buf[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    q.count = 0;
}
// All 64 lanes call; lanes with `pred` append `value`.
__device__ __forceinline__ void pending_push(PendingQueue& q, bool pred, uint32_t value, uint32_t* counter, uint32_t* queue) {
    const unsigned long long m = __ballot(pred);
    if (!m) return;
    if (q.count + 64u > q.cap) pending_flush(q, counter, queue);
    if (pred) q.buf[q.count + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] = value;
    q.count += static_cast<uint32_t>(__popcll(m));
}

// The last flush of a workgroup's four waves with one atomic instead of four: every wave of the chip ends its kernel
// with a flush, all at about the same time and all on one counter, which serves them one after another.  Every
// thread of the workgroup calls, once, after its wave's last push (`s_meet`: 8 words of LDS).
__device__ __forceinline__ void pending_flush_block(PendingQueue& q, uint32_t* counter, uint32_t* queue, uint32_t* s_meet) {
    const uint32_t wave = threadIdx.x >> 6, lane = __lane_id();
    if (lane == 0) s_meet[wave] = q.count;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c0 = s_meet[0], c1 = s_meet[1], c2 = s_meet[2], c3 = s_meet[3], total = c0 + c1 + c2 + c3;
        const uint32_t base = total ? atomicAdd(counter, total) : 0u;
        s_meet[4] = base;
        s_meet[5] = base + c0;
        s_meet[6] = base + c0 + c1;
        s_meet[7] = base + c0 + c1 + c2;
    }
    __syncthreads();
    const uint32_t base = s_meet[4u + wave];
    for (uint32_t i = lane; i < q.count; i += 64u) queue[base + i] = q.buf[i];
    q.count = 0;
}

// A wave's share of a work queue: entries are reserved a chunk at a time, one atomic on the shared
// cursor per chunk (the cursor is served one atomic at a time for the whole chip); chunks shrink towards
// the end of the queue so that the last entries spread over the waves.
struct QueueCursor {
    uint32_t next, end, seen;   // wave-uniform: reserved range, cursor position as last seen
    bool drained;               // the queue has been handed out completely
};
__device__ __forceinline__ bool cursor_empty(const QueueCursor& q) { return q.drained && q.next >= q.end; }
// All 64 lanes call; lanes with `want` get an entry index in `i` (returns whether they did).
__device__ __forceinline__ bool cursor_take(QueueCursor& q, bool want, uint32_t* head, uint32_t n, uint32_t n_waves, uint32_t max_chunk, uint32_t& i) {
    const unsigned long long fm = __ballot(want);
    if (!fm) return false;
    const uint32_t cnt = static_cast<uint32_t>(__popcll(fm));
    const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(fm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(fm), 0u));
    const uint32_t avail = q.end - q.next;
    i = q.next + r;
    bool have = want && r < avail;
    if (avail >= cnt || q.drained) {
        q.next += min(cnt, avail);
    } else {
        const uint32_t need = cnt - avail;
        const uint32_t left = n > q.seen ? n - q.seen : 0u;
        const uint32_t chunk = max(need, min(max_chunk, left / (2u * n_waves)));
        uint32_t base = 0;
        if (__lane_id() == 0) base = atomicAdd(head, chunk);
        base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
        q.seen = base + chunk;
        const uint32_t end = min(base + chunk, max(n, base));
        if (base + chunk >= n) q.drained = true;
        if (want && r >= avail) {
            i = base + (r - avail);
            have = i < end;
        }
        q.next = min(base + need, end);
        q.end = end;
    }
    return have;
}

// The same over sixteen cursors.  Towards the end of a queue the chunks shrink to what a wave's idle lanes need, one
// atomic per wave and round; on a single address the chip serves those one at a time (tens of nanoseconds each), and
// three thousand waves then wait for their turn instead of marching: a fixed millisecond or two per launch, whatever
// its length.  Entries are dealt to the cursors in blocks of 256 (block b belongs to cursor b mod 16, so every cursor
// starts with some of the queue's first entries, and a full chunk is still 256 neighbours of the queue: rays of
// neighbouring pixels, which walk the same craters); a wave starts on the cursor of its workgroup and moves on when
// that one is handed out.  Which wave processes which entry never matters for the result.
struct SplitCursor {
    uint32_t next, end;         // wave-uniform: reserved range, in the current cursor's own numbering
    uint32_t cur, tried;        // cursor in use, cursors found handed out
};
__device__ __forceinline__ bool split_empty(const SplitCursor& q) { return q.tried >= kMarchCursors && q.next >= q.end; }
// All 64 lanes call; lanes with `want` get an entry index in `i` (returns whether they did).
__device__ __forceinline__ bool split_take(SplitCursor& q, bool want, uint32_t (*heads)[32], uint32_t n, uint32_t n_waves, uint32_t max_chunk, uint32_t& i) {
    const unsigned long long fm = __ballot(want);
    if (!fm) return false;
    const uint32_t cnt = static_cast<uint32_t>(__popcll(fm));
    const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(fm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(fm), 0u));
    const uint32_t avail = q.end - q.next;
    uint32_t l = q.next + r, c = q.cur;   // entry l of cursor c
    bool have = want && r < avail;
    if (avail >= cnt) {
        q.next += cnt;
    } else {
        q.next = q.end;   // the reserve is used up
        const uint32_t need = cnt - avail;
        const uint32_t blocks = (n + 255u) >> 8;
        const uint32_t waves_per_cursor = max(1u, n_waves / kMarchCursors);
        // one reservation for the lanes still without an entry, from the first cursor that has anything left
        while (q.tried < kMarchCursors) {
            const uint32_t len = ((blocks + kMarchCursors - 1u - q.cur) / kMarchCursors) << 8;
            // (a plain look first: how much is left sizes the chunk, and a cursor that is handed out costs the waves
            // that come by later no atomic)
            uint32_t base = __hip_atomic_load(&heads[q.cur][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t left = len > base ? len - base : 0u;
            const uint32_t chunk = max(need, min(max_chunk, left / (2u * waves_per_cursor)));
            if (left) {
                if (__lane_id() == 0) base = atomicAdd(&heads[q.cur][0], chunk);
                base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
            }
            if (base < len) {
                const uint32_t end = min(base + chunk, len);
                if (want && r >= avail) {
                    l = base + (r - avail);
                    c = q.cur;
                    have = l < end;
                }
                q.next = min(base + need, end);
                q.end = end;
                break;
            }
            q.cur = (q.cur + 1u) % kMarchCursors;   // handed out: on to the next cursor
            ++q.tried;
            q.next = q.end = 0u;
        }
    }
    i = (((l >> 8) * kMarchCursors + c) << 8) | (l & 255u);
    return have && i < n;
}

// The same with eight cursors instead of one: entries are handed out in blocks of 256, block c through
// cursor c mod 8, and a workgroup starts on cursor blockIdx mod 8 (workgroups go to the XCDs round-robin,
// so the waves of one XCD share a cursor and work on neighbouring blocks).  A wave whose cursor has run
// out moves on to the next one.  This spreads the atomics over eight addresses; which wave processes
// which entry never matters for the result.
struct XcdCursor {
    uint32_t next, end;   // wave-uniform: reserved range
    uint32_t xcd, tried;  // cursor in use, number of cursors found exhausted
};
__device__ __forceinline__ bool xcd_cursor_empty(const XcdCursor& q) { return q.tried >= 8u && q.next >= q.end; }
__device__ __forceinline__ bool xcd_cursor_take(XcdCursor& q, bool want, uint32_t (*heads)[32], uint32_t n, uint32_t& i) {
    const unsigned long long fm = __ballot(want);
    if (!fm) return false;
    const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(fm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(fm), 0u));
    while (q.next >= q.end && q.tried < 8u) {
        // a plain look first: once a cursor is handed out, the thousands of waves that come by to find that out would
        // otherwise queue up on its atomic (one address serves one atomic at a time), a fixed third of a millisecond
        // at the end of every launch
        uint32_t c = __hip_atomic_load(&heads[q.xcd][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((static_cast<unsigned long long>(c) * 8u + q.xcd) * 256u < n) {
            if (__lane_id() == 0) c = atomicAdd(&heads[q.xcd][0], 1u);
            c = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(c)));
        }
        const unsigned long long begin = (static_cast<unsigned long long>(c) * 8u + q.xcd) * 256u;
        if (begin < n) {
            q.next = static_cast<uint32_t>(begin);
            q.end = min(q.next + 256u, n);
        } else {
            q.xcd = (q.xcd + 1u) & 7u;
            ++q.tried;
        }
    }
    i = q.next + r;
    const bool have = want && i < q.end;
    q.next = min(q.next + static_cast<uint32_t>(__popcll(fm)), q.end);
    return have;
}

// =============================================================================================
// generate + first extend, fused: a primary ray lives in registers from the camera to its closest hit.
// A lane owns one pixel and walks a stripe of its samples (blockIdx.y, blockIdx.y + gridDim.y, ...).
// Only paths that hit an analytic primitive or
// reach an SDF box are written to HBM (ray, hit) and queued for the stages that follow; on the demo
// scene that is ~10 % of the paths, the rest cost no memory traffic at all.
// =============================================================================================
template <int STACK, int CAP, bool COUNT>
__global__ void __launch_bounds__(256) k_generate_extend(CameraRec cam, SceneDev sc, BatchParams bp, PathBuffers pb, CullRects cull) {
    if (*pb.cancel_latch) return;   // the frame was cancelled (as of the last latch)
    __shared__ int s_stack[STACK * 256];
    __shared__ uint32_t s_pend[2 * 4 * CAP];
    int* stack = s_stack + threadIdx.x;
    PendingQueue live{s_pend + (threadIdx.x >> 6) * CAP, CAP, 0u}, to_sdf{s_pend + (4u + (threadIdx.x >> 6)) * CAP, CAP, 0u};
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;   // pixel id; the slot of its sample s is s * n_pix + id
    uint32_t x, y;
    const bool in_image = pixel_of(bp, id, x, y);
    const uint32_t pix = y * bp.width + x;                       // what the random numbers are keyed by
    const uint32_t samples = bp.n_slots / bp.n_pix;
    const float sx0 = static_cast<float>(x) / static_cast<float>(bp.width - 1);
    const float sy0 = static_cast<float>(y) / static_cast<float>(bp.height - 1);
    f3 box_lo = mk3(0, 0, 0), box_hi = mk3(0, 0, 0);
    if (sc.n_sdf) {
        box_lo = mk3(sc.sdf_boxes[0].lo[0], sc.sdf_boxes[0].lo[1], sc.sdf_boxes[0].lo[2]);
        box_hi = mk3(sc.sdf_boxes[0].hi[0], sc.sdf_boxes[0].hi[1], sc.sdf_boxes[0].hi[2]);
    }
    uint32_t n_sdf_rays = 0;
    const uint32_t words_per_sample = (bp.n_pix + 63u) / 64u;
    // pixels from which nothing can be seen trace no rays
    bool visible = false;
    if (in_image)
        for (int k = 0; k < cull.n; ++k)
            visible = visible || (static_cast<int>(x) >= cull.r[k][0] && static_cast<int>(y) >= cull.r[k][1] && static_cast<int>(x) <= cull.r[k][2] &&
                                  static_cast<int>(y) <= cull.r[k][3]);
    const bool wave_sees = __ballot(visible) != 0ull;
    if (!wave_sees && __ballot(in_image)) {
        // nothing for this wave to trace: its words of the live-path bitmap are all that is owed
        const uint32_t word = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) >> 6;
        for (uint32_t s = blockIdx.y + (threadIdx.x & 63u) * gridDim.y; s < samples; s += 64u * gridDim.y)
            pb.live_bits[static_cast<size_t>(s) * words_per_sample + word] = 0ull;
    }
    uint32_t s_next = visible ? blockIdx.y : samples;    // this lane's next sample
    uint32_t s_cur = blockIdx.y - gridDim.y;             // wave-uniform (wraps; advanced before use)
    Trav t;
    t.busy = false;
    TravCounts counts{0u, 0u};
    while (wave_sees) {
        // refill: idle lanes start their next sample
        if (!t.busy && s_next < samples) {
            f3 o, d;
            camera_ray(cam, bp, pix, bp.first_sample + s_next, sx0, sy0, o, d);
            trav_begin(t, o, d, s_next * bp.n_pix + id, sc.nodes != nullptr);
            s_next += gridDim.y;
        }
        if (!__ballot(t.busy)) break;
        s_cur += gridDim.y;   // the sample the wave works on now (its lanes move through the samples together)
        // The wave's 64 rays of one sample run to the end together: neighbouring pixels take about equally
        // long, and their queue entries then name 64 consecutive slots, which is what lets the kernels that
        // read the queue later fetch the paths' state in full cache lines.
        trav_run<COUNT>(sc, t, stack, 0u, counts, bp.trav_quorum_primary);
        // finished rays: keep what hit something or reaches an SDF box
        const bool fin = trav_finished(t);
        bool keep = false, wants_sdf = false;
        if (fin) {
            wants_sdf = reaches_sdf_box(sc, box_lo, box_hi, t.o, t.d, t.inv_d, t.limit, !(bp.flags & kFlagFullMarch));
            keep = t.best != kMissKey || wants_sdf || (sc.n_host && reaches_host_box(sc, t.o, t.inv_d, t.limit));
            if (keep) {
                pb.ray_o[t.slot] = st4(t.o);
                pb.ray_d[t.slot] = st4(t.d);
                pb.hit_key[t.slot] = t.best;
                pb.hit_pos[t.slot] = st4(t.best_pos);
                pb.rad[t.slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // stays black unless a hit adds light
            }
            t.busy = false;
        }
        // which of this wave's 64 paths of the sample carry on: one word of the batch's bitmap (written for
        // every wave and sample, so the bitmap needs no clearing)
        {
            const unsigned long long kept = __ballot(keep);
            if ((threadIdx.x & 63u) == 0u) pb.live_bits[static_cast<size_t>(s_cur) * words_per_sample + (id >> 6)] = kept;
        }
        pending_push(live, keep, t.slot, &pb.counters[0].n_live, pb.queue[0]);
        if (wants_sdf) pb.march_start[t.slot] = sdf_first_march(sc, t.o, t.d, t.inv_d, t.limit);
        pending_push(to_sdf, wants_sdf, t.slot, &pb.counters[0].n_sdf, pb.sdf_queue);
        n_sdf_rays += wants_sdf ? 1u : 0u;
    }
    __syncthreads();   // every wave is done with its traversal stack: its first words serve the flush
    uint32_t* s_meet = reinterpret_cast<uint32_t*>(s_stack);
    pending_flush_block(live, &pb.counters[0].n_live, pb.queue[0], s_meet);
    pending_flush_block(to_sdf, &pb.counters[0].n_sdf, pb.sdf_queue, s_meet);
    wave_add_total(&pb.totals->sdf_rays, n_sdf_rays);
    wave_add_total(&pb.totals->primary_traced, visible ? (samples - blockIdx.y + gridDim.y - 1u) / gridDim.y : 0u);
    if (COUNT) {
        wave_add_total(&pb.totals->node_visits, counts.nodes);
        wave_add_total(&pb.totals->prim_tests, counts.prims);
    }
}

// =============================================================================================
// generate alone: the camera rays of every path of the batch go to the path arena and the live-path queue of bounce 0
// names them all, in slot order; the closest-hit kernel of the later bounces (two rays per lane) then traces them like
// any other queue.  For scenes of many analytic primitives that is faster than the fused kernel above, whose one ray
// per lane in lockstep keeps half the lanes busy: the queue's neighbours are neighbouring pixels of one sample, as
// coherent as rays get, and the 56 bytes a path costs in traffic are little beside a traversal of dozens of nodes.
// =============================================================================================
__global__ void __launch_bounds__(256) k_generate(CameraRec cam, BatchParams bp, PathBuffers pb, CullRects cull) {
    if (*pb.cancel_latch) return;
    const uint32_t words_per_sample = bp.n_pix >> 6;   // n_pix is a multiple of 64 (whole 8 x 8 tiles)
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t traced = 0;
    // a wave walks whole groups of 64 slots: 64 pixel ids of one sample, one word of the live-path bitmap
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < bp.n_slots; slot += stride) {
        const uint32_t s = slot / bp.n_pix, id = slot - s * bp.n_pix;
        uint32_t x = 0, y = 0;
        const bool in_image = pixel_of(bp, id, x, y);
        bool visible = false;
        if (in_image)
            for (int k = 0; k < cull.n; ++k)
                visible = visible || (static_cast<int>(x) >= cull.r[k][0] && static_cast<int>(y) >= cull.r[k][1] && static_cast<int>(x) <= cull.r[k][2] &&
                                      static_cast<int>(y) <= cull.r[k][3]);
        // The queue names every slot of the batch, in order (no compaction: no counter to contend for).  A slot without a
        // ray -- tile padding, a pixel from which nothing can be seen -- gets the null direction, which the closest-hit
        // kernels answer with a miss without tracing anything.
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
        if (visible) {
            camera_ray(cam, bp, y * bp.width + x, bp.first_sample + s, static_cast<float>(x) / static_cast<float>(bp.width - 1),
                       static_cast<float>(y) / static_cast<float>(bp.height - 1), o, d);
            pb.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            ++traced;
        }
        pb.ray_o[slot] = st4(o);
        pb.ray_d[slot] = st4(d);
        pb.queue[0][slot] = slot;
        const unsigned long long vis = __ballot(visible);
        if ((threadIdx.x & 63u) == 0u) pb.live_bits[static_cast<size_t>(s) * words_per_sample + (id >> 6)] = vis;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pb.counters[0].n_live = bp.n_slots;
    wave_add_total(&pb.totals->primary_traced, traced);
}

// =============================================================================================
// extend: closest hit for the rays of the live-path queue (bounces after the first, or rays loaded
// from the host).  Persistent waves; a lane that finishes its ray takes the next one from the queue.
// =============================================================================================
template <int STACK, bool COUNT>
__global__ void __launch_bounds__(256) k_extend(SceneDev sc, BatchParams bp, PathBuffers pb, uint32_t bounce) {
    if (*pb.cancel_latch) return;   // the frame was cancelled (as of the last latch)
    __shared__ int s_stack[STACK * 256];
    // one atomic on the SDF queue's counter per `kCap` entries of a wave (that counter is one address for the
    // whole chip): as many as the traversal stacks leave room for
    constexpr uint32_t kCap = STACK <= 8 ? 1024 : STACK <= 24 ? 512 : STACK <= 32 ? 256 : 128;
    __shared__ uint32_t s_pend[4 * kCap];
    PendingQueue to_sdf{s_pend + (threadIdx.x >> 6) * kCap, kCap, 0u};
    const uint32_t n = pb.counters[bounce].n_live;
    // a short queue needs only its first workgroups (one block of 256 entries each); the rest would just
    // queue up on the cursors to learn that nothing is left
    if (static_cast<unsigned long long>(blockIdx.x) * 256u >= n) return;
    const uint32_t* queue = pb.queue[bounce & 1];
    uint32_t (*heads)[32] = pb.counters[bounce].extend_heads;
    int* stack = s_stack + threadIdx.x;
    f3 box_lo = mk3(0, 0, 0), box_hi = mk3(0, 0, 0);
    if (sc.n_sdf) {
        box_lo = mk3(sc.sdf_boxes[0].lo[0], sc.sdf_boxes[0].lo[1], sc.sdf_boxes[0].lo[2]);
        box_hi = mk3(sc.sdf_boxes[0].hi[0], sc.sdf_boxes[0].hi[1], sc.sdf_boxes[0].hi[2]);
    }
    uint32_t n_sdf_rays = 0;
    XcdCursor cursor{0u, 0u, blockIdx.x & 7u, 0u};
    Trav t;
    t.busy = false;
    TravCounts counts{0u, 0u};
    for (;;) {
        // refill idle lanes from the queue
        uint32_t i = 0;
        if (xcd_cursor_take(cursor, !t.busy && !xcd_cursor_empty(cursor), heads, n, i)) {
            const uint32_t slot = queue[i];
            const f3 d = ld3(pb.ray_d[slot]);
            if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) pb.hit_key[slot] = kMissKey;   // a slot without a ray (k_generate)
            else trav_begin(t, ld3(pb.ray_o[slot]), d, slot, sc.nodes != nullptr);
        }
        if (!__ballot(t.busy)) {
            if (xcd_cursor_empty(cursor)) break;
            continue;
        }
        const uint32_t n_busy = static_cast<uint32_t>(__popcll(__ballot(t.busy)));
        trav_run<COUNT>(sc, t, stack, xcd_cursor_empty(cursor) ? 0u : (bp.trav_keep * n_busy) / 64u, counts, bp.trav_quorum);
        const bool fin = trav_finished(t);
        bool wants_sdf = false;
        if (fin) {
            pb.hit_key[t.slot] = t.best;
            pb.hit_pos[t.slot] = st4(t.best_pos);
            wants_sdf = reaches_sdf_box(sc, box_lo, box_hi, t.o, t.d, t.inv_d, t.limit, !(bp.flags & kFlagFullMarch));
            if (wants_sdf) pb.march_start[t.slot] = sdf_first_march(sc, t.o, t.d, t.inv_d, t.limit);
            t.busy = false;
        }
        pending_push(to_sdf, wants_sdf, t.slot, &pb.counters[bounce].n_sdf, pb.sdf_queue);
        n_sdf_rays += wants_sdf ? 1u : 0u;
    }
    __syncthreads();   // every wave is done with its traversal stack: its first words serve the flush
    pending_flush_block(to_sdf, &pb.counters[bounce].n_sdf, pb.sdf_queue, reinterpret_cast<uint32_t*>(s_stack));
    wave_add_total(&pb.totals->sdf_rays, n_sdf_rays);
    if (COUNT) {
        wave_add_total(&pb.totals->node_visits, counts.nodes);
        wave_add_total(&pb.totals->prim_tests, counts.prims);
    }
}

// The same for scenes of a handful of primitives (the demo scene: a crater ball and a slab): no tree, no persistent
// waves, no wave state machine -- a thread takes a ray, tests every analytic primitive in index order and keeps the
// closest hit, ties to the lower index (ObjectTree::Tree::hit, src/ObjectTree.cpp:22-88, literally).  The machinery
// above costs such rays ten times what their memory traffic does.
__global__ void __launch_bounds__(256) k_extend_few(SceneDev sc, BatchParams bp, PathBuffers pb, uint32_t round) {
    if (*pb.cancel_latch) return;
    constexpr uint32_t kCap = 1024;
    __shared__ uint32_t s_pend[4 * kCap];
    __shared__ uint32_t s_meet[8];
    PendingQueue to_sdf{s_pend + (threadIdx.x >> 6) * kCap, kCap, 0u};
    const uint32_t n = pb.counters[round].n_live;
    const uint32_t* queue = pb.queue[round & 1];
    f3 box_lo = mk3(0, 0, 0), box_hi = mk3(0, 0, 0);
    if (sc.n_sdf) {
        box_lo = mk3(sc.sdf_boxes[0].lo[0], sc.sdf_boxes[0].lo[1], sc.sdf_boxes[0].lo[2]);
        box_hi = mk3(sc.sdf_boxes[0].hi[0], sc.sdf_boxes[0].hi[1], sc.sdf_boxes[0].hi[2]);
    }
    uint32_t n_sdf_rays = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;   // every lane of a wave runs every round: the appends are wave-wide
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool wants_sdf = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = queue[i];
            const f3 o = ld3(pb.ray_o[slot]), d = ld3(pb.ray_d[slot]);
            unsigned long long best = kMissKey;
            f3 best_pos = mk3(0.0f, 0.0f, 0.0f);
            float limit = 3.0e38f;
            const bool has_ray = !(d.x == 0.0f && d.y == 0.0f && d.z == 0.0f);   // (the null direction: a slot without a ray, k_generate)
            if (has_ray) {
                for (uint32_t prim = 0; prim < sc.n_prims; ++prim) {
                    if (sc.prims[prim].kind > 2u) continue;
                    f3 w;
                    float th;
                    if (hit_analytic_from(sc.prims, sc.tri_c, prim, o, d, w, th)) {
                        const unsigned long long key = make_key(th, prim);
                        if (key < best) {
                            best = key;
                            best_pos = w;
                            limit = fmaxf(th, 0.0f) * (1.0f / norm_sq(d)) * 1.00001f + 1e-6f;
                        }
                    }
                }
                const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                wants_sdf = reaches_sdf_box(sc, box_lo, box_hi, o, d, inv_d, limit, !(bp.flags & kFlagFullMarch));
                if (wants_sdf) pb.march_start[slot] = sdf_first_march(sc, o, d, inv_d, limit);
            }
            pb.hit_key[slot] = best;
            pb.hit_pos[slot] = st4(best_pos);
        }
        pending_push(to_sdf, wants_sdf, slot, &pb.counters[round].n_sdf, pb.sdf_queue);
        n_sdf_rays += wants_sdf ? 1u : 0u;
    }
    pending_flush_block(to_sdf, &pb.counters[round].n_sdf, pb.sdf_queue, s_meet);
    wave_add_total(&pb.totals->sdf_rays, n_sdf_rays);
}

// The same with two rays per lane (trav_run_pair): the lane's column of S stack entries is shared by both.
template <int S, bool COUNT, bool DIST>
__global__ void __launch_bounds__(256) k_extend_pair(SceneDev sc, BatchParams bp, PathBuffers pb, uint32_t bounce) {
    if (*pb.cancel_latch) return;   // the frame was cancelled (as of the last latch)
    __shared__ int s_stack[S * 256];
    constexpr uint32_t kCap = S <= 32 ? 256 : 128;   // four blocks share a CU's LDS up to S = 38
    __shared__ uint32_t s_pend[4 * kCap];
    PendingQueue to_sdf{s_pend + (threadIdx.x >> 6) * kCap, kCap, 0u};
    const uint32_t n = pb.counters[bounce].n_live;
    if (static_cast<unsigned long long>(blockIdx.x) * 256u >= n) return;
    const uint32_t* queue = pb.queue[bounce & 1];
    uint32_t (*heads)[32] = pb.counters[bounce].extend_heads;
    int* col = s_stack + threadIdx.x;
    const int rows = bp.pair_rows ? min(static_cast<int>(bp.pair_rows), S) : S;
    f3 box_lo = mk3(0, 0, 0), box_hi = mk3(0, 0, 0);
    if (sc.n_sdf) {
        box_lo = mk3(sc.sdf_boxes[0].lo[0], sc.sdf_boxes[0].lo[1], sc.sdf_boxes[0].lo[2]);
        box_hi = mk3(sc.sdf_boxes[0].hi[0], sc.sdf_boxes[0].hi[1], sc.sdf_boxes[0].hi[2]);
    }
    uint32_t n_sdf_rays = 0;
    XcdCursor cursor{0u, 0u, blockIdx.x & 7u, 0u};
    PairRay r0, r1;
    r0.o = r0.d = r0.inv_d = r1.o = r1.d = r1.inv_d = mk3(0, 0, 0);
    r0.limit = r1.limit = 0.0f;
    r0.best = r1.best = kMissKey;
    pair_idle(r0);
    pair_idle(r1);
    TravCounts counts{0u, 0u};
    // a finished ray leaves its result and makes room; an idle slot takes the next ray of the queue
    auto turn_over = [&](PairRay& t) {
        bool wants_sdf = false;
        const uint32_t slot_was = t.slot;
        if (pair_finished(t)) {
            pb.hit_key[t.slot] = t.best;
            if (t.best == kMissKey) pb.hit_pos[t.slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            wants_sdf = reaches_sdf_box(sc, box_lo, box_hi, t.o, t.d, t.inv_d, t.limit, !(bp.flags & kFlagFullMarch));
            if (wants_sdf) pb.march_start[t.slot] = sdf_first_march(sc, t.o, t.d, t.inv_d, t.limit);
            pair_idle(t);
        }
        pending_push(to_sdf, wants_sdf, slot_was, &pb.counters[bounce].n_sdf, pb.sdf_queue);
        n_sdf_rays += wants_sdf ? 1u : 0u;
        uint32_t i = 0;
        if (xcd_cursor_take(cursor, !pair_busy(t) && !xcd_cursor_empty(cursor), heads, n, i)) {
            const uint32_t slot = queue[i];
            const f3 d = ld3(pb.ray_d[slot]);
            if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) pb.hit_key[slot] = kMissKey;   // a slot without a ray (k_generate)
            else pair_begin(t, ld3(pb.ray_o[slot]), d, slot, sc.nodes != nullptr);
        }
    };
    for (;;) {
        turn_over(r0);
        turn_over(r1);
        if (!__ballot(pair_busy(r0) || pair_busy(r1))) {
            if (xcd_cursor_empty(cursor)) break;
            continue;
        }
        const uint32_t n_busy = static_cast<uint32_t>(__popcll(__ballot(pair_busy(r0))) + __popcll(__ballot(pair_busy(r1))));
        trav_run_pair<S, COUNT, DIST>(sc, pb, r0, r1, col, rows, xcd_cursor_empty(cursor) ? 0u : (bp.trav_keep_pair * n_busy) / 128u, counts, bp.trav_quorum_pair);
    }
    __syncthreads();   // every wave is done with its traversal stack: its first words serve the flush
    pending_flush_block(to_sdf, &pb.counters[bounce].n_sdf, pb.sdf_queue, reinterpret_cast<uint32_t*>(s_stack));
    wave_add_total(&pb.totals->sdf_rays, n_sdf_rays);
    if (COUNT) {
        wave_add_total(&pb.totals->node_visits, counts.nodes);
        wave_add_total(&pb.totals->prim_tests, counts.prims);
        wave_add_total(&pb.totals->trav_restarts, counts.restarts);   // per lane
        if (__lane_id() == 0) {
            atomicAdd(&pb.totals->trav_rounds, static_cast<unsigned long long>(counts.rounds));
            atomicAdd(&pb.totals->trav_leaf_steps, static_cast<unsigned long long>(counts.leaf_steps));
        }
    }
}

#include "craterball.h"
#include "demo_sdf.h"

// =============================================================================================
// sdf_march: SDFObjectCRTP::hitLocalRay (include/Object.hpp:135-167) as a per-lane state machine.
// Persistent waves pull rays from the SDF queue; a lane that finishes pulls the next ray, so the
// expensive near-field evaluations always run with as many lanes as the queue can supply.  Each
// round first lets every lane take its cheap far-field steps (sdf = |p| - 0.5 when that exceeds 0.1),
// then all lanes that still need one evaluate the near field together.
// =============================================================================================
struct March {
    f3 p, dir;          // local frame
    float d;
    uint32_t iter;
    bool first, sign;
    // an earlier state (p, d) of this march and the step counts of Brent's cycle search, see march_step
    f3 seen_p;
    float seen_d;
    uint32_t since, span;
};
__device__ __forceinline__ void march_begin(March& m, f3 p, f3 dir) {
    m.p = p;
    m.dir = dir;
    m.iter = 0;
    m.first = true;
    m.seen_p = p;
    m.seen_d = __uint_as_float(0x7fc00000u);   // NaN bits: equal to no d
    m.since = 0;
    m.span = 1;
}
__device__ __forceinline__ bool same_bits(f3 a, f3 b) {
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) && __float_as_uint(a.z) == __float_as_uint(b.z);
}

// One iteration of the reference's loop body given the SDF value d2 at m.p.  Returns 0 to continue,
// 1 for a hit (m.p is the hit), 2 for a miss.  The reference evaluates the SDF once before the loop
// and again, at the same point, in iteration 0; both values are d2 here.
// `box` is the half extent of the object's localBoundingBox (centred on the origin).
//
// The loop's state is (p, d): the next state is a function of it alone (the SDF is a pure function of p,
// sign and direction are fixed).  A march that can no longer end is therefore cut short, with the
// result the reference reaches after running out its 256 iterations -- a miss:
//   * the backing-up branch leaves p unchanged (d has become too small to move it): the same value comes
//     back, the branch is taken again with a smaller d, for ever;
//   * a state comes back bit for bit: the march is on a cycle none of whose states ended it (found by
//     Brent's search: remember the state at step counts 1, 2, 4, ... and compare every later one with it).
//     Far-field steps (`search` off) stay out of the search: a sequence that repeats still repeats without them,
//     and so the search's state does not have to travel with a march whose far-field steps another kernel takes.
// Where the SDF jumps across zero (crater rims, foam walls) the bisection cannot find a root, and rays
// that start on the surface do this often: 2.8 % of them used to run all 256 iterations, 31 % of all
// evaluations of those launches.  `shortcuts` off (MIPT_FULL_MARCH=1) runs every iteration, for the tests.
__device__ __forceinline__ int march_step(March& m, float d2, f3 box, bool shortcuts, bool search = true) {
    if (m.first) {
        m.d = d2;
        m.sign = d2 > 0.0f;
        m.first = false;
    }
    if ((d2 > 0.0f) != m.sign) {
        m.d *= 0.5f;
        const f3 q = m.p - m.d * m.dir;
        if (shortcuts && same_bits(q, m.p)) return 2;
        m.p = q;
    } else {
        m.d = d2;
        if (fabsf(m.d) < 1e-4f) return 1;
        m.p = m.p + fabsf(m.d) * m.dir;
        if (!inside_rect(m.p, box)) return 2;
    }
    if (shortcuts && search) {
        if (same_bits(m.p, m.seen_p) && __float_as_uint(m.d) == __float_as_uint(m.seen_d)) return 2;
        if (++m.since == m.span) {
            m.seen_p = m.p;
            m.seen_d = m.d;
            m.span <<= 1;
            m.since = 0;
        }
    }
    return ++m.iter >= 256u ? 2 : 0;
}

// =============================================================================================
// The SDF queue in the order of the rays' first query cell.  Camera rays reach the march in screen order: the 64 rays of
// a wave look at the same craters and the per-lane evaluator's loop, which runs for the longest crater list of the wave,
// wastes a tenth of its passes.  Rays of later bounces come in the order their paths survived, a wave's lists have
// nothing to do with one another and the loop runs 18.6 passes for lists of 11 (measured).  A counting sort by the
// lattice cell the ray's first near-field query falls into (which names the list) puts like with like again: keys, a
// histogram, its prefix sums, a scatter by tiles.  The order inside a key's run is whatever the atomics make it; no
// result depends on the order of a queue.  Keys go to the live-path queue that the round's shading will fill later,
// the sorted slots to the SDF-shade queue, which is empty until then.
// =============================================================================================
constexpr uint32_t kSortOther = kCellCube * kCellCube * kCellCube;   // rays whose first march is not on a crater ball, or not set up
__global__ void __launch_bounds__(256) k_sort_keys(SceneDev sc, PathBuffers pb, uint32_t round) {
    if (*pb.cancel_latch) return;
    __shared__ uint32_t s_hist[kSortKeys];
    for (uint32_t t = threadIdx.x; t < kSortKeys; t += blockDim.x) s_hist[t] = 0u;
    __syncthreads();
    const uint32_t n = pb.counters[round].n_sdf;
    uint32_t* keys = pb.queue[(round + 1u) & 1u];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 start = pb.march_start[pb.sdf_queue[i]];
        const uint32_t w = __float_as_uint(start.w);
        uint32_t key = kSortOther;
        if (w & 0x80000000u) {
            const PrimRec* pr = sc.prims + sc.sdf_boxes[w & 0xfffffu].prim;
            if (pr->kind == 3u) {
                const f3 off = mk3(pr->p[0], pr->p[1], pr->p[2]);
                f3 u, query;
                near_frame(mk3(start.x, start.y, start.z), off, u, query);
                key = crater_cell_index(query, off);
            }
        }
        keys[i] = key;
        atomicAdd(&s_hist[key], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < kSortKeys; t += blockDim.x)
        if (s_hist[t]) atomicAdd(&pb.counters[round].sort_hist[t], s_hist[t]);
}
__global__ void __launch_bounds__(1024) k_sort_scan(PathBuffers pb, uint32_t round) {
    constexpr uint32_t kPer = kSortKeys / 1024u;   // consecutive keys per thread
    __shared__ uint32_t s_sum[1024];
    const uint32_t t = threadIdx.x;
    uint32_t mine[kPer], total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
        mine[k] = pb.counters[round].sort_hist[t * kPer + k];
        total += mine[k];
    }
    s_sum[t] = total;
    __syncthreads();
    for (uint32_t step = 1; step < 1024u; step <<= 1) {
        const uint32_t add = t >= step ? s_sum[t - step] : 0u;
        __syncthreads();
        s_sum[t] += add;
        __syncthreads();
    }
    uint32_t at = s_sum[t] - total;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
        pb.counters[round].sort_cursor[t * kPer + k] = at;
        at += mine[k];
    }
}
constexpr uint32_t kSortTile = 8192;   // entries a block ranks at a time: one atomic per key and tile on the global cursors
__global__ void __launch_bounds__(256) k_sort_scatter(PathBuffers pb, uint32_t round) {
    if (*pb.cancel_latch) return;
    __shared__ uint32_t s_count[kSortKeys], s_base[kSortKeys];
    const uint32_t n = pb.counters[round].n_sdf;
    const uint32_t* keys = pb.queue[(round + 1u) & 1u];
    uint32_t* sorted = pb.sdf_shade_queue;
    for (uint32_t tile = blockIdx.x * kSortTile; tile < n; tile += gridDim.x * kSortTile) {
        for (uint32_t t = threadIdx.x; t < kSortKeys; t += blockDim.x) s_count[t] = 0u;
        __syncthreads();
        uint32_t rank[kSortTile / 256], key[kSortTile / 256];
#pragma unroll
        for (uint32_t e = 0; e < kSortTile / 256; ++e) {
            const uint32_t i = tile + e * 256u + threadIdx.x;
            key[e] = i < n ? keys[i] : 0u;
            rank[e] = i < n ? atomicAdd(&s_count[key[e]], 1u) : 0u;
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < kSortKeys; t += blockDim.x)
            if (s_count[t]) s_base[t] = atomicAdd(&pb.counters[round].sort_cursor[t], s_count[t]);
        __syncthreads();
#pragma unroll
        for (uint32_t e = 0; e < kSortTile / 256; ++e) {
            const uint32_t i = tile + e * 256u + threadIdx.x;
            if (i < n) sorted[s_base[key[e]] + rank[e]] = pb.sdf_queue[i];
        }
        __syncthreads();
    }
}

// PLAIN: the scene holds SDF kinds that are evaluated one point per lane (RoughBoxObject,
// RoughSphereObject, WeirdWallObject); scenes without them run the instantiation that does not carry
// that code.
template <bool PLAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PLAIN ? 2 : 3, PLAIN ? 2 : 3))) k_sdf_march(SceneDev sc, BatchParams bp, PathBuffers pb, uint32_t round, const uint32_t* sdf_queue) {
    if (*pb.cancel_latch) return;   // the frame was cancelled (as of the last latch)
    // Work: the marches the previous round's launch handed on (they come first), then this round's rays.
    const uint32_t n_in = (round > 0u && pb.carry[0]) ? min(pb.counters[round - 1u].n_carry, bp.carry_cap) : 0u;
    const uint32_t n = n_in + pb.counters[round].n_sdf;
    if (n == 0) return;
    const CarryRec* carry_in = pb.carry[(round & 1u) ^ 1u];
    CarryRec* carry_out = pb.carry[round & 1u];
    uint32_t (*heads)[32] = pb.counters[round].sdf_heads;
    constexpr int IDLE = 0, SETUP = 1, MARCH = 2;
    int state = IDLE;
    SplitCursor cursor{0u, 0u, blockIdx.x % kMarchCursors, 0u};
    // Per lane, between rounds: the ray's slot, the SDF primitive it is on, its best hit so far and the march.  The ray
    // itself is read from the slot where it is needed (set-up, the end of a march): registers are what limits this
    // kernel to three waves per SIMD.
    uint32_t slot = 0, k = 0, crater_base = 0, crater_count = 0;
    f3 off = mk3(0, 0, 0);
    unsigned long long best = kMissKey;
    bool improved = false;
    // 0: a ray of this round's queue.  Otherwise the ray came in as a handed-on march (shading has passed it by and must
    // see it again): record index + 1, with the top bit set until the march's state has been taken from the record.
    uint32_t resumed = 0;
    uint32_t plain = 0;   // kind of the lane-evaluated object being marched, 0 for a crater ball
    const f3 ball_box = mk3(1.0f, 1.0f, 1.0f);   // CraterBallObject::localBoundingBox PathTracerDemo.cpp:474-479
    uint32_t c_far = 0, c_near = 0, c_noise = 0;
    __shared__ __attribute__((aligned(16))) float s_coop[4 * kNearLdsFloats];
    __shared__ uint32_t s_late[4 * 128];
    PendingQueue late{s_late + (threadIdx.x >> 6) * 128u, 128u, 0u};
    const uint32_t lane = __lane_id();
    float* lds = s_coop + (threadIdx.x >> 6) * kNearLdsFloats;
    const CoopLane cl = make_coop_lane(sc.noise_oct, lane);
    CraterTableRegs crater_regs = no_crater_table();
    March m;
    march_begin(m, mk3(0, 0, 0), mk3(0, 0, 0));
    m.d = 0.0f;
    m.first = m.sign = false;
    const bool shortcuts = !(bp.flags & kFlagFullMarch);
    uint32_t dry_rounds = 0;   // rounds since this wave found the queue handed out
    uint32_t n_late = 0;       // handed-on rays this wave has finished (wave-uniform)

    // The ray in `slot` is done with every SDF primitive: its closest hit goes back to the slot (the position went there
    // when the hit was found).  A ray that came in as a handed-on march joins this round's live paths, after the
    // closest-hit stage and ahead of shading.
    auto ray_done = [&]() {
        if (improved) pb.hit_key[slot] = best;
        state = IDLE;
    };

    // When rays are scarce, spread them over the waves instead of letting the first waves take 64 each:
    // a wave serves its lanes' near-field evaluations one after another, so rays per wave is latency.
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t quota = min(64u, (n + n_waves - 1) / n_waves);
    for (;;) {
        // (1) idle lanes fetch a ray.  The wave reserves queue entries a chunk at a time -- one atomic on the
        // shared cursor per chunk, not per lane: the cursor is served one atomic at a time for the whole
        // chip.  Chunks shrink towards the end of the queue so that the last rays spread over the waves.
        uint32_t i = 0;
        const bool have = split_take(cursor, state == IDLE && lane < quota && !split_empty(cursor), heads, n, n_waves, 256u, i);
        bool late_now = false;
        uint32_t late_slot = 0;
        {
            if (have) {
                const bool carried = i < n_in;
                uint4 rec0 = make_uint4(0u, 0u, 0u, 0u);   // slot, primitive, best hit key
                float4 start = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (carried) {
                    rec0 = *reinterpret_cast<const uint4*>(carry_in + i);
                    slot = rec0.x;
                    k = rec0.y;
                } else {
                    slot = sdf_queue[i - n_in];
                    start = pb.march_start[slot];
                    k = __float_as_uint(start.w) & 0xfffffu;
                }
                best = carried ? ((static_cast<unsigned long long>(rec0.w) << 32) | rec0.z) : pb.hit_key[slot];
                // A handed-on ray's slot holds kCarriedKey, so its key is always written back.  Its march goes on on the
                // primitive it was on (the state follows from the record below).
                improved = carried;
                resumed = carried ? (i + 1u) | 0x80000000u : 0u;
                state = SETUP;
                if (carried || (__float_as_uint(start.w) & 0x80000000u)) {
                    // the march's start is known (k_march_setup): what is left of the set-up is the object's parameters
                    const PrimRec* pr = sc.prims + sc.sdf_boxes[k].prim;
                    off = mk3(pr->p[0], pr->p[1], pr->p[2]);
                    crater_base = __float_as_uint(pr->q[0]);
                    crater_count = __float_as_uint(pr->q[1]);
                    if (PLAIN) plain = pr->kind >= 5u ? pr->kind : 0u;
                    march_begin(m, mk3(start.x, start.y, start.z), lin_mul(pr->inv, ld3(pb.ray_d[slot])));
                    // far-field steps the set-up took: they count as this kernel's, and the first step is behind the march
                    m.iter = (__float_as_uint(start.w) >> 20) & 0xffu;
                    m.first = m.iter == 0u;
                    m.sign = true;
                    c_far += m.iter;
                    state = MARCH;
                }
            }
        }
        // (2) set up the march on the next SDF primitive whose box the ray reaches
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
        float inv_len2 = 0.0f, limit = 3.0e38f;
        if (state == SETUP) {
            o = ld3(pb.ray_o[slot]);
            d = ld3(pb.ray_d[slot]);
            inv_len2 = 1.0f / norm_sq(d);
            if (best != kMissKey) {
                // recover t from the key to cull boxes behind the best hit
                const uint32_t ob = static_cast<uint32_t>(best >> 32);
                const uint32_t fb = (ob & 0x80000000u) ? (ob ^ 0x80000000u) : ~ob;
                limit = fmaxf(__uint_as_float(fb), 0.0f) * inv_len2 * 1.00001f + 1e-6f;
            }
        }
        while (state == SETUP) {
            if (k == sc.n_sdf) {
                ray_done();
                late_now = resumed != 0u;
                late_slot = slot;
                break;
            }
            const SdfBox bx = sc.sdf_boxes[k];
            float tn;
            if (slab(o, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), bx.lo, bx.hi, limit, tn)) {
                const uint32_t prim = bx.prim;
                const PrimRec* pr = sc.prims + prim;
                const f3 lo = aff_pos(pr->inv, o);
                const f3 ld = lin_mul(pr->inv, d);
                if (pr->kind == 4u) {
                    // FractalObject: the whole march here and now; p[0..5] hold its world box
                    f3 hp;
                    if (fractal_march(lo, ld, mk3(pr->p[0], pr->p[1], pr->p[2]), mk3(pr->q[0], pr->q[1], pr->q[2]), hp)) {
                        const f3 w = aff_pos(pr->fwd, hp);
                        const float t = dot(w - o, d);
                        const unsigned long long key = make_key(t, prim);
                        if (key < best) {
                            best = key;
                            pb.hit_pos[slot] = st4(w);
                            improved = true;
                            limit = fmaxf(t, 0.0f) * inv_len2 * 1.00001f + 1e-6f;
                        }
                    }
                    ++k;
                    continue;
                }
                off = mk3(pr->p[0], pr->p[1], pr->p[2]);
                crater_base = __float_as_uint(pr->q[0]);
                crater_count = __float_as_uint(pr->q[1]);
                f3 box = ball_box;
                if (PLAIN) {
                    plain = pr->kind >= 5u ? pr->kind : 0u;
                    if (plain) box = plain_box(plain, off);   // `off` holds the object's first three parameters
                }
                bool ok = true;
                f3 p0 = lo;
                if (!inside_rect(lo, box)) ok = hit_rect(lo, ld, box, p0);   // intersect(ray, bb), box centred at 0
                if (ok) {
                    march_begin(m, p0, ld);
                    state = MARCH;
                    break;
                }
            }
            ++k;
        }
        if (resumed & 0x80000000u) {
            resumed &= 0x7fffffffu;
            if (state == MARCH) {
                const CarryRec* rec = carry_in + (resumed - 1u);
                const float4 a = reinterpret_cast<const float4*>(rec)[1], b = reinterpret_cast<const float4*>(rec)[2];
                const uint4 c = reinterpret_cast<const uint4*>(rec)[3];
                m.p = mk3(a.x, a.y, a.z);
                m.d = a.w;
                m.seen_p = mk3(b.x, b.y, b.z);
                m.seen_d = b.w;
                m.iter = c.x & 0xffffu;
                m.first = (c.x >> 16) & 1u;
                m.sign = (c.x >> 17) & 1u;
                m.since = c.y;
                m.span = c.z;
            }
        }
        if (n_in) {
            pending_push(late, late_now, late_slot, &pb.counters[round].n_live, pb.queue[round & 1u]);
            n_late += static_cast<uint32_t>(__popcll(__ballot(late_now)));
        }
        // leave only when no lane holds a ray and no lane may fetch another one
        if (__all(state == IDLE) && split_empty(cursor)) break;
        // (3) far-field steps are a handful of flops: take them until a near-field evaluation is due
        int result = 0;
        float s = 0.0f;
        if (PLAIN && state == MARCH && plain) {
            // one step of a lane-evaluated object
            ++c_near;
            result = march_step(m, plain_sdf(plain, off, sc.noise_raw, m.p), plain_box(plain, off), shortcuts);
        } else if (state == MARCH) {
            for (;;) {
                s = norm(m.p) - 0.5f;
                if (!(static_cast<double>(s) > 0.1)) break;   // `sdf_sphere > 0.1` compares against a double literal
                ++c_far;
                result = march_step(m, s, ball_box, shortcuts, false);
                if (result) break;
            }
        }
        // (4) near-field evaluations for the crater-ball lanes that need one
        const bool wants = state == MARCH && result == 0 && !(PLAIN && plain);
        float d2 = 0.0f;
        wave_near(sc.craters, sc.crater_masks, static_cast<const float4*>(sc.minor_cells), crater_regs, cl, lds, lane, bp.lane_threshold, wants, m.p, off, s, crater_base, crater_count, d2, c_near, c_noise);
        if (wants) result = march_step(m, d2, ball_box, shortcuts);
        // (5) finished marches fold into the ray's best hit and move on to the next primitive
        late_now = false;
        if (state == MARCH && result != 0) {
            if (result == 1) {
                const uint32_t prim = sc.sdf_boxes[k].prim;
                const PrimRec* pr = sc.prims + prim;
                const f3 w = aff_pos(pr->fwd, m.p);
                const float t = dot(w - ld3(pb.ray_o[slot]), ld3(pb.ray_d[slot]));
                const unsigned long long key = make_key(t, prim);
                if (key < best) {
                    best = key;
                    pb.hit_pos[slot] = st4(w);
                    improved = true;
                }
            }
            ++k;
            state = SETUP;
            // after the last SDF primitive the ray is done here and now, so that the lane fetches its next ray at the
            // top of the next round instead of spending that round finding out
            if (k == sc.n_sdf) {
                ray_done();
                late_now = resumed != 0u;
                late_slot = slot;
            }
        }
        if (n_in) {
            pending_push(late, late_now, late_slot, &pb.counters[round].n_live, pb.queue[round & 1u]);
            n_late += static_cast<uint32_t>(__popcll(__ballot(late_now)));
        }
        // (6) The queue is handed out and this wave has gone on for a while: what is left are the long marches, a few
        // lanes of a wave that pays a whole round of evaluations per step while most of the chip has nothing to do.
        // They go to the next round's launch, which takes them up first and walks them in full waves beside its own
        // rays.  The path falls one round behind (its bounce number is the round minus PathBuffers::ray_d.w).
        if (bp.carry_max_lag && split_empty(cursor) && ++dry_rounds > bp.carry_after) {
            // (the rounds the path has fallen behind so far travel in the w of its direction)
            float4 dv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (state == MARCH) dv = pb.ray_d[slot];
            const uint32_t lag = __float_as_uint(dv.w);
            const bool may = state == MARCH && lag < bp.carry_max_lag;
            const unsigned long long cm = __ballot(may);
            if (cm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&pb.counters[round].n_carry, static_cast<uint32_t>(__popcll(cm)));
                base = __shfl(base, 0);
                const uint32_t at = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(cm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(cm), 0u));
                if (may && at < bp.carry_cap) {
                    CarryRec rec;
                    rec.slot = slot;
                    rec.k = k;
                    rec.best_lo = static_cast<uint32_t>(best);
                    rec.best_hi = static_cast<uint32_t>(best >> 32);
                    rec.p[0] = m.p.x; rec.p[1] = m.p.y; rec.p[2] = m.p.z;
                    rec.d = m.d;
                    rec.seen_p[0] = m.seen_p.x; rec.seen_p[1] = m.seen_p.y; rec.seen_p[2] = m.seen_p.z;
                    rec.seen_d = m.seen_d;
                    rec.iter_flags = m.iter | (m.first ? 1u << 16 : 0u) | (m.sign ? 1u << 17 : 0u);
                    rec.since = m.since;
                    rec.span = m.span;
                    rec.pad = 0u;
                    carry_out[at] = rec;
                    pb.hit_key[slot] = kCarriedKey;
                    dv.w = __uint_as_float(lag + 1u);
                    pb.ray_d[slot] = dv;
                    state = IDLE;
                }
            }
        }
    }
    if (n_in) pending_flush(late, &pb.counters[round].n_live, pb.queue[round & 1u]);
    if (n_late && lane == 0) atomicAdd(&pb.counters[round].n_late, n_late);
    wave_add_total(&pb.totals->sdf_far_evals, c_far);
    wave_add_total(&pb.totals->sdf_near_evals, c_near);
    wave_add_total(&pb.totals->sdf_noise_calls, c_noise);
}

// =============================================================================================
// Parity probe: the signed distance of SDF primitive `prim` at n points of its local frame, through the very
// evaluators the march uses (a wave of 64 points asks together, so the per-lane evaluator serves full waves and
// the cooperative one the last, short wave -- or every wave, as the lane threshold says).
// =============================================================================================
__global__ void __launch_bounds__(256) k_sdf_eval(SceneDev sc, uint32_t prim, uint32_t lane_threshold, const float* pos3, uint32_t n, float* out) {
    __shared__ __attribute__((aligned(16))) float s_coop[4 * kNearLdsFloats];
    const uint32_t lane = __lane_id();
    float* lds = s_coop + (threadIdx.x >> 6) * kNearLdsFloats;
    const CoopLane cl = make_coop_lane(sc.noise_oct, lane);
    CraterTableRegs crater_regs = no_crater_table();
    const PrimRec* pr = sc.prims + prim;
    const uint32_t kind = pr->kind;
    const f3 prm = mk3(pr->p[0], pr->p[1], pr->p[2]);
    const uint32_t crater_base = __float_as_uint(pr->q[0]), crater_count = __float_as_uint(pr->q[1]);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;   // every lane of a wave runs every round: wave_near is wave-wide
    uint32_t c_near = 0, c_noise = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = i < n;
        const f3 p = valid ? mk3(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]) : mk3(1.0f, 0.0f, 0.0f);
        float v = 0.0f, s = 0.0f;
        bool wants = false;
        if (kind == 3u) {
            s = norm(p) - 0.5f;
            v = s;
            wants = valid && !(static_cast<double>(s) > 0.1);   // the far field returns the sphere distance itself
        } else if (kind == 4u) {
            v = fractal_sdf(p);
        } else if (valid) {
            v = plain_sdf(kind, prm, sc.noise_raw, p);
        }
        if (kind == 3u) wave_near(sc.craters, sc.crater_masks, static_cast<const float4*>(sc.minor_cells), crater_regs, cl, lds, lane, lane_threshold, wants, p, prm, s, crater_base, crater_count, v, c_near, c_noise);
        if (valid) out[i] = v;
    }
}

// =============================================================================================
// shade: Scene::trace's per-hit body (src/Scene.cpp:46-75) around Object::deflectRay
// (src/Object.cpp:35-44) and BasicMaterial::deflect.  Survivors are appended to the next bounce's
// queue through the wave's pending buffer.
// =============================================================================================
// Applies one bounce to the path in `slot`; returns whether the path continues.
// `bounce` is the path's own bounce number: the round of the batch minus `lag`, the rounds the path has fallen behind
// (k_sdf_march); the lag travels on in the w of the new direction.
__device__ __forceinline__ bool apply_bounce(const PathBuffers& pb, uint32_t slot, const PrimRegs& pr, f3 world_pos, const Bounce& b,
                                             uint32_t bounce, uint32_t max_bounces, uint32_t lag) {
    const f3 wdir = lin_mul(pr.fwd, b.dir);   // only the direction returns to world space (src/Object.cpp:42)
    const f3 npos = mk3(world_pos.x + 1e-3f * wdir.x, world_pos.y + 1e-3f * wdir.y, world_pos.z + 1e-3f * wdir.z);   // src/Scene.cpp:54
    // a path starts with attenuation 1 and no radiance (src/Scene.cpp:19-20); neither is stored before the first hit
    float4 att = bounce == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : pb.att[slot];
    // radiance += emitted * attenuation (src/Scene.cpp:56-58).  Most surfaces emit nothing: the sum then keeps its bits
    // (x + 0 * a = x for finite a; the sum starts at +0 and never becomes -0), and its 32 bytes of traffic are left out.
    const bool lit = b.emitted.x != 0.0f || b.emitted.y != 0.0f || b.emitted.z != 0.0f;
    const float att_sum = att.x + att.y + att.z;
    if (lit || !(fabsf(att_sum) <= 3.402823466e38f)) {
        float4 rad = bounce == 0 ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : pb.rad[slot];
        rad.x += b.emitted.x * att.x;
        rad.y += b.emitted.y * att.y;
        rad.z += b.emitted.z * att.z;
        pb.rad[slot] = rad;
    }
    att.x *= b.attenuation.x;
    att.y *= b.attenuation.y;
    att.z *= b.attenuation.z;
    const bool alive = !(att.x + att.z + att.y < 1e-6f) && (bounce + 1 < max_bounces);   // r + b + g, src/Scene.cpp:73
    if (alive) {
        pb.att[slot] = att;
        pb.ray_o[slot] = st4(npos);
        pb.ray_d[slot] = st4(wdir, __uint_as_float(lag));
    }
    return alive;
}

__global__ void __launch_bounds__(256) k_shade(SceneDev sc, BatchParams bp, PathBuffers pb, uint32_t round, uint32_t max_bounces) {
    if (*pb.cancel_latch) return;   // the frame was cancelled (as of the last latch)
    constexpr uint32_t kCap = 1024;
    __shared__ uint32_t s_pend[2 * 4 * kCap];
    PendingQueue survivors{s_pend + (threadIdx.x >> 6) * kCap, kCap, 0u}, sdf_hits{s_pend + (4u + (threadIdx.x >> 6)) * kCap, kCap, 0u};
    __shared__ uint32_t s_host[4 * 128];
    PendingQueue host_hits{s_host + (threadIdx.x >> 6) * 128u, 128u, 0u};
    const uint32_t n = pb.counters[round].n_live;
    const uint32_t* queue = pb.queue[round & 1];
    uint32_t* next_queue = pb.queue[(round + 1) & 1];
    const uint32_t stride = gridDim.x * blockDim.x;
    // all lanes of a wave run the same number of iterations so that the wave-wide ballots line up
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool alive = false, to_sdf = false, to_host = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = queue[i];
            const unsigned long long key = pb.hit_key[slot];
            if (key != kMissKey && key != kCarriedKey) {   // a path whose march was handed on is shaded in a later round
                const uint32_t prim = static_cast<uint32_t>(key & 0xffffffffu);
                PrimRegs pr;
                load_prim(sc.prims, prim, pr);
                if (pr.kind == 8u) {   // an object only the host can evaluate: its deflectRay runs there
                    to_host = true;
                } else if (pr.kind >= 3u) {   // SDF objects: normals need SDF evaluations
                    to_sdf = true;
                } else {
                    const f3 wp = ld3(pb.hit_pos[slot]);
                    const float4 dv = pb.ray_d[slot];
                    const f3 d = ld3(dv);
                    const uint32_t lag = __float_as_uint(dv.w), bounce = round - lag;
                    const f3 ld = lin_mul(pr.inv, d);
                    const f3 lp = aff_pos(pr.inv, wp);
                    f3 nrm;
                    if (pr.kind == 1u) {
                        nrm = rect_normal(lp, mk3(pr.p[0], pr.p[1], pr.p[2]));
                    } else {
                        f3 raw = lp;   // Sphere::normal src/Geometry.cpp:43-45
                        if (pr.kind == 2u) {   // Triangle::normal src/Geometry.cpp:18-20
                            const f3 a = mk3(pr.p[0], pr.p[1], pr.p[2]), b = mk3(pr.p[3], pr.p[4], pr.p[5]);
                            const f3 c = mk3(sc.tri_c[3 * prim], sc.tri_c[3 * prim + 1], sc.tri_c[3 * prim + 2]);
                            raw = cross(b - a, c - a);
                        }
                        nrm = unit(raw);   // one normalisation for the lanes of either kind
                    }
                    const uint32_t s_local = slot / bp.n_pix;
                    uint32_t px, py;
                    pixel_of(bp, slot - s_local * bp.n_pix, px, py);
                    Rng rng(bp.seed, py * bp.width + px, bp.first_sample + s_local, 1u + bounce);
                    const Bounce b = deflect(sc.materials[pr.material], ld, nrm, rng);
                    alive = apply_bounce(pb, slot, pr, wp, b, bounce, max_bounces, lag);
                }
            }
        }
        pending_push(survivors, alive, slot, &pb.counters[round + 1].n_live, next_queue);
        pending_push(sdf_hits, to_sdf, slot, &pb.counters[round].n_sdf_shade, pb.sdf_shade_queue);
        if (sc.n_host) pending_push(host_hits, to_host, slot, &pb.counters[round].n_host_shade, pb.host_queue);
    }
    __shared__ uint32_t s_meet[8];
    pending_flush_block(survivors, &pb.counters[round + 1].n_live, next_queue, s_meet);
    pending_flush_block(sdf_hits, &pb.counters[round].n_sdf_shade, pb.sdf_shade_queue, s_meet);
    if (sc.n_host) pending_flush(host_hits, &pb.counters[round].n_host_shade, pb.host_queue);
}

// =============================================================================================
// Objects without a device form (kind 8: a user's Object subclass, include/Object.hpp:45-52).  The device finds
// the rays that reach such an object's padded world box before their best hit; the host calls the object's hitRay
// for them, the closer hits are merged by hit key; hits that end up on such an object are shaded by the host's
// deflectRay and applied here.  Slow by construction (two host round trips per bounce), and exactly the
// reference's semantics: closest of all objects, ties to the lower index.
// =============================================================================================
__global__ void __launch_bounds__(256) k_host_gather(SceneDev sc, PathBuffers pb, uint32_t bounce, uint32_t which) {
    if (*pb.cancel_latch) return;
    __shared__ uint32_t s_pend[4 * 512];
    PendingQueue out{s_pend + (threadIdx.x >> 6) * 512u, 512u, 0u};
    const uint32_t n = pb.counters[bounce].n_live;
    const uint32_t* queue = pb.queue[bounce & 1];
    const SdfBox bx = sc.host_boxes[which];
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool reach = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = queue[i];
            const f3 o = ld3(pb.ray_o[slot]), d = ld3(pb.ray_d[slot]);
            const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            float limit = 3.0e38f, tn;
            const unsigned long long best = pb.hit_key[slot];
            if (best != kMissKey) {
                const uint32_t ob = static_cast<uint32_t>(best >> 32);
                const uint32_t fb = (ob & 0x80000000u) ? (ob ^ 0x80000000u) : ~ob;
                limit = fmaxf(__uint_as_float(fb), 0.0f) * (1.0f / norm_sq(d)) * 1.00001f + 1e-6f;
            }
            reach = slab(o, inv_d, bx.lo, bx.hi, limit, tn);
        }
        pending_push(out, reach, slot, &pb.counters[bounce].n_host_rays, pb.host_queue);
    }
    pending_flush(out, &pb.counters[bounce].n_host_rays, pb.host_queue);
}

__global__ void __launch_bounds__(256) k_host_export_rays(PathBuffers pb, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = pb.host_queue[i];
    const float4 o = pb.ray_o[slot], d = pb.ray_d[slot];
    float* q = out + 6 * static_cast<size_t>(i);
    q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = d.x; q[4] = d.y; q[5] = d.z;
}

// pass 0: the smaller key wins; pass 1: the winner's position follows its key
__global__ void __launch_bounds__(256) k_host_merge(PathBuffers pb, uint32_t n, const uint32_t* idx, const unsigned long long* key, const float* pos, int pass) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = pb.host_queue[idx[i]];
    if (pass == 0) atomicMin(&pb.hit_key[slot], key[i]);
    else if (pb.hit_key[slot] == key[i]) pb.hit_pos[slot] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 0.0f);
}

__global__ void __launch_bounds__(256) k_host_export_hits(PathBuffers pb, uint32_t n, float* out, uint32_t* out_prim) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = pb.host_queue[i];
    const float4 p = pb.hit_pos[slot], d = pb.ray_d[slot];
    float* q = out + 6 * static_cast<size_t>(i);
    q[0] = p.x; q[1] = p.y; q[2] = p.z; q[3] = d.x; q[4] = d.y; q[5] = d.z;
    out_prim[i] = static_cast<uint32_t>(pb.hit_key[slot] & 0xffffffffu);
}

// Scene::trace's per-hit body (src/Scene.cpp:52-75) with the ColorBounce the host's Object::deflectRay returned
// (its direction is already in world space, src/Object.cpp:42).
__global__ void __launch_bounds__(256) k_host_apply(PathBuffers pb, uint32_t n, const float* b9, uint32_t bounce, uint32_t max_bounces) {
    if (*pb.cancel_latch) return;
    __shared__ uint32_t s_pend[4 * 512];
    PendingQueue survivors{s_pend + (threadIdx.x >> 6) * 512u, 512u, 0u};
    uint32_t* next_queue = pb.queue[(bounce + 1) & 1];
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool alive = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = pb.host_queue[i];
            const float* b = b9 + 9 * static_cast<size_t>(i);
            const f3 wdir = mk3(b[6], b[7], b[8]);
            const f3 wp = ld3(pb.hit_pos[slot]);
            const f3 npos = mk3(wp.x + 1e-3f * wdir.x, wp.y + 1e-3f * wdir.y, wp.z + 1e-3f * wdir.z);
            float4 att = bounce == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : pb.att[slot];
            float4 rad = bounce == 0 ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : pb.rad[slot];
            rad.x += b[0] * att.x;
            rad.y += b[1] * att.y;
            rad.z += b[2] * att.z;
            att.x *= b[3];
            att.y *= b[4];
            att.z *= b[5];
            pb.rad[slot] = rad;
            alive = !(att.x + att.z + att.y < 1e-6f) && (bounce + 1 < max_bounces);
            if (alive) {
                pb.att[slot] = att;
                pb.ray_o[slot] = st4(npos);
                pb.ray_d[slot] = st4(wdir);
            }
        }
        pending_push(survivors, alive, slot, &pb.counters[bounce + 1].n_live, next_queue);
    }
    pending_flush(survivors, &pb.counters[bounce + 1].n_live, next_queue);
}

// ---- the lane-evaluated SDF kinds' normals and per-hit materials, shared by shading and the parity probe ----------
// Forward-difference normal (signedDistanceNormal, include/Object.hpp:118-132) of a lane-evaluated object at `base`,
// by the four lanes of a quad: lane e of the quad evaluates point e (all four lanes call with the same arguments).
__device__ __forceinline__ f3 plain_quad_normal(uint32_t plain, f3 prm, const float* noise_raw, f3 base, uint32_t e) {
    const float delta = 1e-3f;
    const f3 q = mk3(base.x + (e == 1u ? delta : 0.0f), base.y + (e == 2u ? delta : 0.0f), base.z + (e == 3u ? delta : 0.0f));
    const float w = plain_sdf(plain, prm, noise_raw, q);
    const float w0 = dpp_f<0x00>(0.0f, w), w1 = dpp_f<0x55>(0.0f, w), w2 = dpp_f<0xAA>(0.0f, w), w3 = dpp_f<0xFF>(0.0f, w);
    return unit(mk3(w1 - w0, w2 - w0, w3 - w0) * (1.0f / delta));
}
// The normals a hit at `local_hit` needs: the surface normal and, for RoughSphereObject, whose material is made of
// normals at two other points (PathTracerDemo.cpp:187-188), those two.  Quad-cooperative like plain_quad_normal.
struct PlainNormals {
    f3 n, n1, n2;
};
__device__ __forceinline__ PlainNormals plain_normals(uint32_t plain, f3 prm, const float* noise_raw, f3 local_hit, uint32_t e) {
    PlainNormals r;
    r.n = plain_quad_normal(plain, prm, noise_raw, local_hit, e);
    r.n1 = r.n2 = mk3(0, 0, 0);
    if (plain == 6u) {
        r.n1 = plain_quad_normal(plain, prm, noise_raw, (-local_hit) * 1.724674f + mk3(3.474577f, 1.734674f, 2.713652f), e);
        r.n2 = plain_quad_normal(plain, prm, noise_raw, local_hit * 2.446744f + mk3(1.734674f, 2.713652f, 4.137371f), e);
    }
    return r;
}
// The procedural materials of PathTracerDemo.cpp:184-199 (RoughSphereObject), 248-259 (RoughBoxObject), 330-341
// (WeirdWallObject): what the object's deflectLocalRay makes of its `material` member at this hit.
__device__ __forceinline__ void plain_material(uint32_t plain, const PlainNormals& pn, f3 local_hit, MaterialRec& mat) {
    f3 a;
    if (plain == 6u) {
        a = mk3(0.5f, 0.5f, 0.5f) + 0.5f * pn.n1;
        const f3 n2 = mk3(0.5f, 0.5f, 0.5f) + 0.5f * pn.n2;
        mat.diffuse = n2.x;
        mat.specular = 1.0f - n2.x;
        mat.sharpness = n2.y;
        const float k = 0.125f * n2.z;
        mat.emitted[0] = k * a.x;
        mat.emitted[1] = k * a.y;
        mat.emitted[2] = k * a.z;
    } else {
        a = foam_albedo(local_hit, plain == 5u ? 0.2f : 0.018347f);
    }
    mat.albedo[0] = a.x;
    mat.albedo[1] = a.y;
    mat.albedo[2] = a.z;
}

// Parity probe: the material a hit at each of n local points of SDF primitive `prim` is shaded with -- 9 floats per point
// {reflected absorption r g b, emitted luminance r g b, diffuse, specular, sharpness} -- through the code shading uses.
// One quad per point.
__global__ void __launch_bounds__(256) k_sdf_material(SceneDev sc, uint32_t prim, const float* pos3, uint32_t n, float* out9) {
    const PrimRec* pr = sc.prims + prim;
    const uint32_t kind = pr->kind, plain = kind >= 5u ? kind : 0u;
    const f3 prm = mk3(pr->p[0], pr->p[1], pr->p[2]);
    const uint32_t e = threadIdx.x & 3u;
    const uint32_t quads = (gridDim.x * blockDim.x) >> 2;
    const uint32_t rounds = (n + quads - 1u) / quads;   // whole quads run every round: the normals are quad-cooperative
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * quads + ((blockIdx.x * blockDim.x + threadIdx.x) >> 2);
        const bool valid = i < n;
        const f3 p = valid ? mk3(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]) : mk3(0.1f, 0.2f, 0.3f);
        MaterialRec mat = sc.materials[pr->material];
        if (plain) {
            const PlainNormals pn = plain_normals(plain, prm, sc.noise_raw, p, e);
            plain_material(plain, pn, p, mat);
        }
        if (valid && e == 0u) {
            float* o = out9 + 9 * static_cast<size_t>(i);
            o[0] = mat.albedo[0]; o[1] = mat.albedo[1]; o[2] = mat.albedo[2];
            o[3] = mat.emitted[0]; o[4] = mat.emitted[1]; o[5] = mat.emitted[2];
            o[6] = mat.diffuse; o[7] = mat.specular; o[8] = mat.sharpness;
        }
    }
}

// SDF hits: the normal is a forward difference of four SDF evaluations (signedDistanceNormal,
// include/Object.hpp:118-132): at the hit and at +delta along x, y, z.  The four lanes of a quad take
// the four points of one hit, so a wave shades 16 hits per round; the quad's first lane then deflects
// the path.  A wave takes sixteen CONSECUTIVE hits of the queue per round: neighbours in the queue are neighbours on
// the object (camera rays' hits come in screen order), so the wave's sixteen crater lists are nearly one list and the
// evaluator's loop, which runs for the longest of them, wastes little (dealt round-robin, a wave's hits were 3,000
// entries apart: 21.7 ms per frame instead of 18.7).
template <bool PLAIN>
__device__ __forceinline__ void shade_sdf_body(const SceneDev& sc, const BatchParams& bp, const PathBuffers& pb, uint32_t round, uint32_t max_bounces) {
    if (*pb.cancel_latch) return;   // the frame was cancelled (as of the last latch)
    const uint32_t n = pb.counters[round].n_sdf_shade;
    uint32_t* next_queue = pb.queue[(round + 1) & 1];
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t per_round = n_waves * 16u;
    const uint32_t rounds = (n + per_round - 1) / per_round;
    uint32_t c_evals = 0, c_near = 0, c_noise = 0;
    __shared__ __attribute__((aligned(16))) float s_near[4 * kNearLdsFloats];
    constexpr uint32_t kCap = 128;   // small: three blocks of this kernel must fit the CU's LDS
    __shared__ uint32_t s_pend[4 * kCap];
    PendingQueue survivors{s_pend + (threadIdx.x >> 6) * kCap, kCap, 0u};
    const uint32_t lane = __lane_id();
    float* lds = s_near + (threadIdx.x >> 6) * kNearLdsFloats;
    const CoopLane cl = make_coop_lane(sc.noise_oct, lane);
    CraterTableRegs crater_regs = no_crater_table();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t e = lane & 3u;   // which of the four points
    const float delta = 1e-3f;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * per_round + wave * 16u + (lane >> 2);
        const bool valid = i < n;
        uint32_t slot = 0, crater_base = 0, crater_count = 0;
        PrimRegs pr;
        f3 wp = mk3(0, 0, 0), ld = mk3(0, 0, 0), p = mk3(0, 0, 0), off = mk3(0, 0, 0), local_hit = mk3(0, 0, 0);
        float s = 0.0f;
        uint32_t lag = 0;   // rounds the path has fallen behind: its bounce number is round - lag
        bool fractal = false;
        uint32_t plain = 0;   // kind of a lane-evaluated object, else 0
        if (valid) {
            slot = pb.sdf_shade_queue[i];
            const uint32_t prim = static_cast<uint32_t>(pb.hit_key[slot] & 0xffffffffu);
            load_prim(sc.prims, prim, pr);
            wp = ld3(pb.hit_pos[slot]);
            const float4 dv = pb.ray_d[slot];
            const f3 d = ld3(dv);
            lag = __float_as_uint(dv.w);
            ld = lin_mul(pr.inv, d);
            const f3 lp = aff_pos(pr.inv, wp);
            fractal = pr.kind == 4u;
            if (PLAIN) plain = pr.kind >= 5u ? pr.kind : 0u;
            local_hit = lp;
            if (!fractal) {
                off = mk3(pr.p[0], pr.p[1], pr.p[2]);
                crater_base = __float_as_uint(pr.p[3]);
                crater_count = __float_as_uint(pr.p[4]);
                p = mk3(lp.x + (e == 1u ? delta : 0.0f), lp.y + (e == 2u ? delta : 0.0f), lp.z + (e == 3u ? delta : 0.0f));
                s = norm(p) - 0.5f;
                ++c_evals;
            }
        }
        // the far field returns the sphere distance itself (`sdf_sphere > 0.1` compares against a double literal)
        const bool wants = valid && !fractal && !(PLAIN && plain) && !(static_cast<double>(s) > 0.1);
        float v = s;
        // Lane-evaluated kinds: each lane of the quad evaluates its own point of the forward difference
        // (a quad shares one hit, so the branch is uniform over the quad and the broadcasts are safe).
        PlainNormals pn{mk3(0, 0, 1), mk3(0, 0, 0), mk3(0, 0, 0)};
        if (PLAIN && plain) {
            pn = plain_normals(plain, off, sc.noise_raw, local_hit, e);
            if (plain == 6u) c_evals += 2;
        }
        wave_near(sc.craters, sc.crater_masks, static_cast<const float4*>(sc.minor_cells), crater_regs, cl, lds, lane, bp.lane_threshold, wants, p, off, s, crater_base, crater_count, v, c_near, c_noise);
        const float v0 = dpp_f<0x00>(0.0f, v), v1 = dpp_f<0x55>(0.0f, v), v2 = dpp_f<0xAA>(0.0f, v), v3 = dpp_f<0xFF>(0.0f, v);   // quad broadcasts
        bool alive = false;
        if (valid && e == 0u) {
            f3 nrm;
            if (fractal) {
                nrm = fractal_normal(local_hit);   // six cheap evaluations on the quad's first lane
            } else if (PLAIN && plain) {
                nrm = pn.n;
            } else {
                const f3 g = mk3(v1 - v0, v2 - v0, v3 - v0) * (1.0f / delta);
                nrm = unit(g);
            }
            const uint32_t s_local = slot / bp.n_pix;
            uint32_t px, py;
            pixel_of(bp, slot - s_local * bp.n_pix, px, py);
            const uint32_t bounce = round - lag;
            Rng rng(bp.seed, py * bp.width + px, bp.first_sample + s_local, 1u + bounce);
            Bounce b;
            if (PLAIN && plain) {
                MaterialRec mat = sc.materials[pr.material];
                plain_material(plain, pn, local_hit, mat);
                b = deflect(mat, ld, nrm, rng);
            } else {
                b = deflect(sc.materials[pr.material], ld, nrm, rng);
            }
            alive = apply_bounce(pb, slot, pr, wp, b, bounce, max_bounces, lag);
        }
        pending_push(survivors, alive, slot, &pb.counters[round + 1].n_live, next_queue);
    }
    __shared__ uint32_t s_meet[8];
    pending_flush_block(survivors, &pb.counters[round + 1].n_live, next_queue, s_meet);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pb.totals->sdf_shades, static_cast<unsigned long long>(n));
    if (rounds) {
        wave_add_total(&pb.totals->shade_near_evals, c_evals);
        wave_add_total(&pb.totals->shade_noise_calls, c_noise);
    }
}

// The same for scenes whose SDF objects are crater balls and fractals only, 64 hits per wave and round: every lane
// loads one hit and, at the end, deflects its path; in between, four passes of evaluations, in each of which the
// quads work on the points of sixteen of the hits (a quad asks the lane that holds its hit for the point, the
// lane collects the four values from its quad).  Loading and deflecting thus run with all lanes instead of one lane
// of each quad.
__device__ __forceinline__ void shade_sdf_body64(const SceneDev& sc, const BatchParams& bp, const PathBuffers& pb, uint32_t round, uint32_t max_bounces) {
    if (*pb.cancel_latch) return;   // the frame was cancelled (as of the last latch)
    const uint32_t n = pb.counters[round].n_sdf_shade;
    uint32_t* next_queue = pb.queue[(round + 1) & 1];
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t per_round = n_waves * 64u;
    const uint32_t rounds = (n + per_round - 1) / per_round;
    uint32_t c_evals = 0, c_near = 0, c_noise = 0;
    __shared__ __attribute__((aligned(16))) float s_near[4 * kNearLdsFloats];
    constexpr uint32_t kCap = 128;   // small: three blocks of this kernel must fit the CU's LDS
    __shared__ uint32_t s_pend[4 * kCap];
    PendingQueue survivors{s_pend + (threadIdx.x >> 6) * kCap, kCap, 0u};
    const uint32_t lane = __lane_id();
    float* lds = s_near + (threadIdx.x >> 6) * kNearLdsFloats;
    const CoopLane cl = make_coop_lane(sc.noise_oct, lane);
    CraterTableRegs crater_regs = no_crater_table();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t e = lane & 3u;   // which of a hit's four points this lane evaluates
    const float delta = 1e-3f;
    for (uint32_t r = 0; r < rounds; ++r) {
        // ---- this lane's hit ----
        const uint32_t i = r * per_round + wave * 64u + lane;
        const bool valid = i < n;
        // (only what the evaluations need stays in registers across them; the rest of the hit is read again afterwards)
        uint32_t slot = 0, crater_base = 0, crater_count = 0;
        f3 off = mk3(0, 0, 0), local_hit = mk3(0, 0, 0);
        bool fractal = false;
        if (valid) {
            slot = pb.sdf_shade_queue[i];
            const PrimRec* rec = sc.prims + static_cast<uint32_t>(pb.hit_key[slot] & 0xffffffffu);
            local_hit = aff_pos(rec->inv, ld3(pb.hit_pos[slot]));
            fractal = rec->kind == 4u;
            if (!fractal) {
                off = mk3(rec->p[0], rec->p[1], rec->p[2]);
                crater_base = __float_as_uint(rec->q[0]);
                crater_count = __float_as_uint(rec->q[1]);
            }
        }
        // ---- the four values of every crater-ball hit, sixteen hits a pass ----
        const bool needs = valid && !fractal;
        const unsigned long long need_mask = __ballot(needs);
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
#pragma unroll 1
        for (uint32_t pass = 0; pass < 4u; ++pass) {
            if (((need_mask >> (16u * pass)) & 0xffffull) == 0ull) continue;
            const int src = 4 * static_cast<int>(16u * pass + (lane >> 2));   // (byte address of) the lane that holds this quad's hit
            const bool ok = __builtin_amdgcn_ds_bpermute(src, needs ? 1 : 0) != 0;
            const f3 base = mk3(fetch_lane(local_hit.x, src), fetch_lane(local_hit.y, src), fetch_lane(local_hit.z, src));
            const f3 off_q = mk3(fetch_lane(off.x, src), fetch_lane(off.y, src), fetch_lane(off.z, src));
            const uint32_t cb = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(crater_base)));
            const uint32_t cc = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(crater_count)));
            const f3 p = mk3(base.x + (e == 1u ? delta : 0.0f), base.y + (e == 2u ? delta : 0.0f), base.z + (e == 3u ? delta : 0.0f));
            const float s = norm(p) - 0.5f;
            if (ok) ++c_evals;
            // the far field returns the sphere distance itself (`sdf_sphere > 0.1` compares against a double literal)
            const bool wants = ok && !(static_cast<double>(s) > 0.1);
            float v = s;
            wave_near(sc.craters, sc.crater_masks, static_cast<const float4*>(sc.minor_cells), crater_regs, cl, lds, lane, bp.lane_threshold, wants, p, off_q, s, cb, cc, v, c_near, c_noise);
            // lane 16 pass + h collects the values of quad h
            const int quad = 16 * static_cast<int>(lane & 15u);
            const float q0 = fetch_lane(v, quad), q1 = fetch_lane(v, quad + 4), q2 = fetch_lane(v, quad + 8), q3 = fetch_lane(v, quad + 12);
            if ((lane >> 4) == pass) {
                v0 = q0;
                v1 = q1;
                v2 = q2;
                v3 = q3;
            }
        }
        // ---- deflect ----
        bool alive = false;
        if (valid) {
            f3 nrm;
            if (fractal) {
                nrm = fractal_normal(local_hit);
            } else {
                const f3 g = mk3(v1 - v0, v2 - v0, v3 - v0) * (1.0f / delta);
                nrm = unit(g);
            }
            PrimRegs pr;
            load_prim(sc.prims, static_cast<uint32_t>(pb.hit_key[slot] & 0xffffffffu), pr);
            const f3 wp = ld3(pb.hit_pos[slot]);
            const float4 dv = pb.ray_d[slot];
            const uint32_t lag = __float_as_uint(dv.w);
            const f3 ld = lin_mul(pr.inv, ld3(dv));
            const uint32_t s_local = slot / bp.n_pix;
            uint32_t px, py;
            pixel_of(bp, slot - s_local * bp.n_pix, px, py);
            const uint32_t bounce = round - lag;
            Rng rng(bp.seed, py * bp.width + px, bp.first_sample + s_local, 1u + bounce);
            const Bounce b = deflect(sc.materials[pr.material], ld, nrm, rng);
            alive = apply_bounce(pb, slot, pr, wp, b, bounce, max_bounces, lag);
        }
        pending_push(survivors, alive, slot, &pb.counters[round + 1].n_live, next_queue);
    }
    __shared__ uint32_t s_meet[8];
    pending_flush_block(survivors, &pb.counters[round + 1].n_live, next_queue, s_meet);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pb.totals->sdf_shades, static_cast<unsigned long long>(n));
    if (rounds) {
        wave_add_total(&pb.totals->shade_near_evals, c_evals);
        wave_add_total(&pb.totals->shade_noise_calls, c_noise);
    }
}

// Three waves per SIMD for the crater-ball instantiation (the register allocator would otherwise take one
// register too many for that); the instantiation with the lane-evaluated kinds needs more and runs two.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_shade_sdf(SceneDev sc, BatchParams bp, PathBuffers pb, uint32_t round,
                                                                                           uint32_t max_bounces) {
    shade_sdf_body64(sc, bp, pb, round, max_bounces);
}
__global__ void __launch_bounds__(256) k_shade_sdf_plain(SceneDev sc, BatchParams bp, PathBuffers pb, uint32_t round, uint32_t max_bounces) {
    shade_sdf_body<true>(sc, bp, pb, round, max_bounces);
}

__global__ void k_finish_batch(PathBuffers pb, uint32_t rounds, uint32_t n_slots, uint32_t carry_cap) {
    unsigned long long rays = n_slots;   // every path casts its primary ray; later casts are the queue lengths
    for (uint32_t r = 1; r < rounds; ++r) rays += pb.counters[r].n_live;
    // a ray whose march was handed on stands in two rounds' queues: the one it was cast in and the one it was finished in
    unsigned long long carried = 0, late = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        carried += min(pb.counters[r].n_carry, carry_cap);
        late += pb.counters[r].n_late;
    }
    pb.totals->rays += rays - late;
    pb.totals->marches_carried += carried;
}

__global__ void k_latch_cancel(PathBuffers pb) { *pb.cancel_latch = pb.cancel ? *pb.cancel : 0u; }

// frame[pix] += sample 0 + sample 1 + ... of this batch, sequentially in sample order.  Only the paths
// marked in the batch's bitmap carry radiance; the others would add exactly zero and are not read.
__global__ void __launch_bounds__(256) k_accumulate(BatchParams bp, PathBuffers pb, float* frame) {
    if (*pb.cancel_latch) return;   // cancelled: this batch stays out of the frame
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t samples = bp.n_slots / bp.n_pix;
    const uint32_t words_per_sample = (bp.n_pix + 63u) / 64u;
    if (blockIdx.x == 0 && threadIdx.x == 0) pb.totals->samples_done += samples;   // launches on one stream: no two of them overlap
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < bp.n_pix; id += stride) {
        uint32_t x, y;
        if (!pixel_of(bp, id, x, y)) continue;
        const uint32_t pix = y * bp.width + x;
        float r = frame[3 * pix], g = frame[3 * pix + 1], b = frame[3 * pix + 2];
        const unsigned long long* bits = pb.live_bits + (id >> 6);
        const uint32_t bit = id & 63u;
        uint32_t s = 0;
        for (; s + 4u <= samples; s += 4u) {   // four bitmap words in flight
            const unsigned long long m0 = bits[static_cast<size_t>(s) * words_per_sample];
            const unsigned long long m1 = bits[static_cast<size_t>(s + 1u) * words_per_sample];
            const unsigned long long m2 = bits[static_cast<size_t>(s + 2u) * words_per_sample];
            const unsigned long long m3 = bits[static_cast<size_t>(s + 3u) * words_per_sample];
            const unsigned long long m[4] = {m0, m1, m2, m3};
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
                if ((m[k] >> bit) & 1ull) {
                    const float4 c = pb.rad[(s + k) * bp.n_pix + id];
                    r += c.x;
                    g += c.y;
                    b += c.z;
                }
        }
        for (; s < samples; ++s)
            if ((bits[static_cast<size_t>(s) * words_per_sample] >> bit) & 1ull) {
                const float4 c = pb.rad[s * bp.n_pix + id];
                r += c.x;
                g += c.y;
                b += c.z;
            }
        frame[3 * pix] = r;
        frame[3 * pix + 1] = g;
        frame[3 * pix + 2] = b;
    }
}

__global__ void __launch_bounds__(256) k_scale(const float* in, float k, size_t n, float* out) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i] * k;
}

__global__ void __launch_bounds__(256) k_add(float* dst, const float* src, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

// copyToSFImage's conversion (PathTracerDemo.cpp:524-545): clamp to [0, 1], times 255, truncate; bytes R, G, B, A = 255
__global__ void __launch_bounds__(256) k_pack_rgba8(const float* in, size_t n_pix, uint32_t* out) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
        uint32_t px = 0xff000000u;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = in[3 * i + ch];
            const float cl = v < 0.0f ? 0.0f : (1.0f < v ? 1.0f : v);   // std::clamp; a NaN shows as 0
            px |= static_cast<uint32_t>(cl == cl ? cl * 255.0f : 0.0f) << (8 * ch);
        }
        out[i] = px;
    }
}

__global__ void __launch_bounds__(256) k_load_rays(const float* pos, const float* dir, uint32_t n, PathBuffers pb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pb.ray_o[i] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 0.0f);
    pb.ray_d[i] = make_float4(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], 0.0f);
    pb.rad[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    pb.queue[0][i] = i;
    if (i == 0) pb.counters[0].n_live = n;
}

__global__ void __launch_bounds__(256) k_export_rad(PathBuffers pb, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = pb.rad[i];
    out[3 * i] = c.x;
    out[3 * i + 1] = c.y;
    out[3 * i + 2] = c.z;
}

// =============================================================================================
// tone mappers (src/ToneMapper.cpp)
// =============================================================================================
__global__ void __launch_bounds__(256) k_tonemap_filmic(const float* in, float* out, size_t n) {   // :53-72
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float p = in[i];
        const float v = (p * (a * p + b)) / (p * (c * p + d) + e);
        out[i] = fminf(fmaxf(v, 0.0f), 1.0f);
    }
}

constexpr int kReinhardBlocks = 1024;

// Pass 1 of ReinhardToneMapper :6-21: per-block partial sum of log(lum + 1e-6), min and max of the
// luminance (which reads the blue channel twice, as the reference does).
__global__ void __launch_bounds__(256) k_reinhard_reduce(const float* in, size_t n_pix, float* scratch) {
    __shared__ float s_sum[256], s_min[256], s_max[256];
    float sum = 0.0f, mn = 3.402823466e38f, mx = -3.402823466e38f;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
        const float lum = 0.27f * in[3 * i] + 0.67f * in[3 * i + 2] + 0.06f * in[3 * i + 2];
        mn = fminf(mn, lum);
        mx = fmaxf(mx, lum);
        sum += logf(lum + 1e-6f);
    }
    s_sum[threadIdx.x] = sum;
    s_min[threadIdx.x] = mn;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
            s_min[threadIdx.x] = fminf(s_min[threadIdx.x], s_min[threadIdx.x + w]);
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scratch[3 * blockIdx.x] = s_sum[0];
        scratch[3 * blockIdx.x + 1] = s_min[0];
        scratch[3 * blockIdx.x + 2] = s_max[0];
    }
}

// Pass 2: one block folds the partials and derives k = alpha / avgLum (:23-30).
__global__ void __launch_bounds__(256) k_reinhard_key(float* scratch, int n_blocks, size_t n_pix) {
    __shared__ double s_sum[256];
    __shared__ float s_min[256], s_max[256];
    double sum = 0.0;
    float mn = 3.402823466e38f, mx = -3.402823466e38f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        sum += scratch[3 * i];
        mn = fminf(mn, scratch[3 * i + 1]);
        mx = fmaxf(mx, scratch[3 * i + 2]);
    }
    s_sum[threadIdx.x] = sum;
    s_min[threadIdx.x] = mn;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
            s_min[threadIdx.x] = fminf(s_min[threadIdx.x], s_min[threadIdx.x + w]);
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float acc = static_cast<float>(s_sum[0]);
        const float avg = expf(acc / static_cast<float>(n_pix));
        const float log_avg = log2f(avg), log_min = log2f(s_min[0] + 1e-6f), log_max = log2f(s_max[0] + 1e-6f);
        const float alpha = 0.18f * powf(4.0f, (2.0f * log_avg - log_min - log_max) / (log_max - log_min));
        scratch[3 * n_blocks] = alpha / avg;
    }
}

__global__ void __launch_bounds__(256) k_reinhard_map(const float* in, float* out, size_t n, const float* key) {   // :34-49
    const float k = *key;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float sc = in[i] * k;
        out[i] = sc / (1.0f + sc);
    }
}

// =============================================================================================
// launchers
// =============================================================================================
static inline uint32_t grid_for(size_t n, uint32_t cap = 256u * 32u) {
    const size_t b = (n + 255) / 256;
    return static_cast<uint32_t>(b < 1 ? 1 : (b > cap ? cap : b));
}

void launch_generate_extend(hipStream_t st, const CameraRec& cam, const SceneDev& sc, const BatchParams& bp, const PathBuffers& pb, const CullRects& cull,
                            int stack_need, bool count) {
    // one thread per pixel; enough sample stripes to fill the chip when the image is small
    const uint32_t bx = (bp.n_pix + 255u) / 256u, samples = bp.n_slots / bp.n_pix;
    const uint32_t by = std::max(1u, std::min(samples, (4096u + bx - 1u) / bx));
    const dim3 grid(bx, by), block(256);
    // (traversal stack, pending-queue capacity): LDS per block = STACK KiB + 32 CAP bytes, at most 64 KiB; four blocks
    // share a CU's 160 KiB up to 40 KiB each, three up to 53 KiB
#define MIPT_LAUNCH_GE(STACK, CAP)                                                                                        \
    do {                                                                                                                 \
        if (count) hipLaunchKernelGGL((k_generate_extend<STACK, CAP, true>), grid, block, 0, st, cam, sc, bp, pb, cull);  \
        else hipLaunchKernelGGL((k_generate_extend<STACK, CAP, false>), grid, block, 0, st, cam, sc, bp, pb, cull);       \
    } while (0)
    if (stack_need <= 8) MIPT_LAUNCH_GE(8, 1024);
    else if (stack_need <= 24) MIPT_LAUNCH_GE(24, 512);
    else if (stack_need <= 32) MIPT_LAUNCH_GE(32, 256);
    else if (stack_need <= 36) MIPT_LAUNCH_GE(36, 128);   // 36 + 4 KiB: the last size of which four blocks fit a CU (the 1M-object tree needs 35)
    else if (stack_need <= 48) MIPT_LAUNCH_GE(48, 128);
    else MIPT_LAUNCH_GE(60, 128);
#undef MIPT_LAUNCH_GE
}

void launch_generate(hipStream_t st, const CameraRec& cam, const BatchParams& bp, const PathBuffers& pb, const CullRects& cull) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(bp.n_slots, 256u * 16u)), dim3(256), 0, st, cam, bp, pb, cull);
}

void launch_extend(hipStream_t st, const SceneDev& sc, const BatchParams& bp, const PathBuffers& pb, uint32_t bounce, int stack_need, bool count) {
    // persistent waves: every wave drains the queue
    const dim3 grid(256u * 4u), block(256);
#define MIPT_LAUNCH_EX(STACK)                                                                          \
    do {                                                                                              \
        if (count) hipLaunchKernelGGL((k_extend<STACK, true>), grid, block, 0, st, sc, bp, pb, bounce); \
        else hipLaunchKernelGGL((k_extend<STACK, false>), grid, block, 0, st, sc, bp, pb, bounce);      \
    } while (0)
#define MIPT_LAUNCH_EXP(STACK)                                                                               \
    do {                                                                                                   \
        if (count) hipLaunchKernelGGL((k_extend_pair<STACK, true, true>), grid, block, 0, st, sc, bp, pb, bounce);          \
        else if (dist) hipLaunchKernelGGL((k_extend_pair<STACK, false, true>), grid, block, 0, st, sc, bp, pb, bounce);    \
        else hipLaunchKernelGGL((k_extend_pair<STACK, false, false>), grid, block, 0, st, sc, bp, pb, bounce);             \
    } while (0)
    // stack entries that carry their entry distance: the tree's links must fit 20 bits (the counting build is only
    // instantiated for such trees; larger ones are counted by the one-ray kernels)
    if (sc.n_prims <= 8u && !count && !bp.no_few) {   // a handful of primitives: tested one after the other, no tree
        hipLaunchKernelGGL(k_extend_few, dim3(grid_for(bp.n_slots, 256u * 8u)), dim3(256), 0, st, sc, bp, pb, bounce);
        return;
    }
    const bool dist = sc.n_prims <= (1u << 20) && sc.n_nodes <= (1u << 20) && !bp.pair_plain_entries;
    if (bp.trav_quorum_pair && stack_need > 8 && (dist || !count)) {
        // two rays per lane: one column of S entries for both (S >= the tree's need + 3: the second ray asks for three free rows before every visit, and must find them when it is alone)
        if (stack_need + 3 <= 32) MIPT_LAUNCH_EXP(32);
        else if (stack_need + 3 <= 36) MIPT_LAUNCH_EXP(36);
        else if (stack_need + 3 <= 38) MIPT_LAUNCH_EXP(38);   // 38 KiB of stacks + 2 KiB of pending queue: the last size of which four blocks fit a CU
        else if (stack_need + 3 <= 40) MIPT_LAUNCH_EXP(40);
        else MIPT_LAUNCH_EXP(63);
        return;
    }
#undef MIPT_LAUNCH_EXP
    if (stack_need <= 8) MIPT_LAUNCH_EX(8);
    else if (stack_need <= 24) MIPT_LAUNCH_EX(24);
    else if (stack_need <= 32) MIPT_LAUNCH_EX(32);
    else if (stack_need <= 48) MIPT_LAUNCH_EX(48);
    else MIPT_LAUNCH_EX(60);
#undef MIPT_LAUNCH_EX
}

void launch_crater_cells(hipStream_t st, const float off[3], const int base[3], int n, float* out) {
    hipLaunchKernelGGL(k_crater_cells, dim3((n * n * n + 255) / 256), dim3(256), 0, st, off[0], off[1], off[2], base[0], base[1], base[2], n, reinterpret_cast<float4*>(out));
}
void launch_minor_cells(hipStream_t st, const float off[3], void* out) {
    hipLaunchKernelGGL(k_minor_cells, dim3((kMinorCellsPerBall + 255u) / 256u), dim3(256), 0, st, off[0], off[1], off[2], static_cast<float4*>(out));
}
size_t minor_cells_bytes_per_ball() { return static_cast<size_t>(kMinorCellsPerBall) * sizeof(float4); }

void launch_sdf_march(hipStream_t st, const SceneDev& sc, const BatchParams& bp, const PathBuffers& pb, uint32_t bounce) {
    // persistent waves: enough blocks to fill the chip a few times over, every wave drains the queue
    const uint32_t* queue = pb.sdf_queue;
    if (bounce >= 1u && bounce <= bp.sort_marches) {
        hipLaunchKernelGGL(k_sort_keys, dim3(256u * 4u), dim3(256), 0, st, sc, pb, bounce);
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, st, pb, bounce);
        hipLaunchKernelGGL(k_sort_scatter, dim3(256u * 4u), dim3(256), 0, st, pb, bounce);
        queue = pb.sdf_shade_queue;
    }
    if (sc.plain_sdf) hipLaunchKernelGGL(k_sdf_march<true>, dim3(256u * 4u), dim3(256), 0, st, sc, bp, pb, bounce, queue);
    else hipLaunchKernelGGL(k_sdf_march<false>, dim3(256u * 4u), dim3(256), 0, st, sc, bp, pb, bounce, queue);
}

void launch_sdf_eval(hipStream_t st, const SceneDev& sc, uint32_t prim, uint32_t lane_threshold, const float* pos3, uint32_t n, float* out) {
    hipLaunchKernelGGL(k_sdf_eval, dim3(grid_for(n, 256u * 4u)), dim3(256), 0, st, sc, prim, lane_threshold, pos3, n, out);
}

void launch_sdf_material(hipStream_t st, const SceneDev& sc, uint32_t prim, const float* pos3, uint32_t n, float* out9) {
    hipLaunchKernelGGL(k_sdf_material, dim3(grid_for(static_cast<size_t>(n) * 4u, 256u * 4u)), dim3(256), 0, st, sc, prim, pos3, n, out9);
}

void launch_shade(hipStream_t st, const SceneDev& sc, const BatchParams& bp, const PathBuffers& pb, uint32_t bounce, uint32_t max_bounces) {
    hipLaunchKernelGGL(k_shade, dim3(grid_for(bp.n_slots, 256u * 16u)), dim3(256), 0, st, sc, bp, pb, bounce, max_bounces);
}

void launch_shade_sdf(hipStream_t st, const SceneDev& sc, const BatchParams& bp, const PathBuffers& pb, uint32_t bounce, uint32_t max_bounces) {
    if (sc.plain_sdf) hipLaunchKernelGGL(k_shade_sdf_plain, dim3(256u * 4u), dim3(256), 0, st, sc, bp, pb, bounce, max_bounces);
    else hipLaunchKernelGGL(k_shade_sdf, dim3(256u * 4u), dim3(256), 0, st, sc, bp, pb, bounce, max_bounces);
}

void launch_host_gather(hipStream_t st, const SceneDev& sc, const PathBuffers& pb, uint32_t bounce, uint32_t which) {
    hipLaunchKernelGGL(k_host_gather, dim3(256u * 8u), dim3(256), 0, st, sc, pb, bounce, which);
}
void launch_host_export_rays(hipStream_t st, const PathBuffers& pb, uint32_t n, float* out6) {
    hipLaunchKernelGGL(k_host_export_rays, dim3((n + 255) / 256), dim3(256), 0, st, pb, n, out6);
}
void launch_host_merge(hipStream_t st, const PathBuffers& pb, uint32_t n, const uint32_t* idx, const unsigned long long* key, const float* pos3) {
    hipLaunchKernelGGL(k_host_merge, dim3((n + 255) / 256), dim3(256), 0, st, pb, n, idx, key, pos3, 0);
    hipLaunchKernelGGL(k_host_merge, dim3((n + 255) / 256), dim3(256), 0, st, pb, n, idx, key, pos3, 1);
}
void launch_host_export_hits(hipStream_t st, const PathBuffers& pb, uint32_t n, float* out6, uint32_t* out_prim) {
    hipLaunchKernelGGL(k_host_export_hits, dim3((n + 255) / 256), dim3(256), 0, st, pb, n, out6, out_prim);
}
void launch_host_apply(hipStream_t st, const PathBuffers& pb, uint32_t n, const float* bounce9, uint32_t bounce, uint32_t max_bounces) {
    hipLaunchKernelGGL(k_host_apply, dim3(grid_for(n, 256u * 8u)), dim3(256), 0, st, pb, n, bounce9, bounce, max_bounces);
}

void launch_finish_batch(hipStream_t st, const PathBuffers& pb, uint32_t rounds, uint32_t n_slots, uint32_t carry_cap) {
    hipLaunchKernelGGL(k_finish_batch, dim3(1), dim3(1), 0, st, pb, rounds, n_slots, carry_cap);
}

void launch_latch_cancel(hipStream_t st, const PathBuffers& pb) { hipLaunchKernelGGL(k_latch_cancel, dim3(1), dim3(1), 0, st, pb); }

void launch_accumulate(hipStream_t st, const BatchParams& bp, const PathBuffers& pb, float* frame) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(bp.n_pix)), dim3(256), 0, st, bp, pb, frame);
}

void launch_scale(hipStream_t st, const float* in, float k, size_t n, float* out) {
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(256), 0, st, in, k, n, out);
}

void launch_add(hipStream_t st, float* dst, const float* src, size_t n) {
    hipLaunchKernelGGL(k_add, dim3(grid_for(n)), dim3(256), 0, st, dst, src, n);
}

void launch_pack_rgba8(hipStream_t st, const float* in, size_t n_pix, uint32_t* out) {
    hipLaunchKernelGGL(k_pack_rgba8, dim3(grid_for(n_pix)), dim3(256), 0, st, in, n_pix, out);
}

void launch_load_rays(hipStream_t st, const float* pos, const float* dir, uint32_t n, const PathBuffers& pb) {
    hipLaunchKernelGGL(k_load_rays, dim3((n + 255) / 256), dim3(256), 0, st, pos, dir, n, pb);
}

void launch_export_rad(hipStream_t st, const PathBuffers& pb, uint32_t n, float* out) {
    hipLaunchKernelGGL(k_export_rad, dim3((n + 255) / 256), dim3(256), 0, st, pb, n, out);
}

void launch_tonemap_filmic(hipStream_t st, const float* in, float* out, size_t n) {
    hipLaunchKernelGGL(k_tonemap_filmic, dim3(grid_for(n)), dim3(256), 0, st, in, out, n);
}

size_t reinhard_scratch_floats() { return 3 * kReinhardBlocks + 4; }

// Reinhard partials for callers that cannot allocate (mipt_tonemap_device_noalloc): a file-scope array in
// device memory, 12 KiB on every GPU the library runs on, held until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress; kernels get the address that call returns.
static __device__ __align__(256) float scratch[3 * kReinhardBlocks + 4];

hipError_t reinhard_static_scratch(float** out) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    *out = e == hipSuccess ? static_cast<float*>(p) : nullptr;
    return e;
}

void launch_tonemap_reinhard(hipStream_t st, const float* in, float* out, size_t width, size_t height, float* scratch, size_t) {
    const size_t n_pix = width * height;
    const int blocks = static_cast<int>(grid_for(n_pix, kReinhardBlocks));
    hipLaunchKernelGGL(k_reinhard_reduce, dim3(blocks), dim3(256), 0, st, in, n_pix, scratch);
    hipLaunchKernelGGL(k_reinhard_key, dim3(1), dim3(256), 0, st, scratch, blocks, n_pix);
    hipLaunchKernelGGL(k_reinhard_map, dim3(grid_for(3 * n_pix)), dim3(256), 0, st, in, out, 3 * n_pix, scratch + 3 * blocks);
}

}   // namespace mipt
