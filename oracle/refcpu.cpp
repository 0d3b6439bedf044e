// TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the timstr/pathtracer hot path.
// See refcpu.hpp for what this library is and who may use it.  Build with
// -ffp-contract=off: the reference is compiled for baseline x86-64 (no fused multiply-add), and
// every expression below keeps the reference's operand order so that results are bit-identical.
//
// File:line citations are relative to the reference checkout.
#include "refcpu.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <optional>
#include <random>
#include <thread>
#include <utility>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// Vec / Pos / Linear / Affine  (include/LinearAlgebra.hpp, src/LinearAlgebra.cpp)
// One 3-float type serves for both Vec and Pos: the reference's two types carry identical
// arithmetic, the distinction is only static typing.
// ------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }   // :283-297
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }   // :287-305
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }                        // :266-268
inline V3 operator*(V3 a, float t) { return {a.x * t, a.y * t, a.z * t}; }      // :270-272
inline V3 operator*(float t, V3 a) { return a * t; }                            // :274-276
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // :307-309
inline V3 cross(V3 u, V3 v) {                                                    // :311-313
    return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
inline float norm_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }        // :21-23
inline float norm(V3 a) { return std::sqrt(norm_sq(a)); }                       // :17-19
inline V3 unit(V3 a) {                                                           // :25-29
    const float l = norm(a);
    return {a.x / l, a.y / l, a.z / l};
}
inline V3 div(V3 a, float t) { return a * (1.0f / t); }                         // :278-281

struct Lin {
    float m[9];   // row-major, m[3*i+j]  (:180-184)
};
struct Aff {
    Lin l;
    V3 t;
};

inline V3 col(const Lin& A, int j) { return {A.m[j], A.m[3 + j], A.m[6 + j]}; }         // :164-168
inline V3 row(const Lin& A, int i) { return {A.m[3 * i], A.m[3 * i + 1], A.m[3 * i + 2]}; } // :158-162
inline V3 mul(const Lin& A, V3 v) {                                                       // :380-382
    return col(A, 0) * v.x + col(A, 1) * v.y + col(A, 2) * v.z;
}
inline Lin mul(const Lin& A, const Lin& B) {                                              // :389-395
    Lin r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = dot(row(A, i), col(B, j));
    return r;
}
inline V3 xform_vec(const Aff& T, V3 v) { return mul(T.l, v); }                           // :397-399
inline V3 xform_pos(const Aff& T, V3 p) { return mul(T.l, p) + T.t; }                     // :401-403
inline Aff mul(const Aff& A, const Aff& B) {                                              // :413-415
    return Aff{mul(A.l, B.l), mul(A.l, B.t) + A.t};
}
inline Lin identity() { return Lin{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }

Lin rotation_x(float a) {   // :90-98
    const float c = std::cos(a), s = std::sin(a);
    return Lin{{1.0f, 0.0f, 0.0f, 0.0f, c, -s, 0.0f, s, c}};
}
Lin rotation_y(float a) {   // :100-108
    const float c = std::cos(a), s = std::sin(a);
    return Lin{{c, 0.0f, s, 0.0f, 1.0f, 0.0f, -s, 0.0f, c}};
}
Lin rotation_z(float a) {   // :110-118
    const float c = std::cos(a), s = std::sin(a);
    return Lin{{c, -s, 0.0f, s, c, 0.0f, 0.0f, 0.0f, 1.0f}};
}

bool inverse(const Lin& A, Lin& out) {   // :201-233
    const float a = A.m[0], b = A.m[1], c = A.m[2], d = A.m[3], e = A.m[4], f = A.m[5], g = A.m[6],
                h = A.m[7], i = A.m[8];
    const float det = (a * e * i) + (b * f * g) + (c * d * h) - (c * e * g) - (b * d * i) - (a * f * h);
    if (det < 1e-6f) return false;
    const float k = 1.0f / det;   // operator/(Linear,float) multiplies by the reciprocal (:375-378)
    const float adj[9] = {e * i - f * h, c * h - b * i, b * f - c * e, f * g - d * i, a * i - c * g,
                          c * d - a * f, d * h - e * g, b * g - a * h, a * e - b * d};
    for (int n = 0; n < 9; ++n) out.m[n] = k * adj[n];   // operator*(Linear,float): t * A[n] (:357-369)
    return true;
}
bool inverse(const Aff& T, Aff& out) {   // :258-264: linv * Translation(-t)
    Lin li;
    if (!inverse(T.l, li)) return false;
    const Aff tr{identity(), -T.t};
    out = mul(Aff{li, v3(0.0f, 0.0f, 0.0f)}, tr);   // Linear*Affine = Affine(L)*A (:409-411)
    return true;
}

Aff load_aff(const float* p) {
    Aff a;
    std::memcpy(a.l.m, p, 9 * sizeof(float));
    a.t = v3(p[9], p[10], p[11]);
    return a;
}
void store_aff(const Aff& a, float* p) {
    std::memcpy(p, a.l.m, 9 * sizeof(float));
    p[9] = a.t.x;
    p[10] = a.t.y;
    p[11] = a.t.z;
}

// ------------------------------------------------------------------------------------------
// Random numbers.  The reference draws everything through std::uniform_real_distribution<float>
// on std::default_random_engine (src/RandomNumberGenerator.cpp:3-6).  MinstdRng is that, verbatim.
// CounterRng is the stateless generator the device path uses; it exists here so that an image can be
// compared path by path.  Both expose uniform(a, b) with the library's (b - a) * u + a form.
// ------------------------------------------------------------------------------------------
struct MinstdRng {
    std::default_random_engine eng;
    explicit MinstdRng(uint64_t seed) : eng(static_cast<std::default_random_engine::result_type>(seed)) {}
    float uniform(float a, float b) {
        std::uniform_real_distribution<float> d{a, b};
        return d(eng);
    }
    void begin_stage(uint32_t) {}
};

inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
constexpr uint64_t kGolden = 0x9e3779b97f4a7c15ull;

inline uint64_t counter_path_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    return mix64(((static_cast<uint64_t>(pixel) << 32) | sample) + seed * kGolden);
}
inline float counter_u01(uint64_t key, uint32_t stage, uint32_t k) {
    const uint64_t ctr = static_cast<uint64_t>(stage) * 64u + k + 1u;
    const uint64_t r = mix64(key + ctr * kGolden);
    return static_cast<float>(static_cast<uint32_t>(r >> 40)) * (1.0f / 16777216.0f);
}

struct CounterRng {
    uint64_t key;
    uint32_t stage = 0, k = 0;
    CounterRng(uint64_t seed, uint32_t pixel, uint32_t sample) : key(counter_path_key(seed, pixel, sample)) {}
    void begin_stage(uint32_t s) {
        stage = s;
        k = 0;
    }
    float uniform(float a, float b) { return (b - a) * counter_u01(key, stage, k++) + a; }
};

// ------------------------------------------------------------------------------------------
// Geometry  (src/Geometry.cpp)
// ------------------------------------------------------------------------------------------
constexpr float kGeoEps = 1e-3f;   // :8

struct Ray {
    V3 dir, pos;
};
struct Box {
    V3 center, half;
};

Box box_from_corners(V3 a, V3 b) {   // AxisAlignedBox(Pos,Pos) :348-358
    return Box{a + 0.5f * (b - a),
               v3(0.5f * std::abs(a.x - b.x), 0.5f * std::abs(a.y - b.y), 0.5f * std::abs(a.z - b.z))};
}

Box box_containing(const Box& a, const Box& b) {   // :105-140
    const V3 a0 = a.center - a.half, a1 = a.center + a.half;
    const V3 b0 = b.center - b.half, b1 = b.center + b.half;
    return box_from_corners(v3(std::min(a0.x, b0.x), std::min(a0.y, b0.y), std::min(a0.z, b0.z)),
                            v3(std::max(a1.x, b1.x), std::max(a1.y, b1.y), std::max(a1.z, b1.z)));
}

inline V3 mirror(V3 in, V3 n) { return in - (2.0f * (dot(in, n) * n)); }   // bounce() :142-144

std::optional<V3> hit_triangle(const Ray& r, V3 a, V3 b, V3 c) {   // :146-171
    const V3 e1 = b - a, e2 = c - a;
    const V3 h = cross(r.dir, e2);
    const float det = dot(e1, h);
    if (std::abs(det) < kGeoEps) return std::nullopt;
    const float f = 1.0f / det;
    const V3 s = r.pos - a;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return std::nullopt;
    const V3 q = cross(s, e1);
    const float v = f * dot(r.dir, q);
    if (v < 0.0f || u + v > 1.0f) return std::nullopt;
    const float t = f * dot(e2, q);
    if (t > kGeoEps) return r.pos + t * r.dir;
    return std::nullopt;
}

std::optional<V3> hit_sphere(const Ray& r, float radius) {   // :173-201
    const float a = norm_sq(r.dir);
    const float b = 2.0f * dot(r.dir, r.pos);
    const float c = norm_sq(r.pos) - (radius * radius);
    const float disc = b * b - 4.0f * a * c;
    if (disc <= 0.0) return std::nullopt;
    const float sq = std::sqrt(disc);
    const float t0 = (-b - sq) / (2.0f * a);
    const float t1 = (-b + sq) / (2.0f * a);
    float t;
    if (t0 > kGeoEps) {
        t = (t1 > kGeoEps) ? std::min(t0, t1) : t0;
    } else {
        if (t1 > kGeoEps) t = t1;
        else return std::nullopt;
    }
    return r.pos + t * r.dir;
}

std::optional<V3> hit_rect(const Ray& r, V3 half) {   // intersect(Ray, Rectangle) :203-252
    const V3 lo = -r.pos - half;   // negative-facing corner relative to the ray origin
    const V3 hi = -r.pos + half;
    const float d[3] = {r.dir.x, r.dir.y, r.dir.z};
    const float l[3] = {lo.x, lo.y, lo.z};
    const float h[3] = {hi.x, hi.y, hi.z};
    bool have = false;
    float best = 0.0f;
    for (int k = 0; k < 3; ++k) {          // x, y, z; for each the low corner, then the high one
        for (int side = 0; side < 2; ++side) {
            if (std::abs(d[k]) < kGeoEps) continue;
            const float t = (side == 0 ? l[k] : h[k]) / d[k];
            if (!(t > 0.0f)) continue;
            bool ok = true;
            for (int j = 0; j < 3 && ok; ++j) {
                if (j == k) continue;
                const float v = d[j] * t;
                ok = (v >= l[j]) && (v <= h[j]);
            }
            if (!ok) continue;
            best = have ? std::min(best, t) : t;
            have = true;
        }
    }
    if (!have) return std::nullopt;
    return r.pos + (best * r.dir);
}

std::optional<V3> hit_aabb(const Ray& r, const Box& b) {   // :254-261
    const Ray s{r.dir, r.pos - b.center};
    auto p = hit_rect(s, b.half);
    if (!p) return std::nullopt;
    return *p + b.center;
}

inline bool inside_rect(V3 p, V3 half) {   // :267-272
    return (std::abs(p.x) <= half.x) && (std::abs(p.y) <= half.y) && (std::abs(p.z) <= half.z);
}
inline bool inside_aabb(V3 p, const Box& b) { return inside_rect(p - b.center, b.half); }   // :274-276

V3 rect_normal(V3 p, V3 half) {   // Rectangle::normal :77-103
    const float ax = std::abs(p.x / half.x), ay = std::abs(p.y / half.y), az = std::abs(p.z / half.z);
    auto sgn = [](float v) { return v > 0.0f ? 1.0f : -1.0f; };
    if (ax > ay) {
        if (ax > az) return v3(sgn(p.x), 0.0f, 0.0f);
        return v3(0.0f, 0.0f, sgn(p.z));
    }
    if (ay > az) return v3(0.0f, sgn(p.y), 0.0f);
    return v3(0.0f, 0.0f, sgn(p.z));
}

template <class Rng>
std::pair<float, float> random_point_in_circle(Rng& rng) {   // :278-288
    for (;;) {
        const float x = rng.uniform(-1.0f, 1.0f);
        const float y = rng.uniform(-1.0f, 1.0f);
        if (x * x + y * y <= 1.0f) return {x, y};
    }
}

std::pair<V3, V3> orthogonal_pair(V3 v) {   // :290-299
    v = unit(v);
    const V3 xa = v3(1.0f, 0.0f, 0.0f), ya = v3(0.0f, 1.0f, 0.0f);
    const bool aligned_to_x = (norm_sq(cross(v, xa)) < 0.1);   // double literal in the reference
    const V3 w = aligned_to_x ? ya : xa;
    const V3 v1 = cross(w, v);
    const V3 v2 = cross(v1, v);
    return {unit(v1), unit(v2)};
}

template <class Rng>
V3 cosine_hemisphere(V3 n, Rng& rng) {   // randomPointOnHemisphereCosine :320-341
    n = unit(n);
    for (;;) {
        const float x = rng.uniform(-1.0f, 1.0f);
        const float y = rng.uniform(-1.0f, 1.0f);
        const float l2 = (x * x) + (y * y);
        if (l2 > 1.0f) continue;
        const float z = std::sqrt(1.0f - l2);
        const auto [v1, v2] = orthogonal_pair(n);
        return (z * n) + (x * v1) + (y * v2);
    }
}

// ------------------------------------------------------------------------------------------
// BasicMaterial::deflect  (src/BasicMaterial.cpp:97-159)
// ------------------------------------------------------------------------------------------
struct Material {
    float diffuse, specular, sharpness;
    V3 albedo, emitted;
    float transmittance, ior;
    V3 internal;
};
Material load_material(const float* p) {
    return Material{p[0], p[1], p[2], v3(p[3], p[4], p[5]), v3(p[6], p[7], p[8]), p[9], p[10],
                    v3(p[11], p[12], p[13])};
}
struct Bounce {
    V3 emitted, attenuation, dir, normal;
};

template <class Rng>
Bounce deflect(const Material& m, V3 in, V3 n, Rng& rng) {
    const float s = dot(in, n);
    if (s >= 0.0f) {   // struck from inside (:100-121)
        const V3 v = unit(in + ((dot(in, n) * (1.0f - m.ior)) * n));
        if (dot(v, n) >= 0.0f) return Bounce{v3(0, 0, 0), v3(1, 1, 1), v, n};
        return Bounce{v3(0.0f, 1.0f, 1.0f), v3(0, 0, 0), mirror(in, -n), n};
    }
    const float reflection = m.diffuse + m.specular;
    const float options = reflection + m.transmittance;
    const float which = rng.uniform(0.0f, options);   // :125-126
    if (which < reflection) {
        if (which < m.diffuse) {
            const V3 v = cosine_hemisphere(n, rng);
            return Bounce{m.emitted, m.albedo, v, n};
        }
        const V3 sc = (1.0f - m.sharpness) * cosine_hemisphere(n, rng);
        const V3 v = unit(mirror(in, n) + sc);
        return Bounce{m.emitted, m.albedo, v, n};
    }
    const V3 v = unit(in + ((dot(in, n) * (1.0f - 1.0f / m.ior)) * n));
    return Bounce{m.emitted, m.albedo, v, n};
}

// ------------------------------------------------------------------------------------------
// Demo procedural helpers and CraterBallObject  (PathTracerDemo.cpp:26-116, 352-483)
// ------------------------------------------------------------------------------------------
inline float tri_wave(float x) {   // triangle() :26-28
    return -1.0f + 4.0f * std::abs(x - 0.25f - std::floor(x + 0.25f));
}

float fbm_noise(const Aff& T, V3 p) {   // noise() :56-70
    float v = 0.0f;
    float k = 0.02f;
    const float kk = 0.9f;
    for (int i = 0; i < 8; ++i) {
        p = xform_pos(T, p);
        v += k * (tri_wave(p.x) + tri_wave(p.y) + tri_wave(p.z));
        p.x *= 1.1f;
        p.y *= 1.1f;
        p.z *= 1.1f;
        k *= kk;
    }
    return v;
}

// for_each_nearby_scattered_point :79-116.  The hash relies on wrapping int arithmetic; the jitter
// goes through double because of the `-1.0` literal and is narrowed back by Pos's float constructor.
template <class F>
void scattered_points(V3 location, float scatter, int radius, F&& f) {
    const V3 origin = v3(std::floor(location.x), std::floor(location.y), std::floor(location.z));
    const int lo = -radius + 1, hi = radius;
    for (int i = lo; i <= hi; ++i)
        for (int j = lo; j <= hi; ++j)
            for (int k = lo; k <= hi; ++k) {
                const V3 g = origin + v3(static_cast<float>(i), static_cast<float>(j), static_cast<float>(k));
                uint32_t h = static_cast<uint32_t>(static_cast<int>(g.x)) * 885889u +
                             static_cast<uint32_t>(static_cast<int>(g.y)) * 745273u +
                             static_cast<uint32_t>(static_cast<int>(g.z)) * 1220689u;
                for (int r = 0; r < 4; ++r) {
                    h |= (h & 0x0f0f0f0fu) << 4;
                    h *= 31u;
                }
                const V3 c = v3(
                    static_cast<float>(g.x + scatter * 0.5f * (-1.0 + static_cast<float>(h & 0x3ff) / 512.0f)),
                    static_cast<float>(g.y + scatter * 0.5f * (-1.0 + static_cast<float>((h >> 10) & 0x3ff) / 512.0f)),
                    static_cast<float>(g.z + scatter * 0.5f * (-1.0 + static_cast<float>((h >> 20) & 0x3ff) / 512.0f)));
                f(c, h);
            }
}

struct SdfCounters {
    uint64_t evals = 0, near_evals = 0;
};

// voronoi() :118-154 with its default arguments (scatter 1, identity transform: the inverse of the identity
// and the products with it return their operand unchanged).
struct Voronoi {
    float sqr, next_sqr;
    uint32_t hash;
};
Voronoi voronoi(V3 location) {
    Voronoi v{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), 0u};
    scattered_points(location, 1.0f, 2, [&](V3 c, uint32_t h) {
        const float d = norm_sq(location - c);
        if (d < v.sqr) {
            v.next_sqr = std::min(v.sqr, v.next_sqr);
            v.sqr = d;
            v.hash = h;
        } else {
            v.next_sqr = std::min(d, v.next_sqr);
        }
    });
    return v;
}

float rect_sdf(V3 half, V3 p) {   // Rectangle::signedDistance src/Geometry.cpp:66-75
    const float dx = std::abs(p.x) - half.x, dy = std::abs(p.y) - half.y, dz = std::abs(p.z) - half.z;
    return norm(v3(std::max(dx, 0.0f), std::max(dy, 0.0f), std::max(dz, 0.0f))) + std::min(std::max(std::max(dx, dy), dz), 0.0f);
}

// RoughBoxObject :228-270: a box eroded by noise, cut by the walls of a voronoi foam.
float roughbox_sdf(V3 half, const Aff& NT, V3 pos) {   // :237-246
    const float voronoi_scale = 0.03723f;
    const Voronoi v = voronoi(div(pos, voronoi_scale));
    return std::max(rect_sdf(half, pos) - 0.3f + 2.0f * fbm_noise(NT, 1.0f * pos),
                    (std::sqrt(v.sqr) - std::sqrt(v.next_sqr) + 0.1f) * voronoi_scale);
}
// deflectLocalRay :248-259: the cell of a coarser foam colours the surface.
V3 foam_albedo(V3 pos, float voronoi_scale) {   // the same expression colours WeirdWallObject :333-339 (scale 0.018347)
    const Voronoi v = voronoi(div(pos, voronoi_scale));
    return v3(static_cast<float>(0.6f + 0.4f * static_cast<float>(v.hash & 0xff) / 256.0),
              static_cast<float>(0.3f + 0.5f * static_cast<float>((v.hash >> 8) & 0xff) / 256.0),
              static_cast<float>(0.1f + 0.4f * static_cast<float>((v.hash >> 16) & 0xff) / 256.0));
}

// voronoi() :118-154 in general: scatter and a Linear that shears the distance metric.
Voronoi voronoi(V3 location, float scatter, const Lin& transform) {
    Lin inv;
    inverse(transform, inv);   // `*transform.inverse()`
    location = mul(inv, location);
    Voronoi v{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), 0u};
    scattered_points(location, scatter, 2, [&](V3 c, uint32_t h) {
        const float d = norm_sq(mul(transform, location - c));
        if (d < v.sqr) {
            v.next_sqr = std::min(v.sqr, v.next_sqr);
            v.sqr = d;
            v.hash = h;
        } else {
            v.next_sqr = std::min(d, v.next_sqr);
        }
    });
    return v;
}

inline float smin(float a, float b, float k) {   // :72-76
    return -std::log2(std::exp2(-k * a) + std::exp2(-k * b)) / k;
}

// RoughSphereObject :156-226: a sphere blended with a box, cratered and bumped by noise.
float roughsphere_sdf(float radius, const Aff& NT, V3 pos) {   // :173-182
    const float r = 0.8f * radius;
    const float sd_box = rect_sdf(v3(r, r, r), pos);
    const float sd_sphere = norm(pos) - radius;   // Sphere::signedDistance src/Geometry.cpp:39-41
    const float sd = smin(sd_box, sd_sphere, 8.0f);
    const float craters = std::max(0.0f, -1.0f + 3.0f * fbm_noise(NT, 1.0f * pos));   // :165-167
    const float bumps = -0.15f + 0.5f * fbm_noise(NT, 4.0f * pos);                    // :169-171
    return std::min(sd + craters, sd + craters + bumps);
}
void roughsphere_material(float radius, const Aff& NT, V3 pos, Material& m);

// WeirdWallObject :273-349: a ground slab, a brick wall with an arch, all eroded by noise.
float weirdwall_sdf(const Aff& NT, V3 pos) {   // :281-328
    const float pi = 3.141592654f;
    const float sdf_ground_clean = rect_sdf(v3(50.0f, 0.2f, 10.0f), pos - v3(0.0f, 3.0f, 0.0f));
    const float sdf_wall_proxy = rect_sdf(v3(50.0f, 4.0f, 0.4f), pos);
    const float sdf_proxy = smin(sdf_ground_clean - 0.2f, sdf_wall_proxy, 2.0f);
    if (sdf_proxy > 0.1f) return sdf_proxy;

    const float sdf_ground = sdf_ground_clean + fbm_noise(NT, pos);
    const float dist_xy = std::hypot(pos.x, pos.y);
    const float angle_xy = std::atan2(pos.y, pos.x);
    const float ring_band = std::abs(dist_xy - 1.5f) - (std::abs(angle_xy + 0.5f * pi) < 0.18f ? 0.5f : 0.3f);
    const float spoke_spacing = 18.0f;
    const float scaled_spoke_angle = angle_xy * spoke_spacing * 0.5f / pi;
    const float spoke_angle = (scaled_spoke_angle - std::round(scaled_spoke_angle)) / spoke_spacing;
    const float ring_spokes = std::abs(spoke_angle * dist_xy) * (2.0f * pi) - 0.06f;
    const float arch_ring = std::max(ring_band, -ring_spokes);
    const float pillar_columns = std::abs(std::abs(pos.x) - 1.5f) - 0.3f;
    const float slice_spacing = 1.5f;
    const float pillar_slices = std::abs(pos.y * slice_spacing - std::round(pos.y * slice_spacing)) / slice_spacing - 0.06f;
    const float arch_pillars = std::max(pillar_columns, -pillar_slices);
    const float sdf_arch = std::max(pos.y < 0.0f ? arch_ring : arch_pillars, std::abs(pos.z) - 0.3f);
    const float arch_center = pos.y < 0.0f ? dist_xy : std::abs(pos.x);

    const Voronoi bricks = voronoi(pos + 1.5f * v3(fbm_noise(NT, pos), fbm_noise(NT, pos + v3(10.0f, 10.0f, 10.0f)), 0.0f), 0.3f,
                                   Lin{{1.0f, 0.5f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, 1.0f}});
    const float bricks_everywhere = std::max(-std::abs(std::sqrt(bricks.sqr) - std::sqrt(bricks.next_sqr)) + 0.05f, std::abs(pos.z) - 0.2f);
    const float brick_wall = -smin(-std::max(bricks_everywhere, -arch_center + 1.85f), 3.0f + pos.y + std::exp(-0.125f * (pos.x * pos.x)), 2.0f);

    return smin(std::min(brick_wall, sdf_arch), sdf_ground, 3.0f) + 0.23f * fbm_noise(NT, pos * 5.1712f) + 0.06f * fbm_noise(NT, pos * 14.81552f);
}
float craterball_sdf(V3 off, const Aff& NT, V3 pos, SdfCounters* cnt) {   // :361-466
    if (cnt) cnt->evals++;
    const float sdf_sphere = norm(pos) - 0.5f;
    if (sdf_sphere > 0.1) return sdf_sphere;
    if (cnt) cnt->near_evals++;

    float crater = 0.0f, tiny = 0.0f, fallout = 0.0f, min_age = 1.0f;
    {   // big craters :375-409
        const float scale = 2.0f;
        const V3 query = (-unit(pos) * scale) + off;
        scattered_points(query, 0.6f, 2, [&](V3 point, uint32_t hash) {
            const float t = static_cast<float>(hash & 0xffff) / static_cast<float>(0xffff);
            const float age = static_cast<float>((hash >> 16) & 0xffff) / static_cast<float>(0xffff);
            const float size = 0.05f + 0.4f * t * t;
            const V3 on_sphere = (unit(point - off) * scale) + off;
            const float dist = norm(query - on_sphere);
            if (norm_sq(on_sphere - point) < 0.1f) {
                const float c = 20.0f * (dist - size);
                float depth = std::tanh(c);
                if (c > 1e-3f) depth = 0.9f * t * depth / (0.02f / c + c * c);
                crater -= 0.02f * depth;
                if (dist >= size) {
                    const float lin = std::max(0.0f, 1.2f - dist / (size * 5.0f));
                    const float falloff =
                        std::pow(lin, 2.0f + 3.0f * fbm_noise(NT, on_sphere + 3.0f * size * unit(query - on_sphere)));
                    fallout += 0.1f * falloff * (0.2f + 0.4f * depth) *
                               fbm_noise(NT, on_sphere + 40.0f * size * unit(query - on_sphere));
                } else {
                    min_age = std::min(age, min_age);
                }
            }
        });
    }
    {   // small craters :412-437
        const float scale = 10.0f;
        const V3 query = (-unit(pos) * scale) + off;
        scattered_points(query, 1.0f, 1, [&](V3 point, uint32_t hash) {
            const float t = static_cast<float>(hash & 0xffff) / static_cast<float>(0xffff);
            const float age = static_cast<float>((hash >> 16) & 0xffff) / static_cast<float>(0xffff);
            if (age > min_age) return;
            const float size = 0.05f + 0.35f * t * t;
            const V3 on_sphere = (unit(point - off) * scale) + off;
            if (norm_sq(on_sphere - point) < 0.1f) {
                const float dist = norm(query - on_sphere);
                const float c = 50.0f * (dist - size);
                if (c < 0.0f) {
                    min_age = std::min(age, min_age);
                    crater = std::max(-0.005f * std::tanh(c), crater);
                }
            }
        });
    }
    {   // tiny craters :440-463
        const float scale = 60.0f;
        const V3 query = (-unit(pos) * scale) + off;
        scattered_points(query, 1.0f, 1, [&](V3 point, uint32_t hash) {
            const float t = static_cast<float>(hash & 0xffff) / static_cast<float>(0xffff);
            const float age = static_cast<float>((hash >> 16) & 0xffff) / static_cast<float>(0xffff);
            const float size = 0.02f + 0.35f * t;
            const V3 on_sphere = (unit(point - off) * scale) + off;
            if (norm_sq(on_sphere - point) < 0.1f) {
                if (age > min_age) return;
                tiny = std::max(-0.004f * size * (-0.5f + 0.5f * std::tanh(50.0f * (norm(query - on_sphere) - size))), tiny);
            }
        });
    }
    return sdf_sphere + crater + tiny + fallout + 0.01f * fbm_noise(NT, (pos + off) * 20.0f) +
           0.4f * fbm_noise(NT, (pos + off) * 1.5f);
}

// ------------------------------------------------------------------------------------------
// Objects  (include/Object.hpp, src/Object.cpp)
// ------------------------------------------------------------------------------------------
struct Prim {
    uint32_t kind;
    Aff fwd, inv;
    float p[9];
    uint32_t material;
};

struct SceneData {
    std::vector<Prim> prims;
    std::vector<Material> mats;
    Aff noise;
    std::vector<uint32_t> order;   // closest-hit visiting order (identity, or the reference tree's leaf order)
};

const Box kCraterBox{v3(0, 0, 0), v3(1.0f, 1.0f, 1.0f)};   // CraterBallObject::localBoundingBox :474-479

template <class Sdf>
std::optional<V3> sdf_march(const Ray& r, const Box& bb, Sdf&& sdf) {   // SDFObjectCRTP::hitLocalRay include/Object.hpp:135-167
    std::optional<V3> p0;
    if (inside_aabb(r.pos, bb)) p0 = r.pos;
    else p0 = hit_aabb(r, bb);
    if (!p0) return std::nullopt;
    V3 p = *p0;
    float d = sdf(p);
    const bool sign = d > 0.0f;
    for (int i = 0; i < 256; ++i) {
        const float d2 = sdf(p);
        if ((d2 > 0.0f) != sign) {
            d *= 0.5;
            p = p - d * r.dir;
            continue;
        }
        d = d2;
        if (std::abs(d) < 1e-4f) return p;
        p = p + std::abs(d) * r.dir;
        if (!inside_aabb(p, bb)) return std::nullopt;
    }
    return std::nullopt;
}

template <class Sdf>
V3 sdf_normal(V3 pos, Sdf&& sdf) {   // signedDistanceNormal include/Object.hpp:118-132
    const float delta = 1e-3f;
    const float d0 = sdf(pos);
    return unit(div(v3(sdf(pos + v3(delta, 0, 0)) - d0, sdf(pos + v3(0, delta, 0)) - d0, sdf(pos + v3(0, 0, delta)) - d0), delta));
}

// deflectLocalRay :184-199: the material is made of SDF normals taken at two other points.
void roughsphere_material(float radius, const Aff& NT, V3 pos, Material& m) {
    auto nrm = [&](V3 q) { return sdf_normal(q, [&](V3 x) { return roughsphere_sdf(radius, NT, x); }); };
    V3 n1 = nrm((-pos) * 1.724674f + v3(3.474577f, 1.734674f, 2.713652f));
    V3 n2 = nrm(pos * 2.446744f + v3(1.734674f, 2.713652f, 4.137371f));
    n1 = v3(0.5f, 0.5f, 0.5f) + 0.5f * n1;
    n2 = v3(0.5f, 0.5f, 0.5f) + 0.5f * n2;
    m.albedo = n1;
    m.diffuse = n2.x;
    m.specular = 1.0f - n2.x;
    m.sharpness = n2.y;
    const float k = 0.125f * n2.z;
    m.emitted = v3(k * m.albedo.x, k * m.albedo.y, k * m.albedo.z);   // operator*(float, Color) src/Color.cpp:25-31
}

// FractalObject (src/Object.cpp:160-252): a 3 x 3 x 3 lattice of spheres of radius 0.2.
float fractal_sdf(V3 p) {   // :198-252
    auto f = [](float v) {
        const float l = 0.5f, r = 2.0f;
        if (v < -l) return v + l;
        if (v > l) return v - l;
        v *= r;
        return (v - std::round(v)) / r;
    };
    return norm(v3(f(p.x), f(p.y), f(p.z))) - 0.2f;
}

// :160-175 -- the march starts where the LOCAL ray enters the object's WORLD bounding box
// (`intersect(ray, getBoundingBox())`), takes up to 256 steps of the signed distance and never leaves early.
std::optional<V3> fractal_march(const Ray& r, const Box& world_box) {
    const auto p0 = hit_aabb(r, world_box);
    if (!p0) return std::nullopt;
    V3 p = *p0;
    for (int i = 0; i < 256; ++i) {
        const float d = fractal_sdf(p);
        if (d < 1e-3f) return p;
        p = p + d * r.dir;
    }
    return std::nullopt;
}

V3 fractal_normal(V3 pos) {   // :177-189, central differences
    const float delta = 1e-3f;
    const V3 dx = v3(delta, 0, 0), dy = v3(0, delta, 0), dz = v3(0, 0, delta);
    return unit(div(v3(fractal_sdf(pos + dx) - fractal_sdf(pos - dx), fractal_sdf(pos + dy) - fractal_sdf(pos - dy),
                       fractal_sdf(pos + dz) - fractal_sdf(pos - dz)), delta));
}

Box world_bbox(uint32_t kind, const Aff& T, const float* p);

std::optional<V3> hit_local(uint32_t kind, const float* p, const Aff& noise, const Aff& fwd, const Ray& r, SdfCounters* cnt) {
    switch (kind) {
        case ORC_FRACTAL: return fractal_march(r, world_bbox(kind, fwd, p));
        case ORC_ROUGHBOX: {
            const V3 half = v3(p[0], p[1], p[2]);
            return sdf_march(r, Box{v3(0, 0, 0), 2.0f * half}, [&](V3 q) { return roughbox_sdf(half, noise, q); });   // localBoundingBox :261-266
        }
        case ORC_ROUGHSPHERE: {
            const float h = 1.2f * p[0];   // localBoundingBox :216-221: `1.2f * radius * Vec(1, 1, 1)`
            return sdf_march(r, Box{v3(0, 0, 0), v3(h * 1.0f, h * 1.0f, h * 1.0f)}, [&](V3 q) { return roughsphere_sdf(p[0], noise, q); });
        }
        case ORC_WEIRDWALL:
            return sdf_march(r, Box{v3(0, 0, 0), v3(50.0f, 20.0f, 20.0f)}, [&](V3 q) { return weirdwall_sdf(noise, q); });   // :343-348
        case ORC_SPHERE: return hit_sphere(r, p[0]);                                   // src/Object.cpp:116-118
        case ORC_BOX: return hit_rect(r, v3(p[0], p[1], p[2]));                        // :141-143
        case ORC_TRIANGLE: return hit_triangle(r, v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), v3(p[6], p[7], p[8]));   // :88-90
        case ORC_CRATERBALL: {
            const V3 off = v3(p[0], p[1], p[2]);
            return sdf_march(r, kCraterBox, [&](V3 q) { return craterball_sdf(off, noise, q, cnt); });
        }
    }
    return std::nullopt;
}

V3 normal_local(uint32_t kind, const float* p, const Aff& noise, V3 pos, SdfCounters* cnt) {
    switch (kind) {
        case ORC_SPHERE: return unit(pos);                                             // Sphere::normal src/Geometry.cpp:43-45
        case ORC_BOX: return rect_normal(pos, v3(p[0], p[1], p[2]));
        case ORC_TRIANGLE: {                                                           // Triangle::normal :18-20
            const V3 a = v3(p[0], p[1], p[2]), b = v3(p[3], p[4], p[5]), c = v3(p[6], p[7], p[8]);
            return unit(cross(b - a, c - a));
        }
        case ORC_CRATERBALL: {
            const V3 off = v3(p[0], p[1], p[2]);
            return sdf_normal(pos, [&](V3 q) { return craterball_sdf(off, noise, q, cnt); });
        }
        case ORC_FRACTAL: return fractal_normal(pos);
        case ORC_ROUGHBOX: {
            const V3 half = v3(p[0], p[1], p[2]);
            return sdf_normal(pos, [&](V3 q) { return roughbox_sdf(half, noise, q); });
        }
        case ORC_ROUGHSPHERE: return sdf_normal(pos, [&](V3 q) { return roughsphere_sdf(p[0], noise, q); });
        case ORC_WEIRDWALL: return sdf_normal(pos, [&](V3 q) { return weirdwall_sdf(noise, q); });
    }
    return v3(0, 0, 1);
}

Box local_bbox(uint32_t kind, const float* p) {   // *::getLocalBoundingBox src/Object.cpp:96-158
    switch (kind) {
        case ORC_SPHERE: return Box{v3(0, 0, 0), v3(p[0] + 1e-3f, p[0] + 1e-3f, p[0] + 1e-3f)};
        case ORC_BOX: return Box{v3(0, 0, 0), v3(p[0] + 1e-3f, p[1] + 1e-3f, p[2] + 1e-3f)};
        case ORC_TRIANGLE: {
            const V3 lo = v3(std::min({p[0], p[3], p[6]}) - 1e-3f, std::min({p[1], p[4], p[7]}) - 1e-3f,
                             std::min({p[2], p[5], p[8]}) - 1e-3f);
            const V3 hi = v3(std::max({p[0], p[3], p[6]}) + 1e-3f, std::max({p[1], p[4], p[7]}) + 1e-3f,
                             std::max({p[2], p[5], p[8]}) + 1e-3f);
            return box_from_corners(lo, hi);
        }
        case ORC_FRACTAL: return Box{v3(0, 0, 0), v3(2.0f, 2.0f, 2.0f)};   // src/Object.cpp:191-196
        case ORC_ROUGHBOX: return Box{v3(0, 0, 0), 2.0f * v3(p[0], p[1], p[2])};
        case ORC_ROUGHSPHERE: {
            const float h = 1.2f * p[0];
            return Box{v3(0, 0, 0), v3(h * 1.0f, h * 1.0f, h * 1.0f)};
        }
        case ORC_WEIRDWALL: return Box{v3(0, 0, 0), v3(50.0f, 20.0f, 20.0f)};
        default: return kCraterBox;
    }
}

Box world_bbox(uint32_t kind, const Aff& T, const float* p) {   // Object::getBoundingBox src/Object.cpp:46-80
    const Box bb = local_bbox(kind, p);
    const float big = std::numeric_limits<float>::max(), small = std::numeric_limits<float>::lowest();
    V3 lo = v3(big, big, big), hi = v3(small, small, small);
    const V3 h = bb.half;
    const V3 corners[8] = {v3(h.x, h.y, h.z),  v3(h.x, h.y, -h.z),  v3(h.x, -h.y, h.z),  v3(h.x, -h.y, -h.z),
                           v3(-h.x, h.y, h.z), v3(-h.x, h.y, -h.z), v3(-h.x, -h.y, h.z), v3(-h.x, -h.y, -h.z)};
    for (const V3& c : corners) {
        const V3 u = xform_pos(T, bb.center + c);
        lo = v3(std::min(lo.x, u.x), std::min(lo.y, u.y), std::min(lo.z, u.z));
        hi = v3(std::max(hi.x, u.x), std::max(hi.y, u.y), std::max(hi.z, u.z));
    }
    return box_from_corners(lo, hi);
}

// Object::hitRay src/Object.cpp:22-33 -- the local direction is not renormalised.
std::optional<V3> hit_object(const SceneData& s, const Prim& o, const Ray& r, SdfCounters* cnt) {
    const Ray local{xform_vec(o.inv, r.dir), xform_pos(o.inv, r.pos)};
    auto h = hit_local(o.kind, o.p, s.noise, o.fwd, local, cnt);
    if (!h) return std::nullopt;
    return xform_pos(o.fwd, *h);
}

// Object::deflectRay src/Object.cpp:35-44 -- only the direction is taken back to world space.
template <class Rng>
Bounce deflect_object(const SceneData& s, const Prim& o, const Ray& r, Rng& rng, SdfCounters* cnt) {
    const Ray local{xform_vec(o.inv, r.dir), xform_pos(o.inv, r.pos)};
    const V3 n = normal_local(o.kind, o.p, s.noise, local.pos, cnt);
    Material m = s.mats[o.material];
    if (o.kind == ORC_ROUGHBOX) m.albedo = foam_albedo(local.pos, 0.2f);   // `m.setReflectedAbsorption(...)` :251-257
    if (o.kind == ORC_WEIRDWALL) m.albedo = foam_albedo(local.pos, 0.018347f);   // :333-339
    if (o.kind == ORC_ROUGHSPHERE) roughsphere_material(o.p[0], s.noise, local.pos, m);
    Bounce b = deflect(m, local.dir, n, rng);
    b.dir = xform_vec(o.fwd, b.dir);
    return b;
}

struct Hit {
    int32_t prim = -1;
    V3 pos{0, 0, 0};
    float t = 0.0f;
};

// ObjectTree::Tree::hit src/ObjectTree.cpp:22-88.  The tree never culls: every leaf is tested and the
// pairwise `tb < ta` merges return the smallest t, ties going to the earlier leaf.  A scan in leaf
// order with a strict `<` is the same function.
Hit closest_hit(const SceneData& s, const Ray& r, SdfCounters* cnt) {
    Hit best;
    for (uint32_t idx : s.order) {
        auto p = hit_object(s, s.prims[idx], r, cnt);
        if (!p) continue;
        const float t = dot(*p - r.pos, r.dir);   // :43-44
        if (best.prim < 0 || t < best.t) {
            best.prim = static_cast<int32_t>(idx);
            best.pos = *p;
            best.t = t;
        }
    }
    return best;
}

struct TraceCounters {
    uint64_t rays = 0;
    SdfCounters sdf;
};

// Scene::trace src/Scene.cpp:18-78
template <class Rng>
V3 trace(const SceneData& s, Ray ray, uint32_t depth, Rng& rng, TraceCounters* tc) {
    V3 color = v3(0, 0, 0), att = v3(1.0f, 1.0f, 1.0f);
    for (uint32_t i = 0; i < depth; ++i) {
        rng.begin_stage(1 + i);
        if (tc) tc->rays++;
        const Hit h = closest_hit(s, ray, tc ? &tc->sdf : nullptr);
        if (h.prim < 0) return color;
        ray.pos = h.pos;
        const Bounce b = deflect_object(s, s.prims[h.prim], ray, rng, tc ? &tc->sdf : nullptr);
        ray.pos = v3(ray.pos.x + 1e-3f * b.dir.x, ray.pos.y + 1e-3f * b.dir.y, ray.pos.z + 1e-3f * b.dir.z);   // :54
        color.x += b.emitted.x * att.x;
        color.y += b.emitted.y * att.y;
        color.z += b.emitted.z * att.z;
        att.x *= b.attenuation.x;
        att.y *= b.attenuation.y;
        att.z *= b.attenuation.z;
        ray.dir = b.dir;
        if (att.x + att.z + att.y < 1e-6f) return color;   // r + b + g, :73
    }
    return color;
}

// Camera::getViewRay src/Camera.cpp:65-81
struct Cam {
    Aff T;
    float aspect, fov, focal, blur;
};
Cam load_cam(const float* c) { return Cam{load_aff(c), c[12], c[13], c[14], c[15]}; }

Ray view_ray_with_lens(const Cam& c, float sx, float sy, float rx, float ry) {
    const float x = sx * 2.0f - 1.0f;
    const float y = sy * 2.0f - 1.0f;
    const V3 sp = c.aspect > 1.0f ? v3(x, y / c.aspect, 0.0f) : v3(x * c.aspect, y, 0.0f);
    const float blur_rad = c.blur * std::max(c.aspect, 1.0f / c.aspect);
    const V3 blur = v3(rx * blur_rad, ry * blur_rad, 0.0f);
    const float fov_scale = std::tan(c.fov * 3.141592654f / 180.0f);
    const V3 view = v3(fov_scale * sp.x, fov_scale * sp.y, 1.0f) + div(blur, c.focal);
    return Ray{unit(xform_vec(c.T, view)), xform_pos(c.T, sp - blur)};
}
// The reference always draws a lens point.  The counter generator skips the draw when the blur radius
// is zero: the lens point is then multiplied by zero and cannot change the ray.
Ray view_ray(const Cam& c, float sx, float sy, MinstdRng& rng) {
    const auto [rx, ry] = random_point_in_circle(rng);
    return view_ray_with_lens(c, sx, sy, rx, ry);
}
Ray view_ray(const Cam& c, float sx, float sy, CounterRng& rng) {
    float rx = 0.0f, ry = 0.0f;
    if (c.blur != 0.0f) std::tie(rx, ry) = random_point_in_circle(rng);
    return view_ray_with_lens(c, sx, sy, rx, ry);
}

// ------------------------------------------------------------------------------------------
// ObjectTree::Tree::makeTree  (src/ObjectTree.cpp:90-184) -- leaf order only
// ------------------------------------------------------------------------------------------
inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline float volume(V3 half) { return half.x * half.y * half.z * 8.0f; }   // Rectangle::volume src/Geometry.cpp:62-64

bool tree_order(const std::vector<Box>& boxes, std::vector<uint32_t> objs, std::vector<uint32_t>& out) {
    if (objs.empty()) return true;
    if (objs.size() <= 2) {
        for (uint32_t o : objs) out.push_back(o);
        return true;
    }
    int best_dim = -1;
    size_t best_idx = 0;
    float best_cost = std::numeric_limits<float>::max();
    const int dims[3] = {0, 2, 1};   // x, z, y (:159-161)
    for (int dim : dims) {
        for (size_t i = 1; i < objs.size(); ++i) {
            const float m = axis_of(boxes[objs[i]].center, dim);
            std::optional<Box> a, b;
            for (size_t j = 0; j < objs.size(); ++j) {
                const Box& bb2 = boxes[objs[j]];
                if (axis_of(bb2.center, dim) < m) a = a ? box_containing(*a, bb2) : bb2;
                else b = b ? box_containing(*b, bb2) : bb2;
            }
            if (!(a && b)) continue;
            const float v_inner = volume(a->half) + volume(b->half);
            const float v_total = volume(box_containing(*a, *b).half);
            const float volume_cost = v_inner / v_total;
            const float ratio = static_cast<float>(i) / static_cast<float>(objs.size() - i);
            const float split_cost = std::max(ratio, 1.0f / ratio);
            const float cost = 0.5f * volume_cost + (1.0f - 0.5f) * split_cost;
            if (cost < best_cost) {
                best_cost = cost;
                best_dim = dim;
                best_idx = i;
            }
        }
    }
    if (best_dim < 0) return false;
    const float m = axis_of(boxes[objs[best_idx]].center, best_dim);
    std::vector<uint32_t> a, b;
    for (uint32_t o : objs) (axis_of(boxes[o].center, best_dim) < m ? a : b).push_back(o);
    return tree_order(boxes, std::move(a), out) && tree_order(boxes, std::move(b), out);
}

// ------------------------------------------------------------------------------------------
// Renderer::doWork  (src/Renderer.cpp:54-137)
// ------------------------------------------------------------------------------------------
struct RenderJob {
    const SceneData* scene;
    Cam cam;
    uint32_t W, H, bounces, spp, first_sample;
    uint64_t seed;
    bool full_width;
    float* out;
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> paths{0}, rays{0}, evals{0}, near_evals{0};
};

template <bool kCounter>
void render_worker(RenderJob& job, uint32_t worker) {
    const float max_w = static_cast<float>(job.W - 1), max_h = static_cast<float>(job.H - 1);
    const float spp_inv = 1.0f / static_cast<float>(job.spp);
    const float dx = 1.0f / max_w, dy = 1.0f / max_h;
    MinstdRng thread_rng(job.seed + worker);
    TraceCounters tc;
    uint64_t paths = 0;
    const size_t per_task = 8;   // s_pixelsPerTask include/Renderer.hpp:39
    const size_t tasks_per_row = job.full_width ? (job.W + per_task - 1) / per_task : job.W / per_task;
    for (;;) {
        const size_t task = job.next.fetch_add(1);
        const size_t x0 = per_task * (task % tasks_per_row);
        const size_t y = task / tasks_per_row;
        if (y >= job.H) break;
        const size_t x1 = std::min<size_t>(x0 + per_task, job.W);
        const float py = static_cast<float>(y) / max_h;
        for (size_t x = x0; x < x1; ++x) {
            const float px = static_cast<float>(x) / max_w;
            V3 acc = v3(0, 0, 0);
            for (uint32_t i = 0; i < job.spp; ++i) {
                V3 c;
                if constexpr (kCounter) {
                    CounterRng rng(job.seed, static_cast<uint32_t>(y * job.W + x), job.first_sample + i);
                    rng.begin_stage(0);
                    const float sx = px + rng.uniform(-0.5f * dx, 0.5f * dx);
                    const float sy = py + rng.uniform(-0.5f * dy, 0.5f * dy);
                    const Ray r = view_ray(job.cam, sx, sy, rng);
                    c = trace(*job.scene, r, job.bounces, rng, &tc);
                } else {
                    const float sx = px + thread_rng.uniform(-0.5f * dx, 0.5f * dx);
                    const float sy = py + thread_rng.uniform(-0.5f * dy, 0.5f * dy);
                    const Ray r = view_ray(job.cam, sx, sy, thread_rng);
                    c = trace(*job.scene, r, job.bounces, thread_rng, &tc);
                }
                ++paths;
                acc.x += c.x;
                acc.y += c.y;
                acc.z += c.z;
            }
            float* px_out = job.out + 3 * (y * job.W + x);
            px_out[0] = acc.x * spp_inv;
            px_out[1] = acc.y * spp_inv;
            px_out[2] = acc.z * spp_inv;
        }
    }
    job.paths += paths;
    job.rays += tc.rays;
    job.evals += tc.sdf.evals;
    job.near_evals += tc.sdf.near_evals;
}

}   // namespace

struct orc_scene {
    SceneData d;
};

// Rays that do not share a random-number sequence are independent: large scenes (every ray scans every
// object) spread them over the host's cores.  Results do not depend on the split.
template <class F>
static void for_each_ray(uint32_t n, uint64_t work_per_ray, F&& body) {
    unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (static_cast<uint64_t>(n) * work_per_ray < (1u << 22) || n < 2 * threads) threads = 1;
    if (threads == 1) {
        for (uint32_t i = 0; i < n; ++i) body(i);
        return;
    }
    std::vector<std::thread> pool;
    std::atomic<uint32_t> next{0};
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            for (;;) {
                const uint32_t i0 = next.fetch_add(16);
                if (i0 >= n) return;
                for (uint32_t i = i0; i < std::min(n, i0 + 16); ++i) body(i);
            }
        });
    for (auto& th : pool) th.join();
}

extern "C" {

void orc_rotation_x(float a, float out[9]) { std::memcpy(out, rotation_x(a).m, sizeof(float) * 9); }
void orc_rotation_y(float a, float out[9]) { std::memcpy(out, rotation_y(a).m, sizeof(float) * 9); }
void orc_rotation_z(float a, float out[9]) { std::memcpy(out, rotation_z(a).m, sizeof(float) * 9); }

void orc_linear_mul(const float a[9], const float b[9], float out[9]) {
    Lin A, B;
    std::memcpy(A.m, a, sizeof(A.m));
    std::memcpy(B.m, b, sizeof(B.m));
    const Lin r = mul(A, B);
    std::memcpy(out, r.m, sizeof(r.m));
}

void orc_affine_mul(const float a[12], const float b[12], float out[12]) {
    store_aff(mul(load_aff(a), load_aff(b)), out);
}

int orc_affine_inverse(const float fwd[12], float inv[12]) {
    Aff r;
    if (!inverse(load_aff(fwd), r)) return 0;
    store_aff(r, inv);
    return 1;
}

void orc_noise_transform(const float t[3], const float r[3], float out[12]) {
    // Translation(t) * RotationX(r1) * RotationY(r2) * RotationZ(r3); Affine*Linear keeps the
    // translation (src/LinearAlgebra.cpp:405-407).
    Aff a{identity(), v3(t[0], t[1], t[2])};
    a.l = mul(a.l, rotation_x(r[0]));
    a.l = mul(a.l, rotation_y(r[1]));
    a.l = mul(a.l, rotation_z(r[2]));
    store_aff(a, out);
}

orc_scene* orc_scene_create(uint32_t n, const uint32_t* kind, const float* fwd, const float* inv, const float* params,
                            const uint32_t* material, uint32_t nm, const float* materials, const float* noise12) {
    auto* s = new orc_scene;
    s->d.prims.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        Prim& p = s->d.prims[i];
        p.kind = kind[i];
        p.fwd = load_aff(fwd + 12 * i);
        p.inv = load_aff(inv + 12 * i);
        std::memcpy(p.p, params + 9 * i, sizeof(p.p));
        p.material = material[i];
    }
    s->d.mats.resize(nm);
    for (uint32_t i = 0; i < nm; ++i) s->d.mats[i] = load_material(materials + 14 * i);
    if (noise12) s->d.noise = load_aff(noise12);
    else s->d.noise = Aff{identity(), v3(0, 0, 0)};
    s->d.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) s->d.order[i] = i;
    return s;
}

void orc_scene_free(orc_scene* s) { delete s; }

int orc_scene_use_reference_tree_order(orc_scene* s, uint32_t* out_order) {
    std::vector<Box> boxes;
    boxes.reserve(s->d.prims.size());
    for (const Prim& p : s->d.prims) boxes.push_back(world_bbox(p.kind, p.fwd, p.p));
    std::vector<uint32_t> all(s->d.prims.size()), out;
    for (uint32_t i = 0; i < all.size(); ++i) all[i] = i;
    if (!tree_order(boxes, all, out)) return -1;
    s->d.order = out;
    if (out_order) std::memcpy(out_order, out.data(), out.size() * sizeof(uint32_t));
    return 0;
}

void orc_world_bbox(uint32_t kind, const float fwd[12], const float params[9], float out6[6]) {
    const Box b = world_bbox(kind, load_aff(fwd), params);
    out6[0] = b.center.x;
    out6[1] = b.center.y;
    out6[2] = b.center.z;
    out6[3] = b.half.x;
    out6[4] = b.half.y;
    out6[5] = b.half.z;
}

void orc_hit_local(uint32_t kind, const float params[9], const float noise12[12], uint32_t n, const float* pos,
                   const float* dir, uint8_t* out_hit, float* out_pos) {
    const Aff nt = noise12 ? load_aff(noise12) : Aff{identity(), v3(0, 0, 0)};
    for (uint32_t i = 0; i < n; ++i) {
        const Ray r{v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2])};
        const auto h = hit_local(kind, params, nt, Aff{identity(), v3(0, 0, 0)}, r, nullptr);   // identity transformation
        out_hit[i] = h ? 1 : 0;
        const V3 p = h ? *h : v3(0, 0, 0);
        out_pos[3 * i] = p.x;
        out_pos[3 * i + 1] = p.y;
        out_pos[3 * i + 2] = p.z;
    }
}

void orc_normal_local(uint32_t kind, const float params[9], const float noise12[12], uint32_t n, const float* pos,
                      float* out_n) {
    const Aff nt = noise12 ? load_aff(noise12) : Aff{identity(), v3(0, 0, 0)};
    for (uint32_t i = 0; i < n; ++i) {
        const V3 q = normal_local(kind, params, nt, v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), nullptr);
        out_n[3 * i] = q.x;
        out_n[3 * i + 1] = q.y;
        out_n[3 * i + 2] = q.z;
    }
}

void orc_sdf_craterball(const float off[3], const float noise12[12], uint32_t n, const float* pos, float* out_d) {
    const Aff nt = load_aff(noise12);
    for (uint32_t i = 0; i < n; ++i)
        out_d[i] = craterball_sdf(v3(off[0], off[1], off[2]), nt, v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), nullptr);
}

// signedDistance of the demo's RoughBoxObject (kind 5), RoughSphereObject (6) and WeirdWallObject (7) at local points
int orc_sdf_value(uint32_t kind, const float params[9], const float noise12[12], uint32_t n, const float* pos, float* out_d) {
    const Aff nt = load_aff(noise12);
    for (uint32_t i = 0; i < n; ++i) {
        const V3 p = v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        if (kind == ORC_ROUGHBOX) out_d[i] = roughbox_sdf(v3(params[0], params[1], params[2]), nt, p);
        else if (kind == ORC_ROUGHSPHERE) out_d[i] = roughsphere_sdf(params[0], nt, p);
        else if (kind == ORC_WEIRDWALL) out_d[i] = weirdwall_sdf(nt, p);
        else return 0;
    }
    return 1;
}

void orc_noise(const float noise12[12], uint32_t n, const float* pos, float* out_v) {
    const Aff nt = load_aff(noise12);
    for (uint32_t i = 0; i < n; ++i) out_v[i] = fbm_noise(nt, v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
}

void orc_camera_rays(const float cam16[16], uint32_t n, const float* sx, const float* sy, uint32_t rng_mode,
                     uint64_t seed, float* out_dir, float* out_pos) {
    const Cam c = load_cam(cam16);
    MinstdRng seq(seed);
    for (uint32_t i = 0; i < n; ++i) {
        Ray r;
        if (rng_mode == ORC_RNG_COUNTER) {
            CounterRng rng(seed, i, 0);
            rng.begin_stage(0);
            rng.k = 2;   // k = 0, 1 are the pixel jitter draws
            r = view_ray(c, sx[i], sy[i], rng);
        } else {
            r = view_ray(c, sx[i], sy[i], seq);
        }
        out_dir[3 * i] = r.dir.x;
        out_dir[3 * i + 1] = r.dir.y;
        out_dir[3 * i + 2] = r.dir.z;
        out_pos[3 * i] = r.pos.x;
        out_pos[3 * i + 1] = r.pos.y;
        out_pos[3 * i + 2] = r.pos.z;
    }
}

void orc_closest_hit(const orc_scene* s, uint32_t n, const float* pos, const float* dir, int32_t* out_prim,
                     float* out_t, float* out_pos) {
    for_each_ray(n, s->d.prims.size(), [&](uint32_t i) {
        const Ray r{v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2])};
        const Hit h = closest_hit(s->d, r, nullptr);
        out_prim[i] = h.prim;
        out_t[i] = h.t;
        if (out_pos) {
            out_pos[3 * i] = h.pos.x;
            out_pos[3 * i + 1] = h.pos.y;
            out_pos[3 * i + 2] = h.pos.z;
        }
    });
}

void orc_deflect(const float mat14[14], uint32_t n, const float* inbound, const float* normal, uint32_t rng_mode,
                 uint64_t seed, float* out12) {
    const Material m = load_material(mat14);
    MinstdRng seq(seed);
    for (uint32_t i = 0; i < n; ++i) {
        const V3 in = v3(inbound[3 * i], inbound[3 * i + 1], inbound[3 * i + 2]);
        const V3 nn = v3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        Bounce b;
        if (rng_mode == ORC_RNG_COUNTER) {
            CounterRng rng(seed, i, 0);
            rng.begin_stage(1);
            b = deflect(m, in, nn, rng);
        } else {
            b = deflect(m, in, nn, seq);
        }
        const V3 f[4] = {b.emitted, b.attenuation, b.dir, b.normal};
        for (int k = 0; k < 4; ++k) {
            out12[12 * i + 3 * k] = f[k].x;
            out12[12 * i + 3 * k + 1] = f[k].y;
            out12[12 * i + 3 * k + 2] = f[k].z;
        }
    }
}

void orc_trace(const orc_scene* s, uint32_t n, const float* pos, const float* dir, uint32_t depth, uint32_t rng_mode,
               uint64_t seed, float* out_rgb) {
    if (rng_mode == ORC_RNG_COUNTER) {   // every ray has its own random stream
        for_each_ray(n, s->d.prims.size(), [&](uint32_t i) {
            const Ray r{v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2])};
            CounterRng rng(seed, i, 0);
            const V3 c = trace(s->d, r, depth, rng, nullptr);
            out_rgb[3 * i] = c.x;
            out_rgb[3 * i + 1] = c.y;
            out_rgb[3 * i + 2] = c.z;
        });
        return;
    }
    MinstdRng seq(seed);   // one sequence drawn ray after ray, as the reference's single worker draws it
    for (uint32_t i = 0; i < n; ++i) {
        const Ray r{v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2])};
        const V3 c = trace(s->d, r, depth, seq, nullptr);
        out_rgb[3 * i] = c.x;
        out_rgb[3 * i + 1] = c.y;
        out_rgb[3 * i + 2] = c.z;
    }
}

void orc_primary_hits(const orc_scene* s, const float cam16[16], uint32_t W, uint32_t H, int32_t* out_prim,
                      float* out_t) {
    const Cam c = load_cam(cam16);
    const float max_w = static_cast<float>(W - 1), max_h = static_cast<float>(H - 1);
    // (an SDF primitive costs far more than its share of the object count: price every ray highly enough to use the cores)
    for_each_ray(W * H, s->d.prims.size() + 4096, [&](uint32_t i) {
        const uint32_t y = i / W, x = i - y * W;
        const float py = static_cast<float>(y) / max_h;
        const float px = static_cast<float>(x) / max_w;
        const Ray r = view_ray_with_lens(c, px, py, 0.0f, 0.0f);
        const Hit h = closest_hit(s->d, r, nullptr);
        out_prim[i] = h.prim;
        out_t[i] = h.prim >= 0 ? h.t : 0.0f;
    });
}

int orc_render(const orc_scene* s, const float cam16[16], uint32_t W, uint32_t H, uint32_t bounces, uint32_t spp,
               uint32_t rng_mode, uint64_t seed, uint32_t first_sample, uint32_t threads, uint32_t full_width,
               float* out_rgb, uint64_t* stats4) {
    if (W < 2 || H < 2 || spp == 0 || bounces == 0) return -1;
    if (!full_width && W < 8) return -1;   // the reference divides by zero here (src/Renderer.cpp:122-123)
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    std::fill(out_rgb, out_rgb + static_cast<size_t>(3) * W * H, 0.0f);   // Image is zero-filled (src/Image.cpp:8-13)
    RenderJob job;
    job.scene = &s->d;
    job.cam = load_cam(cam16);
    job.W = W;
    job.H = H;
    job.bounces = bounces;
    job.spp = spp;
    job.first_sample = first_sample;
    job.seed = seed;
    job.full_width = full_width != 0;
    job.out = out_rgb;
    std::vector<std::thread> pool;
    for (uint32_t k = 0; k < threads; ++k) {
        if (rng_mode == ORC_RNG_COUNTER) pool.emplace_back([&job, k] { render_worker<true>(job, k); });
        else pool.emplace_back([&job, k] { render_worker<false>(job, k); });
    }
    for (auto& t : pool) t.join();
    if (stats4) {
        stats4[0] = job.paths;
        stats4[1] = job.rays;
        stats4[2] = job.evals;
        stats4[3] = job.near_evals;
    }
    return 0;
}

void orc_tonemap(uint32_t kind, uint32_t W, uint32_t H, const float* in, float* out) {
    auto at = [&](uint32_t i, uint32_t j) { return in + 3 * (static_cast<size_t>(j) * W + i); };   // img(x=i, y=j)
    if (kind == 1) {   // FilmicToneMapper src/ToneMapper.cpp:53-72
        const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
        for (size_t n = 0; n < static_cast<size_t>(3) * W * H; ++n) {
            const float p = in[n];
            out[n] = std::clamp((p * (a * p + b)) / (p * (c * p + d) + e), 0.0f, 1.0f);
        }
        return;
    }
    // ReinhardToneMapper :6-51 -- column-major visiting order, luminance reads b twice.
    float acc = 0.0f;
    float min_l = std::numeric_limits<float>::max(), max_l = std::numeric_limits<float>::lowest();
    for (uint32_t i = 0; i < W; ++i)
        for (uint32_t j = 0; j < H; ++j) {
            const float* c = at(i, j);
            const float lum = 0.27f * c[0] + 0.67f * c[2] + 0.06f * c[2];
            min_l = std::min(min_l, lum);
            max_l = std::max(max_l, lum);
            acc += std::log(lum + 1e-6f);
        }
    const float avg = std::exp(acc / static_cast<float>(static_cast<size_t>(W) * H));
    const float log_avg = std::log2(avg), log_min = std::log2(min_l + 1e-6f), log_max = std::log2(max_l + 1e-6f);
    const float alpha = 0.18f * std::pow(4.0f, (2.0f * log_avg - log_min - log_max) / (log_max - log_min));
    const float k = alpha / avg;
    for (size_t n = 0; n < static_cast<size_t>(3) * W * H; ++n) {
        const float sc = in[n] * k;
        out[n] = sc / (1.0f + sc);
    }
}

void orc_image_bytes(uint32_t W, uint32_t H, const float* rgb, uint8_t* out) {
    const uint64_t w = W, h = H;
    std::memcpy(out, &w, 8);
    std::memcpy(out + 8, &h, 8);
    std::memcpy(out + 16, rgb, static_cast<size_t>(12) * W * H);
}

float orc_counter_uniform01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stage, uint32_t k) {
    return counter_u01(counter_path_key(seed, pixel, sample), stage, k);
}

}   // extern "C"
