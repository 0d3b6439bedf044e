// Host side of include/pathtracer/pathtracer.hpp.
//
// Two halves: (1) the per-ray host functions of the public API, written to the reference's arithmetic
// (operation order, epsilons, quirks) so that values computed on the host agree with the device and
// with the reference; (2) the glue that flattens a Scene and drives the GPU through the C ABI
// (include/mipt.h).  Built with -ffp-contract=off.  Citations are file:line of the reference checkout.
#include "pathtracer/pathtracer.hpp"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <stdexcept>

#include "mipt.h"

// ================================================================================================
// random numbers
// ================================================================================================
// Weak, so that a program may link its own definition (a fixed seed for reproducible host sampling).
__attribute__((weak)) std::default_random_engine& randomEngine() noexcept {   // src/RandomNumberGenerator.cpp:3-6
    static thread_local std::default_random_engine engine{std::random_device{}()};
    return engine;
}

// ================================================================================================
// linear algebra: delegated to the C ABI's host helpers, which keep the reference's operation order
// ================================================================================================
namespace {
Linear linear_from(const float* m) { return Linear(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]); }
void affine_to(const Affine& a, float* o) {
    std::memcpy(o, a.linear.data(), 9 * sizeof(float));
    o[9] = a.translation.x;
    o[10] = a.translation.y;
    o[11] = a.translation.z;
}
Affine affine_from(const float* m) { return Affine(linear_from(m), Vec(m[9], m[10], m[11])); }
Linear rotation(char axis, float angle) {
    float m[9];
    mipt_rotation(axis, 1, &angle, m);
    return linear_from(m);
}
}   // namespace

Linear Linear::RotationX(float a) noexcept { return rotation('x', a); }
Linear Linear::RotationY(float a) noexcept { return rotation('y', a); }
Linear Linear::RotationZ(float a) noexcept { return rotation('z', a); }

Linear Linear::Rotation(Vec u, float a) noexcept {   // axis-angle, src/LinearAlgebra.cpp:120-130
    const float c = std::cos(a), s = std::sin(a), omc = 1.0f - c;
    return Linear(c + u.x * u.x * omc, u.x * u.y * omc - u.z * s, u.x * u.z * omc + u.y * s,
                  u.y * u.x * omc + u.z * s, c + u.y * u.y * omc, u.y * u.z * omc - u.x * s,
                  u.z * u.x * omc - u.y * s, u.z * u.y * omc + u.x * s, c + u.z * u.z * omc);
}

float Linear::determinant() const noexcept {   // :201-205
    const auto& m = m_data;
    return (m[0] * m[4] * m[8]) + (m[1] * m[5] * m[6]) + (m[2] * m[3] * m[7]) - (m[2] * m[4] * m[6]) - (m[1] * m[3] * m[8]) - (m[0] * m[5] * m[7]);
}

std::optional<Linear> Linear::inverse() const noexcept {
    float f[12] = {}, inv[12];
    std::memcpy(f, m_data.data(), sizeof(float) * 9);
    if (mipt_affine_inverse(1, f, inv) != MIPT_OK) return std::nullopt;
    return linear_from(inv);
}

Linear& Linear::operator*=(const Linear& other) noexcept {
    *this = (*this) * other;
    return *this;
}

std::optional<Affine> Affine::inverse() const noexcept {   // :258-264
    float f[12], inv[12];
    affine_to(*this, f);
    if (mipt_affine_inverse(1, f, inv) != MIPT_OK) return std::nullopt;
    return affine_from(inv);
}

Linear operator-(const Linear& A) noexcept {
    Linear r;
    for (std::size_t i = 0; i < 9; ++i) r[i] = -A[i];
    return r;
}
Linear operator+(const Linear& A, const Linear& B) noexcept {
    Linear r;
    for (std::size_t i = 0; i < 9; ++i) r[i] = A[i] + B[i];
    return r;
}
Linear operator-(const Linear& A, const Linear& B) noexcept {
    Linear r;
    for (std::size_t i = 0; i < 9; ++i) r[i] = A[i] - B[i];
    return r;
}
Linear operator*(const Linear& A, float t) noexcept {
    Linear r;
    for (std::size_t i = 0; i < 9; ++i) r[i] = t * A[i];
    return r;
}
Linear operator*(float t, const Linear& A) noexcept { return A * t; }
Linear operator/(const Linear& A, float t) noexcept { return A * (1.0f / t); }
Linear operator*(const Linear& A, const Linear& B) noexcept {   // rows times columns, :389-395
    Linear r;
    for (std::size_t i = 0; i < 3; ++i)
        for (std::size_t j = 0; j < 3; ++j) r(i, j) = A.row(i) * B.column(j);
    return r;
}
Affine operator*(const Affine& A, const Linear& L) noexcept { return Affine(A.linear * L, A.translation); }
Affine operator*(const Linear& L, const Affine& A) noexcept { return Affine(L) * A; }
Affine operator*(const Affine& A, const Affine& B) noexcept { return Affine(A.linear * B.linear, A.linear * B.translation + A.translation); }

// ================================================================================================
// geometry (src/Geometry.cpp)
// ================================================================================================
namespace {
constexpr float kPi = 3.141592653589793f;
constexpr float kEps = 1e-3f;   // :8
}

float Sphere::surfaceArea() const noexcept { return 4.0f * kPi * radius * radius; }
float Sphere::volume() const noexcept { return (4.0f / 3.0f) * kPi * radius * radius * radius; }
float Rectangle::surfaceArea() const noexcept { return 8.0f * (halfSize.x * halfSize.y + halfSize.x * halfSize.z + halfSize.y * halfSize.z); }

float Rectangle::signedDistance(Pos p) const noexcept {   // :66-75
    const float dx = std::abs(p.x) - halfSize.x, dy = std::abs(p.y) - halfSize.y, dz = std::abs(p.z) - halfSize.z;
    return Vec(std::max(dx, 0.0f), std::max(dy, 0.0f), std::max(dz, 0.0f)).norm() + std::min(std::max(std::max(dx, dy), dz), 0.0f);
}

// Largest |p_k / half_k| wins; ties prefer z, then y; the sign of 0 counts as negative (:77-103).
Vec Rectangle::normal(Pos p) const noexcept {
    const float ax = std::abs(p.x / halfSize.x), ay = std::abs(p.y / halfSize.y), az = std::abs(p.z / halfSize.z);
    const auto sgn = [](float v) { return v > 0.0f ? 1.0f : -1.0f; };
    if (ax > ay && ax > az) return Vec(sgn(p.x), 0.0f, 0.0f);
    if (!(ax > ay) && ay > az) return Vec(0.0f, sgn(p.y), 0.0f);
    return Vec(0.0f, 0.0f, sgn(p.z));
}

AxisAlignedBox::AxisAlignedBox(Pos corner, Pos opposite) noexcept   // :348-358
    : center(corner + 0.5f * (opposite - corner)),
      halfSize(Vec(0.5f * std::abs(corner.x - opposite.x), 0.5f * std::abs(corner.y - opposite.y), 0.5f * std::abs(corner.z - opposite.z))) {}

AxisAlignedBox boxContaining(const AxisAlignedBox& a, const AxisAlignedBox& b) noexcept {   // :105-140
    const Pos a0 = a.center - a.halfSize, a1 = a.center + a.halfSize, b0 = b.center - b.halfSize, b1 = b.center + b.halfSize;
    return AxisAlignedBox(Pos(std::min(a0.x, b0.x), std::min(a0.y, b0.y), std::min(a0.z, b0.z)),
                          Pos(std::max(a1.x, b1.x), std::max(a1.y, b1.y), std::max(a1.z, b1.z)));
}

Vec bounce(const Vec& inbound, const Vec& normal) noexcept { return inbound - (2.0f * ((inbound * normal) * normal)); }   // :142-144

std::optional<Pos> intersect(const Ray& ray, const Triangle& tri) noexcept {   // Moeller-Trumbore, :146-171
    const Vec e1 = tri.b - tri.a, e2 = tri.c - tri.a;
    const Vec h = ray.dir ^ e2;
    const float det = e1 * h;
    if (std::abs(det) < kEps) return std::nullopt;
    const float f = 1.0f / det;
    const Vec s = ray.pos - tri.a;
    const float u = f * (s * h);
    if (u < 0.0f || u > 1.0f) return std::nullopt;
    const Vec q = s ^ e1;
    const float v = f * (ray.dir * q);
    if (v < 0.0f || u + v > 1.0f) return std::nullopt;
    const float t = f * (e2 * q);
    if (t > kEps) return ray.pos + t * ray.dir;
    return std::nullopt;
}

std::optional<Pos> intersect(const Ray& ray, const Sphere& sphere) noexcept {   // :173-201
    const float a = ray.dir.normSquared();
    const float b = 2.0f * (ray.dir * ray.pos.toVec());
    const float c = ray.pos.toVec().normSquared() - (sphere.radius * sphere.radius);
    const float disc = b * b - 4.0f * a * c;
    if (disc <= 0.0f) return std::nullopt;
    const float root = std::sqrt(disc);
    const float t0 = (-b - root) / (2.0f * a), t1 = (-b + root) / (2.0f * a);
    float t;
    if (t0 > kEps) t = t1 > kEps ? std::min(t0, t1) : t0;
    else if (t1 > kEps) t = t1;
    else return std::nullopt;
    return ray.pos + t * ray.dir;
}

// Six face planes; a face is skipped when the direction component is below 1e-3 in magnitude, t > 0
// carries no epsilon, and from inside the exit face is returned (:203-252).
std::optional<Pos> intersect(const Ray& ray, const Rectangle& box) noexcept {
    const Vec lo = -ray.pos.toVec() - box.halfSize, hi = -ray.pos.toVec() + box.halfSize;
    const float d[3] = {ray.dir.x, ray.dir.y, ray.dir.z}, l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    std::optional<float> best;
    for (int k = 0; k < 3; ++k) {
        if (std::abs(d[k]) < kEps) continue;
        for (const float plane : {l[k], h[k]}) {
            const float t = plane / d[k];
            if (!(t > 0.0f)) continue;
            bool within = true;
            for (int j = 0; j < 3; ++j) {
                if (j == k) continue;
                const float v = d[j] * t;
                within = within && v >= l[j] && v <= h[j];
            }
            if (within) best = best ? std::min(*best, t) : t;
        }
    }
    if (!best) return std::nullopt;
    return ray.pos + ((*best) * ray.dir);
}

std::optional<Pos> intersect(const Ray& ray, const AxisAlignedBox& b) noexcept {   // :254-261
    const auto p = intersect(Ray(ray.dir, ray.pos - b.center.toVec()), Rectangle(b.halfSize));
    if (!p) return std::nullopt;
    return (*p) + b.center.toVec();
}

bool inside(const Pos& p, const Sphere& s) noexcept { return p.toVec().normSquared() <= (s.radius * s.radius); }
bool inside(const Pos& p, const Rectangle& b) noexcept {
    return std::abs(p.x) <= b.halfSize.x && std::abs(p.y) <= b.halfSize.y && std::abs(p.z) <= b.halfSize.z;
}
bool inside(const Pos& p, const AxisAlignedBox& b) noexcept { return inside(p - b.center.toVec(), Rectangle(b.halfSize)); }

std::pair<float, float> randomPointInCircle() noexcept {   // :278-288
    std::uniform_real_distribution<float> dist{-1.0f, 1.0f};
    for (;;) {
        const float x = dist(randomEngine()), y = dist(randomEngine());
        if (x * x + y * y <= 1.0f) return {x, y};
    }
}

std::pair<Vec, Vec> orthogonalPair(Vec v) noexcept {   // :290-299
    v = v.unit();
    const bool alongX = (v ^ Vec(1.0f, 0.0f, 0.0f)).normSquared() < 0.1;
    const Vec w = alongX ? Vec(0.0f, 1.0f, 0.0f) : Vec(1.0f, 0.0f, 0.0f);
    const Vec v1 = w ^ v, v2 = v1 ^ v;
    return {v1.unit(), v2.unit()};
}

Vec randomPointOnHemisphereUniform(Vec normal) noexcept {   // :301-318
    normal = normal.unit();
    std::uniform_real_distribution<float> dist{-1.0f, 1.0f};
    for (;;) {
        const float x = dist(randomEngine()), y = dist(randomEngine()), z = dist(randomEngine());
        const float l2 = (x * x) + (y * y) + (z * z);
        if (l2 > 1.0f || l2 < kEps) continue;
        const Vec v(x, y, z);
        if ((v * normal) <= 0.0f) continue;
        return v.unit();
    }
}

Vec randomPointOnHemisphereCosine(Vec normal) noexcept {   // :320-341
    normal = normal.unit();
    std::uniform_real_distribution<float> dist{-1.0f, 1.0f};
    for (;;) {
        const float x = dist(randomEngine()), y = dist(randomEngine());
        const float l2 = (x * x) + (y * y);
        if (l2 > 1.0f) continue;
        const float z = std::sqrt(1.0f - l2);
        const auto [v1, v2] = orthogonalPair(normal);
        return (z * normal) + (x * v1) + (y * v2);
    }
}

// ================================================================================================
// material (src/BasicMaterial.cpp)
// ================================================================================================
BasicMaterial::BasicMaterial() noexcept : m_p{0.3f, 0.3f, 0.9f, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.5f, 0.9f, 0.9f, 0.9f} {}   // :9-18

// The setters assert their ranges as the reference's do (src/BasicMaterial.cpp:54-95); like the reference's default
// build, this library is compiled without NDEBUG, so the asserts are live.
namespace {
inline bool unit_range(float v) { return v >= 0.0f && v <= 1.0f; }
}
void BasicMaterial::setDiffuseReflection(float r) noexcept { assert(unit_range(r)); m_p[0] = r; }
void BasicMaterial::setSpecularReflection(float r) noexcept { assert(unit_range(r)); m_p[1] = r; }
void BasicMaterial::setSpecularSharpness(float r) noexcept { assert(unit_range(r)); m_p[2] = r; }
void BasicMaterial::setReflectedAbsorption(Color c) noexcept {
    assert(unit_range(c.r) && unit_range(c.g) && unit_range(c.b));
    m_p[3] = c.r; m_p[4] = c.g; m_p[5] = c.b;
}
void BasicMaterial::setEmittedLuminance(Color c) noexcept { m_p[6] = c.r; m_p[7] = c.g; m_p[8] = c.b; }
void BasicMaterial::setTransmittance(float r) noexcept { assert(unit_range(r)); m_p[9] = r; }
void BasicMaterial::setIndexOfRefraction(float i) noexcept { assert(i >= 1.0f); m_p[10] = i; }
void BasicMaterial::setInternalAbsorption(Color c) noexcept {
    assert(unit_range(c.r) && unit_range(c.g) && unit_range(c.b));
    m_p[11] = c.r; m_p[12] = c.g; m_p[13] = c.b;
}

// Inside hits are deterministic; outside hits pick diffuse / specular / transmission with probabilities
// proportional to the three weights; every lobe returns the emitted luminance and the reflected
// absorption without pdf or cosine factors (:97-159).
ColorBounce BasicMaterial::deflect(const Vec& inbound, const Vec& normal) const noexcept {
    const float ior = m_p[10];
    if (inbound * normal >= 0.0f) {
        const Vec v = (inbound + ((inbound * normal) * (1.0f - ior) * normal)).unit();
        if (v * normal >= 0.0f) return ColorBounce(Color(0.0f, 0.0f, 0.0f), Color(1.0f, 1.0f, 1.0f), v, normal);
        return ColorBounce(Color(0.0f, 1.0f, 1.0f), Color(0.0f, 0.0f, 0.0f), bounce(inbound, -normal), normal);
    }
    const float reflection = m_p[0] + m_p[1];
    const float options = reflection + m_p[9];
    std::uniform_real_distribution<float> dist{0.0f, options};
    const float which = dist(randomEngine());
    Vec out;
    if (which < reflection) {
        if (which < m_p[0]) out = randomPointOnHemisphereCosine(normal);
        else out = (bounce(inbound, normal) + (1.0f - m_p[2]) * randomPointOnHemisphereCosine(normal)).unit();
    } else {
        out = (inbound + ((inbound * normal) * (1.0f - 1.0f / ior) * normal)).unit();
    }
    return ColorBounce(emittedLuminance(), reflectedAbsorption(), out, normal);
}

// ================================================================================================
// objects (src/Object.cpp)
// ================================================================================================
void Object::setTransformation(const Affine& t) noexcept {
    m_transformation = t;
    const auto inv = t.inverse();
    if (!inv) {   // the reference asserts; a determinant below 1e-6 also covers mirrors and small scales
        std::fprintf(stderr, "pathtracer: Object::setTransformation: transformation is not invertible (determinant < 1e-6)\n");
        std::abort();
    }
    m_inverseTransformation = *inv;
}

std::optional<Pos> Object::hitRay(const Ray& ray) const noexcept {   // :22-33
    const Ray local(m_inverseTransformation * ray.dir, m_inverseTransformation * ray.pos);
    const auto hit = hitLocalRay(local);
    if (!hit) return std::nullopt;
    return m_transformation * (*hit);
}

ColorBounce Object::deflectRay(const Ray& ray) const noexcept {   // :35-44
    const Ray local(m_inverseTransformation * ray.dir, m_inverseTransformation * ray.pos);
    ColorBounce b = deflectLocalRay(local);
    b.rayDirection = m_transformation * b.rayDirection;
    return b;
}

AxisAlignedBox Object::getBoundingBox() const noexcept {   // :46-80
    const AxisAlignedBox bb = getLocalBoundingBox();
    const float big = std::numeric_limits<float>::max(), low = std::numeric_limits<float>::lowest();
    Pos lo(big, big, big), hi(low, low, low);
    for (int s = 0; s < 8; ++s) {
        const Vec corner((s & 4) ? -bb.halfSize.x : bb.halfSize.x, (s & 2) ? -bb.halfSize.y : bb.halfSize.y, (s & 1) ? -bb.halfSize.z : bb.halfSize.z);
        const Pos u = m_transformation * (bb.center + corner);
        lo = Pos(std::min(lo.x, u.x), std::min(lo.y, u.y), std::min(lo.z, u.z));
        hi = Pos(std::max(hi.x, u.x), std::max(hi.y, u.y), std::max(hi.z, u.z));
    }
    return AxisAlignedBox(lo, hi);
}

TriangleObject::TriangleObject(Triangle g, BasicMaterial m) : geometry(g), material(m) {}
std::optional<Pos> TriangleObject::hitLocalRay(const Ray& ray) const noexcept { return intersect(ray, geometry); }
ColorBounce TriangleObject::deflectLocalRay(const Ray& ray) const noexcept { return material.deflect(ray.dir, geometry.normal()); }
AxisAlignedBox TriangleObject::getLocalBoundingBox() const noexcept {   // padded by 1e-3, :96-108
    const auto& g = geometry;
    return AxisAlignedBox(Pos(std::min({g.a.x, g.b.x, g.c.x}) - 1e-3f, std::min({g.a.y, g.b.y, g.c.y}) - 1e-3f, std::min({g.a.z, g.b.z, g.c.z}) - 1e-3f),
                          Pos(std::max({g.a.x, g.b.x, g.c.x}) + 1e-3f, std::max({g.a.y, g.b.y, g.c.y}) + 1e-3f, std::max({g.a.z, g.b.z, g.c.z}) + 1e-3f));
}
void TriangleObject::describeForDevice(mipt_host::DevicePrimitive& o) const noexcept {
    o.kind = MIPT_TRIANGLE;
    const float p[9] = {geometry.a.x, geometry.a.y, geometry.a.z, geometry.b.x, geometry.b.y, geometry.b.z, geometry.c.x, geometry.c.y, geometry.c.z};
    std::memcpy(o.params, p, sizeof(p));
    o.material = &material;
    o.type_name = "TriangleObject";
}

SphereObject::SphereObject(Sphere g, BasicMaterial m) noexcept : geometry(g), material(m) {}
std::optional<Pos> SphereObject::hitLocalRay(const Ray& ray) const noexcept { return intersect(ray, geometry); }
ColorBounce SphereObject::deflectLocalRay(const Ray& ray) const noexcept { return material.deflect(ray.dir, geometry.normal(ray.pos)); }
AxisAlignedBox SphereObject::getLocalBoundingBox() const noexcept {
    const float h = geometry.radius + 1e-3f;
    return AxisAlignedBox(Pos(0.0f, 0.0f, 0.0f), Vec(h, h, h));
}
void SphereObject::describeForDevice(mipt_host::DevicePrimitive& o) const noexcept {
    o.kind = MIPT_SPHERE;
    o.params[0] = geometry.radius;
    o.material = &material;
    o.type_name = "SphereObject";
}

BoxObject::BoxObject(Rectangle g, BasicMaterial m) noexcept : geometry(g), material(m) {}
std::optional<Pos> BoxObject::hitLocalRay(const Ray& ray) const noexcept { return intersect(ray, geometry); }
ColorBounce BoxObject::deflectLocalRay(const Ray& ray) const noexcept { return material.deflect(ray.dir, geometry.normal(ray.pos)); }
AxisAlignedBox BoxObject::getLocalBoundingBox() const noexcept {
    return AxisAlignedBox(Pos(0.0f, 0.0f, 0.0f), Vec(geometry.halfSize.x + 1e-3f, geometry.halfSize.y + 1e-3f, geometry.halfSize.z + 1e-3f));
}
void BoxObject::describeForDevice(mipt_host::DevicePrimitive& o) const noexcept {
    o.kind = MIPT_BOX;
    o.params[0] = geometry.halfSize.x;
    o.params[1] = geometry.halfSize.y;
    o.params[2] = geometry.halfSize.z;
    o.material = &material;
    o.type_name = "BoxObject";
}

// FractalObject (src/Object.cpp:160-252).  Its hitLocalRay intersects the WORLD bounding box with the
// local ray, as the reference does.
std::optional<Pos> FractalObject::hitLocalRay(const Ray& ray) const noexcept {
    const auto start = intersect(ray, getBoundingBox());
    if (!start) return std::nullopt;
    Pos p = *start;
    for (std::size_t i = 0; i < 256; ++i) {
        const float d = signedDistance(p);
        if (d < 1e-3f) return p;
        p = p + d * ray.dir;
    }
    return std::nullopt;
}
ColorBounce FractalObject::deflectLocalRay(const Ray& ray) const noexcept {
    const float delta = 1e-3f;
    const Vec dx(delta, 0.0f, 0.0f), dy(0.0f, delta, 0.0f), dz(0.0f, 0.0f, delta);
    const Vec n = (Vec(signedDistance(ray.pos + dx) - signedDistance(ray.pos - dx), signedDistance(ray.pos + dy) - signedDistance(ray.pos - dy),
                       signedDistance(ray.pos + dz) - signedDistance(ray.pos - dz)) /
                   delta)
                      .unit();
    return material.deflect(ray.dir, n);
}
AxisAlignedBox FractalObject::getLocalBoundingBox() const noexcept { return AxisAlignedBox(Pos(0.0f, 0.0f, 0.0f), Vec(2.0f, 2.0f, 2.0f)); }
void FractalObject::describeForDevice(mipt_host::DevicePrimitive& o) const noexcept {
    o.kind = MIPT_FRACTAL;
    o.material = &material;
    o.type_name = "FractalObject";
}
float FractalObject::signedDistance(const Pos& p) const noexcept {
    const auto fold = [](float v) {
        if (v < -0.5f) return v + 0.5f;
        if (v > 0.5f) return v - 0.5f;
        v *= 2.0f;
        return (v - std::round(v)) / 2.0f;
    };
    return Vec(fold(p.x), fold(p.y), fold(p.z)).norm() - 0.2f;
}

// ================================================================================================
// the demo's SDF classes, recognised without their cooperation
// ================================================================================================
bool mipt_host::describeKnownDemoSdf(const char* name, std::size_t derived_size, std::size_t base_size, const void* object, DevicePrimitive& out) noexcept {
    // Itanium mangling of a global class: <length><name>.  The data members lie directly behind the
    // SDFObjectCRTP base: CraterBallObject a Vec (PathTracerDemo.cpp:481-482); RoughBoxObject a Rectangle, which is
    // its halfSize Vec (:268-269); RoughSphereObject a Sphere, which is its radius, then a BasicMaterial (:223-225);
    // WeirdWallObject nothing.
    static_assert(sizeof(Rectangle) == sizeof(Vec), "Rectangle is its halfSize");
    static_assert(sizeof(Sphere) == sizeof(float), "Sphere is its radius");
    const auto pad8 = [](std::size_t n) { return (n + 7) / 8 * 8; };
    const char* members = static_cast<const char*>(object) + base_size;
    if (std::strcmp(name, "15WeirdWallObject") == 0) {
        if (derived_size != base_size) return false;
        out.kind = MIPT_WEIRDWALL;
        return true;
    }
    if (std::strcmp(name, "17RoughSphereObject") == 0) {
        if (derived_size != pad8(base_size + sizeof(Sphere) + sizeof(BasicMaterial))) return false;
        std::memcpy(&out.params[0], members, sizeof(float));
        out.kind = MIPT_ROUGHSPHERE;
        return true;
    }
    const bool ball = std::strcmp(name, "16CraterBallObject") == 0, box = std::strcmp(name, "14RoughBoxObject") == 0;
    if (!ball && !box) return false;
    if (derived_size != pad8(base_size + sizeof(Vec))) return false;
    float off[3];
    std::memcpy(off, members, sizeof(off));
    out.kind = ball ? MIPT_CRATERBALL : MIPT_ROUGHBOX;
    out.params[0] = off[0];
    out.params[1] = off[1];
    out.params[2] = off[2];
    return true;
}

// ---- CraterBall: host evaluation of the SDF (PathTracerDemo.cpp:26-116, 361-466) ----------------
namespace {
float triangle_wave(float x) { return -1.0f + 4.0f * std::abs(x - 0.25f - std::floor(x + 0.25f)); }

float fbm(const Affine& T, Pos p) {   // noise(): 8 octaves, weight 0.02 * 0.9^i, lacunarity 1.1
    float v = 0.0f, k = 0.02f;
    for (int i = 0; i < 8; ++i) {
        p = T * p;
        v += k * (triangle_wave(p.x) + triangle_wave(p.y) + triangle_wave(p.z));
        p = Pos(p.x * 1.1f, p.y * 1.1f, p.z * 1.1f);
        k *= 0.9f;
    }
    return v;
}

// Hashed, jittered lattice points around `location`; the jitter expression is evaluated in double
// and narrowed, as the `-1.0` literal in the reference makes it.
template <class F>
void lattice_points(Pos location, float scatter, int radius, F&& f) {
    const Pos origin(std::floor(location.x), std::floor(location.y), std::floor(location.z));
    for (int i = -radius + 1; i <= radius; ++i)
        for (int j = -radius + 1; j <= radius; ++j)
            for (int k = -radius + 1; k <= radius; ++k) {
                const Pos g = origin + Vec(static_cast<float>(i), static_cast<float>(j), static_cast<float>(k));
                std::uint32_t h = static_cast<std::uint32_t>(static_cast<int>(g.x)) * 885889u + static_cast<std::uint32_t>(static_cast<int>(g.y)) * 745273u +
                                  static_cast<std::uint32_t>(static_cast<int>(g.z)) * 1220689u;
                for (int r = 0; r < 4; ++r) {
                    h |= (h & 0x0f0f0f0fu) << 4;
                    h *= 31u;
                }
                const auto jit = [&](float c, std::uint32_t bits) { return static_cast<float>(c + scatter * 0.5f * (-1.0 + static_cast<float>(bits & 0x3ffu) / 512.0f)); };
                f(Pos(jit(g.x, h), jit(g.y, h >> 10), jit(g.z, h >> 20)), h);
            }
}
}   // namespace

CraterBall::CraterBall(BasicMaterial mat, Vec offset, Affine noise) : SDFObjectCRTP(mat), m_offset(offset), m_noise(noise) {}

float CraterBall::signedDistance(const Pos& pos) const noexcept {
    const float sphere = pos.toVec().norm() - 0.5f;
    if (sphere > 0.1) return sphere;
    const Vec inward = -pos.toVec().unit();
    float crater = 0.0f, tiny = 0.0f, fallout = 0.0f, min_age = 1.0f;
    const auto unit16 = [](std::uint32_t v) { return static_cast<float>(v & 0xffffu) / static_cast<float>(0xffff); };
    const auto project = [&](Pos point, float scale) { return m_offset + ((point - m_offset).toVec().unit() * scale).toPos(); };
    {   // big craters
        const Pos query = m_offset + (inward * 2.0f).toPos();
        lattice_points(query, 0.6f, 2, [&](Pos point, std::uint32_t hash) {
            const float t = unit16(hash), age = unit16(hash >> 16), size = 0.05f + 0.4f * t * t;
            const Pos on = project(point, 2.0f);
            const float dist = (query - on).norm();
            if (!((on - point).normSquared() < 0.1f)) return;
            const float c = 20.0f * (dist - size);
            float depth = std::tanh(c);
            if (c > 1e-3f) depth = 0.9f * t * depth / (0.02f / c + c * c);
            crater -= 0.02f * depth;
            if (dist >= size) {
                const float lin = std::max(0.0f, 1.2f - dist / (size * 5.0f));
                const Vec away = (query - on).unit();
                const float falloff = std::pow(lin, 2.0f + 3.0f * fbm(m_noise, on + 3.0f * size * away));
                fallout += 0.1f * falloff * (0.2f + 0.4f * depth) * fbm(m_noise, on + 40.0f * size * away);
            } else {
                min_age = std::min(age, min_age);
            }
        });
    }
    {   // small craters
        const Pos query = m_offset + (inward * 10.0f).toPos();
        lattice_points(query, 1.0f, 1, [&](Pos point, std::uint32_t hash) {
            const float t = unit16(hash), age = unit16(hash >> 16);
            if (age > min_age) return;
            const float size = 0.05f + 0.35f * t * t;
            const Pos on = project(point, 10.0f);
            if (!((on - point).normSquared() < 0.1f)) return;
            const float c = 50.0f * ((query - on).norm() - size);
            if (c < 0.0f) {
                min_age = std::min(age, min_age);
                crater = std::max(-0.005f * std::tanh(c), crater);
            }
        });
    }
    {   // tiny craters
        const Pos query = m_offset + (inward * 60.0f).toPos();
        lattice_points(query, 1.0f, 1, [&](Pos point, std::uint32_t hash) {
            const float t = unit16(hash), age = unit16(hash >> 16), size = 0.02f + 0.35f * t;
            const Pos on = project(point, 60.0f);
            if (!((on - point).normSquared() < 0.1f) || age > min_age) return;
            tiny = std::max(-0.004f * size * (-0.5f + 0.5f * std::tanh(50.0f * ((query - on).norm() - size))), tiny);
        });
    }
    return sphere + crater + tiny + fallout + 0.01f * fbm(m_noise, ((pos + m_offset).toVec() * 20.0f).toPos()) +
           0.4f * fbm(m_noise, ((pos + m_offset).toVec() * 1.5f).toPos());
}

bool mipt_host::DeviceSdf<CraterBall>::describe(const CraterBall& ball, DevicePrimitive& out) noexcept {
    out.kind = MIPT_CRATERBALL;
    out.params[0] = ball.offset().x;
    out.params[1] = ball.offset().y;
    out.params[2] = ball.offset().z;
    static_assert(sizeof(Affine) == 12 * sizeof(float), "Affine is 12 packed floats");
    out.noise = reinterpret_cast<const float*>(&ball.noise());
    return true;
}

// The demo defines `const Affine& noiseTransform() noexcept` in its own translation unit; when this
// library is linked into such a program the weak reference resolves to it.
const Affine& noiseTransform() noexcept __attribute__((weak));

// ================================================================================================
// object tree (src/ObjectTree.cpp): closest hit over all leaves; built by median splits
// ================================================================================================
namespace ObjectTree {

InternalNode::InternalNode(std::unique_ptr<Node> a, std::unique_ptr<Node> b) noexcept
    : Node(boxContaining(a->boundingBox, b->boundingBox)), m_childA(std::move(a)), m_childB(std::move(b)) {}

std::optional<std::pair<Pos, const Object*>> InternalNode::hit(const Ray& ray) const noexcept {   // :22-63
    const auto a = m_childA->hit(ray), b = m_childB->hit(ray);
    if (a && b) {
        const float ta = (a->first - ray.pos) * ray.dir, tb = (b->first - ray.pos) * ray.dir;
        return tb < ta ? b : a;
    }
    return a ? a : b;
}

LeafNode::LeafNode(const Object* object) noexcept : Node(object->getBoundingBox()), m_object(object) {}

std::optional<std::pair<Pos, const Object*>> LeafNode::hit(const Ray& ray) const noexcept {
    const auto p = m_object->hitRay(ray);
    if (!p) return std::nullopt;
    return std::pair<Pos, const Object*>{*p, m_object};
}

Tree::Tree(std::vector<const Object*> objects) : m_root(makeTree(std::move(objects), 0.5f)) {}

std::optional<std::pair<Pos, const Object*>> Tree::hit(const Ray& ray) const noexcept {
    if (!m_root) return std::nullopt;   // the reference dereferences a null root here
    return m_root->hit(ray);
}

// The reference searches every (object, axis) split with an O(n^2) cost scan per level; because
// traversal never culls, the topology only decides how exact ties are broken.  A split in the middle
// of the insertion order keeps that order (ties go to the earlier object) and builds in O(n).
std::unique_ptr<Node> Tree::makeTree(std::vector<const Object*> objects, float cost) {
    if (objects.empty()) return nullptr;
    if (objects.size() == 1) return std::make_unique<LeafNode>(objects[0]);
    const auto mid = objects.begin() + static_cast<std::ptrdiff_t>(objects.size() / 2);
    return std::make_unique<InternalNode>(makeTree(std::vector<const Object*>(objects.begin(), mid), cost),
                                          makeTree(std::vector<const Object*>(mid, objects.end()), cost));
}

}   // namespace ObjectTree

// ================================================================================================
// GPU context shared by Renderer, Scene uploads and the tone mappers
// ================================================================================================
namespace mipt_host {

[[noreturn]] void fail(const char* what, int status) {
    std::fprintf(stderr, "pathtracer: %s: %s (status %d). The render path runs on the GPU only; there is no CPU fallback.\n", what,
                 mipt_strerror(status), status);
    std::abort();
}

mipt_ctx* shared_context() {
    static std::mutex m;
    static mipt_ctx* ctx = nullptr;
    std::lock_guard<std::mutex> lock(m);
    if (!ctx) {
        int device = 0;
        if (const char* e = std::getenv("MIPT_DEVICE")) device = std::atoi(e);
        const int rc = mipt_init(device, nullptr, &ctx);
        if (rc != MIPT_OK) fail("mipt_init", rc);
    }
    return ctx;
}

struct SceneDeviceState {
    // Renderer::render takes its scene by const reference and the reference's is safe for concurrent readers
    // (src/Renderer.cpp:139-172); here a render refreshes the device copy first, so renders of one scene are serialised.
    std::mutex mutex;
    mipt_scene* scene = nullptr;
    // when the renderer drives several GPUs (MIPT_GPUS): the scene on every device of that renderer's node
    mipt_node* node = nullptr;
    mipt_node_scene* node_scene = nullptr;
    std::vector<uint32_t> kind, material;
    std::vector<float> fwd, inv, params, materials;
    float noise[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    // objects without a device form, by primitive index (null for device primitives): evaluated through their own
    // virtuals by the callbacks below
    std::vector<const Object*> host_objects;
    // device twins of SDF objects that have been checked against the object's own signedDistance (twin_matches)
    struct TwinVerdict {
        const Object* object;
        int kind;
        float params[9], noise[12], material[14];
        bool ok;
    };
    std::vector<TwinVerdict> twin_cache;
    ~SceneDeviceState() {
        if (scene) mipt_scene_free(scene);
        if (node_scene) mipt_node_scene_free(node, node_scene);
    }
};

struct RendererState {
    std::mutex mutex;
    std::uint64_t seed = 0x5eed;
    std::uint64_t frame = 0;
    bool started = false;
    // Every Renderer drives the GPU through a context of its own (its own stream, working set and cancel flag), so
    // cancelRender() stops this renderer's frame and no other's; scenes and tone mappers use the shared context of
    // the same device.
    std::atomic<mipt_ctx*> ctx{nullptr};
    // MIPT_GPUS = N (or "all"): this renderer divides every frame's samples among N GPUs of the node and combines
    // them with one RCCL all-reduce (mipt_init_all); unset or 1: the single device MIPT_DEVICE names.
    std::atomic<mipt_node*> node{nullptr};
    bool node_tried = false;
    mipt_node* gpus() {   // caller holds `mutex`
        if (!node_tried) {
            node_tried = true;
            if (const char* e = std::getenv("MIPT_GPUS")) {
                const int n = std::strcmp(e, "all") == 0 ? 0 : std::atoi(e);
                if (n == 0 || n > 1) {
                    mipt_node* nd = nullptr;
                    const int rc = mipt_init_all(n, nullptr, 0, &nd);
                    if (rc != MIPT_OK) fail("mipt_init_all (MIPT_GPUS)", rc);
                    node.store(nd, std::memory_order_release);
                }
            }
        }
        return node.load(std::memory_order_acquire);
    }
    mipt_ctx* context() {   // caller holds `mutex`
        mipt_ctx* c = ctx.load(std::memory_order_acquire);
        if (!c) {
            int device = 0;
            if (const char* e = std::getenv("MIPT_DEVICE")) device = std::atoi(e);
            const int rc = mipt_init(device, nullptr, &c);
            if (rc != MIPT_OK) fail("mipt_init", rc);
            ctx.store(c, std::memory_order_release);
        }
        return c;
    }
    ~RendererState() {
        if (mipt_ctx* c = ctx.load()) mipt_shutdown(c);
        if (mipt_node* nd = node.load()) mipt_node_shutdown(nd);
    }
};

// The callbacks of mipt_scene_upload_host for objects the device does not know: the object's own non-virtual wrappers
// (src/Object.cpp:22-44), which call the user's hitLocalRay / deflectLocalRay.  Called from the library's worker threads.
namespace {
int host_object_hit(void* user, uint32_t prim, const float pos[3], const float dir[3], float out[3]) {
    const auto* st = static_cast<const SceneDeviceState*>(user);
    const auto hit = st->host_objects[prim]->hitRay(Ray(Vec(dir[0], dir[1], dir[2]), Pos(pos[0], pos[1], pos[2])));
    if (!hit) return 0;
    out[0] = hit->x;
    out[1] = hit->y;
    out[2] = hit->z;
    return 1;
}
void host_object_deflect(void* user, uint32_t prim, const float pos[3], const float dir[3], float out9[9]) {
    const auto* st = static_cast<const SceneDeviceState*>(user);
    const ColorBounce b = st->host_objects[prim]->deflectRay(Ray(Vec(dir[0], dir[1], dir[2]), Pos(pos[0], pos[1], pos[2])));
    out9[0] = b.emitted.r; out9[1] = b.emitted.g; out9[2] = b.emitted.b;
    out9[3] = b.attenuation.r; out9[4] = b.attenuation.g; out9[5] = b.attenuation.b;
    out9[6] = b.rayDirection.x; out9[7] = b.rayDirection.y; out9[8] = b.rayDirection.z;
}
}   // namespace

// Is the device twin that `d` names the class this object really is?  A demo class is recognised by its name and size
// (describeKnownDemoSdf), and a program is free to change what a class of that name computes (include/Object.hpp:107-177
// lets Derived::signedDistance and deflectLocalRay be anything): the device would go on rendering the formula it knows.
// So before a twin is used, the object's own signedDistance is compared with the device's (mipt_sdf_eval) at several
// hundred points of its local box -- short sphere-traced walks towards the surface, so that most of them lie in the
// near field, plus points spread over the box -- and its deflectRay with the device's per-hit material
// (mipt_sdf_material_eval) at the walks' ends.  One primitive, identity transform, on the shared context.
bool twin_matches(const Object& object, const DevicePrimitive& d, const float noise[12], const char** why) {
    static std::string reason;
    std::uint32_t rng = 0x9e3779b9u;
    const auto u01 = [&rng]() {
        rng = rng * 1664525u + 1013904223u;
        return static_cast<float>(rng >> 8) * (1.0f / 16777216.0f);
    };
    const float* bc = d.probe_box;
    const float* bh = d.probe_box + 3;
    const auto in_box = [&](const float p[3]) {
        return std::abs(p[0] - bc[0]) <= bh[0] && std::abs(p[1] - bc[1]) <= bh[1] && std::abs(p[2] - bc[2]) <= bh[2];
    };
    std::vector<float> pts, want, ends;
    const auto visit = [&](const float p[3]) {
        const float v = d.probe_sdf(d.probe_object, p);
        pts.insert(pts.end(), p, p + 3);
        want.push_back(v);
        return v;
    };
    for (int walk = 0; walk < 48; ++walk) {
        // from a point of the box's surface towards a point near its centre
        float p[3], target[3], dir[3];
        const int face = static_cast<int>(u01() * 6.0f) % 6;
        for (int k = 0; k < 3; ++k) {
            p[k] = bc[k] + bh[k] * (2.0f * u01() - 1.0f);
            target[k] = bc[k] + 0.3f * bh[k] * (2.0f * u01() - 1.0f);
        }
        p[face / 2] = bc[face / 2] + ((face & 1) ? bh[face / 2] : -bh[face / 2]) * 0.999f;
        float len = 0.0f;
        for (int k = 0; k < 3; ++k) {
            dir[k] = target[k] - p[k];
            len += dir[k] * dir[k];
        }
        len = std::sqrt(len);
        if (!(len > 0.0f)) continue;
        for (int k = 0; k < 3; ++k) dir[k] /= len;
        for (int step = 0; step < 14 && in_box(p); ++step) {
            const float v = visit(p);
            if (!(std::abs(v) <= 3.0e38f)) break;
            if (std::abs(v) < 2e-4f) {
                ends.insert(ends.end(), p, p + 3);
                break;
            }
            const float adv = std::max(std::abs(v), 1e-3f);
            for (int k = 0; k < 3; ++k) p[k] += adv * dir[k];
        }
    }
    for (int k = 0; k < 128; ++k) {
        const float p[3] = {bc[0] + bh[0] * (2.0f * u01() - 1.0f), bc[1] + bh[1] * (2.0f * u01() - 1.0f), bc[2] + bh[2] * (2.0f * u01() - 1.0f)};
        visit(p);
    }
    // the twin, alone in a scene of its own
    const std::uint32_t kind = static_cast<std::uint32_t>(d.kind), material = 0;
    const float identity[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    mipt_scene_desc desc{};
    desc.n_prims = 1;
    desc.kind = &kind;
    desc.fwd = identity;
    desc.inv = identity;
    desc.params = d.params;
    desc.material = &material;
    desc.n_materials = 1;
    desc.materials = d.material->record();
    desc.noise_transform = noise;
    mipt_scene* twin = nullptr;
    int rc = mipt_scene_upload(shared_context(), &desc, &twin);
    if (rc != MIPT_OK) fail("mipt_scene_upload (device twin check)", rc);
    const std::uint32_t n = static_cast<std::uint32_t>(want.size());
    std::vector<float> got(n);
    rc = mipt_sdf_eval(shared_context(), twin, 0, n, pts.data(), got.data());
    if (rc != MIPT_OK) fail("mipt_sdf_eval (device twin check)", rc);
    std::uint32_t off = 0;
    float worst = 0.0f;
    for (std::uint32_t i = 0; i < n; ++i) {
        const float diff = std::abs(got[i] - want[i]);
        if (!(diff <= 2e-5f * std::max(1.0f, std::abs(want[i])))) ++off;
        if (diff > worst || diff != diff) worst = diff;
    }
    bool ok = off * 100u <= n;   // a point on a crater's rim or a cell wall may fall on the other side in float arithmetic
    if (!ok) reason = "signedDistance differs from the device's at " + std::to_string(off) + " of " + std::to_string(n) + " probe points (largest difference " + std::to_string(worst) + ")";
    // the per-hit material, where the walks ended on the surface: struck from outside along the inward normal, every
    // lobe of BasicMaterial::deflect returns the material's emission and reflected absorption (src/BasicMaterial.cpp:123-159)
    const std::uint32_t n_hits = static_cast<std::uint32_t>(ends.size() / 3);
    if (ok && n_hits) {
        std::vector<float> mats(static_cast<std::size_t>(n_hits) * 9);
        rc = mipt_sdf_material_eval(shared_context(), twin, 0, n_hits, ends.data(), mats.data());
        if (rc != MIPT_OK) fail("mipt_sdf_material_eval (device twin check)", rc);
        std::uint32_t bad = 0;
        const Affine& T = object.transformation();
        for (std::uint32_t i = 0; i < n_hits; ++i) {
            const float* p = &ends[3 * static_cast<std::size_t>(i)];
            const float delta = 1e-3f, v0 = d.probe_sdf(d.probe_object, p);
            float g[3];
            for (int k = 0; k < 3; ++k) {
                float q[3] = {p[0], p[1], p[2]};
                q[k] += delta;
                g[k] = d.probe_sdf(d.probe_object, q) - v0;
            }
            const Vec nrm = Vec(g[0], g[1], g[2]).unit();
            if (!(nrm.x == nrm.x)) continue;
            const ColorBounce b = object.deflectRay(Ray(T * (-1.0f * nrm), T * Pos(p[0], p[1], p[2])));
            const float* m = &mats[9 * static_cast<std::size_t>(i)];
            const float host6[6] = {b.attenuation.r, b.attenuation.g, b.attenuation.b, b.emitted.r, b.emitted.g, b.emitted.b};
            bool same = true;
            for (int k = 0; k < 6; ++k) same = same && std::abs(host6[k] - m[k]) <= 0.02f * std::max(1.0f, std::abs(host6[k]));
            bad += same ? 0u : 1u;
        }
        ok = bad * 4u <= n_hits;   // forward-difference normals and cell-wall points: a few hits may disagree
        if (!ok) reason = "deflectLocalRay's material differs from the device's at " + std::to_string(bad) + " of " + std::to_string(n_hits) + " probe hits";
    }
    mipt_scene_free(twin);
    if (why) *why = reason.c_str();
    return ok;
}

// Flattens the objects (Scene::updateGeometry's job in this design) and uploads when anything changed.
// The caller holds the scene's device-state lock until it has finished rendering from the returned scene.
mipt_scene* sync_scene(const Scene& scene, mipt_node* node = nullptr) {
    SceneDeviceState& st = scene.deviceState();
    const auto& objs = scene.objects();
    const std::size_t n = objs.size();
    if (n == 0) fail("Renderer::render: the scene has no objects", MIPT_ERR_ARGUMENT);
    std::vector<uint32_t> kind(n), material(n);
    std::vector<float> fwd(12 * n), inv(12 * n), params(9 * n), materials(14 * n);
    bool any_sdf = false;
    const float* own_noise = nullptr;
    std::vector<const Object*> host_objects(n, nullptr);
    bool any_host = false;
    static const bool check_twins = !(std::getenv("MIPT_TRUST_TWINS") && std::atoi(std::getenv("MIPT_TRUST_TWINS")) != 0);
    static const BasicMaterial unused_material;   // the material row of a host-evaluated object is never read
    for (std::size_t i = 0; i < n; ++i) {
        DevicePrimitive d;
        objs[i]->describeForDevice(d);
        if (d.kind >= 0 && d.material && d.probe_sdf && check_twins) {
            // a device twin of an SDF class: trusted once it has been seen to compute what the object computes
            float nz[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
            if (d.noise) std::memcpy(nz, d.noise, sizeof(nz));
            else if (&noiseTransform) affine_to(noiseTransform(), nz);
            const SceneDeviceState::TwinVerdict* known = nullptr;
            for (const auto& v : st.twin_cache)
                if (v.object == objs[i].get() && v.kind == d.kind && std::memcmp(v.params, d.params, sizeof(v.params)) == 0 &&
                    std::memcmp(v.noise, nz, sizeof(nz)) == 0 && std::memcmp(v.material, d.material->record(), sizeof(v.material)) == 0)
                    known = &v;
            bool ok;
            if (known) {
                ok = known->ok;
            } else {
                const char* why = "";
                ok = twin_matches(*objs[i], d, nz, &why);
                SceneDeviceState::TwinVerdict v{};
                v.object = objs[i].get();
                v.kind = d.kind;
                std::memcpy(v.params, d.params, sizeof(v.params));
                std::memcpy(v.noise, nz, sizeof(nz));
                std::memcpy(v.material, d.material->record(), sizeof(v.material));
                v.ok = ok;
                st.twin_cache.push_back(v);
                if (!ok)
                    std::fprintf(stderr, "pathtracer: %s has the name of a class the device knows, but it is not that class: its %s.  It is evaluated by its own "
                                         "virtuals on the host.\n", d.type_name, why);
            }
            if (!ok) d.kind = -1;
        }
        if (d.kind < 0 || !d.material) {
            // Any other Object subclass (the reference lets a user override hitLocalRay / deflectLocalRay / getLocalBoundingBox,
            // include/Object.hpp:45-52): no device form -- rays that reach its world box are evaluated by the object's own
            // virtuals on host threads, between the device stages (the slow path; said once per scene upload).
            d.kind = MIPT_HOST_OBJECT;
            d.material = &unused_material;
            const AxisAlignedBox box = objs[i]->getBoundingBox();
            const float p[9] = {box.center.x, box.center.y, box.center.z, box.halfSize.x, box.halfSize.y, box.halfSize.z, 0.0f, 0.0f, 0.0f};
            std::memcpy(d.params, p, sizeof(p));
            host_objects[i] = objs[i].get();
            any_host = true;
        }
        kind[i] = static_cast<uint32_t>(d.kind);
        any_sdf |= d.kind == MIPT_CRATERBALL || d.kind >= MIPT_ROUGHBOX;
        if (d.noise && !own_noise) own_noise = d.noise;
        material[i] = static_cast<uint32_t>(i);
        affine_to(objs[i]->transformation(), &fwd[12 * i]);
        affine_to(objs[i]->inverseTransformation(), &inv[12 * i]);
        std::memcpy(&params[9 * i], d.params, sizeof(d.params));
        std::memcpy(&materials[14 * i], d.material->record(), 14 * sizeof(float));
    }
    float noise[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    if (own_noise) {
        std::memcpy(noise, own_noise, sizeof(noise));   // one noise transform per scene: the first CraterBall's
    } else if (any_sdf) {
        if (!&noiseTransform) {
            std::fprintf(stderr, "pathtracer: the scene holds one of the demo's SDF objects but the program defines no noiseTransform()\n");
            std::abort();
        }
        affine_to(noiseTransform(), noise);
    }
    const bool uploaded = node ? (st.node_scene && st.node == node) : st.scene != nullptr;
    const bool same = uploaded && kind == st.kind && material == st.material && fwd == st.fwd && inv == st.inv && params == st.params &&
                      materials == st.materials && std::memcmp(noise, st.noise, sizeof(noise)) == 0 && host_objects == st.host_objects;
    if (same) return st.scene;
    if (st.scene) {
        mipt_scene_free(st.scene);
        st.scene = nullptr;
    }
    if (st.node_scene) {
        mipt_node_scene_free(st.node, st.node_scene);
        st.node_scene = nullptr;
    }
    mipt_scene_desc desc{};
    desc.n_prims = static_cast<uint32_t>(n);
    desc.kind = kind.data();
    desc.fwd = fwd.data();
    desc.inv = inv.data();
    desc.params = params.data();
    desc.material = material.data();
    desc.n_materials = static_cast<uint32_t>(n);
    desc.materials = materials.data();
    desc.noise_transform = noise;
    st.host_objects = std::move(host_objects);
    int rc;
    const mipt_host_callbacks host_cb{host_object_hit, host_object_deflect, &st};
    if (any_host) {
        std::size_t count = 0;
        for (const Object* o : st.host_objects) count += o ? 1 : 0;
        std::fprintf(stderr, "pathtracer: %zu object(s) of this scene have no device form (first: %s); their rays are evaluated by their own virtuals on the "
                             "host between the GPU stages -- the slow path\n", count, typeid(**std::find_if(st.host_objects.begin(), st.host_objects.end(), [](const Object* o) { return o != nullptr; })).name());
    }
    if (node) {
        st.node = node;
        rc = mipt_node_scene_upload_host(node, &desc, any_host ? &host_cb : nullptr, &st.node_scene);
    } else if (any_host) {
        rc = mipt_scene_upload_host(shared_context(), &desc, std::getenv("MIPT_DEVICE_BVH") && std::atoi(std::getenv("MIPT_DEVICE_BVH")) ? static_cast<uint32_t>(MIPT_UPLOAD_DEVICE_BVH) : 0u, &host_cb,
                                    &st.scene);
    } else {
        rc = mipt_scene_upload(shared_context(), &desc, &st.scene);
    }
    if (rc != MIPT_OK) fail("mipt_scene_upload", rc);
    st.kind = std::move(kind);
    st.material = std::move(material);
    st.fwd = std::move(fwd);
    st.inv = std::move(inv);
    st.params = std::move(params);
    st.materials = std::move(materials);
    std::memcpy(st.noise, noise, sizeof(noise));
    return st.scene;
}

}   // namespace mipt_host

// ================================================================================================
// scene (src/Scene.cpp)
// ================================================================================================
Scene::Scene() : m_objectTree(std::vector<const Object*>{}), m_device(std::make_unique<mipt_host::SceneDeviceState>()) {}
Scene::~Scene() = default;
Scene::Scene(Scene&&) noexcept = default;
Scene& Scene::operator=(Scene&&) noexcept = default;

void Scene::addObject(std::unique_ptr<Object> object) noexcept { m_objects.push_back(std::move(object)); }

Color Scene::trace(Ray ray, std::size_t depth) const noexcept {   // :18-78
    Color color;
    Color att(1.0f, 1.0f, 1.0f);
    for (std::size_t i = 0; i < depth; ++i) {
        const auto hit = m_objectTree.hit(ray);
        if (!hit) return color;
        ray.pos = hit->first;
        const ColorBounce b = hit->second->deflectRay(ray);
        ray.pos += 1e-3f * b.rayDirection;
        color.r += b.emitted.r * att.r;
        color.g += b.emitted.g * att.g;
        color.b += b.emitted.b * att.b;
        att.r *= b.attenuation.r;
        att.g *= b.attenuation.g;
        att.b *= b.attenuation.b;
        ray.dir = b.rayDirection;
        if (att.r + att.b + att.g < 1e-6f) return color;
    }
    return color;
}

void Scene::updateGeometry() {   // :80-87
    std::vector<const Object*> v;
    v.reserve(m_objects.size());
    for (const auto& o : m_objects) v.push_back(o.get());
    m_objectTree = ObjectTree::Tree{std::move(v)};
}

// ================================================================================================
// camera, settings, image
// ================================================================================================
Camera::Camera(Affine transform, float aspectRatio, float fieldOfView) noexcept
    : m_transform(transform), m_aspectRatio(aspectRatio), m_fieldOfView(fieldOfView), m_focalDistance(10.0f), m_focalBlurRadius(0.0f) {}
void Camera::setAspectRatio(float r) noexcept { m_aspectRatio = r; }
void Camera::setFieldOfView(float f) noexcept { m_fieldOfView = f; }
void Camera::setFocalDistance(float d) noexcept { m_focalDistance = d; }
void Camera::setFocalBlurRadius(float r) noexcept { m_focalBlurRadius = r; }

Ray Camera::getViewRay(float screenX, float screenY) const noexcept {   // src/Camera.cpp:65-81
    const float x = screenX * 2.0f - 1.0f, y = screenY * 2.0f - 1.0f;
    const Pos sp = m_aspectRatio > 1.0f ? Pos(x, y / m_aspectRatio) : Pos(x * m_aspectRatio, y);
    const auto [rx, ry] = randomPointInCircle();
    const float blurRad = m_focalBlurRadius * std::max(m_aspectRatio, 1.0f / m_aspectRatio);
    const Vec blur(rx * blurRad, ry * blurRad, 0.0f);
    const float fovScale = std::tan(m_fieldOfView * 3.141592654f / 180.0f);
    const Vec view = Vec(fovScale * sp.x, fovScale * sp.y, 1.0f) + (blur / m_focalDistance);
    return Ray((m_transform * view).unit(), m_transform * (sp - blur));
}

RenderSettings::RenderSettings(std::size_t w, std::size_t h) noexcept : m_width(w), m_height(h), m_numBounces(8), m_samplesPerPixel(1) {
    assert(m_width > 0 && m_height > 0);
}
// asserted as in src/RenderSettings.cpp:5-47
void RenderSettings::setSize(size_t w, size_t h) noexcept { assert(w > 0 && h > 0); m_width = w; m_height = h; }
void RenderSettings::setNumBounces(std::size_t n) noexcept { assert(n > 0); m_numBounces = n; }
void RenderSettings::setSamplesPerPixel(std::size_t s) noexcept { assert(s > 0); m_samplesPerPixel = s; }

Image::Image(std::size_t w, std::size_t h) : m_width(w), m_height(h), m_data(w * h, Color{}) {}

void Image::save(const std::string& path) const {   // src/Image.cpp:48-60
    std::ofstream f(path, std::ios::out | std::ios::binary);
    const std::uint64_t w = m_width, h = m_height;
    f.write(reinterpret_cast<const char*>(&w), sizeof(w));
    f.write(reinterpret_cast<const char*>(&h), sizeof(h));
    f.write(reinterpret_cast<const char*>(raw()), static_cast<std::streamsize>(m_data.size() * sizeof(Color)));
}

Image Image::load(const std::string& path) {   // :62-81
    std::ifstream f(path, std::ios::in | std::ios::binary);
    if (!f) throw std::runtime_error("Failed to open file: \"" + path + "\"");
    std::uint64_t w = 0, h = 0;
    f.read(reinterpret_cast<char*>(&w), sizeof(w));
    f.read(reinterpret_cast<char*>(&h), sizeof(h));
    Image img(w, h);
    f.read(reinterpret_cast<char*>(img.raw()), static_cast<std::streamsize>(img.m_data.size() * sizeof(Color)));
    return img;
}

void Image::fill(const Color& c) noexcept { std::fill(m_data.begin(), m_data.end(), c); }

Image& Image::operator+=(const Image& other) noexcept {
    for (size_t i = 0; i < m_data.size(); ++i) m_data[i] += other.m_data[i];
    return *this;
}

// ================================================================================================
// renderer (src/Renderer.cpp)
// ================================================================================================
Renderer::Renderer() noexcept : m_state(new mipt_host::RendererState) {}   // (noexcept as in the reference: failing to allocate terminates)
Renderer::~Renderer() = default;

void Renderer::startThreadPool(size_t) {
    std::lock_guard<std::mutex> lock(m_state->mutex);
    mipt_host::shared_context();
    m_state->context();
    m_state->started = true;
}
void Renderer::stopThreadPool() { m_state->started = false; }

namespace {
mipt_camera pack_camera(const Camera& camera) {
    mipt_camera cam{};
    affine_to(camera.transform(), cam.transform);
    cam.aspect = camera.aspectRatio();
    cam.fov_degrees = camera.fieldOfView();
    cam.focal_distance = camera.focalDistance();
    cam.blur_radius = camera.focalBlurRadius();
    return cam;
}
// settings of the renderer's next frame; advances its frame counter (caller holds the renderer's lock)
mipt_settings next_frame_settings(mipt_host::RendererState& rs, const RenderSettings& settings) {
    mipt_settings st{};
    st.width = static_cast<uint32_t>(settings.width());
    st.height = static_cast<uint32_t>(settings.height());
    st.bounces = static_cast<uint32_t>(settings.numBounces());
    st.spp = static_cast<uint32_t>(settings.samplesPerPixel());
    st.seed = rs.seed;
    st.first_sample = static_cast<uint32_t>(rs.frame * settings.samplesPerPixel());
    rs.frame += 1;
    return st;
}
}   // namespace

Image Renderer::render(const Scene& scene, const Camera& camera, const RenderSettings& settings) {
    std::lock_guard<std::mutex> lock(m_state->mutex);
    std::lock_guard<std::mutex> scene_lock(scene.deviceState().mutex);
    if (mipt_node* node = m_state->gpus()) {   // the samples divided among the node's GPUs, one all-reduce
        mipt_host::sync_scene(scene, node);
        const mipt_camera cam = pack_camera(camera);
        const mipt_settings st = next_frame_settings(*m_state, settings);
        Image img(settings.width(), settings.height());
        const int rc = mipt_node_render(node, scene.deviceState().node_scene, &cam, &st, -1, img.raw(), nullptr);
        if (rc != MIPT_OK && rc != MIPT_ERR_CANCELLED) mipt_host::fail("mipt_node_render", rc);
        return img;
    }
    mipt_ctx* ctx = m_state->context();
    mipt_scene* dev = mipt_host::sync_scene(scene);
    const mipt_camera cam = pack_camera(camera);
    const mipt_settings st = next_frame_settings(*m_state, settings);
    Image img(settings.width(), settings.height());
    const int rc = mipt_render(ctx, dev, &cam, &st, img.raw(), nullptr);
    if (rc != MIPT_OK && rc != MIPT_ERR_CANCELLED) mipt_host::fail("mipt_render", rc);
    return img;
}

void Renderer::cancelRender() noexcept {   // any thread, lock-free (src/Renderer.cpp:174-176)
    if (mipt_ctx* c = m_state->ctx.load(std::memory_order_acquire)) mipt_cancel(c);
    if (mipt_node* nd = m_state->node.load(std::memory_order_acquire)) mipt_node_cancel(nd);
}

void Renderer::setSeed(std::uint64_t seed) noexcept {
    m_state->seed = seed;
    m_state->frame = 0;
}

// ================================================================================================
// progressive display with the accumulator on the GPU (PathTracerDemo.cpp:934-974)
// ================================================================================================
ProgressiveFrame::ProgressiveFrame(std::size_t width, std::size_t height) : m_width(width), m_height(height) {
    const int rc = mipt_device_floats_alloc(mipt_host::shared_context(), width * height * 3, &m_device);
    if (rc != MIPT_OK) mipt_host::fail("mipt_device_floats_alloc", rc);
}

ProgressiveFrame::~ProgressiveFrame() { mipt_device_floats_free(mipt_host::shared_context(), m_device); }

void ProgressiveFrame::reset() {
    const int rc = mipt_device_floats_zero(mipt_host::shared_context(), m_device, m_width * m_height * 3);
    if (rc != MIPT_OK) mipt_host::fail("mipt_device_floats_zero", rc);
    m_count = 0;
    m_samples = 0;
}

bool ProgressiveFrame::add(Renderer& renderer, const Scene& scene, const Camera& camera, const RenderSettings& settings) {
    if (settings.width() != m_width || settings.height() != m_height) mipt_host::fail("ProgressiveFrame::add: the settings' size is not the frame's", MIPT_ERR_ARGUMENT);
    std::lock_guard<std::mutex> lock(renderer.m_state->mutex);
    mipt_ctx* ctx = renderer.m_state->context();
    std::lock_guard<std::mutex> scene_lock(scene.deviceState().mutex);
    mipt_scene* dev = mipt_host::sync_scene(scene);
    const mipt_camera cam = pack_camera(camera);
    const mipt_settings st = next_frame_settings(*renderer.m_state, settings);
    mipt_stats stats{};
    const int rc = mipt_render_accumulate(ctx, dev, &cam, &st, m_device, &stats);
    if (rc == MIPT_ERR_CANCELLED) {
        // the sample batches that pass had finished are in the sums: they are counted, so that the mean stays a mean
        m_samples += stats.samples_done;
        m_count += stats.samples_done ? 1 : 0;
        return false;
    }
    if (rc != MIPT_OK) mipt_host::fail("mipt_render_accumulate", rc);
    m_samples += settings.samplesPerPixel();
    m_count += 1;
    return true;
}

// Checkpoint: the radiance SUMS in Image's raw format (src/Image.cpp:48-60: u64 width, u64 height, row-major f32 r, g, b),
// so Image::load reads the file -- the reference's too, which stops after the pixels -- followed by five u64: a tag,
// the samples per pixel and passes accumulated, and the renderer's seed and frame counter (which sample indices come next).
namespace {
constexpr std::uint64_t kCheckpointTag = 0x3143434154504d4dull;   // "MMPTACC1"
}

void ProgressiveFrame::save(const std::string& path, const Renderer& renderer) const {
    Image sums(m_width, m_height);
    const int rc = mipt_device_floats_read(mipt_host::shared_context(), m_device, 1.0f, m_width * m_height * 3, sums.raw());
    if (rc != MIPT_OK) mipt_host::fail("mipt_device_floats_read", rc);
    sums.save(path);
    std::ofstream f(path, std::ios::out | std::ios::binary | std::ios::app);
    std::uint64_t seed, frame;
    {
        std::lock_guard<std::mutex> lock(renderer.m_state->mutex);
        seed = renderer.m_state->seed;
        frame = renderer.m_state->frame;
    }
    const std::uint64_t tail[5] = {kCheckpointTag, m_samples, m_count, seed, frame};
    f.write(reinterpret_cast<const char*>(tail), sizeof(tail));
}

bool ProgressiveFrame::load(const std::string& path, Renderer& renderer) {
    const Image sums = Image::load(path);   // throws std::runtime_error when the file cannot be opened, as Image::load does
    if (sums.width() != m_width || sums.height() != m_height) return false;
    std::ifstream f(path, std::ios::in | std::ios::binary);
    f.seekg(static_cast<std::streamoff>(16 + m_width * m_height * 12));
    std::uint64_t tail[5] = {};
    f.read(reinterpret_cast<char*>(tail), sizeof(tail));
    if (!f || tail[0] != kCheckpointTag) return false;   // a plain image, not a checkpoint
    const int rc = mipt_device_floats_write(mipt_host::shared_context(), m_device, m_width * m_height * 3, sums.raw());
    if (rc != MIPT_OK) mipt_host::fail("mipt_device_floats_write", rc);
    m_samples = tail[1];
    m_count = static_cast<std::size_t>(tail[2]);
    std::lock_guard<std::mutex> lock(renderer.m_state->mutex);
    renderer.m_state->seed = tail[3];
    renderer.m_state->frame = tail[4];
    return true;
}

Image ProgressiveFrame::image() const {
    Image img(m_width, m_height);
    if (m_samples == 0) return img;
    const int rc = mipt_device_floats_read(mipt_host::shared_context(), m_device, 1.0f / static_cast<float>(m_samples), m_width * m_height * 3, img.raw());
    if (rc != MIPT_OK) mipt_host::fail("mipt_device_floats_read", rc);
    return img;
}

void ProgressiveFrame::present(std::uint8_t* rgba, int toneMapper) const {
    const float k = m_samples ? 1.0f / static_cast<float>(m_samples) : 0.0f;
    const int rc = mipt_present_rgba8(mipt_host::shared_context(), m_device, k, toneMapper, rgba, m_width, m_height);
    if (rc != MIPT_OK) mipt_host::fail("mipt_present_rgba8", rc);
}

// ================================================================================================
// tone mappers (src/ToneMapper.cpp), on the GPU
// ================================================================================================
namespace {
Image tonemap(int kind, const Image& img) noexcept {
    Image out(img.width(), img.height());
    const int rc = mipt_tonemap(mipt_host::shared_context(), kind, img.raw(), out.raw(), img.width(), img.height());
    if (rc != MIPT_OK) mipt_host::fail("mipt_tonemap", rc);
    return out;
}
}   // namespace

Image ReinhardToneMapper::operator()(const Image& img) const noexcept { return tonemap(0, img); }
Image FilmicToneMapper::operator()(const Image& img) const noexcept { return tonemap(1, img); }
