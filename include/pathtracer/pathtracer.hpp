// Host-side scene and render API of the MI355X path-tracing core.
//
// The public (and, for subclassing, protected virtual) signatures are those of timstr/pathtracer's
// headers, so code written against the reference -- PathTracerDemo.cpp included -- compiles against
// this header unchanged; the reference's per-class header names exist next to this file and simply
// include it.  What differs is everything behind the signatures:
//
//   * Scene::updateGeometry flattens the objects into primitive / material records and uploads them
//     through the C ABI (include/mipt.h); Renderer::render runs the HIP wavefront kernels and returns
//     the Image; the tone mappers run on the GPU.  Renderer::render has no CPU path: without a usable
//     GPU, or with an Object subclass the device does not know, it reports the reason and aborts.
//   * The per-ray functions the API exposes (intersect, Object::hitRay, Scene::trace, Camera::getViewRay,
//     BasicMaterial::deflect ...) remain host functions with the reference's arithmetic.
//
// Citations are file:line of the reference checkout.
#pragma once

#include <array>
#include <atomic>
#include <barrier>   // the reference's Renderer.hpp includes <barrier> and <thread>; programs written against it
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <optional>
#include <random>
#include <string>
#include <thread>    // (PathTracerDemo.cpp uses std::thread through that include)
#include <typeinfo>
#include <utility>
#include <vector>

// ================================================================================================
// random numbers (include/RandomNumberGenerator.hpp): the host functions that sample (Camera blur,
// BasicMaterial::deflect, hemisphere samplers) draw from this thread-local engine, as the reference's do.
// The device path does not use it: it addresses random numbers by (seed, pixel, sample, bounce, draw).
// ================================================================================================
std::default_random_engine& randomEngine() noexcept;

// ================================================================================================
// linear algebra (include/LinearAlgebra.hpp)
// ================================================================================================
class Pos;

class Vec {
public:
    float x, y, z;

    explicit Vec(float _x = 0.0f, float _y = 0.0f, float _z = 0.0f) noexcept : x(_x), y(_y), z(_z) {}

    float norm() const noexcept { return std::sqrt(normSquared()); }
    float normSquared() const noexcept { return x * x + y * y + z * z; }
    Vec unit() const noexcept;
    Vec abs() const noexcept { return Vec(std::abs(x), std::abs(y), std::abs(z)); }
    Pos toPos() const noexcept;

    Vec& operator-=(const Vec& o) noexcept { x -= o.x; y -= o.y; z -= o.z; return *this; }
    Vec& operator+=(const Vec& o) noexcept { x += o.x; y += o.y; z += o.z; return *this; }
};

class Pos {
public:
    float x, y, z;

    explicit Pos(float _x = 0.0f, float _y = 0.0f, float _z = 0.0f) noexcept : x(_x), y(_y), z(_z) {}

    Pos& operator+=(const Vec& v) noexcept { x += v.x; y += v.y; z += v.z; return *this; }
    Pos& operator-=(const Vec& v) noexcept { x -= v.x; y -= v.y; z -= v.z; return *this; }
    Vec toVec() const noexcept { return Vec(x, y, z); }
};

inline Pos Vec::toPos() const noexcept { return Pos(x, y, z); }
inline Vec Vec::unit() const noexcept {
    const float l = norm();
    return Vec(x / l, y / l, z / l);
}

// Row-major 3x3: element (i, j) is m_data[3 * i + j].
class Linear {
public:
    static Linear Identity() noexcept { return Linear(); }
    static Linear RotationX(float angle) noexcept;
    static Linear RotationY(float angle) noexcept;
    static Linear RotationZ(float angle) noexcept;
    static Linear Rotation(Vec axis, float angle) noexcept;
    static Linear ScaleX(float k) { return Linear(k, 0, 0, 0, 1, 0, 0, 0, 1); }
    static Linear ScaleY(float k) { return Linear(1, 0, 0, 0, k, 0, 0, 0, 1); }
    static Linear ScaleZ(float k) { return Linear(1, 0, 0, 0, 1, 0, 0, 0, k); }
    static Linear Scale(float k) { return Linear(k, 0, 0, 0, k, 0, 0, 0, k); }

    Linear() noexcept : m_data{1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f} {}
    Linear(Vec c1, Vec c2, Vec c3) noexcept : m_data{c1.x, c2.x, c3.x, c1.y, c2.y, c3.y, c1.z, c2.z, c3.z} {}
    Linear(float a, float b, float c, float d, float e, float f, float g, float h, float i) noexcept : m_data{a, b, c, d, e, f, g, h, i} {}

    Vec row(std::size_t i) const noexcept { return Vec(m_data[3 * i], m_data[3 * i + 1], m_data[3 * i + 2]); }
    Vec column(std::size_t i) const noexcept { return Vec(m_data[i], m_data[3 + i], m_data[6 + i]); }

    float operator[](std::size_t idx) const noexcept { return m_data[idx]; }
    float& operator[](std::size_t idx) noexcept { return m_data[idx]; }
    float operator()(std::size_t i, std::size_t j) const noexcept { return m_data[3 * i + j]; }
    float& operator()(std::size_t i, std::size_t j) noexcept { return m_data[3 * i + j]; }

    Linear transpose() const noexcept {
        return Linear(m_data[0], m_data[3], m_data[6], m_data[1], m_data[4], m_data[7], m_data[2], m_data[5], m_data[8]);
    }
    float determinant() const noexcept;
    std::optional<Linear> inverse() const noexcept;   // nullopt when the determinant is below 1e-6

    Linear& operator*=(const Linear& other) noexcept;

    const float* data() const noexcept { return m_data.data(); }

private:
    std::array<float, 9> m_data;
};

class Affine {
public:
    Affine() noexcept = default;
    explicit Affine(const Linear& _linear) noexcept : linear(_linear) {}
    explicit Affine(const Linear& _linear, const Vec& _translation) noexcept : linear(_linear), translation(_translation) {}

    static Affine Translation(const Vec& t) noexcept { return Affine(Linear(), t); }
    static Affine Translation(float x, float y, float z) noexcept { return Affine(Linear(), Vec(x, y, z)); }

    std::optional<Affine> inverse() const noexcept;

    Linear linear;
    Vec translation;
};

inline Vec operator-(const Vec& v) noexcept { return Vec(-v.x, -v.y, -v.z); }
inline Vec operator*(const Vec& v, float t) noexcept { return Vec(v.x * t, v.y * t, v.z * t); }
inline Vec operator*(float t, const Vec& v) noexcept { return v * t; }
inline Vec operator/(const Vec& v, float t) noexcept { return v * (1.0f / t); }
inline Vec operator+(const Vec& u, const Vec& v) noexcept { return Vec(u.x + v.x, u.y + v.y, u.z + v.z); }
inline Vec operator-(const Vec& u, const Vec& v) noexcept { return Vec(u.x - v.x, u.y - v.y, u.z - v.z); }
inline Pos operator+(const Pos& p, const Vec& v) noexcept { return Pos(p.x + v.x, p.y + v.y, p.z + v.z); }
inline Pos operator+(const Vec& v, const Pos& p) noexcept { return p + v; }
inline Pos operator-(const Pos& p, const Vec& v) noexcept { return Pos(p.x - v.x, p.y - v.y, p.z - v.z); }
inline Vec operator-(const Pos& p, const Pos& q) noexcept { return Vec(p.x - q.x, p.y - q.y, p.z - q.z); }
inline float operator*(const Vec& u, const Vec& v) noexcept { return u.x * v.x + u.y * v.y + u.z * v.z; }   // dot
inline Vec operator^(const Vec& u, const Vec& v) noexcept {                                                  // cross
    return Vec(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}

Linear operator-(const Linear& A) noexcept;
Linear operator+(const Linear& A, const Linear& B) noexcept;
Linear operator-(const Linear& A, const Linear& B) noexcept;
Linear operator*(const Linear& A, float t) noexcept;
Linear operator*(float t, const Linear& A) noexcept;
Linear operator/(const Linear& A, float t) noexcept;
// A sum of scaled columns, in this order (src/LinearAlgebra.cpp:380-382): the device code does the same.
inline Vec operator*(const Linear& A, const Vec& v) noexcept { return A.column(0) * v.x + A.column(1) * v.y + A.column(2) * v.z; }
inline Pos operator*(const Linear& A, const Pos& p) noexcept { return (A * p.toVec()).toPos(); }
Linear operator*(const Linear& A, const Linear& B) noexcept;
inline Vec operator*(const Affine& T, const Vec& v) { return T.linear * v; }                  // the translation does not apply
inline Pos operator*(const Affine& T, const Pos& p) { return T.linear * p + T.translation; }
Affine operator*(const Affine& A, const Linear& L) noexcept;
Affine operator*(const Linear& L, const Affine& A) noexcept;
Affine operator*(const Affine& A, const Affine& B) noexcept;

// ================================================================================================
// colour (include/Color.hpp, include/ColorBounce.hpp)
// ================================================================================================
class Color {
public:
    explicit Color(float _r = 0.0f, float _g = 0.0f, float _b = 0.0f) noexcept : r(_r), g(_g), b(_b) {}

    Color& operator+=(const Color& o) noexcept { r += o.r; g += o.g; b += o.b; return *this; }

    float r, g, b;
};

inline Color operator+(const Color& l, const Color& r) noexcept { return Color(l.r + r.r, l.g + r.g, l.b + r.b); }
inline Color operator*(float k, const Color& c) noexcept { return Color(k * c.r, k * c.g, k * c.b); }

class ColorBounce {
public:
    ColorBounce(Color _emitted, Color _attenuation, Vec _rayDirection, Vec _normal) noexcept
        : emitted(_emitted), attenuation(_attenuation), rayDirection(_rayDirection), normal(_normal) {}

    Color emitted;
    Color attenuation;
    Vec rayDirection;
    Vec normal;
};

// ================================================================================================
// geometry (include/Geometry.hpp)
// ================================================================================================
class Triangle {
public:
    Pos a, b, c;

    Triangle(Pos _a, Pos _b, Pos _c) noexcept : a(_a), b(_b), c(_c) {}

    Vec normal() const noexcept { return ((b - a) ^ (c - a)).unit(); }
    float area() const noexcept { return ((b - a) ^ (c - a)).norm() * 0.5f; }
};

class Sphere {   // centred on the local origin
public:
    float radius;

    Sphere(float _radius) noexcept : radius(_radius) {}

    float surfaceArea() const noexcept;
    float volume() const noexcept;
    float signedDistance(Pos p) const noexcept { return p.toVec().norm() - radius; }
    Vec normal(Pos p) const noexcept { return p.toVec().unit(); }
};

class Rectangle {   // a box centred on the local origin
public:
    Vec halfSize;

    Rectangle(Vec _halfSize) noexcept : halfSize(_halfSize) {}

    float surfaceArea() const noexcept;
    float volume() const noexcept { return halfSize.x * halfSize.y * halfSize.z * 8.0f; }
    float signedDistance(Pos p) const noexcept;
    Vec normal(Pos p) const noexcept;
};

class AxisAlignedBox {
public:
    Pos center;
    Vec halfSize;

    AxisAlignedBox(Pos corner, Pos oppositeCorner) noexcept;
    AxisAlignedBox(Pos _center, Vec _halfSize) noexcept : center(_center), halfSize(_halfSize) {}
};

class Ray {
public:
    Ray() noexcept = default;
    Ray(Vec _direction, Pos _position) noexcept : dir(_direction), pos(_position) {}

    Vec dir;
    Pos pos;
};

AxisAlignedBox boxContaining(const AxisAlignedBox& a, const AxisAlignedBox& b) noexcept;
Vec bounce(const Vec& inbound, const Vec& normal) noexcept;
std::optional<Pos> intersect(const Ray&, const Triangle&) noexcept;
std::optional<Pos> intersect(const Ray&, const Sphere&) noexcept;
std::optional<Pos> intersect(const Ray&, const Rectangle&) noexcept;
std::optional<Pos> intersect(const Ray&, const AxisAlignedBox&) noexcept;
bool inside(const Pos&, const Sphere&) noexcept;
bool inside(const Pos&, const Rectangle&) noexcept;
bool inside(const Pos&, const AxisAlignedBox&) noexcept;
std::pair<float, float> randomPointInCircle() noexcept;
std::pair<Vec, Vec> orthogonalPair(Vec v) noexcept;
Vec randomPointOnHemisphereUniform(Vec normal) noexcept;
Vec randomPointOnHemisphereCosine(Vec normal) noexcept;

// ================================================================================================
// material (include/BasicMaterial.hpp)
// ================================================================================================
class BasicMaterial {
public:
    BasicMaterial() noexcept;

    float diffuseReflection() const noexcept { return m_p[0]; }
    float specularReflection() const noexcept { return m_p[1]; }
    float specularSharpness() const noexcept { return m_p[2]; }
    Color reflectedAbsorption() const noexcept { return Color(m_p[3], m_p[4], m_p[5]); }
    Color emittedLuminance() const noexcept { return Color(m_p[6], m_p[7], m_p[8]); }
    float transmittance() const noexcept { return m_p[9]; }
    float indexOfRefraction() const noexcept { return m_p[10]; }
    Color internalAbsorption() const noexcept { return Color(m_p[11], m_p[12], m_p[13]); }

    void setDiffuseReflection(float) noexcept;
    void setSpecularReflection(float) noexcept;
    void setSpecularSharpness(float) noexcept;
    void setReflectedAbsorption(Color) noexcept;
    void setEmittedLuminance(Color) noexcept;
    void setTransmittance(float) noexcept;
    void setIndexOfRefraction(float) noexcept;
    void setInternalAbsorption(Color) noexcept;

    ColorBounce deflect(const Vec& inbound, const Vec& normal) const noexcept;

    // The 14 floats as the device's material table stores them (mipt_scene_desc::materials).
    const float* record() const noexcept { return m_p; }

private:
    float m_p[14];
};

// ================================================================================================
// objects (include/Object.hpp)
// ================================================================================================
namespace mipt_host {
// What the flatten step needs to know about one object; filled by Object::describeForDevice.
struct DevicePrimitive {
    int kind = -1;              // MIPT_SPHERE ... (include/mipt.h); -1: no device counterpart
    float params[9] = {};
    const BasicMaterial* material = nullptr;
    const char* type_name = "";
    const float* noise = nullptr;   // 12 floats: set by SDF kinds that carry their own noise transform
    // Set by SDFObjectCRTP: the object's own signedDistance and local bounding box.  A device twin is only trusted after
    // it has returned the object's own values at probe points (a class recognised by name may have been edited).
    float (*probe_sdf)(const void* object, const float local_pos[3]) = nullptr;
    const void* probe_object = nullptr;
    float probe_box[6] = {};    // localBoundingBox(): centre, half size
};
}   // namespace mipt_host

class Object {
public:
    Object() noexcept = default;
    virtual ~Object() noexcept = default;

    const Affine& transformation() const noexcept { return m_transformation; }
    const Affine& inverseTransformation() const noexcept { return m_inverseTransformation; }
    void setTransformation(const Affine&) noexcept;

    // World-frame wrappers around the local-frame virtuals (the local ray direction is not renormalised).
    std::optional<Pos> hitRay(const Ray& ray) const noexcept;
    ColorBounce deflectRay(const Ray& ray) const noexcept;
    AxisAlignedBox getBoundingBox() const noexcept;

    // Device description of this object; the default says "unknown to the device".
    virtual void describeForDevice(mipt_host::DevicePrimitive& out) const noexcept { out.kind = -1; out.type_name = typeid(*this).name(); }

protected:
    virtual std::optional<Pos> hitLocalRay(const Ray& ray) const noexcept = 0;
    virtual ColorBounce deflectLocalRay(const Ray& ray) const noexcept = 0;
    virtual AxisAlignedBox getLocalBoundingBox() const noexcept = 0;

private:
    Affine m_transformation;
    Affine m_inverseTransformation;
};

class TriangleObject : public Object {
public:
    Triangle geometry;
    BasicMaterial material;

    TriangleObject(Triangle _geometry, BasicMaterial _material = {});

    std::optional<Pos> hitLocalRay(const Ray& ray) const noexcept override;
    ColorBounce deflectLocalRay(const Ray& ray) const noexcept override;
    AxisAlignedBox getLocalBoundingBox() const noexcept override;
    void describeForDevice(mipt_host::DevicePrimitive& out) const noexcept override;
};

class SphereObject : public Object {
public:
    Sphere geometry;
    BasicMaterial material;

    SphereObject(Sphere _geometry, BasicMaterial _material = {}) noexcept;

    void describeForDevice(mipt_host::DevicePrimitive& out) const noexcept override;

private:
    std::optional<Pos> hitLocalRay(const Ray& ray) const noexcept override;
    ColorBounce deflectLocalRay(const Ray& ray) const noexcept override;
    AxisAlignedBox getLocalBoundingBox() const noexcept override;
};

class BoxObject : public Object {
public:
    Rectangle geometry;
    BasicMaterial material;

    BoxObject(Rectangle _geometry, BasicMaterial _material = {}) noexcept;

    void describeForDevice(mipt_host::DevicePrimitive& out) const noexcept override;

private:
    std::optional<Pos> hitLocalRay(const Ray& ray) const noexcept override;
    ColorBounce deflectLocalRay(const Ray& ray) const noexcept override;
    AxisAlignedBox getLocalBoundingBox() const noexcept override;
};

namespace mipt_host {
// A user SDF class opts in to a device twin by specialising this trait:
//     template <> struct DeviceSdf<MyBall> { static bool describe(const MyBall&, DevicePrimitive&); };
// The primary template knows nothing.
template <typename Derived>
struct DeviceSdf {
    static bool describe(const Derived&, DevicePrimitive&) noexcept { return false; }
};
// Recognises the reference demo's own SDF classes (CraterBallObject, RoughBoxObject, RoughSphereObject,
// WeirdWallObject; PathTracerDemo.cpp:156-483), which cannot opt in because the demo is compiled
// unchanged: a class is identified by its name and size, and its parameters (the noise offset, the
// Rectangle, the Sphere: floats that follow the SDFObjectCRTP base) are read in place.
bool describeKnownDemoSdf(const char* mangled_name, std::size_t derived_size, std::size_t base_size, const void* object,
                          DevicePrimitive& out) noexcept;
}   // namespace mipt_host

// Sphere tracer over Derived::signedDistance (CRTP).  Required members of Derived:
//   float Derived::signedDistance(const Pos&) const noexcept;
//   AxisAlignedBox Derived::localBoundingBox() const noexcept;
template <typename Derived>
class SDFObjectCRTP : public Object {
public:
    SDFObjectCRTP(BasicMaterial mat) : material(mat) {}

    BasicMaterial material;

    void describeForDevice(mipt_host::DevicePrimitive& out) const noexcept override {
        const auto* self = static_cast<const Derived*>(this);
        out.type_name = typeid(Derived).name();
        out.material = &material;
        out.kind = -1;
        out.probe_object = self;
        out.probe_sdf = [](const void* object, const float p[3]) noexcept -> float {
            return static_cast<const Derived*>(object)->signedDistance(Pos(p[0], p[1], p[2]));
        };
        const AxisAlignedBox bb = self->localBoundingBox();
        const float box[6] = {bb.center.x, bb.center.y, bb.center.z, bb.halfSize.x, bb.halfSize.y, bb.halfSize.z};
        for (int k = 0; k < 6; ++k) out.probe_box[k] = box[k];
        if (mipt_host::DeviceSdf<Derived>::describe(*self, out)) return;
        mipt_host::describeKnownDemoSdf(typeid(Derived).name(), sizeof(Derived), sizeof(SDFObjectCRTP<Derived>), self, out);
    }

protected:
    // Forward-difference gradient of the SDF, delta = 1e-3, normalised (include/Object.hpp:118-132).
    Vec signedDistanceNormal(const Pos& pos) const noexcept {
        const float delta = 1e-3f;
        const auto* self = static_cast<const Derived*>(this);
        const float d0 = self->signedDistance(pos);
        return (Vec(self->signedDistance(pos + Vec(delta, 0.0f, 0.0f)) - d0, self->signedDistance(pos + Vec(0.0f, delta, 0.0f)) - d0,
                    self->signedDistance(pos + Vec(0.0f, 0.0f, delta)) - d0) /
                delta)
            .unit();
    }

private:
    // include/Object.hpp:135-167: enter the bounding box, then at most 256 steps; a sign change of the
    // SDF halves the last step and backs up; |d| < 1e-4 is a hit; leaving the box is a miss.
    std::optional<Pos> hitLocalRay(const Ray& ray) const noexcept override {
        const auto* self = static_cast<const Derived*>(this);
        const AxisAlignedBox bb = self->localBoundingBox();
        std::optional<Pos> start;
        if (inside(ray.pos, bb)) start = ray.pos;
        else start = intersect(ray, bb);
        if (!start) return std::nullopt;
        Pos p = *start;
        float d = self->signedDistance(p);
        const bool sign = d > 0.0f;
        for (std::size_t i = 0; i < 256; ++i) {
            const float d2 = self->signedDistance(p);
            if ((d2 > 0.0f) != sign) {
                d *= 0.5f;
                p -= d * ray.dir;
                continue;
            }
            d = d2;
            if (std::abs(d) < 1e-4f) return p;
            p = p + std::abs(d) * ray.dir;
            if (!inside(p, bb)) return std::nullopt;
        }
        return std::nullopt;
    }

    ColorBounce deflectLocalRay(const Ray& ray) const noexcept override { return material.deflect(ray.dir, this->signedDistanceNormal(ray.pos)); }

    AxisAlignedBox getLocalBoundingBox() const noexcept override { return static_cast<const Derived*>(this)->localBoundingBox(); }
};

// The demo scene's crater ball (PathTracerDemo.cpp:352-483) as a library type, for programs that do not
// carry their own copy of the demo's classes: a radius-0.5 sphere with three scales of hashed craters
// and fBm noise, sphere-traced inside the box [-1, 1]^3.  `noise` is the demo's noiseTransform().
class CraterBall : public SDFObjectCRTP<CraterBall> {
public:
    CraterBall(BasicMaterial mat, Vec offset, Affine noise);

    float signedDistance(const Pos& pos) const noexcept;
    AxisAlignedBox localBoundingBox() const noexcept { return AxisAlignedBox(Pos(0.0f, 0.0f, 0.0f), Vec(1.0f, 1.0f, 1.0f)); }

    const Vec& offset() const noexcept { return m_offset; }
    const Affine& noise() const noexcept { return m_noise; }

private:
    Vec m_offset;
    Affine m_noise;
};

namespace mipt_host {
template <>
struct DeviceSdf<CraterBall> {
    static bool describe(const CraterBall& ball, DevicePrimitive& out) noexcept;
};
}   // namespace mipt_host

// 5 x 5 x 5 lattice of spheres as an SDF (src/Object.cpp:160-252).
class FractalObject : public Object {
public:
    BasicMaterial material;

private:
    std::optional<Pos> hitLocalRay(const Ray& ray) const noexcept override;
    ColorBounce deflectLocalRay(const Ray& ray) const noexcept override;
    AxisAlignedBox getLocalBoundingBox() const noexcept override;
    void describeForDevice(mipt_host::DevicePrimitive&) const noexcept override;
    float signedDistance(const Pos&) const noexcept;
};

// ================================================================================================
// object tree (include/ObjectTree.hpp): the host-side closest-hit structure behind Scene::trace.
// Like the reference's it returns the closest hit over ALL its objects (ties go to the first leaf).
// ================================================================================================
namespace ObjectTree {

class Node {
public:
    Node(AxisAlignedBox bb) noexcept : boundingBox(bb) {}
    virtual ~Node() noexcept = default;

    virtual std::optional<std::pair<Pos, const Object*>> hit(const Ray&) const noexcept = 0;

    AxisAlignedBox const boundingBox;
};

class InternalNode : public Node {
public:
    InternalNode(std::unique_ptr<Node> childA, std::unique_ptr<Node> childB) noexcept;

private:
    std::optional<std::pair<Pos, const Object*>> hit(const Ray&) const noexcept override;

    std::unique_ptr<Node> const m_childA;
    std::unique_ptr<Node> const m_childB;
};

class LeafNode : public Node {
public:
    LeafNode(const Object* object) noexcept;

private:
    std::optional<std::pair<Pos, const Object*>> hit(const Ray&) const noexcept override;

    const Object* const m_object;
};

class Tree {
public:
    Tree(std::vector<const Object*> objects);

    std::optional<std::pair<Pos, const Object*>> hit(const Ray&) const noexcept;

private:
    static std::unique_ptr<Node> makeTree(std::vector<const Object*> objects, float volumeVersusSplitCost);

    std::unique_ptr<Node> m_root;
};

}   // namespace ObjectTree

// ================================================================================================
// scene (include/Scene.hpp)
// ================================================================================================
namespace mipt_host {
struct SceneDeviceState;   // uploaded scene + its context, owned by Scene
}

class Scene {
public:
    Scene();
    ~Scene();
    Scene(Scene&&) noexcept;
    Scene& operator=(Scene&&) noexcept;

    void addObject(std::unique_ptr<Object> object) noexcept;

    template <typename T, typename... Args>
    T& addObject(Args&&... args) {
        auto up = std::make_unique<T>(std::forward<Args>(args)...);
        auto& r = *up;
        addObject(std::move(up));
        return r;
    }

    // One path on the host: at most `depth` ray casts; emission is gathered at every hit (src/Scene.cpp:18-78).
    Color trace(Ray ray, std::size_t depth) const noexcept;

    // Rebuilds the host tree and marks the device copy stale; the next Renderer::render flattens the
    // objects and uploads them (so transformations changed after this call are still picked up).
    void updateGeometry();

    // Used by Renderer.
    const std::vector<std::unique_ptr<Object>>& objects() const noexcept { return m_objects; }
    mipt_host::SceneDeviceState& deviceState() const noexcept { return *m_device; }

private:
    std::vector<std::unique_ptr<Object>> m_objects;
    ObjectTree::Tree m_objectTree;
    std::unique_ptr<mipt_host::SceneDeviceState> m_device;
};

// ================================================================================================
// camera, settings, image (include/Camera.hpp, include/RenderSettings.hpp, include/Image.hpp)
// ================================================================================================
class Camera {
public:
    Camera(Affine transform, float aspectRatio = 1.0f, float fieldOfView = 30.0f) noexcept;

    const Affine& transform() const noexcept { return m_transform; }
    Affine& transform() noexcept { return m_transform; }
    float aspectRatio() const noexcept { return m_aspectRatio; }
    float fieldOfView() const noexcept { return m_fieldOfView; }
    float focalDistance() const noexcept { return m_focalDistance; }
    float focalBlurRadius() const noexcept { return m_focalBlurRadius; }

    void setTransform(Affine t) noexcept { m_transform = t; }
    void setAspectRatio(float) noexcept;
    void setFieldOfView(float) noexcept;
    void setFocalDistance(float) noexcept;
    void setFocalBlurRadius(float) noexcept;

    // screenX, screenY in [0, 1]; the field of view acts as a half angle; the lens point is drawn even
    // when the blur radius is zero (src/Camera.cpp:65-81).
    Ray getViewRay(float screenX, float screenY) const noexcept;

private:
    Affine m_transform;
    float m_aspectRatio;
    float m_fieldOfView;
    float m_focalDistance;
    float m_focalBlurRadius;
};

class RenderSettings {
public:
    RenderSettings(std::size_t width = 256, std::size_t height = 256) noexcept;

    std::size_t width() const noexcept { return m_width; }
    std::size_t height() const noexcept { return m_height; }
    std::size_t numBounces() const noexcept { return m_numBounces; }
    std::size_t samplesPerPixel() const noexcept { return m_samplesPerPixel; }

    void setSize(size_t width, size_t height) noexcept;
    void setNumBounces(std::size_t) noexcept;
    void setSamplesPerPixel(std::size_t) noexcept;

private:
    std::size_t m_width, m_height, m_numBounces, m_samplesPerPixel;
};

class Image {
public:
    Image(std::size_t width, std::size_t height);

    std::size_t width() const noexcept { return m_width; }
    std::size_t height() const noexcept { return m_height; }

    // Raw file: u64 width, u64 height, then row-major f32 r, g, b (src/Image.cpp:48-81).
    void save(const std::string& path) const;
    static Image load(const std::string& path);   // throws std::runtime_error if the file cannot be opened

    const Color& operator()(std::size_t x, std::size_t y) const noexcept { return m_data[y * m_width + x]; }
    Color& operator()(std::size_t x, std::size_t y) noexcept { return m_data[y * m_width + x]; }

    void fill(const Color&) noexcept;
    Image& operator+=(const Image& other) noexcept;

    // Row-major r, g, b floats: the layout of the device's radiance buffer.
    float* raw() noexcept { return reinterpret_cast<float*>(m_data.data()); }
    const float* raw() const noexcept { return reinterpret_cast<const float*>(m_data.data()); }

private:
    size_t m_width;
    size_t m_height;
    std::vector<Color> m_data;
};
static_assert(sizeof(Color) == 3 * sizeof(float), "Image::raw relies on Color being three packed floats");

// ================================================================================================
// renderer (include/Renderer.hpp)
// ================================================================================================
namespace mipt_host {
struct RendererState;
}

class Renderer {
public:
    Renderer() noexcept;
    ~Renderer();

    Renderer(const Renderer&) = delete;
    Renderer(Renderer&&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    Renderer& operator=(Renderer&&) = delete;

    // The reference starts CPU workers here; this binds the GPU context (numThreads is ignored).
    void startThreadPool(size_t numThreads = 0);
    void stopThreadPool();

    // One frame of settings.samplesPerPixel() samples per pixel on the GPU.  Calls are serialised.
    Image render(const Scene&, const Camera&, const RenderSettings&);

    // Callable from any thread while render() runs: the frame stops at the next sample batch and
    // render() returns what was accumulated, like the reference's partially filled image.
    void cancelRender() noexcept;

    // Seed of the device's counter-based generator for the next frames (the reference seeds from
    // std::random_device).  Each render() call advances a frame counter so that progressive
    // accumulation of successive frames keeps drawing new samples.
    void setSeed(std::uint64_t seed) noexcept;

private:
    friend class ProgressiveFrame;
    std::unique_ptr<mipt_host::RendererState> m_state;
};

// The demo's progressive display (PathTracerDemo.cpp:934-974: `acc += rendered; ++count; scaled = scaleImage(acc,
// 1 / count); copyToSFImage(scaled, img)`) with the accumulator kept on the GPU: per pass nothing comes back to the host
// but the 4 bytes per pixel that go on screen.  Not part of the reference's API; a program that keeps the reference's
// loop works unchanged, this is the faster way to write it.
class ProgressiveFrame {
public:
    ProgressiveFrame(std::size_t width, std::size_t height);
    ~ProgressiveFrame();
    ProgressiveFrame(const ProgressiveFrame&) = delete;
    ProgressiveFrame& operator=(const ProgressiveFrame&) = delete;

    std::size_t width() const noexcept { return m_width; }
    std::size_t height() const noexcept { return m_height; }
    // passes accumulated since the last reset (the demo's `count`)
    std::size_t count() const noexcept { return m_count; }
    // forget what was accumulated (the demo's renderReset)
    void reset();
    // One more pass of settings.samplesPerPixel() samples, added on the device.  settings' size must be the frame's.
    // Returns false when the pass was cancelled (Renderer::cancelRender): the sample batches that pass had finished are
    // in the frame and counted (samples() says how many per pixel), so image() and present() still show a mean; the
    // demo resets its accumulator after every cancel anyway (PathTracerDemo.cpp:976-990).
    bool add(Renderer&, const Scene&, const Camera&, const RenderSettings&);
    // samples per pixel accumulated since the last reset
    std::uint64_t samples() const noexcept { return m_samples; }
    // mean over everything accumulated, as the demo's `scaled`
    Image image() const;
    // copyToSFImage's bytes (PathTracerDemo.cpp:524-545: clamp to [0, 1], times 255, truncated; R, G, B, 255) of the mean,
    // through no tone mapper (-1, as the demo shows it), Reinhard (0) or filmic (1); rgba holds width * height * 4 bytes
    void present(std::uint8_t* rgba, int toneMapper = -1) const;
    // Checkpoint and resume.  save() writes the radiance sums in Image's raw format (Image::load -- the reference's too --
    // reads the file as an image of sums) followed by the sample count and the renderer's position in its sample sequence;
    // load() puts them back (false, and nothing changed, if the file is a plain image or of another size; throws as
    // Image::load does if it cannot be opened).  A frame that is saved, loaded and continued equals the uninterrupted
    // one bit for bit.
    void save(const std::string& path, const Renderer&) const;
    bool load(const std::string& path, Renderer&);

private:
    std::size_t m_width, m_height, m_count = 0;
    std::uint64_t m_samples = 0;   // samples per pixel accumulated
    float* m_device = nullptr;     // width * height * 3 radiance sums on the GPU
};

// ================================================================================================
// tone mappers (include/ToneMapper.hpp) -- evaluated on the GPU
// ================================================================================================
class ReinhardToneMapper {
public:
    Image operator()(const Image&) const noexcept;
};

class FilmicToneMapper {
public:
    Image operator()(const Image&) const noexcept;
};
