#include "hostmath.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

// Built with -ffp-contract=off: a fused multiply-add here would change the last bit of an inverse.
namespace mipt {

void rotation(char axis, float angle, float o[9]) {   // src/LinearAlgebra.cpp:90-118
    const float c = std::cos(angle), s = std::sin(angle);
    const float rx[9] = {1.0f, 0.0f, 0.0f, 0.0f, c, -s, 0.0f, s, c};
    const float ry[9] = {c, 0.0f, s, 0.0f, 1.0f, 0.0f, -s, 0.0f, c};
    const float rz[9] = {c, -s, 0.0f, s, c, 0.0f, 0.0f, 0.0f, 1.0f};
    std::memcpy(o, axis == 'x' ? rx : (axis == 'y' ? ry : rz), sizeof(rx));
}

// Linear * Vec is a sum of scaled columns: ((m0*x + m1*y) + m2*z) per row (:380-382).
static inline void lin_vec(const float* m, const float* v, float* o) {
    const float x = v[0], y = v[1], z = v[2];
    o[0] = (m[0] * x + m[1] * y) + m[2] * z;
    o[1] = (m[3] * x + m[4] * y) + m[5] * z;
    o[2] = (m[6] * x + m[7] * y) + m[8] * z;
}

void linear_mul(const float a[9], const float b[9], float o[9]) {   // row(i) . column(j), :389-395
    float r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    std::memcpy(o, r, sizeof(r));
}

void affine_mul(const float a[12], const float b[12], float o[12]) {   // :413-415
    float r[12], t[3];
    linear_mul(a, b, r);
    lin_vec(a, b + 9, t);
    r[9] = t[0] + a[9];
    r[10] = t[1] + a[10];
    r[11] = t[2] + a[11];
    std::memcpy(o, r, sizeof(r));
}

bool affine_inverse(const float f[12], float o[12]) {   // Linear::inverse :207-233, Affine::inverse :258-264
    const float a = f[0], b = f[1], c = f[2], d = f[3], e = f[4], ff = f[5], g = f[6], h = f[7], i = f[8];
    const float det = (a * e * i) + (b * ff * g) + (c * d * h) - (c * e * g) - (b * d * i) - (a * ff * h);
    if (det < 1e-6f) return false;
    const float k = 1.0f / det;
    const float adj[9] = {e * i - ff * h, c * h - b * i, b * ff - c * e, ff * g - d * i, a * i - c * g,
                          c * d - a * ff, d * h - e * g, b * g - a * h, a * e - b * d};
    float li[12];
    for (int n = 0; n < 9; ++n) li[n] = k * adj[n];
    li[9] = li[10] = li[11] = 0.0f;
    // linv * Translation(-t): the product with the identity matrix is carried out as written
    const float tr[12] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, -f[9], -f[10], -f[11]};
    affine_mul(li, tr, o);
    return true;
}

void affine_point(const float a[12], const float p[3], float o[3]) {   // :401-403
    float t[3];
    lin_vec(a, p, t);
    o[0] = t[0] + a[9];
    o[1] = t[1] + a[10];
    o[2] = t[2] + a[11];
}

bool world_bounds(uint32_t kind, const float fwd[12], const float p[9], float lo[3], float hi[3]) {
    float c[3] = {0.0f, 0.0f, 0.0f}, h[3];
    switch (kind) {   // *::getLocalBoundingBox src/Object.cpp:96-158, PathTracerDemo.cpp:474-479
        case 0: h[0] = h[1] = h[2] = p[0] + 1e-3f; break;
        case 1: h[0] = p[0] + 1e-3f; h[1] = p[1] + 1e-3f; h[2] = p[2] + 1e-3f; break;
        case 2: {
            for (int k = 0; k < 3; ++k) {
                const float mn = std::min({p[k], p[3 + k], p[6 + k]}) - 1e-3f;
                const float mx = std::max({p[k], p[3 + k], p[6 + k]}) + 1e-3f;
                c[k] = mn + 0.5f * (mx - mn);          // AxisAlignedBox(Pos, Pos) src/Geometry.cpp:348-358
                h[k] = 0.5f * std::abs(mn - mx);
            }
            break;
        }
        case 3: h[0] = h[1] = h[2] = 1.0f; break;
        case 4: h[0] = h[1] = h[2] = 2.0f; break;   // FractalObject::getLocalBoundingBox src/Object.cpp:191-196
        case 5: h[0] = 2.0f * p[0]; h[1] = 2.0f * p[1]; h[2] = 2.0f * p[2]; break;   // RoughBoxObject::localBoundingBox PathTracerDemo.cpp:261-266
        case 6: h[0] = h[1] = h[2] = 1.2f * p[0]; break;                              // RoughSphereObject::localBoundingBox :216-221
        case 7: h[0] = 50.0f; h[1] = 20.0f; h[2] = 20.0f; break;                      // WeirdWallObject::localBoundingBox :343-348
        case 8:   // an object evaluated on the host brings its world box along (Object::getBoundingBox: centre, half size)
            for (int k = 0; k < 3; ++k) {
                lo[k] = p[k] - p[3 + k];
                hi[k] = p[k] + p[3 + k];
            }
            return true;
        default: return false;
    }
    for (int k = 0; k < 3; ++k) {
        lo[k] = std::numeric_limits<float>::max();
        hi[k] = std::numeric_limits<float>::lowest();
    }
    static const float sg[8][3] = {{1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}, {-1, 1, 1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
    for (const auto& s : sg) {
        const float corner[3] = {c[0] + s[0] * h[0], c[1] + s[1] * h[1], c[2] + s[2] * h[2]};
        float u[3];
        affine_point(fwd, corner, u);
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], u[k]);
            hi[k] = std::max(hi[k], u[k]);
        }
    }
    return true;
}

}   // namespace mipt
