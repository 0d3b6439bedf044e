// Host-side affine helpers that follow the reference's operation order (src/LinearAlgebra.cpp), so
// that transforms built here carry the same bits as ones built through the reference's classes.
// Layout everywhere: 9 floats Linear row-major (m[3*r+c]) followed by the translation.
#pragma once
#include <stdint.h>

namespace mipt {

void rotation(char axis, float angle, float out9[9]);
void linear_mul(const float a[9], const float b[9], float out[9]);
void affine_mul(const float a[12], const float b[12], float out[12]);
bool affine_inverse(const float fwd[12], float inv[12]);   // false when det < 1e-6
void affine_point(const float a[12], const float p[3], float out[3]);
// Object::getBoundingBox (src/Object.cpp:46-80): padded local box -> 8 corners -> world min / max.
bool world_bounds(uint32_t kind, const float fwd[12], const float params[9], float lo[3], float hi[3]);

}   // namespace mipt
