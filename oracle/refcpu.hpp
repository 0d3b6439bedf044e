// TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the timstr/pathtracer hot path.
//
// This header declares the C interface of liboracle.so.  Nothing under
// pathtracer_amd/ or include/ may include, link or call it: it exists so that
// tests, the smoke check and the benchmark's cpu_baseline leg have an
// independent statement of what the reference computes.  Every function cites
// the reference file:line (relative to the reference checkout) it follows.
//
// Pinning: the reference ships no tests or golden vectors.  The restatement is
// pinned by (a) bit-exact comparison against the reference's own sources
// compiled into oracle/_ref/libref.so (oracle/Makefile, tests/test_oracle_vs_ref.py,
// runs where the reference checkout is present) and (b) the fixtures under
// tests/golden/ that were captured from that compiled reference.
//
// Flat scene format (shared with the device library's C ABI, restated here so
// that the oracle does not depend on product headers):
//   kind[i]      0 sphere, 1 box, 2 triangle, 3 crater-ball SDF
//   fwd[12*i..]  Object::m_transformation: Linear row-major (m[3*r+c]) then translation
//   inv[12*i..]  Object::m_inverseTransformation in the same layout
//   params[9*i..] sphere {radius}; box {half x,y,z}; triangle {a,b,c}; crater ball {offset x,y,z}
//   material[i]  row of materials
//   materials[14*m..] diffuse, specular, sharpness, albedo rgb, emitted rgb,
//                transmittance, index of refraction, internal absorption rgb
//                (member order of BasicMaterial, include/BasicMaterial.hpp:32-39)
//   noise12      the demo's noiseTransform() (PathTracerDemo.cpp:37-54), same affine layout
//   cam[16]      Camera::m_transform (12), aspect, field of view (degrees), focal distance, blur radius
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

enum { ORC_SPHERE = 0, ORC_BOX = 1, ORC_TRIANGLE = 2, ORC_CRATERBALL = 3, ORC_FRACTAL = 4, ORC_ROUGHBOX = 5, ORC_ROUGHSPHERE = 6, ORC_WEIRDWALL = 7 };
enum { ORC_RNG_MINSTD = 0, ORC_RNG_COUNTER = 1 };

// ---- linear algebra helpers (src/LinearAlgebra.cpp) --------------------------------------
void orc_rotation_x(float angle, float out9[9]);                      // :90-98
void orc_rotation_y(float angle, float out9[9]);                      // :100-108
void orc_rotation_z(float angle, float out9[9]);                      // :110-118
void orc_linear_mul(const float a9[9], const float b9[9], float out9[9]);   // :389-395
void orc_affine_mul(const float a12[12], const float b12[12], float out12[12]); // :413-415
int  orc_affine_inverse(const float fwd12[12], float inv12[12]);      // :207-264; 0 when det < 1e-6
void orc_noise_transform(const float t3[3], const float r3[3], float out12[12]); // PathTracerDemo.cpp:48-51

// ---- scene -----------------------------------------------------------------------------
orc_scene* orc_scene_create(uint32_t n_prims, const uint32_t* kind, const float* fwd, const float* inv,
                            const float* params, const uint32_t* material, uint32_t n_materials,
                            const float* materials, const float* noise12);
void orc_scene_free(orc_scene*);
// Leaf order of the reference's own tree builder (src/ObjectTree.cpp:90-184).  After this call
// closest-hit ties resolve as in the reference (first leaf in tree order); before it they resolve to
// the lowest primitive index.  Super-quadratic: meant for a few hundred primitives.  Returns 0 on
// success, -1 when the reference builder would assert (no split found).
int  orc_scene_use_reference_tree_order(orc_scene*, uint32_t* out_order /*nullable, n_prims*/);

// ---- per-function entry points ------------------------------------------------------------
// Object::getBoundingBox (src/Object.cpp:46-80): out6 = center xyz, half xyz.
void orc_world_bbox(uint32_t kind, const float fwd12[12], const float params9[9], float out6[6]);
// *Object::hitLocalRay (src/Object.cpp:88-90,116-118,141-143; include/Object.hpp:135-167)
void orc_hit_local(uint32_t kind, const float params9[9], const float noise12[12], uint32_t n,
                   const float* ray_pos3, const float* ray_dir3, uint8_t* out_hit, float* out_pos3);
// Surface normal as *Object::deflectLocalRay passes it to the material
// (src/Object.cpp:92-94,120-122,145-147; include/Object.hpp:118-132)
void orc_normal_local(uint32_t kind, const float params9[9], const float noise12[12], uint32_t n,
                      const float* pos3, float* out_n3);
// CraterBallObject::signedDistance (PathTracerDemo.cpp:361-466) and noise (:56-70)
void orc_sdf_craterball(const float offset3[3], const float noise12[12], uint32_t n, const float* pos3, float* out_d);
void orc_noise(const float noise12[12], uint32_t n, const float* pos3, float* out_v);
// Camera::getViewRay (src/Camera.cpp:65-81).  rng_mode MINSTD: one engine seeded `seed`, rays in order.
// rng_mode COUNTER: ray i draws its lens point from stream (seed, path=i, sample 0, stage 0, k>=2) when
// the blur radius is non-zero and draws nothing otherwise.
void orc_camera_rays(const float cam16[16], uint32_t n, const float* sx, const float* sy,
                     uint32_t rng_mode, uint64_t seed, float* out_dir3, float* out_pos3);
// ObjectTree::Tree::hit (src/ObjectTree.cpp:22-88): out_prim = -1 on a miss;
// out_t = (hit - ray.pos) . ray.dir as InternalNode::hit computes it.
void orc_closest_hit(const orc_scene*, uint32_t n, const float* ray_pos3, const float* ray_dir3,
                     int32_t* out_prim, float* out_t, float* out_pos3);
// BasicMaterial::deflect (src/BasicMaterial.cpp:97-159); out12 = emitted, attenuation, direction, normal.
// MINSTD: one engine seeded `seed`, calls in order.  COUNTER: call i uses stream (seed, i, 0, stage 1).
void orc_deflect(const float mat14[14], uint32_t n, const float* inbound3, const float* normal3,
                 uint32_t rng_mode, uint64_t seed, float* out12);
// Scene::trace (src/Scene.cpp:18-78) for explicit rays.  MINSTD: one engine, rays in order.
// COUNTER: ray i is path (pixel=i, sample=0); bounce b draws from stage 1+b.
void orc_trace(const orc_scene*, uint32_t n, const float* ray_pos3, const float* ray_dir3, uint32_t depth,
               uint32_t rng_mode, uint64_t seed, float* out_rgb);
// Pixel-centre primary rays (sx = x/(W-1), sy = y/(H-1), no jitter; blur radius must be 0):
// Renderer::doWork's px/py (src/Renderer.cpp:72-89) -> getViewRay -> Tree::hit.
void orc_primary_hits(const orc_scene*, const float cam16[16], uint32_t W, uint32_t H,
                      int32_t* out_prim, float* out_t);
// Renderer::render (src/Renderer.cpp:54-172).  full_width = 0 reproduces the 8-pixel task split
// (columns >= 8*floor(W/8) stay black); 1 renders every column.  MINSTD: worker k owns an engine seeded
// seed+k and samples are drawn in the reference's order (bit-identical to the reference with one
// thread).  COUNTER: path (pixel, first_sample+i) uses the counter generator, so the image does not
// depend on the thread count.  stats4 (nullable) = paths, rays, sdf evaluations, near-field sdf evaluations.
int  orc_render(const orc_scene*, const float cam16[16], uint32_t W, uint32_t H, uint32_t bounces, uint32_t spp,
                uint32_t rng_mode, uint64_t seed, uint32_t first_sample, uint32_t threads, uint32_t full_width,
                float* out_rgb, uint64_t* stats4);
// ReinhardToneMapper / FilmicToneMapper (src/ToneMapper.cpp:6-72); kind 0 Reinhard, 1 filmic.
void orc_tonemap(uint32_t kind, uint32_t W, uint32_t H, const float* in_rgb, float* out_rgb);
// Image::save byte stream (src/Image.cpp:48-60): u64 W, u64 H, row-major f32 rgb.  out has 16 + 12*W*H bytes.
void orc_image_bytes(uint32_t W, uint32_t H, const float* rgb, uint8_t* out);
// The counter generator itself: uniform in [0,1) with 24 random bits.
float orc_counter_uniform01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stage, uint32_t k);

#ifdef __cplusplus
}
#endif
