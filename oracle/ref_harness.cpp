// TEST INFRASTRUCTURE -- C wrapper around the reference's own classes.
//
// Built by oracle/Makefile into oracle/_ref/libref.so from the reference sources where they lie
// (the reference checkout is never copied into this repository).  It exposes the same calls as
// liboracle.so with a `ref_` prefix so that tests/test_oracle_vs_ref.py can compare the restatement
// with the real implementation bit for bit and tests/golden/make_golden.py can capture fixtures.
//
// Two things are substituted at link time, neither changes arithmetic:
//   * randomEngine() (src/RandomNumberGenerator.cpp is left out): thread-local engines seeded from a
//     counter the harness controls instead of std::random_device, so runs are reproducible;
//   * DEMO_SDF_INC is a file holding PathTracerDemo.cpp lines 26-512 (procedural helpers and the SDF
//     object classes, no SFML dependency), produced by the Makefile in a temporary directory.
#include <Camera.hpp>
#include <Geometry.hpp>
#include <Image.hpp>
#include <LinearAlgebra.hpp>
#include <Object.hpp>
#include <ObjectTree.hpp>
#include <RandomNumberGenerator.hpp>
#include <Renderer.hpp>
#include <RenderSettings.hpp>
#include <Scene.hpp>
#include <ToneMapper.hpp>

#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <thread>
#include <unistd.h>
#include <vector>

static std::atomic<uint64_t> g_next_seed{1};

std::default_random_engine& randomEngine() noexcept {
    static thread_local auto re = std::default_random_engine{
        static_cast<std::default_random_engine::result_type>(g_next_seed.fetch_add(1))};
    return re;
}

#ifdef DEMO_SDF_INC
#include DEMO_SDF_INC
#else
// Built against this repository's own headers without the demo's source block: the crater ball is the
// library's CraterBall type and its noise transform is set by ref_set_noise().
static Affine g_noise;
class CraterBallObject : public CraterBall {
public:
    CraterBallObject(BasicMaterial m, Vec offset) : CraterBall(m, offset, g_noise) {}
};
#endif

namespace {

Affine load_aff(const float* p) {
    return Affine(Linear(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]), Vec(p[9], p[10], p[11]));
}
void store_aff(const Affine& a, float* p) {
    for (int i = 0; i < 9; ++i) p[i] = a.linear[i];
    p[9] = a.translation.x;
    p[10] = a.translation.y;
    p[11] = a.translation.z;
}
BasicMaterial load_material(const float* p) {
    auto m = BasicMaterial{};
    m.setDiffuseReflection(p[0]);
    m.setSpecularReflection(p[1]);
    m.setSpecularSharpness(p[2]);
    m.setReflectedAbsorption(Color{p[3], p[4], p[5]});
    m.setEmittedLuminance(Color{p[6], p[7], p[8]});
    m.setTransmittance(p[9]);
    m.setIndexOfRefraction(p[10]);
    m.setInternalAbsorption(Color{p[11], p[12], p[13]});
    return m;
}
std::unique_ptr<Object> make_object(uint32_t kind, const float* p, const BasicMaterial& m) {
    switch (kind) {
        case 0: return std::make_unique<SphereObject>(Sphere(p[0]), m);
        case 1: return std::make_unique<BoxObject>(Rectangle(Vec(p[0], p[1], p[2])), m);
        case 2: return std::make_unique<TriangleObject>(Triangle(Pos(p[0], p[1], p[2]), Pos(p[3], p[4], p[5]), Pos(p[6], p[7], p[8])), m);
        case 3: return std::make_unique<CraterBallObject>(m, Vec(p[0], p[1], p[2]));
        case 4: {
            auto o = std::make_unique<FractalObject>();
            o->material = m;
            return o;
        }
#ifdef DEMO_SDF_INC
        case 5: return std::make_unique<RoughBoxObject>(Rectangle(Vec(p[0], p[1], p[2])), m);
        case 6: return std::make_unique<RoughSphereObject>(Sphere(p[0]), m, m);
        case 7: return std::make_unique<WeirdWallObject>(m);
#endif
    }
    return nullptr;
}
Camera load_cam(const float* c) {
    auto cam = Camera(load_aff(c), c[12], c[13]);
    cam.setFocalDistance(c[14]);
    cam.setFocalBlurRadius(c[15]);
    return cam;
}

struct RefScene {
    Scene scene;
    std::vector<const Object*> objects;   // insertion order
    std::unique_ptr<ObjectTree::Tree> tree;
};

}   // namespace

extern "C" {

#ifdef DEMO_SDF_INC
// Initialises the demo's process-wide noise transform (PathTracerDemo.cpp:37-54) from an engine
// seeded `seed` on the calling thread and returns it.  Only the first call in a process draws.
void ref_init_noise(uint64_t seed, float out12[12]) {
    randomEngine().seed(static_cast<std::default_random_engine::result_type>(seed));
    store_aff(noiseTransform(), out12);
}
void ref_noise(uint32_t n, const float* pos, float* out_v) {
    for (uint32_t i = 0; i < n; ++i) out_v[i] = noise(Pos(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
}
#else
void ref_set_noise(const float in12[12]) { g_noise = load_aff(in12); }
#endif

void ref_rotation_x(float a, float out[9]) { auto l = Linear::RotationX(a); for (int i = 0; i < 9; ++i) out[i] = l[i]; }
void ref_rotation_y(float a, float out[9]) { auto l = Linear::RotationY(a); for (int i = 0; i < 9; ++i) out[i] = l[i]; }
void ref_rotation_z(float a, float out[9]) { auto l = Linear::RotationZ(a); for (int i = 0; i < 9; ++i) out[i] = l[i]; }
void ref_affine_mul(const float a[12], const float b[12], float out[12]) { store_aff(load_aff(a) * load_aff(b), out); }
int ref_affine_inverse(const float fwd[12], float inv[12]) {
    auto r = load_aff(fwd).inverse();
    if (!r) return 0;
    store_aff(*r, inv);
    return 1;
}
void ref_noise_transform(const float t[3], const float r[3], float out[12]) {
    store_aff(Affine::Translation(Vec(t[0], t[1], t[2])) * Linear::RotationX(r[0]) * Linear::RotationY(r[1]) * Linear::RotationZ(r[2]), out);
}

void* ref_scene_create(uint32_t n, const uint32_t* kind, const float* fwd, const float* /*inv: recomputed by setTransformation*/,
                       const float* params, const uint32_t* material, uint32_t /*nm*/, const float* materials) {
    auto* s = new RefScene;
    for (uint32_t i = 0; i < n; ++i) {
        auto o = make_object(kind[i], params + 9 * i, load_material(materials + 14 * material[i]));
        o->setTransformation(load_aff(fwd + 12 * i));
        s->objects.push_back(o.get());
        s->scene.addObject(std::move(o));
    }
    return s;
}
// Runs the reference's tree builder (super-quadratic).  Needed by closest_hit / trace / render.
void ref_scene_build_tree(void* h) {
    auto* s = static_cast<RefScene*>(h);
    s->scene.updateGeometry();
    s->tree = std::make_unique<ObjectTree::Tree>(s->objects);
}
void ref_scene_free(void* h) { delete static_cast<RefScene*>(h); }

void ref_scene_inverse(void* h, uint32_t i, float out12[12]) {
    store_aff(static_cast<RefScene*>(h)->objects[i]->inverseTransformation(), out12);
}

void ref_world_bbox(uint32_t kind, const float fwd[12], const float params[9], float out6[6]) {
    auto o = make_object(kind, params, BasicMaterial{});
    o->setTransformation(load_aff(fwd));
    const auto b = o->getBoundingBox();
    out6[0] = b.center.x; out6[1] = b.center.y; out6[2] = b.center.z;
    out6[3] = b.halfSize.x; out6[4] = b.halfSize.y; out6[5] = b.halfSize.z;
}

// Local-frame hit through Object::hitRay with the default (identity) transformation.
void ref_hit_local(uint32_t kind, const float params[9], uint32_t n, const float* pos, const float* dir,
                   uint8_t* out_hit, float* out_pos) {
    auto o = make_object(kind, params, BasicMaterial{});
    for (uint32_t i = 0; i < n; ++i) {
        const auto r = Ray(Vec(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), Pos(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        const auto h = o->hitRay(r);
        out_hit[i] = h ? 1 : 0;
        out_pos[3 * i] = h ? h->x : 0.0f;
        out_pos[3 * i + 1] = h ? h->y : 0.0f;
        out_pos[3 * i + 2] = h ? h->z : 0.0f;
    }
}

// Local-frame normal: ColorBounce::normal from Object::deflectRay with the identity transformation and
// an inbound direction that points against nothing in particular (the normal does not depend on it).
void ref_normal_local(uint32_t kind, const float params[9], uint32_t n, const float* pos, float* out_n) {
    auto o = make_object(kind, params, BasicMaterial{});
    for (uint32_t i = 0; i < n; ++i) {
        const auto r = Ray(Vec(0.0f, 0.0f, 1.0f), Pos(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        const auto b = o->deflectRay(r);
        out_n[3 * i] = b.normal.x;
        out_n[3 * i + 1] = b.normal.y;
        out_n[3 * i + 2] = b.normal.z;
    }
}

void ref_sdf_craterball(const float off[3], uint32_t n, const float* pos, float* out_d) {
    const auto o = CraterBallObject(BasicMaterial{}, Vec(off[0], off[1], off[2]));
    for (uint32_t i = 0; i < n; ++i) out_d[i] = o.signedDistance(Pos(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
}

// signedDistance of the demo program's SDF classes (kinds 5-7: PathTracerDemo.cpp:237-246, 173-182, 281-328) at local points
int ref_sdf_value(uint32_t kind, const float params[9], uint32_t n, const float* pos, float* out_d) {
#ifdef DEMO_SDF_INC
    const BasicMaterial m{};
    auto eval = [&](const auto& o) {
        for (uint32_t i = 0; i < n; ++i) out_d[i] = o.signedDistance(Pos(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
    };
    if (kind == 5) { eval(RoughBoxObject(Rectangle(Vec(params[0], params[1], params[2])), m)); return 1; }
    if (kind == 6) { eval(RoughSphereObject(Sphere(params[0]), m, m)); return 1; }
    if (kind == 7) { eval(WeirdWallObject(m)); return 1; }
#endif
    (void)kind; (void)params; (void)n; (void)pos; (void)out_d;
    return 0;
}

void ref_camera_rays(const float cam16[16], uint32_t n, const float* sx, const float* sy, uint64_t seed,
                     float* out_dir, float* out_pos) {
    const auto cam = load_cam(cam16);
    randomEngine().seed(static_cast<std::default_random_engine::result_type>(seed));
    for (uint32_t i = 0; i < n; ++i) {
        const auto r = cam.getViewRay(sx[i], sy[i]);
        out_dir[3 * i] = r.dir.x; out_dir[3 * i + 1] = r.dir.y; out_dir[3 * i + 2] = r.dir.z;
        out_pos[3 * i] = r.pos.x; out_pos[3 * i + 1] = r.pos.y; out_pos[3 * i + 2] = r.pos.z;
    }
}

void ref_closest_hit(void* h, uint32_t n, const float* pos, const float* dir, int32_t* out_prim, float* out_t,
                     float* out_pos) {
    auto* s = static_cast<RefScene*>(h);
    for (uint32_t i = 0; i < n; ++i) {
        const auto r = Ray(Vec(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), Pos(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        const auto hit = s->tree->hit(r);
        if (!hit) {
            out_prim[i] = -1;
            out_t[i] = 0.0f;
            if (out_pos) out_pos[3 * i] = out_pos[3 * i + 1] = out_pos[3 * i + 2] = 0.0f;
            continue;
        }
        int32_t idx = -1;
        for (size_t k = 0; k < s->objects.size(); ++k)
            if (s->objects[k] == hit->second) idx = static_cast<int32_t>(k);
        out_prim[i] = idx;
        out_t[i] = (hit->first - r.pos) * r.dir;
        if (out_pos) { out_pos[3 * i] = hit->first.x; out_pos[3 * i + 1] = hit->first.y; out_pos[3 * i + 2] = hit->first.z; }
    }
}

void ref_deflect(const float mat14[14], uint32_t n, const float* inbound, const float* normal, uint64_t seed,
                 float* out12) {
    const auto m = load_material(mat14);
    randomEngine().seed(static_cast<std::default_random_engine::result_type>(seed));
    for (uint32_t i = 0; i < n; ++i) {
        const auto b = m.deflect(Vec(inbound[3 * i], inbound[3 * i + 1], inbound[3 * i + 2]),
                                 Vec(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]));
        float* o = out12 + 12 * i;
        o[0] = b.emitted.r; o[1] = b.emitted.g; o[2] = b.emitted.b;
        o[3] = b.attenuation.r; o[4] = b.attenuation.g; o[5] = b.attenuation.b;
        o[6] = b.rayDirection.x; o[7] = b.rayDirection.y; o[8] = b.rayDirection.z;
        o[9] = b.normal.x; o[10] = b.normal.y; o[11] = b.normal.z;
    }
}

void ref_trace(void* h, uint32_t n, const float* pos, const float* dir, uint32_t depth, uint64_t seed, float* out_rgb) {
    auto* s = static_cast<RefScene*>(h);
    randomEngine().seed(static_cast<std::default_random_engine::result_type>(seed));
    for (uint32_t i = 0; i < n; ++i) {
        const auto r = Ray(Vec(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), Pos(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        const auto c = s->scene.trace(r, depth);
        out_rgb[3 * i] = c.r; out_rgb[3 * i + 1] = c.g; out_rgb[3 * i + 2] = c.b;
    }
}

void ref_primary_hits(void* h, const float cam16[16], uint32_t W, uint32_t H, int32_t* out_prim, float* out_t) {
    auto* s = static_cast<RefScene*>(h);
    const auto cam = load_cam(cam16);
    const auto maxW = static_cast<float>(W - 1), maxH = static_cast<float>(H - 1);
    for (uint32_t y = 0; y < H; ++y) {
        const auto py = static_cast<float>(y) / maxH;
        for (uint32_t x = 0; x < W; ++x) {
            const auto px = static_cast<float>(x) / maxW;
            const auto r = cam.getViewRay(px, py);
            const auto hit = s->tree->hit(r);
            int32_t idx = -1;
            float t = 0.0f;
            if (hit) {
                for (size_t k = 0; k < s->objects.size(); ++k)
                    if (s->objects[k] == hit->second) idx = static_cast<int32_t>(k);
                t = (hit->first - r.pos) * r.dir;
            }
            out_prim[y * W + x] = idx;
            out_t[y * W + x] = t;
        }
    }
}

// Renderer::render with `threads` workers; worker engines take seeds seed, seed+1, ... in the order
// the workers first draw (deterministic for one thread only).
void ref_render(void* h, const float cam16[16], uint32_t W, uint32_t H, uint32_t bounces, uint32_t spp, uint64_t seed,
                uint32_t threads, float* out_rgb) {
    auto* s = static_cast<RefScene*>(h);
    const auto cam = load_cam(cam16);
    auto settings = RenderSettings(W, H);
    settings.setNumBounces(bounces);
    settings.setSamplesPerPixel(spp);
    g_next_seed.store(seed);
    {
        auto r = Renderer{};
        r.startThreadPool(threads);
        const auto img = r.render(s->scene, cam, settings);
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                const auto& c = img(x, y);
                float* o = out_rgb + 3 * (static_cast<size_t>(y) * W + x);
                o[0] = c.r; o[1] = c.g; o[2] = c.b;
            }
        // The reference's teardown can miss a barrier phase if the pool is stopped while a worker is
        // still between its two barrier waits (src/Renderer.cpp:109-113 vs :45-46); give it time.
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
    }
}

void ref_tonemap(uint32_t kind, uint32_t W, uint32_t H, const float* in, float* out) {
    auto img = Image(W, H);
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const float* p = in + 3 * (static_cast<size_t>(y) * W + x);
            img(x, y) = Color{p[0], p[1], p[2]};
        }
    const auto res = kind == 1 ? FilmicToneMapper{}(img) : ReinhardToneMapper{}(img);
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const auto& c = res(x, y);
            float* o = out + 3 * (static_cast<size_t>(y) * W + x);
            o[0] = c.r; o[1] = c.g; o[2] = c.b;
        }
}

// Image::save byte stream via a temporary file.
int ref_image_bytes(uint32_t W, uint32_t H, const float* rgb, uint8_t* out) {
    auto img = Image(W, H);
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const float* p = rgb + 3 * (static_cast<size_t>(y) * W + x);
            img(x, y) = Color{p[0], p[1], p[2]};
        }
    char path[] = "/tmp/ref_image_XXXXXX";
    const int fd = mkstemp(path);
    if (fd < 0) return -1;
    close(fd);
    img.save(path);
    auto f = std::ifstream(path, std::ios::binary);
    f.read(reinterpret_cast<char*>(out), 16 + static_cast<std::streamsize>(12) * W * H);
    const bool ok = static_cast<bool>(f);
    f.close();
    unlink(path);
    return ok ? 0 : -1;
}

}   // extern "C"
