// The demo's default scene and camera, rendered headlessly through the C++ scene / render API:
// a crater ball lit by a large emissive slab, viewed through a 10-degree lens from 30 units away.
//
//   demo_headless [width height spp bounces seed out.img [passes [resume_after [user]]]]
//
// resume_after > 0: after that many passes the accumulator is saved (ProgressiveFrame::save, `out`.ckpt), everything
// is torn down, and a new Renderer and ProgressiveFrame load the checkpoint and render the remaining passes.
//
// With a ninth argument "user" the light is not a BoxObject but an Object subclass defined in this file (UserSlab): the
// library has no device form for it and evaluates it through its virtuals on the host (the slow path).
//
// Writes the frame with Image::save (u64 width, u64 height, row-major f32 rgb) and prints timing.
// Renderer::render runs on the GPU; nothing here has a CPU fallback.
#include <Camera.hpp>
#include <Geometry.hpp>
#include <Image.hpp>
#include <Object.hpp>
#include <Renderer.hpp>
#include <RenderSettings.hpp>
#include <Scene.hpp>
#include <ToneMapper.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>
#include <string>

// What any user of the reference may write: an Object subclass with its own three virtuals (include/Object.hpp:45-52).
// This one is a box like BoxObject, so that a frame lit by it can be compared with the ordinary one.
class UserSlab : public Object {
public:
    UserSlab(Rectangle geometry, BasicMaterial material) noexcept : m_geometry(geometry), m_material(material) {}

private:
    std::optional<Pos> hitLocalRay(const Ray& ray) const noexcept override { return intersect(ray, m_geometry); }
    ColorBounce deflectLocalRay(const Ray& ray) const noexcept override { return m_material.deflect(ray.dir, m_geometry.normal(ray.pos)); }
    AxisAlignedBox getLocalBoundingBox() const noexcept override { return AxisAlignedBox{Pos(0.0f, 0.0f, 0.0f), m_geometry.halfSize + Vec(1e-3f, 1e-3f, 1e-3f)}; }
    Rectangle m_geometry;
    BasicMaterial m_material;
};

int main(int argc, char** argv) {
    const std::size_t width = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 256;
    const std::size_t height = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 256;
    const std::size_t spp = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 16;
    const std::size_t bounces = argc > 4 ? std::strtoul(argv[4], nullptr, 10) : 8;
    const unsigned long long seed = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 1;
    const std::string out = argc > 6 ? argv[6] : "demo_headless.img";
    // more than one pass: the demo's progressive loop with the accumulator on the GPU (ProgressiveFrame); writes the
    // mean image to `out` and the displayed bytes to `out`.rgba
    const std::size_t passes = argc > 7 ? std::strtoul(argv[7], nullptr, 10) : 1;
    const std::size_t resume_after = argc > 8 ? std::strtoul(argv[8], nullptr, 10) : 0;
    const bool user_light = argc > 9 && std::string(argv[9]) == "user";
    // "cancel:<ms>": one ProgressiveFrame pass that another thread cancels after <ms> milliseconds (Renderer::cancelRender
    // as the demo's window thread calls it, PathTracerDemo.cpp:982); writes the mean of what had been accumulated
    const bool cancel_mode = argc > 9 && std::string(argv[9]).rfind("cancel:", 0) == 0;

    // The demo draws its noise transform at random per process; fix it here.
    const Affine noise = Affine::Translation(Vec(0.417022f, 0.720324f, 0.000114f)) * Linear::RotationX(0.302333f) *
                         Linear::RotationY(0.146756f) * Linear::RotationZ(0.092339f);

    auto scene = Scene{};
    {
        auto m = BasicMaterial{};
        m.setDiffuseReflection(1.0f);
        m.setSpecularReflection(0.0f);
        m.setTransmittance(0.0f);
        m.setReflectedAbsorption(Color{0.5f, 0.5f, 0.5f});
        m.setEmittedLuminance(Color{0.0f, 0.0f, 0.0f});
        auto& ball = scene.addObject<CraterBall>(m, Vec(-8.0f, 15.0f, -6.5f), noise);
        ball.setTransformation(Affine::Translation(Vec{0.0f, 0.5f, -2.0f}));
    }
    {
        auto m = BasicMaterial{};
        m.setDiffuseReflection(1.0f);
        m.setSpecularReflection(0.0f);
        m.setTransmittance(0.0f);
        m.setReflectedAbsorption(Color{1.0f, 1.0f, 1.0f});
        m.setEmittedLuminance(2.0f * Color{1.0f, 1.0f, 1.0f});
        if (user_light) {
            auto& slab = scene.addObject<UserSlab>(Rectangle(Vec(100.0f, 0.1f, 100.0f)), m);
            slab.setTransformation(Affine::Translation(Vec{0.0f, -50.0f, 0.0f}));
        } else {
            auto& slab = scene.addObject<BoxObject>(Rectangle(Vec(100.0f, 0.1f, 100.0f)), m);
            slab.setTransformation(Affine::Translation(Vec{0.0f, -50.0f, 0.0f}));
        }
    }
    scene.updateGeometry();

    auto camera = Camera(Affine{});
    camera.setAspectRatio(static_cast<float>(width) / static_cast<float>(height));
    camera.setFieldOfView(10.0f);
    camera.setFocalDistance(30.0f);
    camera.setFocalBlurRadius(0.0f);
    camera.setTransform(Affine::Translation(0.0f, 0.0f, -30.0f) * Linear::Scale(0.01f));

    auto settings = RenderSettings(width, height);
    settings.setNumBounces(bounces);
    settings.setSamplesPerPixel(spp);

    if (passes > 1 && resume_after > 0 && resume_after < passes) {
        const std::string ckpt = out + ".ckpt";
        {
            auto first = Renderer{};
            first.startThreadPool();
            first.setSeed(seed);
            ProgressiveFrame acc(width, height);
            for (std::size_t i = 0; i < resume_after; ++i) acc.add(first, scene, camera, settings);
            acc.save(ckpt, first);
        }
        auto second = Renderer{};   // a fresh renderer: its seed and frame counter come from the checkpoint
        second.startThreadPool();
        ProgressiveFrame acc(width, height);
        if (!acc.load(ckpt, second)) {
            std::fprintf(stderr, "checkpoint %s did not load\n", ckpt.c_str());
            return 1;
        }
        for (std::size_t i = resume_after; i < passes; ++i) acc.add(second, scene, camera, settings);
        acc.image().save(out);
        std::printf("%zu passes of %zu spp, saved after %zu and resumed from %s (%zu passes counted), wrote %s\n", passes, spp, resume_after, ckpt.c_str(),
                    acc.count(), out.c_str());
        return 0;
    }
    auto renderer = Renderer{};
    renderer.startThreadPool();
    renderer.setSeed(seed);
    if (cancel_mode) {
        const long ms = std::strtol(argv[9] + 7, nullptr, 10);
        ProgressiveFrame acc(width, height);
        settings.setSamplesPerPixel(1);
        acc.add(renderer, scene, camera, settings);   // a first pass, so that allocations and the scene upload are behind us
        acc.reset();
        renderer.setSeed(seed);
        settings.setSamplesPerPixel(spp);
        std::thread canceller([&] {
            std::this_thread::sleep_for(std::chrono::milliseconds(ms));
            renderer.cancelRender();
        });
        const bool finished = acc.add(renderer, scene, camera, settings);
        canceller.join();
        acc.image().save(out);
        std::printf("pass %s: %llu samples per pixel counted, %zu passes, wrote %s\n", finished ? "finished" : "cancelled",
                    static_cast<unsigned long long>(acc.samples()), acc.count(), out.c_str());
        return 0;
    }
    if (passes > 1) {
        ProgressiveFrame acc(width, height);
        std::vector<std::uint8_t> rgba(width * height * 4);
        const auto p0 = std::chrono::steady_clock::now();
        for (std::size_t i = 0; i < passes; ++i) {
            acc.add(renderer, scene, camera, settings);
            acc.present(rgba.data());   // what the window would show after this pass
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count();
        acc.image().save(out);
        std::ofstream f(out + ".rgba", std::ios::out | std::ios::binary);
        f.write(reinterpret_cast<const char*>(rgba.data()), static_cast<std::streamsize>(rgba.size()));
        std::printf("%zu passes of %zu spp at %zux%zu in %.3f s (%.1f ms per pass incl. display bytes), wrote %s and %s.rgba\n", passes, spp, width, height, s,
                    1e3 * s / static_cast<double>(passes), out.c_str(), out.c_str());
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const Image frame = renderer.render(scene, camera, settings);
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const Image mapped = FilmicToneMapper{}(frame);
    frame.save(out);

    double sum = 0.0;
    for (std::size_t y = 0; y < height; ++y)
        for (std::size_t x = 0; x < width; ++x) sum += mapped(x, y).r + mapped(x, y).g + mapped(x, y).b;
    std::printf("rendered %zux%zu, %zu spp, %zu bounces in %.3f s (%.1f Mpaths/s incl. upload), mean mapped value %.6f, wrote %s\n", width,
                height, spp, bounces, seconds, static_cast<double>(width * height * spp) / seconds / 1e6, sum / static_cast<double>(3 * width * height),
                out.c_str());
    return 0;
}
