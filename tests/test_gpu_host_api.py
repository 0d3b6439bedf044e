"""The C++ scene / render API on the GPU: Renderer::render and the tone mappers through
libpathtracer_host.a must return exactly what the C ABI returns for the same flattened scene."""
import os
import subprocess

import numpy as np
import pytest

import golden_cases as G
from oracle import orc
from pathtracer_amd import api, scene as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pathtracer_amd", "lib")
HARNESS = os.path.join(LIB, "libhost_harness.so")
HARNESS_DEMO = os.path.join(ROOT, "oracle", "_ref", "libhost_harness_demo.so")
HARNESS_DEMO_EDITED = os.path.join(ROOT, "oracle", "_ref", "libhost_harness_demo_edited.so")
REF_EDITED = os.path.join(ROOT, "oracle", "_ref", "libref_edited.so")
DEFAULT_SEED = 0x5EED   # Renderer's seed until setSeed is called


@pytest.fixture(scope="module")
def ctx():
    c = api.Context()
    yield c
    c.close()


def _load_image(path):
    raw = open(path, "rb").read()
    w, h = np.frombuffer(raw[:16], np.uint64)
    return np.frombuffer(raw[16:16 + int(w) * int(h) * 12], np.float32).reshape(int(h), int(w), 3)   # (a checkpoint carries a trailer)


def test_headless_demo_program_matches_c_abi(ctx, tmp_path):
    exe = os.path.join(LIB, "demo_headless")
    if not os.path.exists(exe):
        pytest.skip("demo_headless not built (make -C pathtracer_amd/host example)")
    out = tmp_path / "frame.img"
    r = subprocess.run([exe, "96", "64", "4", "8", "7", str(out)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    got = _load_image(out)
    alg = api.HostAlgebra()
    flat = S.demo_scene(alg)
    want, _ = ctx.render(ctx.upload(flat), S.demo_camera(alg, 96 / 64), 96, 64, 8, 4, seed=7)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert want.max() > 0


def test_renderer_class_matches_c_abi(ctx):
    if not os.path.exists(HARNESS):
        pytest.skip("libhost_harness.so not built")
    host = orc.Reference(HARNESS)
    alg = api.HostAlgebra()
    flat, cam = G.mixed_scene(alg), G.parity_camera(alg, 4.0 / 3.0)
    got = host.scene(flat).render(cam, 48, 36, 8, 4, seed=1, threads=1)
    want, _ = ctx.render(ctx.upload(flat), cam, 48, 36, 8, 4, seed=DEFAULT_SEED)
    assert np.array_equal(got, want)
    # a scene with a FractalObject goes to the device as well
    flat, cam = G.fractal_scene(alg), G.fractal_camera(alg)
    got = host.scene(flat).render(cam, 40, 40, 6, 4, seed=1, threads=1)
    want, st = ctx.render(ctx.upload(flat), cam, 40, 40, 6, 4, seed=DEFAULT_SEED)
    assert st["sdf_shades"] > 50
    assert np.array_equal(got, want)
    hdr = np.random.RandomState(3).gamma(0.6, 2.0, (30, 50, 3)).astype(np.float32)
    for kind in (0, 1):
        assert np.array_equal(host.tonemap(kind, hdr), ctx.tonemap(kind, hdr))


def test_unchanged_demo_classes_render_on_the_gpu(ctx):
    """The reference demo's own SDF classes (compiled unchanged into the harness) are recognised by
    Scene flattening, the demo's noiseTransform() is picked up, and the frames equal the C ABI's."""
    if not os.path.exists(HARNESS_DEMO):
        pytest.skip("oracle/_ref/libhost_harness_demo.so not built (needs the reference checkout at build time)")
    host = orc.Reference(HARNESS_DEMO)
    noise = host.init_noise(G.NOISE_SEED)
    alg = api.HostAlgebra()
    flat = S.demo_scene(alg, noise)
    cam = G.demo_closeup_camera(alg)
    got = host.scene(flat).render(cam, 40, 40, 8, 2, seed=1, threads=1)
    want, st = ctx.render(ctx.upload(flat), cam, 40, 40, 8, 2, seed=DEFAULT_SEED)
    assert st["sdf_shades"] > 100
    assert np.array_equal(got, want)
    # RoughBoxObject, RoughSphereObject and WeirdWallObject of the same unchanged block
    for flat, cam in ((G.roughbox_scene(alg, noise), G.roughbox_camera(alg)),
                      (G.roughsphere_scene(alg, noise), G.sdf_camera(alg, -8.0, 14.0)),
                      (G.weirdwall_scene(alg, noise), G.sdf_camera(alg, -14.0, 20.0, -0.5))):
        got = host.scene(flat).render(cam, 32, 32, 4, 2, seed=1, threads=1)
        want, st = ctx.render(ctx.upload(flat), cam, 32, 32, 4, 2, seed=DEFAULT_SEED)
        assert st["sdf_shades"] > 100
        assert np.array_equal(got, want)


def test_an_edited_demo_class_is_not_mistaken_for_its_device_twin(ctx, capfd):
    """The demo file is a playground: a program may change what CraterBallObject::signedDistance computes and keep the
    class's name and size (include/Object.hpp:107-177 lets Derived::signedDistance be anything).  Both harnesses below
    are built from the reference's demo block with ONE constant changed on the way through the pipe (the sphere's
    radius, 0.5 -> 0.42; oracle/Makefile): libref_edited.so is the reference's own renderer with that class,
    libhost_harness_demo_edited.so the same block compiled against this repository's headers.  The scene flattening
    must find that the device's twin does not compute what the object computes (probe points through
    mipt_sdf_eval), say so, and evaluate the object through its own virtuals: the frame then matches the reference's
    frame of the EDITED class, which the device twin's frame does not."""
    if not (os.path.exists(HARNESS_DEMO_EDITED) and os.path.exists(REF_EDITED)):
        pytest.skip("the edited-class harnesses are not built (need the reference checkout at build time)")
    host, ref = orc.Reference(HARNESS_DEMO_EDITED), orc.Reference(REF_EDITED)
    noise = host.init_noise(G.NOISE_SEED)
    assert np.array_equal(noise, ref.init_noise(G.NOISE_SEED))
    alg = api.HostAlgebra()
    flat = S.demo_scene(alg, noise)
    cam = G.demo_closeup_camera(alg)
    # the edited class itself: the same values from either build, and not the device twin's
    rs = np.random.RandomState(1)
    d = rs.normal(size=(500, 3))
    pts = (d / np.linalg.norm(d, axis=1)[:, None] * rs.uniform(0.45, 0.58, (500, 1))).astype(np.float32)
    own = host.sdf_craterball(flat.params[0, :3], pts)
    assert np.array_equal(own, ref.sdf_craterball(flat.params[0, :3], pts))
    assert np.median(np.abs(own - ctx.sdf_eval(ctx.upload(flat), 0, pts))) > 0.05
    w = h = 48
    capfd.readouterr()
    got = host.scene(flat).render(cam, w, h, 4, 32, seed=1, threads=1)
    said = capfd.readouterr().err
    assert "is not that class" in said and "evaluated by its own virtuals on the host" in said, said
    a = ref.scene(flat).render(cam, w, h, 4, 32, seed=1001, threads=1)
    b = ref.scene(flat).render(cam, w, h, 4, 32, seed=2002, threads=1)
    twin, _ = ctx.render(ctx.upload(flat), cam, w, h, 4, 32, seed=5)   # what the device's CraterBallObject renders
    rmse = lambda x, y: float(np.sqrt(np.mean((x - y) ** 2)))
    floor = rmse(a, b)
    print(f"edited class: RMSE host path / reference {rmse(got, a):.4f}, reference seeds {floor:.4f}, device twin / reference {rmse(twin, a):.4f}; "
          f"means {got.mean():.4f} / {a.mean():.4f} / {twin.mean():.4f}")
    assert floor > 0 and rmse(got, a) <= 1.25 * floor and rmse(got, b) <= 1.25 * floor
    assert abs(got.mean() / (0.5 * (a.mean() + b.mean())) - 1.0) <= 0.05
    assert rmse(twin, a) >= 1.8 * floor   # the statistic does tell the two classes apart


def test_progressive_frame_keeps_the_accumulator_on_the_device(ctx, tmp_path):
    """ProgressiveFrame (the demo's `acc += rendered; ++count; scaled = acc / count; copyToSFImage`, PathTracerDemo.cpp:934-974,
    with the accumulator on the GPU): three passes of 4 spp are the 12-spp frame bit for bit, and the display bytes are
    copyToSFImage's of that frame."""
    exe = os.path.join(LIB, "demo_headless")
    if not os.path.exists(exe):
        pytest.skip("demo_headless not built (make -C pathtracer_amd/host example)")
    out = tmp_path / "frame.img"
    r = subprocess.run([exe, "96", "64", "4", "8", "7", str(out), "3"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    got = _load_image(out)
    rgba = np.fromfile(str(out) + ".rgba", np.uint8).reshape(64, 96, 4)
    alg = api.HostAlgebra()
    want, _ = ctx.render(ctx.upload(S.demo_scene(alg)), S.demo_camera(alg, 96 / 64), 96, 64, 8, 12, seed=7)
    assert np.array_equal(got, want) and want.max() > 0
    assert np.array_equal(rgba[..., :3], (np.clip(want, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8))
    assert (rgba[..., 3] == 255).all()


def test_a_cancelled_progressive_pass_is_counted_by_what_it_added(ctx, tmp_path):
    """Renderer::cancelRender during ProgressiveFrame::add (the demo's window thread does that on every camera move,
    PathTracerDemo.cpp:982): the sample batches the pass had finished are in the sums AND in the count, so the mean is the
    frame of exactly that many samples, bit for bit.  MIPT_BATCH_PATHS makes the pass many batches long."""
    import re
    exe = os.path.join(LIB, "demo_headless")
    if not os.path.exists(exe):
        pytest.skip("demo_headless not built (make -C pathtracer_amd/host example)")
    out = tmp_path / "cancelled.img"
    w, h, spp = 256, 256, 2048
    env = dict(os.environ, MIPT_BATCH_PATHS=str(64 * 64 * 32))   # slots exist only for the tiles that see the ball: a few samples per batch
    r = subprocess.run([exe, str(w), str(h), str(spp), "8", "21", str(out), "1", "0", "cancel:25"], capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode == 0, r.stderr
    m = re.search(r"pass (finished|cancelled): (\d+) samples per pixel counted, (\d+) passes", r.stdout)
    assert m, r.stdout
    done = int(m.group(2))
    print(r.stdout.strip())
    got = _load_image(out)
    if m.group(1) == "finished":
        assert done == spp                      # the cancel came too late on this machine: nothing to tell apart
    else:
        assert done < spp and int(m.group(3)) == (1 if done else 0)
    if done == 0:
        assert not got.any()
        return
    alg = api.HostAlgebra()
    want, _ = ctx.render(ctx.upload(S.demo_scene(alg)), S.demo_camera(alg, w / h), w, h, 8, done, seed=21)
    assert np.array_equal(got, want) and want.max() > 0


def test_progressive_frame_resumes_from_a_checkpoint(ctx, tmp_path):
    """ProgressiveFrame::save / load (Image's raw format, src/Image.cpp:48-81, + the sample count and the renderer's
    position in its sample sequence): four passes of 3 spp, saved after the second and resumed by a new Renderer and a
    new frame, equal the uninterrupted 12-spp frame bit for bit; the checkpoint itself reads as an Image of sums."""
    exe = os.path.join(LIB, "demo_headless")
    if not os.path.exists(exe):
        pytest.skip("demo_headless not built (make -C pathtracer_amd/host example)")
    out = tmp_path / "resumed.img"
    r = subprocess.run([exe, "96", "64", "3", "8", "9", str(out), "4", "2"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    assert "(4 passes counted)" in r.stdout
    got = _load_image(out)
    alg = api.HostAlgebra()
    ds = ctx.upload(S.demo_scene(alg))
    cam = S.demo_camera(alg, 96 / 64)
    want, _ = ctx.render(ds, cam, 96, 64, 8, 12, seed=9)
    assert np.array_equal(got, want) and want.max() > 0
    half, _ = ctx.render(ds, cam, 96, 64, 8, 6, seed=9)
    sums = _load_image(str(out) + ".ckpt")            # the file's head is a plain Image: the sums after two passes
    assert np.allclose(sums, half * 6.0, rtol=1e-6, atol=1e-7)
    tail = np.fromfile(str(out) + ".ckpt", np.uint64, offset=16 + 96 * 64 * 12)
    assert tail.shape == (5,) and tail[1] == 6 and tail[2] == 2 and tail[3] == 9 and tail[4] == 2


def test_renderer_over_the_nodes_gpus_equals_the_single_gpu_frame(tmp_path):
    """MIPT_GPUS=all: Renderer::render divides the frame's samples among every GPU of the node and combines them with one
    RCCL all-reduce (mipt_init_all / mipt_node_render).  On this one-GPU box that is a node of one rank, and the frame
    is the single-GPU frame bit for bit."""
    exe = os.path.join(LIB, "demo_headless")
    if not os.path.exists(exe):
        pytest.skip("demo_headless not built (make -C pathtracer_amd/host example)")
    outs = []
    for gpus in (None, "all"):
        out = tmp_path / f"frame_{gpus}.img"
        env = {k: v for k, v in os.environ.items() if k != "MIPT_GPUS"}
        if gpus:
            env["MIPT_GPUS"] = gpus
        r = subprocess.run([exe, "96", "64", "8", "8", "11", str(out)], capture_output=True, text=True, timeout=100, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(_load_image(out))
    assert np.array_equal(outs[0], outs[1]) and outs[0].max() > 0
