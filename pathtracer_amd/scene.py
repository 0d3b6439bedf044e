"""Flat scene description and the synthetic scene generators of the benchmark configurations.

A FlatScene is what Scene::updateGeometry hands to the device library: per primitive a kind, the
forward and inverse affine transform (Linear row-major, then translation), nine geometry floats and
a material row; plus the material table and the demo's noise transform.  The arrays are exactly the
arguments of mipt_scene_upload (include/mipt.h).

Transforms are composed through an `alg` object (rotation / affine_mul / affine_inverse /
noise_transform).  By default that is the device library's host-side helper set, which follows the
reference's operation order (src/LinearAlgebra.cpp) so that inverses match
Object::setTransformation bit for bit.
"""
from dataclasses import dataclass

import numpy as np

SPHERE, BOX, TRIANGLE, CRATERBALL, FRACTAL, ROUGHBOX, ROUGHSPHERE, WEIRDWALL = 0, 1, 2, 3, 4, 5, 6, 7

IDENTITY12 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float32)


def material(diffuse=0.3, specular=0.3, sharpness=0.9, albedo=(1.0, 1.0, 1.0), emitted=(0.0, 0.0, 0.0),
             transmittance=0.0, ior=1.5, internal=(0.9, 0.9, 0.9)):
    """A BasicMaterial row; the defaults are BasicMaterial() (src/BasicMaterial.cpp:9-18)."""
    return np.array([diffuse, specular, sharpness, *albedo, *emitted, transmittance, ior, *internal], np.float32)


def translation(x, y, z):
    t = IDENTITY12.copy()
    t[9:] = (x, y, z)
    return t


def scale(k):
    t = IDENTITY12.copy()
    t[0] = t[4] = t[8] = k
    return t


def linear_to_affine(l9):
    t = np.zeros(12, np.float32)
    t[:9] = l9
    return t


def camera(transform12, aspect=1.0, fov=30.0, focal=10.0, blur=0.0):
    """Camera record: transform, aspect ratio, field of view (degrees), focal distance, blur radius.
    Defaults follow Camera::Camera (src/Camera.cpp:8-15)."""
    return np.concatenate([np.asarray(transform12, np.float32), np.array([aspect, fov, focal, blur], np.float32)])


@dataclass
class FlatScene:
    kind: np.ndarray        # (n,) uint32
    fwd: np.ndarray         # (n, 12) float32
    inv: np.ndarray         # (n, 12) float32
    params: np.ndarray      # (n, 9) float32
    material: np.ndarray    # (n,) uint32
    materials: np.ndarray   # (m, 14) float32
    noise: np.ndarray       # (12,) float32

    @property
    def n_prims(self):
        return int(self.kind.shape[0])


class SceneBuilder:
    def __init__(self, alg, noise=None):
        self.alg = alg
        self.kind, self.fwd, self.params, self.mat_idx, self.mats = [], [], [], [], []
        self.noise = IDENTITY12.copy() if noise is None else np.asarray(noise, np.float32)

    def _add(self, kind, params, mat, fwd):
        p = np.zeros(9, np.float32)
        p[: len(params)] = params
        self.kind.append(kind)
        self.params.append(p)
        self.fwd.append(IDENTITY12.copy() if fwd is None else np.asarray(fwd, np.float32))
        self.mat_idx.append(len(self.mats))
        self.mats.append(np.asarray(mat, np.float32))
        return len(self.kind) - 1

    def sphere(self, radius, mat, fwd=None):
        return self._add(SPHERE, [radius], mat, fwd)

    def box(self, half, mat, fwd=None):
        return self._add(BOX, list(half), mat, fwd)

    def triangle(self, a, b, c, mat, fwd=None):
        return self._add(TRIANGLE, [*a, *b, *c], mat, fwd)

    def craterball(self, offset, mat, fwd=None):
        return self._add(CRATERBALL, list(offset), mat, fwd)

    def roughbox(self, half, mat, fwd=None):
        """The demo's RoughBoxObject (PathTracerDemo.cpp:228-270); needs the scene's noise transform."""
        return self._add(ROUGHBOX, list(half), mat, fwd)

    def roughsphere(self, radius, mat, fwd=None):
        """The demo's RoughSphereObject (PathTracerDemo.cpp:156-226); needs the scene's noise transform."""
        return self._add(ROUGHSPHERE, [radius], mat, fwd)

    def weirdwall(self, mat, fwd=None):
        """The demo's WeirdWallObject (PathTracerDemo.cpp:273-349); needs the scene's noise transform."""
        return self._add(WEIRDWALL, [], mat, fwd)

    def fractal(self, mat, fwd=None):
        """FractalObject (include/Object.hpp:179-192): no parameters beyond its material."""
        return self._add(FRACTAL, [], mat, fwd)

    def build(self):
        fwd = np.stack(self.fwd).astype(np.float32)
        if hasattr(self.alg, "affine_inverse_batch"):
            inv = self.alg.affine_inverse_batch(fwd)
        else:
            rows = []
            for f in fwd:
                r = self.alg.affine_inverse(f)
                if r is None:
                    raise ValueError("transformation is not invertible by the reference's rule (det < 1e-6)")
                rows.append(r)
            inv = np.stack(rows)
        return FlatScene(np.array(self.kind, np.uint32), fwd, inv.astype(np.float32), np.stack(self.params),
                         np.array(self.mat_idx, np.uint32), np.stack(self.mats), self.noise)


# Noise transform used by every fixed demo-scene workload: Translation(t) * Rx * Ry * Rz with these six
# numbers (the demo draws them at random per process, PathTracerDemo.cpp:37-54).
DEMO_NOISE_T = (0.417022, 0.720324, 0.000114)
DEMO_NOISE_R = (0.302333, 0.146756, 0.092339)


def demo_scene(alg, noise=None):
    """The demo's default scene (PathTracerDemo.cpp:564-590): a crater ball and an emissive slab."""
    if noise is None:
        noise = alg.noise_transform(DEMO_NOISE_T, DEMO_NOISE_R)
    b = SceneBuilder(alg, noise)
    ball = material(diffuse=1.0, specular=0.0, transmittance=0.0, albedo=(0.5, 0.5, 0.5), emitted=(0, 0, 0))
    b.craterball((-8.0, 15.0, -6.5), ball, translation(0.0, 0.5, -2.0))
    light = material(diffuse=1.0, specular=0.0, transmittance=0.0, albedo=(1, 1, 1), emitted=(2.0, 2.0, 2.0))
    b.box((100.0, 0.1, 100.0), light, translation(0.0, -50.0, 0.0))
    return b.build()


def demo_camera(alg, aspect=1.0):
    """The demo's camera (PathTracerDemo.cpp:897-903)."""
    t = alg.affine_mul(translation(0.0, 0.0, -30.0), scale(0.01))
    return camera(t, aspect=aspect, fov=10.0, focal=30.0, blur=0.0)


def cornell_scene(alg):
    """The demo's commented-out Cornell box (PathTracerDemo.cpp:637-707)."""
    b = SceneBuilder(alg)
    matte = dict(diffuse=1.0, specular=0.0, transmittance=0.0)
    white = material(albedo=(1, 1, 1), **matte)
    red = material(albedo=(0.8, 0.1, 0.1), **matte)
    green = material(albedo=(0.1, 0.8, 0.1), **matte)
    glow = material(albedo=(1, 1, 1), emitted=(50.0, 40.0, 30.0), **matte)
    mirror = material(diffuse=0.05, specular=0.95, sharpness=0.95, albedo=(0.9, 0.9, 0.9), transmittance=0.0)
    d = 0.01
    b.box((1.0, d, 1.0), glow, translation(0.0, -5.0 + 2.0 * d, 0.0))
    b.box((5.0, 5.0, d), white, translation(0.0, 0.0, 5.0))
    b.box((5.0, d, 5.0), white, translation(0.0, 5.0, 0.0))
    b.box((5.0, d, 5.0), white, translation(0.0, -5.0, 0.0))
    b.box((d, 5.0, 5.0), red, translation(-5.0, 0.0, 0.0))
    b.box((d, 5.0, 5.0), green, translation(5.0, 0.0, 0.0))
    b.sphere(1.5, white, translation(-3.0, 3.5, -2.5))
    b.sphere(1.5, white, translation(3.0, -3.5, -2.5))
    b.sphere(1.5, mirror, translation(3.0, 3.5, 2.5))
    b.sphere(1.5, mirror, translation(-3.0, -3.5, 2.5))
    return b.build()


def cornell_camera(alg, aspect=1.0):
    return camera(translation(0.0, 0.0, -14.0), aspect=aspect, fov=20.0, focal=10.0, blur=0.0)


def random_scene(alg, n, seed=7, radius_scale=1.0, extent=10.0, glass_mirror_mix=False):
    """Alternating spheres and Y-rotated boxes with random diffuse/specular materials and 10 % emitters
    (the pattern of the demo's commented-out random scenes, PathTracerDemo.cpp:607-635, 757-779).
    With glass_mirror_mix the materials cycle mirror / glass / diffuse (20 % of the diffuse emit)."""
    rs = np.random.RandomState(seed)
    u = lambda *shape: rs.random_sample(shape).astype(np.float32)
    kind = np.where(np.arange(n) % 2 == 0, SPHERE, BOX).astype(np.uint32)
    lo, hi = np.float32(0.02 * radius_scale), np.float32(0.12 * radius_scale)
    params = np.zeros((n, 9), np.float32)
    params[:, :3] = lo + (hi - lo) * u(n, 3)
    centres = (np.float32(extent) * (np.float32(2.0) * u(n, 3) - np.float32(1.0))).astype(np.float32)
    angles = np.where(kind == BOX, np.float32(2.0) * u(n) - np.float32(1.0), np.float32(0.0)).astype(np.float32)
    fwd = np.zeros((n, 12), np.float32)
    fwd[:, :9] = alg.rotation_batch("y", angles)
    fwd[kind == SPHERE, :9] = IDENTITY12[:9]
    fwd[:, 9:] = centres
    inv = alg.affine_inverse_batch(fwd)
    mats = np.zeros((n, 14), np.float32)
    if glass_mirror_mix:
        cls = np.arange(n) % 3
        mats[:, 10] = 1.5
        mats[:, 11:14] = 0.9
        m = cls == 0   # mirrors (PathTracerDemo.cpp:660-664)
        mats[m, 0], mats[m, 1], mats[m, 2] = 0.05, 0.95, 0.95
        mats[m, 3:6] = 0.9
        g = cls == 1   # glass (PathTracerDemo.cpp:868-874)
        mats[g, 0], mats[g, 1], mats[g, 2], mats[g, 9], mats[g, 10] = 0.05, 0.05, 1.0, 0.9, 1.55
        mats[g, 3:6] = 0.9
        d = cls == 2
        mats[d, 0], mats[d, 1], mats[d, 2] = 1.0, 0.0, 0.9
        mats[d, 3:6] = (np.float32(0.5) + np.float32(0.5) * u(n))[d, None]
        emit = d & (u(n) < 0.2)
        k = (np.float32(30.0) * u(n) ** 3)[:, None] * u(n, 3)
        mats[emit, 6:9] = k[emit]
    else:
        mats[:, 0] = u(n)
        mats[:, 1] = np.float32(0.3) * u(n)
        mats[:, 2] = u(n)
        mats[:, 3:6] = (np.float32(0.5) + np.float32(0.5) * u(n))[:, None]
        mats[:, 9] = 0.0
        mats[:, 10] = 1.5
        mats[:, 11:14] = 0.9
        emit = u(n) < 0.1
        k = (np.float32(30.0) * u(n) ** 3)[:, None] * u(n, 3)
        mats[emit, 6:9] = k[emit]
    return FlatScene(kind, fwd, inv.astype(np.float32), params, np.arange(n, dtype=np.uint32), mats.astype(np.float32),
                     IDENTITY12.copy())


def random_scene_camera(alg, aspect=16.0 / 9.0):
    return camera(translation(0.0, 0.0, -30.0), aspect=aspect, fov=20.0, focal=30.0, blur=0.0)
