// CraterBallObject::signedDistance (PathTracerDemo.cpp:26-116, 361-466) for whole waves.  Included by
// kernels.hip after its vector helpers.
//
// Two evaluators share every per-term formula below and agree bit for bit, so a point's value does
// not depend on which of them served it (which lanes of which wave ask together is a matter of timing):
//   * coop_near: all 64 lanes work on one point (or a group of 4) -- for waves with few points;
//   * lane_near: one point per lane -- for waves where most lanes hold a point.
// What makes them agree: floating-point contraction is off in this file and every fused multiply-add
// is spelled fmaf; sums over craters are taken in fixed point (integer addition is associative);
// everything else that crosses lanes is a minimum or a maximum.
#pragma once

#pragma clang fp contract(off)

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
// tanh(c) = 1 - 2 / (exp(2c) + 1); absolute error ~1e-7, which is what the SDF terms it feeds need
__device__ __forceinline__ float fast_tanh(float c) {
    const float e = fast_exp2(c * 2.885390081777927f);   // 2 / ln 2
    return fmaf(-2.0f, fast_rcp(e + 1.0f), 1.0f);
}
__device__ __forceinline__ float fnorm2(f3 v) { return fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)); }

// ---- wave-wide reductions on the DPP crossbar (every lane must be active) --------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u(uint32_t identity, uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(identity), static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140, kRowBcast15 = 0x142, kRowBcast31 = 0x143;
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114;

__device__ __forceinline__ int wave_sum_i(int x) {
    uint32_t v = static_cast<uint32_t>(x);
    v += dpp_u<kQuadXor1>(0u, v);
    v += dpp_u<kQuadXor2>(0u, v);
    v += dpp_u<kRowHalfMirror>(0u, v);
    v += dpp_u<kRowMirror>(0u, v);
    v += dpp_u<kRowBcast15, 0xa>(0u, v);
    v += dpp_u<kRowBcast31, 0xc>(0u, v);
    return __builtin_amdgcn_readlane(static_cast<int>(v), 63);
}
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) {
    v = min(v, dpp_u<kQuadXor1>(0xffffffffu, v));
    v = min(v, dpp_u<kQuadXor2>(0xffffffffu, v));
    v = min(v, dpp_u<kRowHalfMirror>(0xffffffffu, v));
    v = min(v, dpp_u<kRowMirror>(0xffffffffu, v));
    v = min(v, dpp_u<kRowBcast15, 0xa>(0xffffffffu, v));
    v = min(v, dpp_u<kRowBcast31, 0xc>(0xffffffffu, v));
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
__device__ __forceinline__ float bcast_lane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t bcast_lane_u(uint32_t v, int lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}
// row-wise (16 lanes) reductions: afterwards every lane of a row holds the row's result
__device__ __forceinline__ float row_max(float v) {
    const float ninf = -3.402823466e38f;
    v = fmaxf(v, dpp_f<kQuadXor1>(ninf, v));
    v = fmaxf(v, dpp_f<kQuadXor2>(ninf, v));
    v = fmaxf(v, dpp_f<kRowHalfMirror>(ninf, v));
    v = fmaxf(v, dpp_f<kRowMirror>(ninf, v));
    return v;
}
__device__ __forceinline__ uint32_t row_min_u(uint32_t v) {
    v = min(v, dpp_u<kQuadXor1>(0xffffffffu, v));
    v = min(v, dpp_u<kQuadXor2>(0xffffffffu, v));
    v = min(v, dpp_u<kRowHalfMirror>(0xffffffffu, v));
    v = min(v, dpp_u<kRowMirror>(0xffffffffu, v));
    return v;
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Sums over craters are carried as integers of 2^-30: every term is below 0.03 in magnitude and at
// most 64 craters contribute, so the sum stays below 2^31.
constexpr float kFixScale = 1073741824.0f;
__device__ __forceinline__ int to_fix(float v) { return static_cast<int>(v * kFixScale); }
__device__ __forceinline__ float from_fix(int v) { return static_cast<float>(v) * (1.0f / kFixScale); }

__device__ __forceinline__ uint32_t mul24_lo(int a, int k) {
    int r;
    asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(a));
    return static_cast<uint32_t>(r);
}
__device__ __forceinline__ uint32_t shl5(uint32_t h) {
    uint32_t r;
    asm("v_lshlrev_b32_e32 %0, 5, %1" : "=v"(r) : "v"(h));
    return r;
}

// One cell of for_each_nearby_scattered_point :79-116: hashed, jittered lattice point.  The reference
// forms the jitter through double; fmaf(scatter/2, m, g) rounds the same exact value once.
// Lattice coordinates and multipliers fit 24 signed bits, so the full-rate 24-bit multiply returns
// the same low 32 bits as the wrapping int multiply of the reference, and h * 31 = (h << 5) - h.
// Both are spelled in asm because the compiler folds the C forms back into quarter-rate v_mul_lo_u32.
constexpr int kHashX = 885889, kHashY = 745273, kHashZ = 1220689;
__device__ __forceinline__ uint32_t cell_hash_seed(f3 g) {   // the sum the four mixing rounds start from
    return mul24_lo(static_cast<int>(g.x), kHashX) + mul24_lo(static_cast<int>(g.y), kHashY) + mul24_lo(static_cast<int>(g.z), kHashZ);
}
__device__ __forceinline__ uint32_t cell_hash_mix(uint32_t h) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h |= (h & 0x0f0f0f0fu) << 4;
        h = shl5(h) - h;
    }
    return h;
}
// The jittered point of a cell for scatter 1 (the small and tiny craters), given the cell minus one half:
// g + 0.5 (m / 512 - 1) = (g - 0.5) + m / 1024, both sides exact before the one rounding of the fused
// multiply-add, so this is scattered_cell(g, 0.5, .) bit for bit with three instructions fewer.
__device__ __forceinline__ f3 jitter_unit(f3 g_minus_half, uint32_t h) {
    return mk3(fmaf(static_cast<float>(h & 0x3ffu), 1.0f / 1024.0f, g_minus_half.x), fmaf(static_cast<float>((h >> 10) & 0x3ffu), 1.0f / 1024.0f, g_minus_half.y),
               fmaf(static_cast<float>((h >> 20) & 0x3ffu), 1.0f / 1024.0f, g_minus_half.z));
}
__device__ __forceinline__ f3 scattered_cell(f3 g, float half_scatter, uint32_t& hash) {
    const uint32_t h = cell_hash_mix(cell_hash_seed(g));
    hash = h;
    const float mx = fmaf(static_cast<float>(h & 0x3ffu), 1.0f / 512.0f, -1.0f);
    const float my = fmaf(static_cast<float>((h >> 10) & 0x3ffu), 1.0f / 512.0f, -1.0f);
    const float mz = fmaf(static_cast<float>((h >> 20) & 0x3ffu), 1.0f / 512.0f, -1.0f);
    return mk3(fmaf(half_scatter, mx, g.x), fmaf(half_scatter, my, g.y), fmaf(half_scatter, mz, g.z));
}

// Projects a lattice point onto the sphere of radius `scale` about `off` and reports whether it lies
// within sqrt(0.1) of that sphere (the `(point_on_sphere - point).normSquared() < 0.1f` tests).
__device__ __forceinline__ bool project_cell(f3 point, f3 off, float scale, f3& on_sphere) {
    const f3 v = point - off;
    const float k = fast_rsq(fnorm2(v)) * scale;
    on_sphere = mk3(fmaf(v.x, k, off.x), fmaf(v.y, k, off.y), fmaf(v.z, k, off.z));
    return fnorm2(on_sphere - point) < 0.1f;
}

// ---- shared per-term formulas -------------------------------------------------------------------
// Direction to the centre and the big-crater query point (:368-373).
__device__ __forceinline__ void near_frame(f3 pos, f3 off, f3& u, f3& query) {
    const float il = fast_rsq(fnorm2(pos));
    u = mk3(-pos.x * il, -pos.y * il, -pos.z * il);
    query = mk3(fmaf(u.x, 2.0f, off.x), fmaf(u.y, 2.0f, off.y), fmaf(u.z, 2.0f, off.z));
}

// The big craters of a ball do not depend on the query: a lattice cell passes the shell test or not,
// and its point on the sphere, size, depth parameter and age follow from its hash alone.  They are
// tabulated once per primitive (about 32 cells pass for any offset), three float4 per crater in lattice
// order: {cell.xyz, size}, {on_sphere.xyz, t}, {age, 1 / (5 size), passes, -}.
__global__ void __launch_bounds__(256) k_crater_cells(float ox, float oy, float oz, int bx, int by, int bz, int n, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n * n) return;
    const int ix = i / (n * n), iy = (i / n) % n, iz = i % n;
    const f3 off = mk3(ox, oy, oz);
    const f3 g = mk3(static_cast<float>(bx + ix), static_cast<float>(by + iy), static_cast<float>(bz + iz));
    uint32_t hash;
    f3 on;
    const f3 point = scattered_cell(g, 0.3f, hash);
    const bool pass = project_cell(point, off, 2.0f, on);
    const float t = static_cast<float>(hash & 0xffffu) * (1.0f / 65535.0f);
    const float size = fmaf(0.4f * t, t, 0.05f);
    out[3 * i] = make_float4(g.x, g.y, g.z, size);
    out[3 * i + 1] = make_float4(on.x, on.y, on.z, t);
    out[3 * i + 2] = make_float4(__uint_as_float(hash >> 16), fast_rcp(size * 5.0f), pass ? 1.0f : 0.0f, 0.0f);
}

// Is crater r0 one of the 4^3 cells about the query?  fl = floor(query) + 0.5; cell - floor(query) must be -1, 0, 1 or 2.
__device__ __forceinline__ bool crater_in_reach(float4 r0, f3 fl) {
    return fabsf(r0.x - fl.x) < 2.0f && fabsf(r0.y - fl.y) < 2.0f && fabsf(r0.z - fl.z) < 2.0f;
}

// One big crater at one query point (:381-408).
struct BigCrater {
    float depth, lin, k1, k2;   // depth profile, linear falloff, the two fallout noise arguments are on + k qv
    f3 qv;
    bool outside;
};
__device__ __forceinline__ BigCrater big_crater(f3 query, float4 r0, float4 r1, float4 r2) {
    BigCrater b;
    const float size = r0.w, t = r1.w;
    b.qv = query - mk3(r1.x, r1.y, r1.z);
    const float dist2 = fnorm2(b.qv);
    const float idist = fast_rsq(fmaxf(dist2, 1e-30f));
    const float dist = dist2 * idist;
    const float c20 = 20.0f * (dist - size);
    const float th = fast_tanh(c20);
    // tanh(c) for c <= 1e-3, else 0.9 t tanh(c) / (0.02 / c + c^2) = 0.9 t tanh(c) c / (0.02 + c^3)
    const float rim = ((0.9f * t) * th) * (c20 * fast_rcp(fmaf(c20 * c20, c20, 0.02f)));
    b.depth = c20 > 1e-3f ? rim : th;
    b.outside = dist >= size;
    b.lin = fmaxf(0.0f, fmaf(-dist, r2.y, 1.2f));
    const float si = size * idist;
    b.k1 = 3.0f * si;
    b.k2 = 40.0f * si;
    return b;
}
__device__ __forceinline__ int big_crater_fix(const BigCrater& b) { return to_fix(-0.02f * b.depth); }

// pow(0, y) = 0 for the y in [1, 3] that occurs, so a crater whose linear falloff is zero deposits
// exactly nothing and its two noise() calls are skipped (lin > 0 is part of `deposits`).
__device__ __forceinline__ int fallout_fix(float n1, float n2, float lin, float depth) {
    const float falloff = fast_exp2(fmaf(3.0f, n1, 2.0f) * fast_log2(lin));
    return to_fix(((0.1f * falloff) * fmaf(0.4f, depth, 0.2f)) * n2);
}

// One cell of the small (scale 10, :412-437) or tiny (scale 60, :440-463) craters.
// `seed` is cell_hash_seed(cell) and `cell_minus_half` the cell's lattice point minus 0.5 on every axis.
struct MinorCell {
    bool pass;
    uint32_t age;
    float size, c50;
};
__device__ __forceinline__ MinorCell minor_cell(f3 q2, uint32_t seed, f3 cell_minus_half, f3 off, float scale, bool tiny) {
    MinorCell m;
    f3 on2;
    const uint32_t h2 = cell_hash_mix(seed);
    const f3 pt2 = jitter_unit(cell_minus_half, h2);
    m.pass = project_cell(pt2, off, scale, on2);
    const float t2 = static_cast<float>(h2 & 0xffffu) * (1.0f / 65535.0f);
    m.age = h2 >> 16;
    const float d2 = __builtin_amdgcn_sqrtf(fnorm2(q2 - on2));
    m.size = tiny ? fmaf(0.35f, t2, 0.02f) : fmaf(0.35f * t2, t2, 0.05f);
    m.c50 = 50.0f * (d2 - m.size);
    return m;
}
// Small craters: `crater = max(crater, -0.005 tanh(c50))` over the cells that take effect is the term of the
// smallest c50 among them (tanh rises with its argument), so one tanh per evaluation serves all eight cells.
constexpr float kNoSmallCrater = 3.402823466e38f;   // "no cell took effect" as a c50
__device__ __forceinline__ float small_total(float least_c50) { return least_c50 < kNoSmallCrater ? -0.005f * fast_tanh(least_c50) : -3.402823466e38f; }
// Tiny craters: (-0.004 size) (0.5 tanh(c50) - 0.5).  From c50 = 10 on, fast_tanh returns exactly 1 (exp2 of
// more than 28 leaves 2 / (e + 1) below a quarter of an ulp of 1) and the term exactly 0, which the running
// maximum, starting from 0, ignores: such cells are skipped.
constexpr float kTinyReach = 10.0f;
__device__ __forceinline__ float tiny_term(const MinorCell& m) { return (-0.004f * m.size) * fmaf(0.5f, fast_tanh(m.c50), -0.5f); }

__device__ __forceinline__ float near_total(float s, int big_fix, float crater_small, float tiny, int fallout, float nf1, float nf2) {
    const float crater = fmaxf(from_fix(big_fix), crater_small);
    return ((((s + crater) + tiny) + from_fix(fallout)) + 0.01f * nf1) + 0.4f * nf2;
}

// The small and tiny craters' lattice cells do not depend on the query either: a cell's hash, its jittered point and
// that point's projection onto the sphere of radius `scale` about the ball's offset follow from the cell and the offset.
// A query (offset + scale u, |u| = 1) falls into cells floor(offset) - scale - 1 .. floor(offset) + scale + 1 and looks
// at each cell and its upper neighbour, so a cube of 2 scale + 4 cells per axis about floor(offset) - scale - 1 covers
// all of them: 24^3 cells at scale 10, 124^3 at scale 60, one float4 per cell -- {on_sphere.xyz, hash bits} for a cell
// that passes the shell test, {+inf, 0, 0, hash bits} for one that does not (its distance from any query is then +inf,
// which fails every later test the way `pass` did).  Made once per ball by k_minor_cells with the very functions the
// evaluators use, so a value read from the table is the value computed on the spot, bit for bit.
#ifndef MIPT_CELLS_IN_FLIGHT
#define MIPT_CELLS_IN_FLIGHT 4
#endif
constexpr int kSmallCube = 24, kTinyCube = 124;
constexpr uint32_t kSmallCells = kSmallCube * kSmallCube * kSmallCube, kTinyCells = kTinyCube * kTinyCube * kTinyCube;
constexpr uint32_t kMinorCellsPerBall = kSmallCells + kTinyCells;   // float4s per ball: the small craters' cube, then the tiny ones'
__global__ void __launch_bounds__(256) k_minor_cells(float ox, float oy, float oz, float4* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kMinorCellsPerBall) return;
    const bool tiny = i >= kSmallCells;
    const uint32_t j = tiny ? i - kSmallCells : i;
    const uint32_t n = tiny ? kTinyCube : kSmallCube;
    const float scale = tiny ? 60.0f : 10.0f;
    const f3 off = mk3(ox, oy, oz);
    const f3 g = mk3((floorf(ox) - (scale + 1.0f)) + static_cast<float>(j / (n * n)), (floorf(oy) - (scale + 1.0f)) + static_cast<float>((j / n) % n),
                     (floorf(oz) - (scale + 1.0f)) + static_cast<float>(j % n));
    const uint32_t h2 = cell_hash_mix(cell_hash_seed(g));
    const f3 pt2 = jitter_unit(mk3(g.x - 0.5f, g.y - 0.5f, g.z - 0.5f), h2);
    f3 on2;
    const bool pass = project_cell(pt2, off, scale, on2);
    out[i] = pass ? make_float4(on2.x, on2.y, on2.z, __uint_as_float(h2)) : make_float4(__uint_as_float(0x7f800000u), 0.0f, 0.0f, __uint_as_float(h2));
}
// Index, within a ball's cube of `n` cells per axis that starts at floor(offset) - first, of the cell a query falls into.
__device__ __forceinline__ uint32_t minor_cell_index(f3 q2, f3 off, float first, int n) {
    const int ix = static_cast<int>(floorf(q2.x) - (floorf(off.x) - first)), iy = static_cast<int>(floorf(q2.y) - (floorf(off.y) - first)),
              iz = static_cast<int>(floorf(q2.z) - (floorf(off.z) - first));
    const int hi = n - 2;   // (a query cannot leave the cube and has its upper neighbour inside: the clamp only keeps a wrong address impossible)
    return static_cast<uint32_t>((min(max(ix, 0), hi) * n + min(max(iy, 0), hi)) * n + min(max(iz, 0), hi));
}
// The same cell as minor_cell() from its table entry.
__device__ __forceinline__ MinorCell minor_cell_from(f3 q2, float4 rec, bool tiny) {
    MinorCell m;
    const uint32_t h2 = __float_as_uint(rec.w);
    m.pass = true;   // a cell that does not pass lies at +inf
    const float t2 = static_cast<float>(h2 & 0xffffu) * (1.0f / 65535.0f);
    m.age = h2 >> 16;
    const float d2 = __builtin_amdgcn_sqrtf(fnorm2(q2 - mk3(rec.x, rec.y, rec.z)));
    m.size = tiny ? fmaf(0.35f, t2, 0.02f) : fmaf(0.35f * t2, t2, 0.05f);
    m.c50 = 50.0f * (d2 - m.size);
    return m;
}

// What each lane contributes to the cooperative parts of an evaluation.
struct CoopLane {
    float si, sj, sk;   // lanes 0-7 / 8-15 of a row: cell of the 2^3 small / tiny-crater neighbourhood (0..1)
    float a_xy, a_z1;   // this lane's element of the two octave matrices (below)
    float kw[4];        // weights of octaves (lane >> 5), 2 + (lane >> 5), 4 + ..., 6 + ...
};

// noise() :56-70 is 8 octaves; octave o samples the triangle wave at q_o = A_o p + b_o (the composed
// affine, 13 floats per octave in `noise_oct`: A row-major, b + 0.25, weight).  For 32 points at once
// that is a 32 x 4 by 4 x 32 matrix product -- rows (octave, coordinate) padded to 4 per octave, columns
// (x, y, z, 1) -- which two chained v_mfma_f32_32x32x2_f32 compute exactly (f32 FMA chains, k = x, y,
// then z, 1) while the vector pipe does other waves' work.  Lane l supplies element [l & 31][l >> 5] of
// each 32 x 2 tile.
__device__ __forceinline__ CoopLane make_coop_lane(const float* noise_oct, uint32_t lane) {
    CoopLane c;
    c.si = static_cast<float>((lane >> 2) & 1u);
    c.sj = static_cast<float>((lane >> 1) & 1u);
    c.sk = static_cast<float>(lane & 1u);
    c.a_xy = c.a_z1 = 0.0f;
    c.kw[0] = c.kw[1] = c.kw[2] = c.kw[3] = 0.0f;
    if (!noise_oct) return c;   // a scene without crater balls carries no octave table (and never evaluates one)
    const uint32_t row = lane & 31u, k = lane >> 5, octave = row >> 2, coord = row & 3u;
    if (coord < 3u) {
        c.a_xy = noise_oct[13u * octave + 3u * coord + k];                                         // x or y coefficient
        c.a_z1 = k == 0u ? noise_oct[13u * octave + 3u * coord + 2u] : noise_oct[13u * octave + 9u + coord];   // z coefficient or b + 0.25
    }
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) c.kw[i] = noise_oct[13u * (2u * i + k) + 12u];
    return c;
}

typedef float mipt_f32x16 __attribute__((ext_vector_type(16)));

// noise() for 32 points at once: the matrix products, then the triangle waves and the sum over octaves.
// `c_xy` is x (lanes 0-31) or y (lanes 32-63) of point (lane & 31), `c_z1` is z (lanes 0-31) or 1; every
// lane gets the value of point (lane & 31).  All 64 lanes must be active.  A lane holds, of its point,
// q (+ 0.25) of octaves (lane >> 5) + 2 i in elements 4 i .. 4 i + 2.
__device__ __forceinline__ float noise32(const CoopLane& cl, float c_xy, float c_z1) {
    mipt_f32x16 q = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    q = __builtin_amdgcn_mfma_f32_32x32x2f32(cl.a_xy, c_xy, q, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f32_32x32x2f32(cl.a_z1, c_z1, q, 0, 0, 0);
    // triangle(v) = -1 + 4 |v - 0.25 - floor(v + 0.25)| = 4 |fract(v + 0.25) - 0.5| - 1  (:26-28)
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = (fabsf(__builtin_amdgcn_fractf(q[4 * i]) - 0.5f) + fabsf(__builtin_amdgcn_fractf(q[4 * i + 1]) - 0.5f)) +
                        fabsf(__builtin_amdgcn_fractf(q[4 * i + 2]) - 0.5f);
        v = i == 0 ? cl.kw[0] * a : fmaf(cl.kw[i], a, v);
    }
    // even octaves on lanes 0-31, odd octaves on lanes 32-63: sum the two halves
    {
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    // noise = sum_i k_i (4 a_i - 3) with sum_i k_i = 0.02 (1 - 0.9^8) / 0.1
    return fmaf(4.0f, v, -3.0f * 0.11390656f);
}

// =============================================================================================
// coop_near: G points (G = 1 or 4) whose sphere distances s = |pos| - 0.5 are already known to be
// <= 0.1 (the far field returns s itself), evaluated by ONE WAVE: off, pos and s are wave-uniform and
// every lane must be active.
//
//   * lane l holds crater l of the ball's table (r0..r2; lanes past the table hold a cell out of
//     anyone's reach), one pass per point;
//   * noise() calls -- two per big crater that deposits fallout, two for the final terms -- run 16 at
//     a time on the matrix core; the calls of all G points share the rounds;
//   * the 8 small-crater and 8 tiny-crater cells of point g run on the 16 lanes of DPP row g; the
//     order dependence of the small craters through min_age is an exclusive prefix minimum.
//
// An evaluation costs a few hundred wave instructions with no data-dependent loop except the noise
// rounds, so the 256-step worst case of a march is a fraction of a millisecond.
// =============================================================================================
constexpr int kCoopGroup = 4;                                    // points one wave evaluates together
constexpr int kCoopPts = 2 * 64 * kCoopGroup + 32;               // fallout points + final points, rounded up to whole rounds of 32
constexpr int kCoopLdsFloats = 4 * kCoopPts;                     // x, y, z, value

template <int G>
__device__ __forceinline__ void coop_near(const CoopLane& cl, float* lds, float4 r0, float4 r1, float4 r2, const f3 (&off)[G], const f3 (&pos)[G],
                                          const float (&s)[G], uint32_t lane, float (&out)[G], uint32_t (&noise_calls)[G]) {
    static_assert(G == 1 || G == kCoopGroup, "one point or a full group");
    float* px = lds;
    float* py = lds + kCoopPts;
    float* pz = lds + 2 * kCoopPts;
    float* pv = lds + 3 * kCoopPts;
    f3 u[G];
    float depth[G], lin[G];
    int big_fix[G];
    uint32_t rank[G], min_age0[G];
    bool deposits[G];
    uint32_t n_fall = 0;   // fallout points staged so far

    // ---- big craters :375-409: one crater per lane, one pass per point -----------------------
#pragma unroll
    for (int g = 0; g < G; ++g) {
        f3 query;
        near_frame(pos[g], off[g], u[g], query);
        const f3 fl = mk3(floorf(query.x) + 0.5f, floorf(query.y) + 0.5f, floorf(query.z) + 0.5f);
        const bool in = crater_in_reach(r0, fl);
        const BigCrater b = big_crater(query, r0, r1, r2);
        depth[g] = b.depth;
        lin[g] = b.lin;
        deposits[g] = in && b.outside && b.lin > 0.0f;
        big_fix[g] = wave_sum_i(in ? big_crater_fix(b) : 0);
        // ages are compared as their 16-bit integers (age = bits / 65535); 0xffff is the initial 1.0
        min_age0[g] = wave_min_u((in && !b.outside) ? __float_as_uint(r2.x) : 0xffffu);
        // stage this point's fallout noise() arguments
        const unsigned long long dmask = __ballot(deposits[g]);
        rank[g] = n_fall + 2u * lanes_below(dmask);
        const uint32_t n_dep = __popcll(dmask);
        n_fall += 2u * n_dep;
        if (deposits[g]) {
            px[rank[g]] = fmaf(b.k1, b.qv.x, r1.x);
            py[rank[g]] = fmaf(b.k1, b.qv.y, r1.y);
            pz[rank[g]] = fmaf(b.k1, b.qv.z, r1.z);
            px[rank[g] + 1] = fmaf(b.k2, b.qv.x, r1.x);
            py[rank[g] + 1] = fmaf(b.k2, b.qv.y, r1.y);
            pz[rank[g] + 1] = fmaf(b.k2, b.qv.z, r1.z);
        }
        noise_calls[g] = 2u * n_dep + 2u;
    }
    // the two final noise() arguments of every point follow the fallout points
    const uint32_t n_pts = n_fall + 2u * G;
    if (lane < 2u * G) {
        f3 q = pos[0] + off[0];
#pragma unroll
        for (int g = 1; g < G; ++g)
            if ((lane >> 1) == static_cast<uint32_t>(g)) q = pos[g] + off[g];
        const float f = (lane & 1u) ? 1.5f : 20.0f;
        px[n_fall + lane] = q.x * f;
        py[n_fall + lane] = q.y * f;
        pz[n_fall + lane] = q.z * f;
    }
    wave_lds_fence();

    // ---- noise(): 32 points per round ------------------------------------------------------------
    // (the next round's arguments are fetched before this round computes, so LDS latency stays hidden;
    // slots past n_pts hold stale data whose values are never used)
    {
        const bool upper = lane >= 32u;
        const float* p_xy = upper ? py : px;
        uint32_t slot = lane & 31u;
        float c_xy = p_xy[slot], c_z = pz[slot];
        for (uint32_t first = 0; first < n_pts; first += 32u) {
            const uint32_t next = slot + 32u < static_cast<uint32_t>(kCoopPts) ? slot + 32u : slot;
            const float n_xy = p_xy[next], n_z = pz[next];
            const float v = noise32(cl, c_xy, upper ? 1.0f : c_z);
            if (!upper) pv[slot] = v;
            slot = next;
            c_xy = n_xy;
            c_z = n_z;
        }
    }
    wave_lds_fence();

    // ---- small craters on lanes 0-7 and tiny craters on lanes 8-15 of row g ---------------------
    const uint32_t row = lane >> 4;
    f3 ur = u[0], offr = off[0];
    uint32_t min_age_r = min_age0[0];
#pragma unroll
    for (int g = 1; g < G; ++g) {
        const bool mine = row == static_cast<uint32_t>(g);
        ur = mk3(mine ? u[g].x : ur.x, mine ? u[g].y : ur.y, mine ? u[g].z : ur.z);
        offr = mk3(mine ? off[g].x : offr.x, mine ? off[g].y : offr.y, mine ? off[g].z : offr.z);
        min_age_r = mine ? min_age0[g] : min_age_r;
        // (kept a chain of selects: read as an array indexed by the row, `u` moves to scratch memory)
        asm volatile("" : "+v"(ur.x), "+v"(ur.y), "+v"(ur.z));
    }
    const bool in_use = row < static_cast<uint32_t>(G);
    const bool tiny_lane = (lane & 8u) != 0u;
    const float scale = tiny_lane ? 60.0f : 10.0f;
    const f3 q2 = mk3(fmaf(ur.x, scale, offr.x), fmaf(ur.y, scale, offr.y), fmaf(ur.z, scale, offr.z));
    const f3 cell = mk3(floorf(q2.x) + cl.si, floorf(q2.y) + cl.sj, floorf(q2.z) + cl.sk);
    const MinorCell mc = minor_cell(q2, cell_hash_seed(cell), mk3(cell.x - 0.5f, cell.y - 0.5f, cell.z - 0.5f), offr, scale, tiny_lane);
    // A small-crater cell takes effect iff it passes the shell and rim tests and its age does not exceed
    // the running minimum, which only cells that took effect lower: the running minimum before cell k is
    // min(min_age0, ages of the earlier candidates), an exclusive prefix minimum over lanes 0-7 of the row.
    const bool cand = in_use && !tiny_lane && mc.pass && mc.c50 < 0.0f && mc.age <= min_age_r;
    const uint32_t xa = cand ? mc.age : 0xffffffffu;
    uint32_t incl = xa;
    incl = min(incl, dpp_u<kRowShr1>(0xffffffffu, incl));
    incl = min(incl, dpp_u<kRowShr2>(0xffffffffu, incl));
    incl = min(incl, dpp_u<kRowShr4>(0xffffffffu, incl));
    const uint32_t excl = dpp_u<kRowShr1>(0xffffffffu, incl);
    const bool takes = cand && mc.age <= excl;
    const float crater_small_r = small_total(-row_max(takes ? -mc.c50 : -kNoSmallCrater));   // the least c50 of the cells that take effect
    const uint32_t min_age1 = min(min_age_r, row_min_u(xa));
    const bool tiny_ok = in_use && tiny_lane && mc.pass && mc.age <= min_age1 && mc.c50 < kTinyReach;
    const float tiny_r = row_max(tiny_ok ? tiny_term(mc) : 0.0f);

    // ---- fallout and the final sum per point -----------------------------------------------
#pragma unroll
    for (int g = 0; g < G; ++g) {
        int f = 0;
        if (deposits[g]) f = fallout_fix(pv[rank[g]], pv[rank[g] + 1], lin[g], depth[g]);
        const int fallout = wave_sum_i(f);
        const float nf1 = pv[n_fall + 2 * g], nf2 = pv[n_fall + 2 * g + 1];
        out[g] = near_total(s[g], big_fix[g], bcast_lane(crater_small_r, 16 * g), bcast_lane(tiny_r, 16 * g), fallout, nf1, nf2);
    }
    wave_lds_fence();
}

// =============================================================================================
// lane_near: one point per lane.  Which craters of the ball's table lie within reach of a point depends
// only on the lattice cell of its query, so a table made at upload maps that cell to a 64-bit mask of
// crater indices; every lane walks ITS OWN mask, fetching crater c's record from lane c (which holds it)
// through the LDS crossbar (ds_bpermute: no memory, no bank conflicts).  The loop runs as long as the
// longest list of the wave (about 17 steps where a scan of the whole table took 26 with 40 % of the
// lanes busy).  The small and tiny craters go cell by cell, and only noise() stays cooperative: every
// lane stages the arguments of its calls in a ring in LDS, and rounds of 16 calls drain the ring whenever
// half of it is full, writing each value to the LDS slot its owner named.
// =============================================================================================
constexpr int kRing = 256;          // staged noise() arguments; one crater step adds at most 128
constexpr int kLaneDeposits = 8;    // fallout deposits one lane can hold; a point with more is left to the caller
constexpr int kLaneLdsFloats = 4 * kRing + 4 * 64 * kLaneDeposits + 2 * 64 + 8;
constexpr int kNearLdsFloats = kLaneLdsFloats > kCoopLdsFloats ? kLaneLdsFloats : kCoopLdsFloats;   // per wave, either evaluator

__device__ __forceinline__ void noise_ring_rounds(const CoopLane& cl, float* lds, uint32_t lane, uint32_t& done, uint32_t staged, bool flush_all) {
    wave_lds_fence();
    const uint32_t* rt = reinterpret_cast<const uint32_t*>(lds + 3 * kRing);
    const bool upper = lane >= 32u;
    const float* r_xy = lds + (upper ? kRing : 0);   // x of the call on lanes 0-31, y on lanes 32-63
    const float* r_z = lds + 2 * kRing;
    // 32 calls per round; the next round's arguments are fetched before this round computes
    uint32_t call = done + (lane & 31u);
    uint32_t slot = call & (kRing - 1);
    float c_xy = r_xy[slot], c_z = r_z[slot];
    uint32_t target = rt[slot];
    while (flush_all ? done < staged : done + 32u <= staged) {
        const uint32_t next = (call + 32u) & (kRing - 1);
        const float n_xy = r_xy[next], n_z = r_z[next];
        const uint32_t next_target = rt[next];
        const float v = noise32(cl, c_xy, upper ? 1.0f : c_z);
        if (!upper && call < staged) lds[target] = v;
        done += 32u;
        call += 32u;
        c_xy = n_xy;
        c_z = n_z;
        target = next_target;
    }
    if (done > staged) done = staged;
    wave_lds_fence();
}

__device__ __forceinline__ float4 bcast_lane4(float4 v, int lane) {
    return make_float4(bcast_lane(v.x, lane), bcast_lane(v.y, lane), bcast_lane(v.z, lane), bcast_lane(v.w, lane));
}
// v of lane (addr / 4), per lane
__device__ __forceinline__ float fetch_lane(float v, int addr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }

constexpr int kCellCube = 9;   // lattice cells per axis about floor(offset) - 4 that a query (offset + 2 u, |u| = 1) can fall into
// Index into a ball's cell -> crater-mask table (kCellCube^3 entries).
__device__ __forceinline__ uint32_t crater_cell_index(f3 query, f3 off) {
    const int ix = static_cast<int>(floorf(query.x) - (floorf(off.x) - 4.0f));
    const int iy = static_cast<int>(floorf(query.y) - (floorf(off.y) - 4.0f));
    const int iz = static_cast<int>(floorf(query.z) - (floorf(off.z) - 4.0f));
    const int hi = kCellCube - 1;
    return static_cast<uint32_t>((min(max(ix, 0), hi) * kCellCube + min(max(iy, 0), hi)) * kCellCube + min(max(iz, 0), hi));
}

// Evaluates the near field at `pos` for every lane with `active` set; all 64 lanes must call and the
// crater table (lane c holds crater c in t0..t2) with its cell masks is the same for all of them.
// Returns false for a lane whose point has more fallout deposits than a lane can hold; its `out` is
// then not written.
__device__ __forceinline__ bool lane_near(float4 t0, float4 t1, float4 t2, const unsigned long long* cell_masks, const float4* minor, const CoopLane& cl, float* lds, bool active,
                                          f3 pos, f3 off, float s, uint32_t lane, float& out, uint32_t& noise_calls) {
    float* rx = lds;
    float* ry = lds + kRing;
    float* rz = lds + 2 * kRing;
    uint32_t* rt = reinterpret_cast<uint32_t*>(lds + 3 * kRing);
    constexpr uint32_t kCol = 4 * kRing;                            // float4 {n1, n2, lin, depth} of deposit j of lane l at kCol + 4 (64 j + l)
    constexpr uint32_t kFin = kCol + 4 * 64 * kLaneDeposits;        // the two final noise values of lane l at kFin + 2 l
    f3 u, query;
    near_frame(pos, off, u, query);
    unsigned long long todo = active ? cell_masks[crater_cell_index(query, off)] : 0ull;   // craters within reach of this lane's query
    // the two final noise() calls of every active lane open the ring
    const unsigned long long am = __ballot(active);
    uint32_t staged = 2u * static_cast<uint32_t>(__popcll(am)), done = 0;
    if (active) {
        const uint32_t r = 2u * lanes_below(am);
        const f3 q = pos + off;
        rx[r] = q.x * 20.0f;
        ry[r] = q.y * 20.0f;
        rz[r] = q.z * 20.0f;
        rt[r] = kFin + 2u * lane;
        rx[r + 1] = q.x * 1.5f;
        ry[r + 1] = q.y * 1.5f;
        rz[r + 1] = q.z * 1.5f;
        rt[r + 1] = kFin + 2u * lane + 1u;
    }

    // ---- big craters from the table ----------------------------------------------------------
    int big_fix = 0;
    uint32_t min_age = 0xffffu, n_dep = 0;
    bool overflow = false;
    while (__ballot(todo != 0ull)) {
        const bool in = todo != 0ull;
        const int from = in ? 4 * __builtin_ctzll(todo) : 0;   // byte address of the lane that holds this lane's next crater
        todo &= todo - 1ull;
        const float4 r0 = make_float4(fetch_lane(t0.x, from), fetch_lane(t0.y, from), fetch_lane(t0.z, from), fetch_lane(t0.w, from));
        const float4 r1 = make_float4(fetch_lane(t1.x, from), fetch_lane(t1.y, from), fetch_lane(t1.z, from), fetch_lane(t1.w, from));
        const float4 r2 = make_float4(fetch_lane(t2.x, from), fetch_lane(t2.y, from), 0.0f, 0.0f);
        bool dep = false;
        BigCrater b;   // (read only where `dep` is set, which implies `in`: left uninitialised so that no lane zeroes it every pass)
        if (in) {
            b = big_crater(query, r0, r1, r2);
            big_fix += big_crater_fix(b);
            if (!b.outside) min_age = min(min_age, __float_as_uint(r2.x));
            dep = b.outside && b.lin > 0.0f;
            if (dep && n_dep >= static_cast<uint32_t>(kLaneDeposits)) {
                overflow = true;
                dep = false;
            }
        }
        const unsigned long long dm = __ballot(dep);
        if (dm) {
            if (dep) {
                const uint32_t call = staged + 2u * lanes_below(dm);
                const uint32_t s0 = call & (kRing - 1), s1 = (call + 1u) & (kRing - 1);
                const uint32_t col = kCol + 4u * (64u * n_dep + lane);
                rx[s0] = fmaf(b.k1, b.qv.x, r1.x);
                ry[s0] = fmaf(b.k1, b.qv.y, r1.y);
                rz[s0] = fmaf(b.k1, b.qv.z, r1.z);
                rt[s0] = col;
                rx[s1] = fmaf(b.k2, b.qv.x, r1.x);
                ry[s1] = fmaf(b.k2, b.qv.y, r1.y);
                rz[s1] = fmaf(b.k2, b.qv.z, r1.z);
                rt[s1] = col + 1u;
                lds[col + 2] = b.lin;
                lds[col + 3] = b.depth;
                ++n_dep;
            }
            staged += 2u * static_cast<uint32_t>(__popcll(dm));
            if (staged - done >= 128u) noise_ring_rounds(cl, lds, lane, done, staged, false);
        }
    }

    // ---- small and tiny craters, cell by cell in the reference's order; remaining noise() calls; fallout -------------
    // With the ball's cell tables, the eight small-crater entries are asked for before the fallout of the big craters is
    // summed and each tiny-crater entry as soon as the small-crater entry whose registers it takes has been used, so
    // that the answers are there when they are needed.
    float least_c50 = kNoSmallCrater, tiny = 0.0f;
    int fallout = 0;
    auto sum_fallout = [&]() {
        float4 e = *reinterpret_cast<const float4*>(lds + kCol + 4u * lane);
        for (uint32_t j = 0; j < n_dep; ++j) {
            const uint32_t jn = j + 1u < static_cast<uint32_t>(kLaneDeposits) ? j + 1u : j;
            const float4 en = *reinterpret_cast<const float4*>(lds + kCol + 4u * (64u * jn + lane));
            fallout += fallout_fix(e.x, e.y, e.z, e.w);
            e = en;
        }
    };
    if (minor) {
        noise_ring_rounds(cl, lds, lane, done, staged, true);
        if (active) {
            const f3 q_small = mk3(fmaf(u.x, 10.0f, off.x), fmaf(u.y, 10.0f, off.y), fmaf(u.z, 10.0f, off.z));
            const f3 q_tiny = mk3(fmaf(u.x, 60.0f, off.x), fmaf(u.y, 60.0f, off.y), fmaf(u.z, 60.0f, off.z));
            const float4* at_small = minor + minor_cell_index(q_small, off, 11.0f, kSmallCube);
            const float4* at_tiny = minor + kSmallCells + minor_cell_index(q_tiny, off, 61.0f, kTinyCube);
            // kInFlight entries in flight at a time: each is replaced, once used, by the entry kInFlight places on
            const auto small_at = [&](int cell) { return at_small[((cell >> 2) * kSmallCube + ((cell >> 1) & 1)) * kSmallCube + (cell & 1)]; };
            const auto tiny_at = [&](int cell) { return at_tiny[((cell >> 2) * kTinyCube + ((cell >> 1) & 1)) * kTinyCube + (cell & 1)]; };
            constexpr int kInFlight = MIPT_CELLS_IN_FLIGHT;
            float4 rec[kInFlight];
#pragma unroll
            for (int cell = 0; cell < kInFlight; ++cell) rec[cell] = small_at(cell);
            sum_fallout();
#pragma unroll
            for (int cell = 0; cell < 8; ++cell) {
                const MinorCell mc = minor_cell_from(q_small, rec[cell % kInFlight], false);
                rec[cell % kInFlight] = cell + kInFlight < 8 ? small_at(cell + kInFlight) : tiny_at(cell + kInFlight - 8);
                if (mc.c50 < 0.0f && mc.age <= min_age) {
                    least_c50 = fminf(least_c50, mc.c50);
                    min_age = mc.age;
                }
            }
#pragma unroll
            for (int cell = 0; cell < 8; ++cell) {
                const MinorCell mc = minor_cell_from(q_tiny, rec[cell % kInFlight], true);
                if (cell + kInFlight < 8) rec[cell % kInFlight] = tiny_at(cell + kInFlight);
                if (mc.age <= min_age && mc.c50 < kTinyReach) tiny = fmaxf(tiny, tiny_term(mc));
            }
        }
    } else {
        if (active) {
#pragma unroll 1
            for (int pass_no = 0; pass_no < 2; ++pass_no) {
                const bool is_tiny = pass_no != 0;
                const float scale = is_tiny ? 60.0f : 10.0f;
                const f3 q2 = mk3(fmaf(u.x, scale, off.x), fmaf(u.y, scale, off.y), fmaf(u.z, scale, off.z));
                const f3 f2 = mk3(floorf(q2.x), floorf(q2.y), floorf(q2.z));
                const uint32_t seed0 = cell_hash_seed(f2);   // the neighbours' seeds differ from it by wave-uniform constants
#pragma unroll 8
                for (int cell = 0; cell < 8; ++cell) {
                    const int dx = cell >> 2, dy = (cell >> 1) & 1, dz = cell & 1;
                    const uint32_t seed = seed0 + static_cast<uint32_t>(dx * kHashX + dy * kHashY + dz * kHashZ);
                    const f3 gm = mk3(f2.x + (dx ? 0.5f : -0.5f), f2.y + (dy ? 0.5f : -0.5f), f2.z + (dz ? 0.5f : -0.5f));
                    const MinorCell mc = minor_cell(q2, seed, gm, off, scale, is_tiny);
                    if (!is_tiny) {
                        if (mc.pass && mc.c50 < 0.0f && mc.age <= min_age) {
                            least_c50 = fminf(least_c50, mc.c50);
                            min_age = mc.age;
                        }
                    } else if (mc.pass && mc.age <= min_age && mc.c50 < kTinyReach) {
                        tiny = fmaxf(tiny, tiny_term(mc));
                    }
                }
            }
        }
        noise_ring_rounds(cl, lds, lane, done, staged, true);
        if (active) sum_fallout();
    }
    if (active && !overflow) {
        const float nf1 = lds[kFin + 2u * lane], nf2 = lds[kFin + 2u * lane + 1u];
        out = near_total(s, big_fix, small_total(least_c50), tiny, fallout, nf1, nf2);
        noise_calls = 2u * n_dep + 2u;
    }
    wave_lds_fence();
    return !overflow;
}

// =============================================================================================
// wave_near: near-field values for every lane with `wants` set (all 64 lanes must call).  Lanes are
// served primitive by primitive; per primitive, one point per lane when at least `lane_threshold`
// lanes wait, otherwise (and for the rare point lane_near leaves over) the whole wave works on groups
// of four points, then on single points.
// =============================================================================================
// A wave keeps the crater table of the ball it served last in registers (lane c holds crater c), so a march
// that stays on one ball -- every march of a scene with one ball -- reads the table from memory once.
struct CraterTableRegs {
    float4 r0, r1;
    float age_bits, inv_5size;   // the two fields of the third record that the evaluators use
    uint32_t base;               // first record of the cached table, 0xffffffff: nothing cached
};
__device__ __forceinline__ CraterTableRegs no_crater_table() {
    return CraterTableRegs{make_float4(1e30f, 1e30f, 1e30f, 1.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f), 0.0f, 1.0f, 0xffffffffu};
}

__device__ __forceinline__ void wave_near(const float* crater_table, const unsigned long long* crater_masks, const float4* minor_cells, CraterTableRegs& cache, const CoopLane& cl, float* lds, uint32_t lane, uint32_t lane_threshold, bool wants,
                                          f3 p, f3 off, float s, uint32_t crater_base, uint32_t crater_count, float& value, uint32_t& n_evals,
                                          uint32_t& n_noise) {
    unsigned long long todo = __ballot(wants);
    while (todo) {
        const int first = __ffsll(static_cast<long long>(todo)) - 1;
        // `crater_count`: the table's length in the low 8 bits, the ball's ordinal among the scene's crater balls above
        const uint32_t base = bcast_lane_u(crater_base, first), packed = bcast_lane_u(crater_count, first);
        const uint32_t count = packed & 0xffu;
        const unsigned long long* cell_masks = crater_masks + static_cast<size_t>(packed >> 8) * (kCellCube * kCellCube * kCellCube);
        const bool mine = wants && crater_base == base;
        unsigned long long need = __ballot(mine);
        todo &= ~need;
        // lane c holds crater c of the ball's table; lanes past the table hold a cell out of anyone's reach
        if (cache.base != base) {
            const float4* table = reinterpret_cast<const float4*>(crater_table) + 3u * base;
            cache = no_crater_table();
            cache.base = base;
            if (lane < count) {
                cache.r0 = table[3u * lane];
                cache.r1 = table[3u * lane + 1u];
                const float4 third = table[3u * lane + 2u];
                cache.age_bits = third.x;
                cache.inv_5size = third.y;
            }
        }
        const float4 r0 = cache.r0, r1 = cache.r1, r2 = make_float4(cache.age_bits, cache.inv_5size, 0.0f, 0.0f);
        if (static_cast<uint32_t>(__popcll(need)) >= lane_threshold) {
            float v = 0.0f;
            uint32_t calls = 0;
            const float4* minor = minor_cells ? minor_cells + static_cast<size_t>(packed >> 8) * kMinorCellsPerBall : nullptr;
            const bool ok = lane_near(r0, r1, r2, cell_masks, minor, cl, lds, mine, p, off, s, lane, v, calls);
            if (mine && ok) {
                value = v;
                n_noise += calls;
                ++n_evals;
            }
            need &= ~__ballot(mine && ok);
        }
        while (__popcll(need) >= kCoopGroup) {
            int src[kCoopGroup];
            f3 P[kCoopGroup], O[kCoopGroup];
            float S[kCoopGroup], val[kCoopGroup];
            uint32_t calls[kCoopGroup];
#pragma unroll
            for (int g = 0; g < kCoopGroup; ++g) {
                src[g] = __ffsll(static_cast<long long>(need)) - 1;
                need &= need - 1ull;
                P[g] = mk3(bcast_lane(p.x, src[g]), bcast_lane(p.y, src[g]), bcast_lane(p.z, src[g]));
                O[g] = mk3(bcast_lane(off.x, src[g]), bcast_lane(off.y, src[g]), bcast_lane(off.z, src[g]));
                S[g] = bcast_lane(s, src[g]);
            }
            coop_near<kCoopGroup>(cl, lds, r0, r1, r2, O, P, S, lane, val, calls);
#pragma unroll
            for (int g = 0; g < kCoopGroup; ++g)
                if (lane == static_cast<uint32_t>(src[g])) {
                    value = val[g];
                    n_noise += calls[g];
                    ++n_evals;
                }
        }
        while (need) {
            const int src = __ffsll(static_cast<long long>(need)) - 1;
            need &= need - 1ull;
            const f3 P[1] = {mk3(bcast_lane(p.x, src), bcast_lane(p.y, src), bcast_lane(p.z, src))};
            const f3 O[1] = {mk3(bcast_lane(off.x, src), bcast_lane(off.y, src), bcast_lane(off.z, src))};
            const float S[1] = {bcast_lane(s, src)};
            float val[1];
            uint32_t calls[1];
            coop_near<1>(cl, lds, r0, r1, r2, O, P, S, lane, val, calls);
            if (lane == static_cast<uint32_t>(src)) {
                value = val[0];
                n_noise += calls[0];
                ++n_evals;
            }
        }
    }
}
