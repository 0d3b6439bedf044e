// C ABI of libmipt.so (include/mipt.h): context, scene upload, the per-frame wavefront loop, tone
// mapping and the host-side affine helpers.
#include "../../include/mipt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "bvh.h"
#include "lbvh.h"
#include "device_types.h"
#include "hostmath.h"
#include "kernels.h"

using namespace mipt;

struct mipt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::atomic<int> cancel{0};
    volatile uint32_t* cancel_host = nullptr;   // pinned, mapped: the kernels read it through cancel_dev
    const volatile uint32_t* cancel_dev = nullptr;
    int last_hip = 0;
    std::mutex mutex;   // render calls are serialised, as Renderer::render is (src/Renderer.cpp:144)

    // working set, grown on demand and kept between calls
    size_t slots_cap = 0, bits_cap = 0;
    PathBuffers pb{};
    void* arena = nullptr;
    uint32_t* vis_tiles = nullptr;   // image tiles that can see anything, for frames whose objects cover part of the screen
    size_t vis_cap = 0;
    std::vector<uint32_t> vis_host;
    hipEvent_t batch_ev[2] = {nullptr, nullptr};   // end of the last two sample batches (at most two are queued at a time)
    uint32_t* host_q = nullptr;      // PathBuffers::host_queue, allocated for scenes with host-evaluated objects
    size_t host_q_cap = 0;
    void* host_stage = nullptr;      // device staging of the host round trips (rays out, hits / bounces back)
    size_t host_stage_cap = 0;       // bytes
    float* frame = nullptr;          // per-call radiance sum (mipt_render)
    size_t frame_cap = 0;
    float* tm_scratch = nullptr;     // Reinhard partials
    float* tm_static = nullptr;      // the same in a file-scope device array (this device's copy), for mipt_tonemap_device_noalloc
    float* stage = nullptr;          // staging for probes / host tone mapping
    size_t stage_cap = 0;
    bool device_bvh = false;        // MIPT_DEVICE_BVH=1: mipt_scene_upload builds the tree on the GPU
    bool full_march = false;        // MIPT_FULL_MARCH=1: no shortcut for SDF marches that cannot end (for the equivalence test)
    uint32_t lane_threshold = 14;   // MIPT_LANE_THRESHOLD: 1 = always one point per lane, 65 = never
    // MIPT_CARRY_LAG (0: off), MIPT_CARRY_AFTER: SDF marches still running when a launch's queue is handed out go to the next
    // round's launch (BatchParams::carry_max_lag, carry_after)
    uint32_t carry_lag = 4, carry_after = 32;
    size_t minor_table_balls = 64;  // MIPT_CELL_TABLE_BALLS: scenes with up to this many crater balls get the small / tiny crater cell tables (0: none)
    uint32_t sort_marches = 2;      // MIPT_SORT_MARCHES: rounds 1 .. this sort their SDF queue by the rays' first query cell (0: never)
    bool no_few = false;            // MIPT_NO_FEW=1 (test aid): the closest-hit kernel for scenes of a handful of primitives is not used
    int split_primary = -1;         // MIPT_SPLIT_PRIMARY: 1 = camera rays are written out and traced by the later bounces' kernel, 0 = by the fused kernel, unset = by scene
    uint32_t carry_cap = 0;         // records in each of the two hand-over buffers of the arena
    int min_stack = 0;   // MIPT_MIN_STACK: the traversal kernels are picked as if every tree needed at least this many stack entries (a test aid)
    bool pair_plain = false;   // MIPT_PAIR_PLAIN=1 (test aid): stack entries carry no entry distance, as in trees too large for the packed form
    bool pair_tight = false;   // MIPT_PAIR_TIGHT=1 (test aid): the two rays of a lane share only as many stack rows as one ray can need (+ 3), so the second ray starts over often
    uint32_t trav_quorum_pair = 32, trav_keep_pair = 76;   // MIPT_PAIR_QUORUM (0: one ray per lane), MIPT_PAIR_KEEP: the same with two rays per lane
    uint32_t trav_quorum = 20, trav_quorum_primary = 1, trav_keep = 40;   // MIPT_LEAF_QUORUM, MIPT_LEAF_QUORUM_PRIMARY, MIPT_KEEP_BUSY: scheduling of the BVH traversal's two phases (tuning aids)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // optional per-launch timing
    bool counting = false;          // mipt_set_counting: the traversal kernels count node visits and primitive tests
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    struct Span { int stage; size_t e0, e1; };
    std::vector<Span> spans;
    size_t ev_used = 0;
};

struct mipt_scene {
    int device = 0;
    SceneDev dev{};
    void* d_prims = nullptr;
    void* d_tri = nullptr;
    void* d_mats = nullptr;
    void* d_nodes = nullptr;
    void* d_leaf_boxes = nullptr;
    void* d_sdf = nullptr;
    void* d_oct = nullptr;
    void* d_craters = nullptr;
    void* d_crater_masks = nullptr;
    void* d_minor_cells = nullptr;   // small / tiny crater cell tables of the crater balls
    uint32_t n_nodes = 0, n_analytic = 0, n_sdf = 0;
    int depth = 0;        // levels of the two-child tree the builders made
    int stack_need = 0;   // traversal stack entries the four-child tree can need
    std::vector<uint8_t> kinds;   // per primitive, for argument checks
    // objects evaluated on the host: their primitive indices in the order of dev.host_boxes, the caller's callbacks
    std::vector<uint32_t> host_prims;
    mipt_host_callbacks host_cb{};
    void* d_host_boxes = nullptr;
    // padded world boxes that together hold everything a ray can hit: the analytic primitives' common
    // box and the SDF primitives' boxes (merged into one if there are many)
    std::vector<std::array<float, 6>> cull_boxes;
};

namespace {

#define HIP_TRY(ctx, expr)                         \
    do {                                           \
        const hipError_t e_ = (expr);              \
        if (e_ != hipSuccess) {                    \
            (ctx)->last_hip = static_cast<int>(e_);\
            return MIPT_ERR_HIP;                   \
        }                                          \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Per-slot arrays of the path arena, in the order ensure_slots lays them out: ray origin, ray direction,
// attenuation, radiance (float4 each), hit key (u64), hit position (float4), two live-path queues, the SDF-march
// queue and the SDF-shade queue (u32 each), the marches' starting points (float4).  The memory budget below is priced from the same table.
constexpr size_t kSlotArrayBytes[11] = {16, 16, 16, 16, 8, 16, 4, 4, 4, 4, 16};
constexpr size_t slot_bytes() {
    size_t b = 0;
    for (size_t v : kSlotArrayBytes) b += v;
    return b;
}
constexpr size_t kMaxSlots = (size_t(1) << 31) - 256;      // slot indices are 32-bit
// Marches a launch can hand to the next (64 bytes each, two buffers): what the waves resident at a launch's end can
// hold, 3,072 waves x 64 lanes, with room to spare.  Marches beyond that are finished where they are.
constexpr size_t kCarryRecords = size_t(1) << 18;
constexpr size_t kDefaultArenaCeiling = size_t(48) << 30;  // bytes; MIPT_BATCH_PATHS asks for more

// Paths in flight per sample batch.  Every bounce of every batch ends in a latency-bound tail (the longest SDF
// marches, a few rays on an otherwise idle chip), so fewer, larger batches are faster; but the arena stays
// allocated until mipt_trim or mipt_shutdown and whatever else lives on this GPU (another context, the caller's
// tensors, RCCL's buffers) needs memory too.  Default: up to 80 % of the memory that is free, at most 48 GiB;
// MIPT_BATCH_PATHS (path slots) overrides both.  Slots exist only for pixels inside the screen bounds of the scene's
// objects, so the demo frame at 1080p, 1024 spp (3 % of the pixels) is one batch of 7 GB.
size_t batch_path_budget(const mipt_ctx* c) {
    if (const char* e = std::getenv("MIPT_BATCH_PATHS")) {
        const long long v = std::atoll(e);
        if (v > 0) return std::min(static_cast<size_t>(v), kMaxSlots);
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return size_t(128) << 20;
    free_b += c->slots_cap * slot_bytes();   // the arena is replaced when it has to grow
    const size_t bytes = std::min(free_b / 10 * 8, kDefaultArenaCeiling);
    const size_t budget = bytes * 8 / (slot_bytes() * 8 + 1);   // one bit of the live-path bitmap per slot
    return std::min(std::max(budget, size_t(1) << 20), kMaxSlots);
}

// `words`: 64-bit words of the live-path bitmap (samples * ceil(pixels / 64)).
int ensure_slots(mipt_ctx* c, size_t slots, size_t words) {
    if (slots <= c->slots_cap && words <= c->bits_cap) return MIPT_OK;
    slots = std::max(slots, c->slots_cap);
    words = std::max(words, c->bits_cap);
    if (c->arena) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(c->arena));
        c->arena = nullptr;
        c->slots_cap = 0;
        c->bits_cap = 0;
    }
    const size_t n = align_up(slots, 256);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    size_t o_arr[11];
    for (int k = 0; k < 11; ++k) o_arr[k] = take(n * kSlotArrayBytes[k]);
    const size_t o_ro = o_arr[0], o_rd = o_arr[1], o_att = o_arr[2], o_rad = o_arr[3], o_key = o_arr[4], o_pos = o_arr[5];
    const size_t o_q0 = o_arr[6], o_q1 = o_arr[7], o_qs = o_arr[8], o_qss = o_arr[9];
    const size_t o_cnt = take(sizeof(BounceCounters) * kCounterRows);
    const size_t o_tot = take(sizeof(Totals));
    const size_t o_bits = take(words * 8);
    const size_t o_latch = take(sizeof(uint32_t));
    const size_t carry_cap = std::min<size_t>(n, kCarryRecords);
    const size_t o_carry0 = take(carry_cap * sizeof(CarryRec)), o_carry1 = take(carry_cap * sizeof(CarryRec));
    void* base = nullptr;
    const hipError_t e = hipMalloc(&base, off);
    if (e != hipSuccess) {
        c->last_hip = static_cast<int>(e);
        return e == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
    }
    char* b = static_cast<char*>(base);
    c->arena = base;
    c->slots_cap = n;
    c->pb.ray_o = reinterpret_cast<float4*>(b + o_ro);
    c->pb.ray_d = reinterpret_cast<float4*>(b + o_rd);
    c->pb.att = reinterpret_cast<float4*>(b + o_att);
    c->pb.rad = reinterpret_cast<float4*>(b + o_rad);
    c->pb.hit_key = reinterpret_cast<unsigned long long*>(b + o_key);
    c->pb.hit_pos = reinterpret_cast<float4*>(b + o_pos);
    c->pb.queue[0] = reinterpret_cast<uint32_t*>(b + o_q0);
    c->pb.queue[1] = reinterpret_cast<uint32_t*>(b + o_q1);
    c->pb.sdf_queue = reinterpret_cast<uint32_t*>(b + o_qs);
    c->pb.sdf_shade_queue = reinterpret_cast<uint32_t*>(b + o_qss);
    c->pb.march_start = reinterpret_cast<float4*>(b + o_arr[10]);
    c->pb.counters = reinterpret_cast<BounceCounters*>(b + o_cnt);
    c->pb.totals = reinterpret_cast<Totals*>(b + o_tot);
    c->pb.live_bits = reinterpret_cast<unsigned long long*>(b + o_bits);
    c->pb.cancel = c->cancel_dev;
    c->pb.cancel_latch = reinterpret_cast<uint32_t*>(b + o_latch);
    c->pb.carry[0] = reinterpret_cast<CarryRec*>(b + o_carry0);
    c->pb.carry[1] = reinterpret_cast<CarryRec*>(b + o_carry1);
    c->carry_cap = static_cast<uint32_t>(carry_cap);
    c->bits_cap = words;
    return MIPT_OK;
}

int ensure_floats(mipt_ctx* c, float** buf, size_t* cap, size_t n) {
    if (n <= *cap) return MIPT_OK;
    if (*buf) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
    }
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(buf), n * sizeof(float));
    if (e != hipSuccess) {
        c->last_hip = static_cast<int>(e);
        return e == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
    }
    *cap = n;
    return MIPT_OK;
}

CameraRec pack_camera(const mipt_camera& cam) {
    CameraRec r;
    std::memcpy(r.lin, cam.transform, sizeof(float) * 9);
    std::memcpy(r.tr, cam.transform + 9, sizeof(float) * 3);
    r.aspect = cam.aspect;
    // the per-ray constants of Camera::getViewRay (src/Camera.cpp:72-77), hoisted with the same float ops
    r.fov_scale = std::tan(cam.fov_degrees * 3.141592654f / 180.0f);
    r.inv_focal = 1.0f / cam.focal_distance;
    r.blur_rad = cam.blur_radius * std::max(cam.aspect, 1.0f / cam.aspect);
    return r;
}

// Screen bounds of the scene's boxes for this camera (pixels, inclusive, two pixels of margin for the
// sub-pixel jitter and rounding).  Camera::getViewRay (src/Camera.cpp:65-81) without lens blur sends the ray
// of screen point (u, v) from T (u, v, 0) along L (k u, k v, 1), k = tan(fov): in the camera's frame every ray
// passes through (0, 0, -1/k), so a point (X, Y, Z) is seen at (u, v) = (X, Y) / (1 + k Z).  A box whose
// corners all have 1 + k Z > 0 projects inside the bounds of its projected corners; anything else, or
// a lens with blur, gives up and returns the whole frame.
CullRects screen_rects(const mipt_scene* sc, const mipt_camera& cam, uint32_t W, uint32_t H) {
    CullRects full{};
    full.n = 1;
    full.r[0][0] = 0;
    full.r[0][1] = 0;
    full.r[0][2] = static_cast<int>(W) - 1;
    full.r[0][3] = static_cast<int>(H) - 1;
    if (cam.blur_radius != 0.0f || sc->cull_boxes.empty() || sc->cull_boxes.size() > 8) return full;
    const float* m = cam.transform;
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(std::fabs(det) > 1e-300)) return full;
    const double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det, (f * g - d * i) / det, (a * i - c * g) / det,
                           (c * d - a * f) / det, (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
    const double k = std::tan(static_cast<double>(cam.fov_degrees) * 3.141592653589793 / 180.0);
    const double aspect = cam.aspect;
    CullRects out{};
    for (const auto& box : sc->cull_boxes) {
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        double pc[8][3];
        for (int corner = 0; corner < 8; ++corner) {
            const double p[3] = {(corner & 1 ? box[3] : box[0]) - static_cast<double>(m[9]), (corner & 2 ? box[4] : box[1]) - static_cast<double>(m[10]),
                                 (corner & 4 ? box[5] : box[2]) - static_cast<double>(m[11])};
            for (int r = 0; r < 3; ++r) pc[corner][r] = inv[3 * r] * p[0] + inv[3 * r + 1] * p[1] + inv[3 * r + 2] * p[2];
            if (!std::isfinite(pc[corner][0]) || !std::isfinite(pc[corner][1]) || !std::isfinite(pc[corner][2])) return full;
        }
        // Rays leave the plane Z = 0 towards +Z, so only the part of the box with Z >= 0 can be hit: project
        // the corners in front of the plane and the points where the box's edges cross it (there 1 + k Z >= 1).
        bool any = false;
        auto project = [&](double X, double Y, double Z) {
            const double w = 1.0 + k * Z;
            // screen point -> pixel: (u, v) = (vx, vy / aspect) for aspect > 1, else (vx * aspect, vy); vx = 2 x / (W - 1) - 1
            const double u = X / w, v = Y / w;
            const double vx = aspect > 1.0 ? u : u / aspect, vy = aspect > 1.0 ? v * aspect : v;
            const double px = (vx + 1.0) * 0.5 * (W - 1.0), py = (vy + 1.0) * 0.5 * (H - 1.0);
            x0 = std::min(x0, px);
            x1 = std::max(x1, px);
            y0 = std::min(y0, py);
            y1 = std::max(y1, py);
            any = true;
        };
        if (!(k >= 0.0)) return full;
        for (int corner = 0; corner < 8; ++corner) {
            if (pc[corner][2] >= 0.0) project(pc[corner][0], pc[corner][1], pc[corner][2]);
            for (int axis = 0; axis < 3; ++axis) {
                const int other = corner ^ (1 << axis);
                if (other < corner) continue;
                const double za = pc[corner][2], zb = pc[other][2];
                if ((za < 0.0) != (zb < 0.0)) {
                    const double t = za / (za - zb);
                    project(pc[corner][0] + t * (pc[other][0] - pc[corner][0]), pc[corner][1] + t * (pc[other][1] - pc[corner][1]), 0.0);
                }
            }
        }
        if (!any) continue;   // wholly behind the plane the rays start from
        const double lim = 1e9;
        const int rx0 = static_cast<int>(std::floor(std::max(x0, -lim))) - 2, rx1 = static_cast<int>(std::ceil(std::min(x1, lim))) + 2;
        const int ry0 = static_cast<int>(std::floor(std::max(y0, -lim))) - 2, ry1 = static_cast<int>(std::ceil(std::min(y1, lim))) + 2;
        if (rx1 < 0 || ry1 < 0 || rx0 > static_cast<int>(W) - 1 || ry0 > static_cast<int>(H) - 1) continue;   // off screen
        int* r = out.r[out.n++];
        r[0] = std::max(rx0, 0);
        r[1] = std::max(ry0, 0);
        r[2] = std::min(rx1, static_cast<int>(W) - 1);
        r[3] = std::min(ry1, static_cast<int>(H) - 1);
    }
    return out;
}

// Brackets one launch with an event pair when profiling is on.
struct StageTimer {
    mipt_ctx* c;
    int stage;
    size_t e0 = 0;
    bool on;
    static size_t next_event(mipt_ctx* c) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return size_t(-1);
            c->ev_pool.push_back(e);
        }
        return c->ev_used++;
    }
    StageTimer(mipt_ctx* ctx, int st) : c(ctx), stage(st), on(ctx->profiling) {
        if (!on) return;
        e0 = next_event(c);
        if (e0 == size_t(-1)) { on = false; return; }
        (void)hipEventRecord(c->ev_pool[e0], c->stream);
    }
    ~StageTimer() {
        if (!on) return;
        const size_t e1 = next_event(c);
        if (e1 == size_t(-1)) return;
        (void)hipEventRecord(c->ev_pool[e1], c->stream);
        c->spans.push_back({stage, e0, e1});
    }
};

// Runs body(i) for i in [0, n) on up to 16 threads (the user's virtuals are re-entrant, as the reference requires of them).
template <class F>
void host_parallel_for(uint32_t n, F&& body) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned threads = n < 256 ? 1u : std::min({16u, hw, n / 128u});
    if (threads <= 1) {
        for (uint32_t i = 0; i < n; ++i) body(i);
        return;
    }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            for (;;) {
                const uint32_t i0 = next.fetch_add(64);
                if (i0 >= n) return;
                for (uint32_t i = i0; i < std::min(n, i0 + 64u); ++i) body(i);
            }
        });
    for (auto& th : pool) th.join();
}

int ensure_host_stage(mipt_ctx* c, size_t bytes) {
    if (bytes <= c->host_stage_cap) return MIPT_OK;
    if (c->host_stage) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(c->host_stage));
        c->host_stage = nullptr;
        c->host_stage_cap = 0;
    }
    const size_t cap = std::max(bytes, size_t(1) << 20);
    const hipError_t e = hipMalloc(&c->host_stage, cap);
    if (e != hipSuccess) {
        c->last_hip = static_cast<int>(e);
        return e == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
    }
    c->host_stage_cap = cap;
    return MIPT_OK;
}

uint32_t ordered_float_bits(float t) {
    uint32_t b;
    std::memcpy(&b, &t, 4);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

// Closest hits on the objects that only the host can evaluate, for the live rays of `bounce` (after the device has
// found their closest hit on everything else): per object, the rays that reach its box come to the host, the
// object's hitRay runs on them, and hits that beat the ray's best are merged by key.  Synchronises the stream.
int host_objects_extend(mipt_ctx* c, const mipt_scene* sc, uint32_t bounce, Totals& host_totals) {
    for (uint32_t which = 0; which < sc->host_prims.size(); ++which) {
        const uint32_t prim = sc->host_prims[which];
        HIP_TRY(c, hipMemsetAsync(&c->pb.counters[bounce].n_host_rays, 0, sizeof(uint32_t), c->stream));
        launch_host_gather(c->stream, sc->dev, c->pb, bounce, which);
        uint32_t n = 0;
        HIP_TRY(c, hipMemcpyAsync(&n, &c->pb.counters[bounce].n_host_rays, sizeof(n), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (n == 0 || c->cancel.load(std::memory_order_relaxed)) continue;
        // staging: rays out (6 floats per ray); back: index, key, position per hit
        const size_t bytes = static_cast<size_t>(n) * (24 + 4 + 8 + 12) + 64;
        const int rc = ensure_host_stage(c, bytes);
        if (rc != MIPT_OK) return rc;
        char* base = static_cast<char*>(c->host_stage);
        float* d_rays = reinterpret_cast<float*>(base);
        unsigned long long* d_key = reinterpret_cast<unsigned long long*>(base + align_up(static_cast<size_t>(n) * 24, 8));
        float* d_pos = reinterpret_cast<float*>(reinterpret_cast<char*>(d_key) + static_cast<size_t>(n) * 8);
        uint32_t* d_idx = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d_pos) + static_cast<size_t>(n) * 12);
        launch_host_export_rays(c->stream, c->pb, n, d_rays);
        std::vector<float> rays(static_cast<size_t>(n) * 6);
        HIP_TRY(c, hipMemcpyAsync(rays.data(), d_rays, rays.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<uint8_t> hit(n, 0);
        std::vector<float> pos(static_cast<size_t>(n) * 3);
        std::vector<unsigned long long> key(n);
        host_parallel_for(n, [&](uint32_t i) {
            const float* o = &rays[6 * static_cast<size_t>(i)];
            const float* d = o + 3;
            float* hp = &pos[3 * static_cast<size_t>(i)];
            if (!sc->host_cb.hit(sc->host_cb.user, prim, o, d, hp)) return;
            // t = (hit - ray.pos) . ray.dir, in the reference's order (src/ObjectTree.cpp:43-44); this file is compiled without contraction
            const float dx = hp[0] - o[0], dy = hp[1] - o[1], dz = hp[2] - o[2];
            const float t = dx * d[0] + dy * d[1] + dz * d[2];
            key[i] = (static_cast<unsigned long long>(ordered_float_bits(t)) << 32) | prim;
            hit[i] = 1;
        });
        host_totals.host_rays += n;
        // compact the hits
        uint32_t m = 0;
        std::vector<uint32_t> idx(n);
        for (uint32_t i = 0; i < n; ++i)
            if (hit[i]) {
                idx[m] = i;
                key[m] = key[i];
                pos[3 * static_cast<size_t>(m)] = pos[3 * static_cast<size_t>(i)];
                pos[3 * static_cast<size_t>(m) + 1] = pos[3 * static_cast<size_t>(i) + 1];
                pos[3 * static_cast<size_t>(m) + 2] = pos[3 * static_cast<size_t>(i) + 2];
                ++m;
            }
        if (m == 0) continue;
        HIP_TRY(c, hipMemcpyAsync(d_key, key.data(), static_cast<size_t>(m) * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_pos, pos.data(), static_cast<size_t>(m) * 12, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_idx, idx.data(), static_cast<size_t>(m) * 4, hipMemcpyHostToDevice, c->stream));
        launch_host_merge(c->stream, c->pb, m, d_idx, d_key, d_pos);
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // the host vectors go out of scope
    }
    return MIPT_OK;
}

// Object::deflectRay on the host for the hits of `bounce` that lie on host-evaluated objects (queued by shade).
int host_objects_shade(mipt_ctx* c, const mipt_scene* sc, uint32_t bounce, uint32_t max_bounces, Totals& host_totals) {
    uint32_t n = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, &c->pb.counters[bounce].n_host_shade, sizeof(n), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n == 0 || c->cancel.load(std::memory_order_relaxed)) return MIPT_OK;
    const int rc = ensure_host_stage(c, static_cast<size_t>(n) * (24 + 4 + 36) + 64);
    if (rc != MIPT_OK) return rc;
    char* base = static_cast<char*>(c->host_stage);
    float* d_hits = reinterpret_cast<float*>(base);
    uint32_t* d_prim = reinterpret_cast<uint32_t*>(base + static_cast<size_t>(n) * 24);
    float* d_b9 = reinterpret_cast<float*>(base + static_cast<size_t>(n) * 28);
    launch_host_export_hits(c->stream, c->pb, n, d_hits, d_prim);
    std::vector<float> hits(static_cast<size_t>(n) * 6), b9(static_cast<size_t>(n) * 9);
    std::vector<uint32_t> prims(n);
    HIP_TRY(c, hipMemcpyAsync(hits.data(), d_hits, hits.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(prims.data(), d_prim, prims.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    host_parallel_for(n, [&](uint32_t i) {
        const float* p = &hits[6 * static_cast<size_t>(i)];
        sc->host_cb.deflect(sc->host_cb.user, prims[i], p, p + 3, &b9[9 * static_cast<size_t>(i)]);
    });
    host_totals.host_shades += n;
    HIP_TRY(c, hipMemcpyAsync(d_b9, b9.data(), b9.size() * 4, hipMemcpyHostToDevice, c->stream));
    launch_host_apply(c->stream, c->pb, n, d_b9, bounce, max_bounces);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIPT_OK;
}

// The bounce loop of one sample batch over the live-path queue of bounce 0.  `first_done` says how far
// the caller already took bounce 0: 0 = nothing, 1 = closest analytic hit found, 2 = SDF march done too,
// 3 = the host-evaluated objects have been asked as well.
// Leaves per-path radiance in pb.rad (slots that never hit anything keep what the caller put there).
// Scenes with host-evaluated objects synchronise the stream twice per bounce; all others only enqueue.
int enqueue_bounces(mipt_ctx* c, const mipt_scene* sc, const BatchParams& bp, uint32_t bounces, int first_done, Totals* host_totals = nullptr) {
    Totals local{};
    Totals& ht = host_totals ? *host_totals : local;
    const bool host = !sc->host_prims.empty();
    // Paths whose SDF march was handed on run behind the others; the rounds after the last bounce are theirs
    // (every kernel returns at once on an empty queue).
    const uint32_t rounds = bounces + bp.carry_max_lag;
    for (uint32_t b = 0; b < rounds; ++b) {
        const int done = b == 0 ? first_done : 0;
        if (b > 0) launch_latch_cancel(c->stream, c->pb);
        if (done < 1) { StageTimer t(c, MIPT_STAGE_EXTEND); launch_extend(c->stream, sc->dev, bp, c->pb, b, std::max(sc->stack_need, c->min_stack), c->counting); }
        if (sc->n_sdf && done < 2) { StageTimer t(c, MIPT_STAGE_SDF_MARCH); launch_sdf_march(c->stream, sc->dev, bp, c->pb, b); }
        if (host && done < 3) {
            const int rc = host_objects_extend(c, sc, b, ht);
            if (rc != MIPT_OK) return rc;
        }
        { StageTimer t(c, MIPT_STAGE_SHADE); launch_shade(c->stream, sc->dev, bp, c->pb, b, bounces); }
        if (sc->n_sdf) { StageTimer t(c, MIPT_STAGE_SHADE_SDF); launch_shade_sdf(c->stream, sc->dev, bp, c->pb, b, bounces); }
        if (host) {
            const int rc = host_objects_shade(c, sc, b, bounces, ht);
            if (rc != MIPT_OK) return rc;
        }
    }
    // every real path casts a primary ray (tile padding does not)
    const uint32_t primaries = bp.tiles_x ? bp.width * bp.height * (bp.n_slots / bp.n_pix) : bp.n_slots;
    launch_finish_batch(c->stream, c->pb, rounds, primaries, bp.carry_cap);
    return MIPT_OK;
}

// PathBuffers::host_queue for scenes with host-evaluated objects: as many entries as the arena has slots.
int ensure_host_queue(mipt_ctx* c, const mipt_scene* sc) {
    if (sc->host_prims.empty()) {
        c->pb.host_queue = c->host_q;
        return MIPT_OK;
    }
    if (c->host_q_cap < c->slots_cap) {
        if (c->host_q) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipFree(c->host_q));
            c->host_q = nullptr;
            c->host_q_cap = 0;
        }
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->host_q), c->slots_cap * sizeof(uint32_t));
        if (e != hipSuccess) {
            c->last_hip = static_cast<int>(e);
            return e == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
        }
        c->host_q_cap = c->slots_cap;
    }
    c->pb.host_queue = c->host_q;
    return MIPT_OK;
}

// Image tiles (8 x 8 pixels) that touch one of the cull rectangles, in row-major order; empty = all of them.
void visible_tiles(const CullRects& cull, uint32_t tiles_x, uint32_t tiles_y, std::vector<uint32_t>& out) {
    out.clear();
    std::vector<uint8_t> mark(static_cast<size_t>(tiles_x) * tiles_y, 0);
    size_t n = 0;
    for (int k = 0; k < cull.n; ++k) {
        const int tx0 = std::max(cull.r[k][0], 0) / 8, ty0 = std::max(cull.r[k][1], 0) / 8;
        const int tx1 = std::min(cull.r[k][2] / 8, static_cast<int>(tiles_x) - 1), ty1 = std::min(cull.r[k][3] / 8, static_cast<int>(tiles_y) - 1);
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) {
                uint8_t& m = mark[static_cast<size_t>(ty) * tiles_x + tx];
                n += m ? 0 : 1;
                m = 1;
            }
    }
    if (n == mark.size()) return;   // the whole frame: no table needed
    out.reserve(std::max<size_t>(n, 1));
    for (size_t t = 0; t < mark.size(); ++t)
        if (mark[t]) out.push_back(static_cast<uint32_t>(t));
    if (out.empty()) out.push_back(0u);   // nothing on screen: one tile keeps every launch well-formed (its pixels trace nothing)
}

// `samples_done` (nullable) receives the samples per pixel that entered d_sum: st->spp, or fewer after a cancel.
int render_into(mipt_ctx* c, const mipt_scene* sc, const mipt_camera* cam, const mipt_settings* st, float* d_sum, mipt_stats* stats,
                uint32_t* samples_done = nullptr) {
    if (samples_done) *samples_done = 0;
    if (!c || !sc || !cam || !st || !d_sum) return MIPT_ERR_ARGUMENT;
    if (st->width < 2 || st->height < 2 || st->spp == 0 || st->bounces == 0) return MIPT_ERR_ARGUMENT;
    if (st->bounces > static_cast<uint32_t>(kMaxBounces)) return MIPT_ERR_TOO_LARGE;
    // pixel ids cover whole 8 x 8 tiles (BatchParams::tiles_x)
    const uint32_t tiles_x = (st->width + 7u) / 8u, tiles_y = (st->height + 7u) / 8u;
    if (static_cast<size_t>(tiles_x) * tiles_y * 64 > (size_t(1) << 31)) return MIPT_ERR_TOO_LARGE;
    HIP_TRY(c, hipSetDevice(c->device));
    c->cancel.store(0);
    if (c->cancel_host) *c->cancel_host = 0u;
    const CameraRec cr = pack_camera(*cam);
    const CullRects cull = screen_rects(sc, *cam, st->width, st->height);
    // path slots only for the tiles that can see anything
    visible_tiles(cull, tiles_x, tiles_y, c->vis_host);
    const bool listed = !c->vis_host.empty();
    const size_t n_pix = (listed ? c->vis_host.size() : static_cast<size_t>(tiles_x) * tiles_y) * 64;
    if (listed) {
        if (c->vis_host.size() > c->vis_cap) {
            if (c->vis_tiles) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                HIP_TRY(c, hipFree(c->vis_tiles));
                c->vis_tiles = nullptr;
                c->vis_cap = 0;
            }
            const size_t cap = std::max<size_t>(c->vis_host.size(), 4096);
            const hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->vis_tiles), cap * sizeof(uint32_t));
            if (e != hipSuccess) {
                c->last_hip = static_cast<int>(e);
                return e == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
            }
            c->vis_cap = cap;
        }
        // (a copy from pageable memory returns once the source has been staged; the table is reused by the next frame only after that)
        HIP_TRY(c, hipMemcpyAsync(c->vis_tiles, c->vis_host.data(), c->vis_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    // equal batches, as few as the memory budget allows; on an allocation failure try smaller ones
    size_t per_batch_max = std::max<size_t>(1, batch_path_budget(c) / n_pix);
    uint32_t per_batch = 0;
    for (;;) {
        const size_t n_batches = (st->spp + per_batch_max - 1) / per_batch_max;
        per_batch = static_cast<uint32_t>((st->spp + n_batches - 1) / n_batches);
        if (n_pix * per_batch > kMaxSlots) return MIPT_ERR_TOO_LARGE;
        const int rc = ensure_slots(c, n_pix * per_batch, static_cast<size_t>(per_batch) * ((n_pix + 63) / 64));
        if (rc == MIPT_OK) break;
        if (rc != MIPT_ERR_OUT_OF_MEMORY || per_batch == 1) return rc;
        (void)hipGetLastError();
        per_batch_max = std::max<size_t>(1, per_batch / 2);
    }
    if (const int rq = ensure_host_queue(c, sc); rq != MIPT_OK) return rq;
    Totals host_totals{};
    HIP_TRY(c, hipMemsetAsync(c->pb.totals, 0, sizeof(Totals), c->stream));
    c->ev_used = 0;
    c->spans.clear();
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    uint32_t done = 0, batch_no = 0;
    int status = MIPT_OK;
    while (done < st->spp) {
        // at most two batches are queued at a time: the stream never runs dry between batches, and a cancel
        // arrives at most two batches late (every kernel of a cancelled frame returns at once anyway)
        if (batch_no >= 2) HIP_TRY(c, hipEventSynchronize(c->batch_ev[batch_no & 1]));
        if (c->cancel.load(std::memory_order_relaxed)) {
            status = MIPT_ERR_CANCELLED;
            break;
        }
        const uint32_t now = std::min(per_batch, st->spp - done);
        BatchParams bp{};
        bp.n_pix = static_cast<uint32_t>(n_pix);
        bp.n_slots = static_cast<uint32_t>(n_pix * now);
        bp.width = st->width;
        bp.height = st->height;
        bp.tiles_x = tiles_x;
        bp.vis_tiles = listed ? c->vis_tiles : nullptr;
        bp.first_sample = st->first_sample + done;
        bp.flags = (st->flags & static_cast<uint32_t>(MIPT_FLAG_NO_JITTER)) | (c->full_march ? kFlagFullMarch : 0u);
        bp.lane_threshold = c->lane_threshold;
    bp.no_few = c->no_few ? 1u : 0u;
    bp.sort_marches = sc->host_prims.empty() ? c->sort_marches : 0u;   // (the host pass works on the queues between the stages)
        // marches are handed on only where nothing synchronises with the host inside a batch
        bp.carry_max_lag = (sc->n_sdf && sc->host_prims.empty()) ? std::min<uint32_t>(c->carry_lag, kCounterRows - 1 - st->bounces) : 0u;
        bp.carry_after = c->carry_after;
        bp.carry_cap = c->carry_cap;
    bp.trav_quorum = c->trav_quorum;
    bp.trav_quorum_primary = c->trav_quorum_primary;
    bp.trav_keep = c->trav_keep;
    bp.trav_quorum_pair = c->trav_quorum_pair;
    bp.trav_keep_pair = c->trav_keep_pair;
    bp.pair_plain_entries = c->pair_plain ? 1u : 0u;
    bp.pair_rows = c->pair_tight ? static_cast<uint32_t>(std::max(sc->stack_need, c->min_stack)) + 3u : 0u;
        bp.seed = st->seed;
        HIP_TRY(c, hipMemsetAsync(c->pb.counters, 0, sizeof(BounceCounters) * kCounterRows, c->stream));
        launch_latch_cancel(c->stream, c->pb);
        // Camera rays: traced by the kernel that makes them (nothing is stored for a ray that hits nothing: the demo scene,
        // where 97 % of the paths end there), or -- scenes of a great many small analytic primitives -- written out and
        // traced by the two-rays-per-lane kernel of the later bounces (measured: 100,000 objects 367.6 -> 337.2 ms per
        // 128-spp frame, 1,000,000 objects 244.3 -> 230.3 at 16 spp in 3840 x 2160; 20,000 larger objects, whose camera
        // rays keep 50 of 64 lanes busy in the fused kernel, gain nothing).
        const bool split_primary = c->split_primary >= 0 ? c->split_primary != 0 : (sc->n_sdf == 0 && sc->host_prims.empty() && sc->n_analytic >= 50000u);
        if (split_primary) {
            { StageTimer t(c, MIPT_STAGE_GENERATE); launch_generate(c->stream, cr, bp, c->pb, cull); }
        } else {
            StageTimer t(c, MIPT_STAGE_GENERATE);
            launch_generate_extend(c->stream, cr, sc->dev, bp, c->pb, cull, std::max(sc->stack_need, c->min_stack), c->counting);
        }
        if (const int rb = enqueue_bounces(c, sc, bp, st->bounces, split_primary ? 0 : 1, &host_totals); rb != MIPT_OK) return rb;
        launch_latch_cancel(c->stream, c->pb);
        { StageTimer t(c, MIPT_STAGE_ACCUMULATE); launch_accumulate(c->stream, bp, c->pb, d_sum); }
        HIP_TRY(c, hipEventRecord(c->batch_ev[batch_no & 1], c->stream));
        done += now;
        ++batch_no;
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    if (c->cancel.load(std::memory_order_relaxed)) status = MIPT_ERR_CANCELLED;   // possibly in the middle of the last batch
    Totals t{};
    HIP_TRY(c, hipMemcpy(&t, c->pb.totals, sizeof(t), hipMemcpyDeviceToHost));
    if (samples_done) *samples_done = static_cast<uint32_t>(t.samples_done);
    if (stats) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        stats->paths = static_cast<uint64_t>(st->width) * st->height * done;
        stats->rays = t.rays;
        stats->sdf_rays = t.sdf_rays;
        stats->sdf_shades = t.sdf_shades;
        stats->sdf_far_evals = t.sdf_far_evals;
        stats->sdf_near_evals = t.sdf_near_evals;
        stats->sdf_noise_calls = t.sdf_noise_calls;
        stats->shade_evals = t.shade_near_evals;
        stats->shade_noise_calls = t.shade_noise_calls;
        stats->primary_traced = t.primary_traced;
        stats->samples_done = t.samples_done;
        stats->node_visits = t.node_visits;
        stats->prim_tests = t.prim_tests;
        stats->trav_rounds = t.trav_rounds;
        stats->trav_leaf_steps = t.trav_leaf_steps;
        stats->trav_restarts = t.trav_restarts;
        stats->marches_carried = t.marches_carried;
        stats->host_rays = host_totals.host_rays;
        stats->host_shades = host_totals.host_shades;
        stats->ms_device = ms;
        for (int k = 0; k < 6; ++k) {
            stats->stage_ms[k] = 0.0;
            stats->stage_launches[k] = 0;
        }
        for (const auto& sp : c->spans) {
            float d = 0.0f;
            if (hipEventElapsedTime(&d, c->ev_pool[sp.e0], c->ev_pool[sp.e1]) == hipSuccess) {
                stats->stage_ms[sp.stage] += d;
                stats->stage_launches[sp.stage] += 1;
            }
        }
    }
    return status;
}

}   // namespace

extern "C" {

int mipt_init(int device, void* stream, mipt_ctx** out) {
    if (!out) return MIPT_ERR_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return MIPT_ERR_NO_DEVICE;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return MIPT_ERR_NO_DEVICE;
    }
    if (device >= count) return MIPT_ERR_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return MIPT_ERR_NO_DEVICE;
    auto* c = new (std::nothrow) mipt_ctx;
    if (!c) return MIPT_ERR_OUT_OF_MEMORY;
    c->device = device;
    if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return MIPT_ERR_NO_DEVICE;
        }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->batch_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->batch_ev[1], hipEventDisableTiming) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->tm_scratch), reinhard_scratch_floats() * sizeof(float)) != hipSuccess) {
        mipt_shutdown(c);
        return MIPT_ERR_NO_DEVICE;
    }
    // this device's copy of the file-scope scratch array (the device is current)
    if (const hipError_t se = reinhard_static_scratch(&c->tm_static); se != hipSuccess || !c->tm_static) {
        mipt_shutdown(c);
        return MIPT_ERR_NO_DEVICE;
    }
    {
        void* host = nullptr;
        void* dev = nullptr;
        if (hipHostMalloc(&host, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) {
            if (host) (void)hipHostFree(host);
            mipt_shutdown(c);
            return MIPT_ERR_NO_DEVICE;
        }
        c->cancel_host = static_cast<volatile uint32_t*>(host);
        *c->cancel_host = 0u;
        c->cancel_dev = static_cast<const volatile uint32_t*>(dev);
    }
    if (const char* e = std::getenv("MIPT_FULL_MARCH")) c->full_march = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_CARRY_LAG")) c->carry_lag = static_cast<uint32_t>(std::min(long(kMaxCarryLag), std::max(0L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_NO_FEW")) c->no_few = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_SORT_MARCHES")) c->sort_marches = static_cast<uint32_t>(std::min(64, std::max(0, std::atoi(e))));
    if (const char* e = std::getenv("MIPT_CELL_TABLE_BALLS")) c->minor_table_balls = static_cast<size_t>(std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("MIPT_SPLIT_PRIMARY")) c->split_primary = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("MIPT_CARRY_AFTER")) c->carry_after = static_cast<uint32_t>(std::min(100000L, std::max(0L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_DEVICE_BVH")) c->device_bvh = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_LEAF_QUORUM")) c->trav_quorum = static_cast<uint32_t>(std::min(64L, std::max(1L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_LEAF_QUORUM_PRIMARY")) c->trav_quorum_primary = static_cast<uint32_t>(std::min(64L, std::max(1L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_MIN_STACK")) c->min_stack = static_cast<int>(std::min(long(kMaxTraversalStack), std::max(0L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_PAIR_QUORUM")) c->trav_quorum_pair = static_cast<uint32_t>(std::min(64L, std::max(0L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_PAIR_PLAIN")) c->pair_plain = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_PAIR_TIGHT")) c->pair_tight = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_PAIR_KEEP")) c->trav_keep_pair = static_cast<uint32_t>(std::min(128L, std::max(0L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_KEEP_BUSY")) c->trav_keep = static_cast<uint32_t>(std::min(64L, std::max(0L, std::atol(e))));
    if (const char* e = std::getenv("MIPT_LANE_THRESHOLD")) {
        const long v = std::atol(e);
        if (v >= 1 && v <= 65) c->lane_threshold = static_cast<uint32_t>(v);
    }
    *out = c;
    return MIPT_OK;
}

void mipt_shutdown(mipt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->arena) (void)hipFree(c->arena);
    if (c->vis_tiles) (void)hipFree(c->vis_tiles);
    if (c->host_q) (void)hipFree(c->host_q);
    if (c->host_stage) (void)hipFree(c->host_stage);
    for (hipEvent_t e : c->batch_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->frame) (void)hipFree(c->frame);
    if (c->stage) (void)hipFree(c->stage);
    if (c->tm_scratch) (void)hipFree(c->tm_scratch);
    if (c->cancel_host) (void)hipHostFree(const_cast<uint32_t*>(c->cancel_host));
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mipt_cancel(mipt_ctx* c) {
    if (!c) return MIPT_ERR_ARGUMENT;
    c->cancel.store(1, std::memory_order_relaxed);
    if (c->cancel_host) *c->cancel_host = 1u;
    return MIPT_OK;
}

int mipt_last_hip_error(const mipt_ctx* c) { return c ? c->last_hip : 0; }

int mipt_trim(mipt_ctx* c) {
    if (!c) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (void** p : {&c->arena, reinterpret_cast<void**>(&c->frame), reinterpret_cast<void**>(&c->stage), reinterpret_cast<void**>(&c->vis_tiles),
                     reinterpret_cast<void**>(&c->host_q), &c->host_stage})
        if (*p) {
            HIP_TRY(c, hipFree(*p));
            *p = nullptr;
        }
    c->slots_cap = c->bits_cap = c->frame_cap = c->stage_cap = c->vis_cap = c->host_q_cap = c->host_stage_cap = 0;
    c->pb.host_queue = nullptr;
    return MIPT_OK;
}

int mipt_set_profiling(mipt_ctx* c, int on) {
    if (!c) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    c->profiling = on != 0;
    return MIPT_OK;
}

int mipt_set_counting(mipt_ctx* c, int on) {
    if (!c) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    c->counting = on != 0;
    return MIPT_OK;
}

const char* mipt_strerror(int s) {
    switch (s) {
        case MIPT_OK: return "ok";
        case MIPT_ERR_ARGUMENT: return "invalid argument";
        case MIPT_ERR_NO_DEVICE: return "no usable HIP device";
        case MIPT_ERR_HIP: return "HIP call failed";
        case MIPT_ERR_NOT_INVERTIBLE: return "transformation determinant below 1e-6";
        case MIPT_ERR_CANCELLED: return "render cancelled";
        case MIPT_ERR_TOO_LARGE: return "size beyond the supported limit";
        case MIPT_ERR_OUT_OF_MEMORY: return "out of memory";
    }
    return "unknown status";
}

int mipt_scene_upload(mipt_ctx* c, const mipt_scene_desc* d, mipt_scene** out) {
    return mipt_scene_upload_host(c, d, c && c->device_bvh ? MIPT_UPLOAD_DEVICE_BVH : 0u, nullptr, out);
}

int mipt_scene_upload_ex(mipt_ctx* c, const mipt_scene_desc* d, uint32_t flags, mipt_scene** out) { return mipt_scene_upload_host(c, d, flags, nullptr, out); }

int mipt_scene_upload_host(mipt_ctx* c, const mipt_scene_desc* d, uint32_t flags, const mipt_host_callbacks* cb, mipt_scene** out) {
    if (!c || !d || !out || (flags & ~static_cast<uint32_t>(MIPT_UPLOAD_DEVICE_BVH))) return MIPT_ERR_ARGUMENT;
    if (cb && (!cb->hit || !cb->deflect)) return MIPT_ERR_ARGUMENT;
    *out = nullptr;
    if (d->n_prims == 0 || !d->kind || !d->fwd || !d->inv || !d->params || !d->material || !d->materials || d->n_materials == 0)
        return MIPT_ERR_ARGUMENT;
    if (d->n_prims > (1u << 30)) return MIPT_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lock(c->mutex);   // the crater tables are computed on the context's stream
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t n = d->n_prims;
    std::vector<PrimRec> prims(n);
    std::vector<float> tri_c;
    std::vector<BuildBox> boxes;
    std::vector<SdfBox> sdf, host_boxes;
    boxes.reserve(n);
    bool any_tri = false, any_sdf = false /* crater balls */, any_plain = false /* SDF kinds evaluated one point per lane */;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t k = d->kind[i];
        if (k > MIPT_HOST_OBJECT || (k == MIPT_HOST_OBJECT && !cb) || d->material[i] >= d->n_materials) return MIPT_ERR_ARGUMENT;
        any_tri |= (k == MIPT_TRIANGLE);
        any_sdf |= (k == MIPT_CRATERBALL);
        any_plain |= (k >= MIPT_ROUGHBOX && k <= MIPT_WEIRDWALL);
    }
    if ((any_sdf || any_plain) && !d->noise_transform) return MIPT_ERR_ARGUMENT;
    if (any_tri) tri_c.assign(static_cast<size_t>(3) * n, 0.0f);
    for (uint32_t i = 0; i < n; ++i) {
        PrimRec& r = prims[i];
        std::memcpy(r.inv, d->inv + 12 * static_cast<size_t>(i), sizeof(r.inv));
        std::memcpy(r.fwd, d->fwd + 12 * static_cast<size_t>(i), sizeof(r.fwd));
        const float* p = d->params + 9 * static_cast<size_t>(i);
        std::memcpy(r.p, p, sizeof(r.p));
        std::memcpy(r.q, p + 3, sizeof(r.q));
        r.kind = d->kind[i];
        r.material = d->material[i];
        if (r.kind == MIPT_TRIANGLE) std::memcpy(&tri_c[3 * static_cast<size_t>(i)], p + 6, sizeof(float) * 3);
        float lo[3], hi[3];
        if (!world_bounds(r.kind, r.fwd, p, lo, hi)) return MIPT_ERR_ARGUMENT;
        if (r.kind == MIPT_ROUGHBOX) {
            // the foam's lattice coordinates (position / 0.03723) must stay exact 24-bit integers
            for (int a = 0; a < 3; ++a)
                if (!(p[a] >= 0.0f && p[a] < 7.0e4f)) return MIPT_ERR_ARGUMENT;
        }
        if (r.kind == MIPT_FRACTAL) {
            // FractalObject marches from where its LOCAL ray enters its WORLD bounding box
            // (src/Object.cpp:161): keep that box as AxisAlignedBox(Pos, Pos) forms it (src/Geometry.cpp:348-358)
            for (int a = 0; a < 3; ++a) {
                r.p[a] = lo[a] + 0.5f * (hi[a] - lo[a]);
                r.q[a] = 0.5f * std::fabs(lo[a] - hi[a]);
            }
        }
        // The reference's hits lie inside its padded boxes (1e-3 in local units); widen a little more
        // so that rounding in the hit position or in the slab test can never cull a hit it would find.
        for (int a = 0; a < 3; ++a) {
            const float m = std::max(std::fabs(lo[a]), std::fabs(hi[a]));
            const float pad = 1e-5f * m + 1e-6f;
            lo[a] -= pad;
            hi[a] += pad;
        }
        if (r.kind == MIPT_HOST_OBJECT) {
            SdfBox s{};
            std::memcpy(s.lo, lo, sizeof(lo));
            std::memcpy(s.hi, hi, sizeof(hi));
            s.prim = i;
            host_boxes.push_back(s);
        } else if (r.kind >= MIPT_CRATERBALL) {
            SdfBox s{};
            std::memcpy(s.lo, lo, sizeof(lo));
            std::memcpy(s.hi, hi, sizeof(hi));
            s.prim = i;
            sdf.push_back(s);
        } else {
            BuildBox b;
            std::memcpy(b.lo, lo, sizeof(lo));
            std::memcpy(b.hi, hi, sizeof(hi));
            b.prim = i;
            boxes.push_back(b);
        }
    }
    std::vector<MaterialRec> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const float* m = d->materials + 14 * static_cast<size_t>(i);
        MaterialRec& r = mats[i];
        r.diffuse = m[0];
        r.specular = m[1];
        r.sharpness = m[2];
        std::memcpy(r.albedo, m + 3, 12);
        std::memcpy(r.emitted, m + 6, 12);
        r.transmittance = m[9];
        r.ior = m[10];
        std::memcpy(r.internal, m + 11, 12);
        r.pad[0] = r.pad[1] = 0.0f;
    }
    const uint32_t n_analytic = static_cast<uint32_t>(boxes.size());
    std::vector<std::array<float, 6>> analytic_boxes;
    if (!boxes.empty()) {
        std::array<float, 6> u{boxes[0].lo[0], boxes[0].lo[1], boxes[0].lo[2], boxes[0].hi[0], boxes[0].hi[1], boxes[0].hi[2]};
        for (const BuildBox& b : boxes)
            for (int a = 0; a < 3; ++a) {
                u[a] = std::min(u[a], b.lo[a]);
                u[3 + a] = std::max(u[3 + a], b.hi[a]);
            }
        analytic_boxes.push_back(u);
    }
    BvhResult bvh;
    if ((flags & MIPT_UPLOAD_DEVICE_BVH) && n_analytic >= 2) {
        // the two-child tree is built on the device, and comes back for the four-child form to be made of it
        void* d_lbvh = nullptr;
        hipError_t be = build_lbvh_device(c->stream, boxes.data(), n_analytic, analytic_boxes[0].data(), analytic_boxes[0].data() + 3, &d_lbvh, &bvh.depth);
        if (be == hipSuccess) {
            bvh.nodes.resize(n_analytic - 1u);
            be = hipMemcpy(bvh.nodes.data(), d_lbvh, bvh.nodes.size() * sizeof(BvhNode), hipMemcpyDeviceToHost);
            (void)hipFree(d_lbvh);
        }
        if (be != hipSuccess) {
            c->last_hip = static_cast<int>(be);
            return be == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
        }
    } else {
        bvh = build_bvh(std::move(boxes));
    }
    const uint32_t n_nodes = static_cast<uint32_t>(bvh.nodes.size());
    // Before anything walks the tree on the host (the collapse recurses once per level, and a tree that came back from the
    // device is not trusted blindly): a sane depth, and links that stay inside the arrays.
    if (bvh.depth < 0 || bvh.depth > 60) return MIPT_ERR_TOO_LARGE;
    for (const BvhNode& nd : bvh.nodes)
        for (const int32_t link : {nd.child0, nd.child1})
            if (link >= 0 ? static_cast<uint32_t>(link) >= n_nodes : static_cast<uint32_t>(~link) >= n) return MIPT_ERR_HIP;
    Bvh4Result bvh4 = collapse_bvh4(bvh.nodes.data(), bvh.nodes.size(), n);
    // PathBuffers::march_start keeps the index of an SDF box in 20 bits
    if (sdf.size() >= (size_t(1) << 20)) return MIPT_ERR_TOO_LARGE;
    // the deepest traversal stack the kernels carry; only a great many coincident boxes get there
    if (bvh4.stack_need > kMaxTraversalStack) return MIPT_ERR_TOO_LARGE;
    // the traversal addresses a node by a 32-bit byte offset (index * 128)
    if (bvh4.nodes.size() > (size_t(1) << 25)) return MIPT_ERR_TOO_LARGE;

    // Big craters of each crater ball (for_each_nearby_scattered_point at scale 2, PathTracerDemo.cpp:375-409):
    // which lattice cells pass the shell test depends on the ball's offset alone.  The device evaluates the
    // cells of a cube that covers the shell (|cell - offset| <= 2 + 0.3 + sqrt(0.1) < 3), with the very
    // arithmetic the kernels use; the cells that pass are kept in lattice order.
    std::vector<float> craters;
    std::vector<unsigned long long> crater_masks;
    std::vector<float> ball_offsets;   // CraterBallObject::offset of every ball, by ordinal
    if (any_sdf) {
        constexpr int kCube = 9;
        constexpr size_t kCells = size_t(kCube) * kCube * kCube;
        float* d_cells = nullptr;
        if (const hipError_t me = hipMalloc(reinterpret_cast<void**>(&d_cells), kCells * 12 * sizeof(float)); me != hipSuccess) {
            c->last_hip = static_cast<int>(me);
            return MIPT_ERR_HIP;
        }
        std::vector<float> cells(kCells * 12);
        for (const SdfBox& sb : sdf) {
            PrimRec& r = prims[sb.prim];
            if (r.kind != MIPT_CRATERBALL) continue;
            int base[3];
            for (int a = 0; a < 3; ++a) {
                if (!(std::fabs(r.p[a]) < 4.0e6f)) {   // lattice coordinates must stay exact 24-bit integers
                    (void)hipFree(d_cells);
                            return MIPT_ERR_ARGUMENT;
                }
                base[a] = static_cast<int>(std::floor(r.p[a])) - 4;
            }
            launch_crater_cells(c->stream, r.p, base, kCube, d_cells);
            hipError_t ce = hipMemcpyAsync(cells.data(), d_cells, cells.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            if (ce == hipSuccess) ce = hipStreamSynchronize(c->stream);
            if (ce != hipSuccess) {
                (void)hipFree(d_cells);
                    c->last_hip = static_cast<int>(ce);
                return MIPT_ERR_HIP;
            }
            const uint32_t first = static_cast<uint32_t>(craters.size() / 12);
            for (size_t i = 0; i < kCells; ++i)
                if (cells[12 * i + 10] != 0.0f) craters.insert(craters.end(), cells.begin() + 12 * i, cells.begin() + 12 * i + 12);
            const uint32_t count = static_cast<uint32_t>(craters.size() / 12) - first;
            if (count > 64) {   // one crater per lane; the shell holds ~32 cells for any offset (23..39 seen)
                (void)hipFree(d_cells);
                    return MIPT_ERR_TOO_LARGE;
            }
            // cell -> crater mask: crater c is visited from a query in cell f iff cell_c - f is -1, 0, 1 or 2 on every
            // axis (for_each_nearby_scattered_point with radius 2, PathTracerDemo.cpp:87-93)
            const uint32_t ordinal = static_cast<uint32_t>(crater_masks.size() / kCells);
            crater_masks.resize(crater_masks.size() + kCells, 0ull);
            unsigned long long* masks = crater_masks.data() + static_cast<size_t>(ordinal) * kCells;
            for (uint32_t cidx = 0; cidx < count; ++cidx) {
                const float* rec = &craters[12 * static_cast<size_t>(first + cidx)];
                const int cell[3] = {static_cast<int>(rec[0]), static_cast<int>(rec[1]), static_cast<int>(rec[2])};
                for (int dx = -1; dx <= 2; ++dx)
                    for (int dy = -1; dy <= 2; ++dy)
                        for (int dz = -1; dz <= 2; ++dz) {
                            const int f[3] = {cell[0] - dx - base[0], cell[1] - dy - base[1], cell[2] - dz - base[2]};
                            if (f[0] < 0 || f[0] >= kCube || f[1] < 0 || f[1] >= kCube || f[2] < 0 || f[2] >= kCube) continue;
                            masks[(static_cast<size_t>(f[0]) * kCube + f[1]) * kCube + f[2]] |= 1ull << cidx;
                        }
            }
            const uint32_t packed = count | (ordinal << 8);
            ball_offsets.insert(ball_offsets.end(), r.p, r.p + 3);
            std::memcpy(&r.q[0], &first, 4);
            std::memcpy(&r.q[1], &packed, 4);
        }
        (void)hipFree(d_cells);
    }

    auto* s = new (std::nothrow) mipt_scene;
    if (!s) return MIPT_ERR_OUT_OF_MEMORY;
    s->device = c->device;
    auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        hipError_t e = hipMalloc(dst, bytes);
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    hipError_t e = upload(&s->d_prims, prims.data(), prims.size() * sizeof(PrimRec));
    if (e == hipSuccess) e = upload(&s->d_tri, tri_c.data(), tri_c.size() * sizeof(float));
    if (e == hipSuccess) e = upload(&s->d_mats, mats.data(), mats.size() * sizeof(MaterialRec));
    std::vector<Bvh4QNode> qnodes(bvh4.nodes.size());   // the 64-byte form the kernels read
    host_parallel_for(static_cast<uint32_t>(qnodes.size()), [&](uint32_t i) { quantize_bvh4_node(bvh4.nodes[i], qnodes[i]); });
    if (const char* nc = std::getenv("MIPT_NO_CULL"); nc && std::atoi(nc) != 0) {
        // test aid: every box of the tree becomes all of space, so that every ray visits every node and tests every
        // primitive -- the reference's own rule (src/ObjectTree.cpp:22-88 never culls) through the same kernels; a frame
        // must not depend on it
        for (Bvh4QNode& n : qnodes)
            for (int a = 0; a < 3; ++a) {
                n.org[a] = -1e18f;
                n.scale[a] = 1e16f;
                n.qlo[a] = 0u;
                n.qhi[a] = 0xffffffffu;
            }
        for (size_t i = 0; i + 7 < bvh4.leaf_boxes.size(); i += 8)
            for (int a = 0; a < 3; ++a) {
                bvh4.leaf_boxes[i + a] = -1e18f;
                bvh4.leaf_boxes[i + 4 + a] = 1e18f;
            }
    }
    if (e == hipSuccess) e = upload(&s->d_nodes, qnodes.data(), qnodes.size() * sizeof(Bvh4QNode));
    if (e == hipSuccess && !bvh4.nodes.empty()) e = upload(&s->d_leaf_boxes, bvh4.leaf_boxes.data(), bvh4.leaf_boxes.size() * sizeof(float));
    if (e == hipSuccess) e = upload(&s->d_sdf, sdf.data(), sdf.size() * sizeof(SdfBox));
    if (e == hipSuccess) e = upload(&s->d_host_boxes, host_boxes.data(), host_boxes.size() * sizeof(SdfBox));
    // noise() (PathTracerDemo.cpp:56-70) applies the transform T = (L, t), samples, scales by 1.1 and
    // repeats; octave i therefore samples q_i = A_i p + b_i with A_0 = L, b_0 = t, A_i = 1.1 L A_{i-1},
    // b_i = 1.1 L b_{i-1} + t.  Composed in double, stored as float (13 per octave: A row-major, b + 0.25,
    // weight), so the 8 octaves of one call can be evaluated side by side.
    float oct[8 * 13 + 12] = {};   // the raw transform follows the octave table
    if (any_sdf || any_plain) {
        const float* nt = d->noise_transform;
        std::memcpy(oct + 8 * 13, nt, 12 * sizeof(float));
        double A[9], bb[3];
        for (int i = 0; i < 9; ++i) A[i] = nt[i];
        for (int i = 0; i < 3; ++i) bb[i] = nt[9 + i];
        float kf = 0.02f;
        for (int o = 0; o < 8; ++o) {
            for (int i = 0; i < 9; ++i) oct[13 * o + i] = static_cast<float>(A[i]);
            for (int i = 0; i < 3; ++i) oct[13 * o + 9 + i] = static_cast<float>(bb[i] + 0.25);   // the triangle wave's phase, folded in
            oct[13 * o + 12] = kf;
            kf *= 0.9f;
            double A2[9], b2[3];
            for (int r = 0; r < 3; ++r) {
                for (int cidx = 0; cidx < 3; ++cidx)
                    A2[3 * r + cidx] = 1.1f * (nt[3 * r] * A[cidx] + nt[3 * r + 1] * A[3 + cidx] + nt[3 * r + 2] * A[6 + cidx]);
                b2[r] = 1.1f * (nt[3 * r] * bb[0] + nt[3 * r + 1] * bb[1] + nt[3 * r + 2] * bb[2]) + nt[9 + r];
            }
            std::memcpy(A, A2, sizeof(A));
            std::memcpy(bb, b2, sizeof(bb));
        }
        if (e == hipSuccess) e = upload(&s->d_oct, oct, sizeof(oct));
        if (e == hipSuccess) e = upload(&s->d_craters, craters.data(), craters.size() * sizeof(float));
        if (e == hipSuccess) e = upload(&s->d_crater_masks, crater_masks.data(), crater_masks.size() * sizeof(unsigned long long));
        // The small and tiny craters' cells of every ball, in the order of the balls' ordinals (the order of `sdf`).  30 MB
        // a ball: scenes with more balls than kMinorTableBalls go without, and the evaluators compute the cells.
        if (e == hipSuccess && !ball_offsets.empty() && ball_offsets.size() <= 3 * c->minor_table_balls) {
            const size_t per_ball = minor_cells_bytes_per_ball();
            e = hipMalloc(&s->d_minor_cells, per_ball * (ball_offsets.size() / 3));
            for (size_t b = 0; e == hipSuccess && b < ball_offsets.size() / 3; ++b) {
                launch_minor_cells(c->stream, &ball_offsets[3 * b], static_cast<char*>(s->d_minor_cells) + per_ball * b);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    if (e != hipSuccess) {
        c->last_hip = static_cast<int>(e);
        mipt_scene_free(s);
        return e == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
    }
    s->dev.prims = static_cast<const PrimRec*>(s->d_prims);
    s->dev.tri_c = static_cast<const float*>(s->d_tri);
    s->dev.materials = static_cast<const MaterialRec*>(s->d_mats);
    s->dev.nodes = static_cast<const Bvh4QNode*>(s->d_nodes);
    s->dev.leaf_boxes = static_cast<const float*>(s->d_leaf_boxes);
    s->dev.sdf_boxes = static_cast<const SdfBox*>(s->d_sdf);
    s->dev.noise_oct = static_cast<const float*>(s->d_oct);
    s->dev.noise_raw = s->d_oct ? static_cast<const float*>(s->d_oct) + 8 * 13 : nullptr;
    s->dev.plain_sdf = any_plain ? 1u : 0u;
    s->dev.craters = static_cast<const float*>(s->d_craters);
    s->dev.crater_masks = static_cast<const unsigned long long*>(s->d_crater_masks);
    s->dev.minor_cells = s->d_minor_cells;
    s->dev.n_sdf = static_cast<uint32_t>(sdf.size());
    s->dev.n_prims = n;
    s->dev.n_nodes = static_cast<uint32_t>(qnodes.size());
    s->dev.host_boxes = static_cast<const SdfBox*>(s->d_host_boxes);
    s->dev.n_host = static_cast<uint32_t>(host_boxes.size());
    for (const SdfBox& hb : host_boxes) s->host_prims.push_back(hb.prim);
    if (cb) s->host_cb = *cb;
    {
        const float big = std::numeric_limits<float>::max();
        std::array<float, 6> all{big, big, big, -big, -big, -big};
        auto grow = [](std::array<float, 6>& b, const float* lo, const float* hi) {
            for (int a = 0; a < 3; ++a) {
                b[a] = std::min(b[a], lo[a]);
                b[3 + a] = std::max(b[3 + a], hi[a]);
            }
        };
        if (!analytic_boxes.empty()) {
            for (const auto& b : analytic_boxes) grow(all, b.data(), b.data() + 3);
            s->cull_boxes.push_back(all);
        }
        if (sdf.size() <= 6) {
            for (const SdfBox& b : sdf) {
                std::array<float, 6> box{b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2]};
                const PrimRec& r = prims[b.prim];
                if (r.kind == MIPT_CRATERBALL) {
                    // a ray that stays farther than 0.6 (local units) from the ball's centre can only miss (see
                    // sdf_prim_can_hit in kernels.hip): the world box of that sphere is all a pixel needs to see
                    for (int a = 0; a < 3; ++a) {
                        const double row = std::sqrt(double(r.fwd[3 * a]) * r.fwd[3 * a] + double(r.fwd[3 * a + 1]) * r.fwd[3 * a + 1] +
                                                     double(r.fwd[3 * a + 2]) * r.fwd[3 * a + 2]);
                        const double half = 0.602 * row * 1.00001 + 1e-6;
                        box[a] = std::max(box[a], static_cast<float>(r.fwd[9 + a] - half));
                        box[3 + a] = std::min(box[3 + a], static_cast<float>(r.fwd[9 + a] + half));
                    }
                }
                s->cull_boxes.push_back(box);
            }
        } else {
            std::array<float, 6> m{big, big, big, -big, -big, -big};
            for (const SdfBox& b : sdf) grow(m, b.lo, b.hi);
            s->cull_boxes.push_back(m);
        }
        if (!host_boxes.empty()) {   // one box around everything the host evaluates
            std::array<float, 6> m{big, big, big, -big, -big, -big};
            for (const SdfBox& b : host_boxes) grow(m, b.lo, b.hi);
            s->cull_boxes.push_back(m);
        }
    }
    s->kinds.assign(d->kind, d->kind + n);
    s->n_nodes = n_nodes;
    s->n_analytic = n_analytic;
    s->n_sdf = static_cast<uint32_t>(sdf.size());
    s->depth = bvh.depth;
    s->stack_need = bvh4.stack_need;
    *out = s;
    return MIPT_OK;
}

void mipt_scene_free(mipt_scene* s) {
    if (!s) return;
    // hipFree waits for work that still uses the memory; the context may already be gone
    (void)hipSetDevice(s->device);
    for (void* p : {s->d_prims, s->d_tri, s->d_mats, s->d_nodes, s->d_leaf_boxes, s->d_sdf, s->d_oct, s->d_craters, s->d_crater_masks, s->d_minor_cells, s->d_host_boxes})
        if (p) (void)hipFree(p);
    delete s;
}

int mipt_scene_info(const mipt_scene* s, uint32_t out4[4]) {
    if (!s || !out4) return MIPT_ERR_ARGUMENT;
    out4[0] = s->n_nodes;
    out4[1] = static_cast<uint32_t>(s->depth);
    out4[2] = s->n_analytic;
    out4[3] = s->n_sdf;
    return MIPT_OK;
}

int mipt_render_accumulate(mipt_ctx* c, const mipt_scene* sc, const mipt_camera* cam, const mipt_settings* st, float* device_sum, mipt_stats* stats) {
    if (!c) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    return render_into(c, sc, cam, st, device_sum, stats);
}

int mipt_render(mipt_ctx* c, const mipt_scene* sc, const mipt_camera* cam, const mipt_settings* st, float* out_rgb, mipt_stats* stats) {
    if (!c || !st || !out_rgb) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    const size_t n = static_cast<size_t>(st->width) * st->height * 3;
    if (n == 0) return MIPT_ERR_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_floats(c, &c->frame, &c->frame_cap, n);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(c, hipMemsetAsync(c->frame, 0, n * sizeof(float), c->stream));   // Image starts black (src/Image.cpp:8-13)
    uint32_t entered = 0;
    rc = render_into(c, sc, cam, st, c->frame, stats, &entered);
    if (rc != MIPT_OK && rc != MIPT_ERR_CANCELLED) return rc;
    // acc * sppInv over the samples that entered the frame: all of them, or the batches finished before a cancel
    launch_scale(c->stream, c->frame, 1.0f / static_cast<float>(std::max(entered, 1u)), n, c->frame);
    HIP_TRY(c, hipMemcpyAsync(out_rgb, c->frame, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return rc;
}

int mipt_scale_device(mipt_ctx* c, const float* in, float k, size_t n, float* out) {
    if (!c || !in || !out) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    launch_scale(c->stream, in, k, n, out);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIPT_OK;
}

int mipt_add_device(mipt_ctx* c, float* dst, const float* src, size_t n) {
    if (!c || !dst || !src) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    launch_add(c->stream, dst, src, n);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIPT_OK;
}

int mipt_device_floats_alloc(mipt_ctx* c, size_t n, float** out) {
    if (!c || !out || n == 0) return MIPT_ERR_ARGUMENT;
    *out = nullptr;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    float* p = nullptr;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(float));
    if (e != hipSuccess) {
        c->last_hip = static_cast<int>(e);
        return e == hipErrorOutOfMemory ? MIPT_ERR_OUT_OF_MEMORY : MIPT_ERR_HIP;
    }
    const hipError_t z = hipMemsetAsync(p, 0, n * sizeof(float), c->stream);
    const hipError_t y = z == hipSuccess ? hipStreamSynchronize(c->stream) : z;
    if (y != hipSuccess) {
        (void)hipFree(p);
        c->last_hip = static_cast<int>(y);
        return MIPT_ERR_HIP;
    }
    *out = p;
    return MIPT_OK;
}

int mipt_device_floats_zero(mipt_ctx* c, float* p, size_t n) {
    if (!c || !p) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(p, 0, n * sizeof(float), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIPT_OK;
}

int mipt_device_floats_read(mipt_ctx* c, const float* p, float k, size_t n, float* host_out) {
    if (!c || !p || !host_out || n == 0) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = ensure_floats(c, &c->stage, &c->stage_cap, n);
    if (rc != MIPT_OK) return rc;
    launch_scale(c->stream, p, k, n, c->stage);
    HIP_TRY(c, hipMemcpyAsync(host_out, c->stage, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return MIPT_OK;
}

int mipt_device_floats_write(mipt_ctx* c, float* p, size_t n, const float* host_in) {
    if (!c || !p || !host_in || n == 0) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(p, host_in, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIPT_OK;
}

void mipt_device_floats_free(mipt_ctx* c, float* p) {
    if (!c || !p) return;
    std::lock_guard<std::mutex> lock(c->mutex);
    if (hipSetDevice(c->device) != hipSuccess) return;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(p);
}

int mipt_present_rgba8(mipt_ctx* c, const float* device_sum, float scale, int tonemap, uint8_t* out_rgba, size_t width, size_t height) {
    if (!c || !device_sum || !out_rgba || width == 0 || height == 0 || tonemap < -1 || tonemap > 1) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n_pix = width * height, n = n_pix * 3;
    // staging: scaled image, mapped image, packed pixels
    const int rc = ensure_floats(c, &c->stage, &c->stage_cap, 2 * n + n_pix);
    if (rc != MIPT_OK) return rc;
    float* scaled = c->stage;
    float* mapped = c->stage + n;
    uint32_t* packed = reinterpret_cast<uint32_t*>(c->stage + 2 * n);
    launch_scale(c->stream, device_sum, scale, n, scaled);
    const float* shown = scaled;
    if (tonemap == 1) {
        launch_tonemap_filmic(c->stream, scaled, mapped, n);
        shown = mapped;
    } else if (tonemap == 0) {
        launch_tonemap_reinhard(c->stream, scaled, mapped, width, height, c->tm_scratch, reinhard_scratch_floats());
        shown = mapped;
    }
    launch_pack_rgba8(c->stream, shown, n_pix, packed);
    HIP_TRY(c, hipMemcpyAsync(out_rgba, packed, n_pix * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return MIPT_OK;
}

int mipt_primary_hits(mipt_ctx* c, const mipt_scene* sc, const mipt_camera* cam, uint32_t width, uint32_t height, int32_t* out_prim, float* out_t) {
    if (!c || !sc || !cam || !out_prim || !out_t || width < 2 || height < 2) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = static_cast<size_t>(width) * height;
    if (n > (size_t(1) << 28)) return MIPT_ERR_TOO_LARGE;
    int rc = ensure_slots(c, n, (n + 63) / 64);
    if (rc != MIPT_OK) return rc;
    rc = ensure_host_queue(c, sc);
    if (rc != MIPT_OK) return rc;
    c->cancel.store(0);
    *c->cancel_host = 0u;
    BatchParams bp{};
    bp.n_pix = bp.n_slots = static_cast<uint32_t>(n);
    bp.width = width;
    bp.height = height;
    bp.flags = MIPT_FLAG_NO_JITTER | (c->full_march ? kFlagFullMarch : 0u);
    bp.lane_threshold = c->lane_threshold;
    bp.no_few = c->no_few ? 1u : 0u;
    bp.sort_marches = sc->host_prims.empty() ? c->sort_marches : 0u;   // (the host pass works on the queues between the stages)
    bp.trav_quorum = c->trav_quorum;
    bp.trav_quorum_primary = c->trav_quorum_primary;
    bp.trav_keep = c->trav_keep;
    bp.trav_quorum_pair = c->trav_quorum_pair;
    bp.trav_keep_pair = c->trav_keep_pair;
    bp.pair_plain_entries = c->pair_plain ? 1u : 0u;
    bp.pair_rows = c->pair_tight ? static_cast<uint32_t>(std::max(sc->stack_need, c->min_stack)) + 3u : 0u;
    mipt_camera c0 = *cam;
    c0.blur_radius = 0.0f;
    HIP_TRY(c, hipMemsetAsync(c->pb.counters, 0, sizeof(BounceCounters) * kCounterRows, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->pb.hit_key, 0xff, n * 8, c->stream));   // the fused kernel stores hits only
    launch_latch_cancel(c->stream, c->pb);
    launch_generate_extend(c->stream, pack_camera(c0), sc->dev, bp, c->pb, screen_rects(sc, c0, width, height), std::max(sc->stack_need, c->min_stack));
    if (sc->n_sdf) launch_sdf_march(c->stream, sc->dev, bp, c->pb, 0);
    if (!sc->host_prims.empty()) {
        Totals unused{};
        const int rh = host_objects_extend(c, sc, 0, unused);
        if (rh != MIPT_OK) return rh;
    }
    std::vector<unsigned long long> keys(n);
    HIP_TRY(c, hipMemcpyAsync(keys.data(), c->pb.hit_key, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    for (size_t i = 0; i < n; ++i) {
        if (keys[i] == ~0ull) {
            out_prim[i] = -1;
            out_t[i] = 0.0f;
            continue;
        }
        const uint32_t ob = static_cast<uint32_t>(keys[i] >> 32);
        const uint32_t fb = (ob & 0x80000000u) ? (ob ^ 0x80000000u) : ~ob;
        out_prim[i] = static_cast<int32_t>(keys[i] & 0xffffffffu);
        std::memcpy(&out_t[i], &fb, 4);
    }
    return MIPT_OK;
}

int mipt_trace_rays(mipt_ctx* c, const mipt_scene* sc, uint32_t n, const float* pos, const float* dir, uint32_t bounces, uint64_t seed,
                    float* out_rgb, int32_t* out_first_prim, float* out_first_t) {
    if (!c || !sc || !pos || !dir || !out_rgb || n == 0 || bounces == 0) return MIPT_ERR_ARGUMENT;
    if (bounces > static_cast<uint32_t>(kMaxBounces) || n > (1u << 28)) return MIPT_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_slots(c, n, 1);
    if (rc != MIPT_OK) return rc;
    rc = ensure_host_queue(c, sc);
    if (rc != MIPT_OK) return rc;
    c->cancel.store(0);
    *c->cancel_host = 0u;
    rc = ensure_floats(c, &c->stage, &c->stage_cap, static_cast<size_t>(n) * 6);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->stage, pos, static_cast<size_t>(n) * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->stage + static_cast<size_t>(n) * 3, dir, static_cast<size_t>(n) * 12, hipMemcpyHostToDevice, c->stream));
    BatchParams bp{};
    bp.n_pix = bp.n_slots = n;
    bp.width = n;
    bp.height = 1;
    bp.lane_threshold = c->lane_threshold;
    bp.no_few = c->no_few ? 1u : 0u;
    bp.sort_marches = sc->host_prims.empty() ? c->sort_marches : 0u;   // (the host pass works on the queues between the stages)
    bp.trav_quorum = c->trav_quorum;
    bp.trav_quorum_primary = c->trav_quorum_primary;
    bp.trav_keep = c->trav_keep;
    bp.trav_quorum_pair = c->trav_quorum_pair;
    bp.trav_keep_pair = c->trav_keep_pair;
    bp.pair_plain_entries = c->pair_plain ? 1u : 0u;
    bp.pair_rows = c->pair_tight ? static_cast<uint32_t>(std::max(sc->stack_need, c->min_stack)) + 3u : 0u;
    bp.flags = c->full_march ? kFlagFullMarch : 0u;
    bp.seed = seed;
    HIP_TRY(c, hipMemsetAsync(c->pb.counters, 0, sizeof(BounceCounters) * kCounterRows, c->stream));
    launch_latch_cancel(c->stream, c->pb);
    launch_load_rays(c->stream, c->stage, c->stage + static_cast<size_t>(n) * 3, n, c->pb);
    std::vector<unsigned long long> keys;
    if (out_first_prim || out_first_t) {
        // first closest hit, before shading overwrites the rays
        launch_extend(c->stream, sc->dev, bp, c->pb, 0, std::max(sc->stack_need, c->min_stack));
        if (sc->n_sdf) launch_sdf_march(c->stream, sc->dev, bp, c->pb, 0);
        if (!sc->host_prims.empty()) {
            Totals unused{};
            const int rh = host_objects_extend(c, sc, 0, unused);
            if (rh != MIPT_OK) return rh;
        }
        keys.resize(n);
        HIP_TRY(c, hipMemcpyAsync(keys.data(), c->pb.hit_key, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    rc = enqueue_bounces(c, sc, bp, bounces, keys.empty() ? 0 : 3);
    if (rc != MIPT_OK) return rc;
    launch_export_rad(c->stream, c->pb, n, c->stage);
    HIP_TRY(c, hipMemcpyAsync(out_rgb, c->stage, static_cast<size_t>(n) * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    for (size_t i = 0; i < keys.size(); ++i) {
        const bool miss = keys[i] == ~0ull;
        const uint32_t ob = static_cast<uint32_t>(keys[i] >> 32);
        const uint32_t fb = (ob & 0x80000000u) ? (ob ^ 0x80000000u) : ~ob;
        if (out_first_prim) out_first_prim[i] = miss ? -1 : static_cast<int32_t>(keys[i] & 0xffffffffu);
        if (out_first_t) {
            if (miss) out_first_t[i] = 0.0f;
            else std::memcpy(&out_first_t[i], &fb, 4);
        }
    }
    return MIPT_OK;
}

// Argument check of the SDF probes (caller holds the context's lock): an SDF primitive with a device form, and for
// the kinds that run noise() octave by octave a scene that carries the noise transform.
static int check_sdf_probe(const mipt_scene* sc, uint32_t prim) {
    if (prim >= sc->dev.n_prims || prim >= sc->kinds.size()) return MIPT_ERR_ARGUMENT;
    const uint32_t kind = sc->kinds[prim];
    if (kind < MIPT_CRATERBALL || kind > MIPT_WEIRDWALL) return MIPT_ERR_ARGUMENT;   // analytic, or evaluated on the host
    if (kind >= MIPT_ROUGHBOX && !sc->dev.noise_raw) return MIPT_ERR_ARGUMENT;
    if (kind == MIPT_CRATERBALL && (!sc->dev.noise_oct || !sc->dev.craters)) return MIPT_ERR_ARGUMENT;
    return MIPT_OK;
}

int mipt_sdf_eval(mipt_ctx* c, const mipt_scene* sc, uint32_t prim, uint32_t n, const float* pos3, float* out) {
    if (!c || !sc || !pos3 || !out || n == 0) return MIPT_ERR_ARGUMENT;
    if (n > (1u << 28)) return MIPT_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lock(c->mutex);
    if (const int rc = check_sdf_probe(sc, prim); rc != MIPT_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = ensure_floats(c, &c->stage, &c->stage_cap, static_cast<size_t>(n) * 4);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->stage, pos3, static_cast<size_t>(n) * 12, hipMemcpyHostToDevice, c->stream));
    launch_sdf_eval(c->stream, sc->dev, prim, c->lane_threshold, c->stage, n, c->stage + static_cast<size_t>(n) * 3);
    HIP_TRY(c, hipMemcpyAsync(out, c->stage + static_cast<size_t>(n) * 3, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return MIPT_OK;
}

int mipt_sdf_material_eval(mipt_ctx* c, const mipt_scene* sc, uint32_t prim, uint32_t n, const float* pos3, float* out9) {
    if (!c || !sc || !pos3 || !out9 || n == 0) return MIPT_ERR_ARGUMENT;
    if (n > (1u << 24)) return MIPT_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lock(c->mutex);
    if (const int rc = check_sdf_probe(sc, prim); rc != MIPT_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = ensure_floats(c, &c->stage, &c->stage_cap, static_cast<size_t>(n) * 12);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->stage, pos3, static_cast<size_t>(n) * 12, hipMemcpyHostToDevice, c->stream));
    launch_sdf_material(c->stream, sc->dev, prim, c->stage, n, c->stage + static_cast<size_t>(n) * 3);
    HIP_TRY(c, hipMemcpyAsync(out9, c->stage + static_cast<size_t>(n) * 3, static_cast<size_t>(n) * 36, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return MIPT_OK;
}

// caller holds the context's lock
static int tonemap_on_device(mipt_ctx* c, int kind, const float* in, float* out, size_t width, size_t height) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (kind == 1) launch_tonemap_filmic(c->stream, in, out, width * height * 3);
    else launch_tonemap_reinhard(c->stream, in, out, width, height, c->tm_scratch, reinhard_scratch_floats());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return MIPT_OK;
}

int mipt_tonemap_device(mipt_ctx* c, int kind, const float* in, float* out, size_t width, size_t height) {
    if (!c || !in || !out || width == 0 || height == 0 || (kind != 0 && kind != 1)) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);   // the Reinhard scratch buffer and the stream belong to the context
    return tonemap_on_device(c, kind, in, out, width, height);
}

int mipt_tonemap_device_noalloc(mipt_ctx* c, int kind, const float* in, float* out, size_t width, size_t height) {
    if (!c || !in || !out || width == 0 || height == 0 || (kind != 0 && kind != 1)) return MIPT_ERR_ARGUMENT;
    if (width > MIPT_NOALLOC_MAX_PIXELS || height > MIPT_NOALLOC_MAX_PIXELS / width) return MIPT_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    // Every partial the key kernel reads is written by this call's reduction first, so the array's old
    // contents (the previous call's) never matter and nothing has to be cleared.
    if (kind == 1) launch_tonemap_filmic(c->stream, in, out, width * height * 3);
    else launch_tonemap_reinhard(c->stream, in, out, width, height, c->tm_static, reinhard_scratch_floats());
    HIP_TRY(c, hipGetLastError());
    return MIPT_OK;
}

int mipt_tonemap(mipt_ctx* c, int kind, const float* in, float* out, size_t width, size_t height) {
    if (!c || !in || !out || width == 0 || height == 0 || (kind != 0 && kind != 1)) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(c->mutex);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = width * height * 3;
    int rc = ensure_floats(c, &c->stage, &c->stage_cap, n);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->stage, in, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = tonemap_on_device(c, kind, c->stage, c->stage, width, height);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(c, hipMemcpy(out, c->stage, n * sizeof(float), hipMemcpyDeviceToHost));
    return MIPT_OK;
}

int mipt_affine_inverse(size_t n, const float* fwd, float* inv) {
    if (!fwd || !inv) return MIPT_ERR_ARGUMENT;
    for (size_t i = 0; i < n; ++i)
        if (!affine_inverse(fwd + 12 * i, inv + 12 * i)) return MIPT_ERR_NOT_INVERTIBLE;
    return MIPT_OK;
}

int mipt_rotation(char axis, size_t n, const float* angles, float* out9) {
    if (!angles || !out9 || (axis != 'x' && axis != 'y' && axis != 'z')) return MIPT_ERR_ARGUMENT;
    for (size_t i = 0; i < n; ++i) rotation(axis, angles[i], out9 + 9 * i);
    return MIPT_OK;
}

int mipt_affine_mul(size_t n, const float* a, const float* b, float* out) {
    if (!a || !b || !out) return MIPT_ERR_ARGUMENT;
    for (size_t i = 0; i < n; ++i) affine_mul(a + 12 * i, b + 12 * i, out + 12 * i);
    return MIPT_OK;
}

int mipt_noise_transform(const float t[3], const float r[3], float out[12]) {
    if (!t || !r || !out) return MIPT_ERR_ARGUMENT;
    // Affine * Linear multiplies the linear parts and keeps the translation (src/LinearAlgebra.cpp:405-407)
    float l[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, rot[9];
    const char axes[3] = {'x', 'y', 'z'};
    for (int k = 0; k < 3; ++k) {
        rotation(axes[k], r[k], rot);
        linear_mul(l, rot, l);
    }
    std::memcpy(out, l, sizeof(l));
    out[9] = t[0];
    out[10] = t[1];
    out[11] = t[2];
    return MIPT_OK;
}

int mipt_world_bounds(uint32_t kind, const float fwd[12], const float params[9], float out6[6]) {
    if (!fwd || !params || !out6) return MIPT_ERR_ARGUMENT;
    return world_bounds(kind, fwd, params, out6, out6 + 3) ? MIPT_OK : MIPT_ERR_ARGUMENT;
}

}   // extern "C"
