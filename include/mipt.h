/* mipt -- MI355X path-tracing core behind timstr/pathtracer's scene and render API.
 *
 * C ABI of libmipt.so (pathtracer_amd/lib/libmipt.so).  The reference has no FFI of its own: its
 * boundary is the public C++ API (Scene, Object, Camera, RenderSettings, Renderer, tone mappers).
 * Each entry point below names the reference interface it stands in for (file:line relative to
 * the reference checkout); the C++ classes of include/pathtracer/ call these and nothing else.
 *
 * All functions return MIPT_OK (0) or a negative mipt_status; nothing throws across this boundary.
 * Host pointers unless a parameter is marked `device`.  One mipt_ctx drives one GPU; a process that
 * drives several GPUs creates one ctx per device (or runs one process per GPU).
 */
#ifndef MIPT_H
#define MIPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MIPT_OK = 0,
    MIPT_ERR_ARGUMENT = -1,     /* null pointer, zero size, unknown kind, W or H < 2 */
    MIPT_ERR_NO_DEVICE = -2,    /* no usable HIP device / HIP runtime failure at init */
    MIPT_ERR_HIP = -3,          /* a HIP call failed; mipt_last_hip_error() has the code */
    MIPT_ERR_NOT_INVERTIBLE = -4, /* determinant < 1e-6: the reference's Linear::inverse refuses it */
    MIPT_ERR_CANCELLED = -5,    /* mipt_cancel() arrived during the call; outputs hold what was finished */
    MIPT_ERR_TOO_LARGE = -6,    /* size beyond a documented limit */
    MIPT_ERR_OUT_OF_MEMORY = -7
} mipt_status;

typedef struct mipt_ctx mipt_ctx;
typedef struct mipt_scene mipt_scene;

/* Primitive kinds: the concrete Object subclasses (include/Object.hpp:60-102, 179-192) and the demo's
 * SDF objects: CraterBallObject (PathTracerDemo.cpp:352-483), RoughBoxObject (:228-270), RoughSphereObject
 * (:156-226) and WeirdWallObject (:273-349). */
enum {
    MIPT_SPHERE = 0, MIPT_BOX = 1, MIPT_TRIANGLE = 2, MIPT_CRATERBALL = 3, MIPT_FRACTAL = 4, MIPT_ROUGHBOX = 5,
    MIPT_ROUGHSPHERE = 6, MIPT_WEIRDWALL = 7,
    /* Any other Object subclass (include/Object.hpp:45-52: hitLocalRay, deflectLocalRay, getLocalBoundingBox overridden by
     * the user).  It has no device form: rays that reach its bounding box are handed to the host, which evaluates the
     * object through the callbacks of mipt_scene_upload_host; params = the object's getBoundingBox() (the padded WORLD
     * box, src/Object.cpp:46-80) as {centre x,y,z, half size x,y,z}; fwd / inv are not used. */
    MIPT_HOST_OBJECT = 8
};

/* What Scene::updateGeometry (src/Scene.cpp:80-87) flattens its objects into.  Arrays are copied
 * by mipt_scene_upload; the caller keeps ownership.
 *   fwd / inv   Object::m_transformation / m_inverseTransformation (include/Object.hpp:55-56):
 *               12 floats each, Linear row-major (m[3*r+c]) then the translation.  inv must be what
 *               Object::setTransformation caches (mipt_affine_inverse computes exactly that).
 *   params      9 floats per primitive: sphere {radius}; box {half x,y,z}; triangle {a, b, c};
 *               crater ball {offset x,y,z}; fractal {} (FractalObject has no parameters);
 *               rough box {Rectangle::halfSize x,y,z} (each below 7e4); rough sphere {Sphere::radius};
 *               weird wall {}
 *   materials   14 floats per row in BasicMaterial's member order (include/BasicMaterial.hpp:32-39):
 *               diffuse, specular, sharpness, reflected absorption rgb, emitted luminance rgb,
 *               transmittance, index of refraction, internal absorption rgb
 *   noise_transform  the demo's noiseTransform() (PathTracerDemo.cpp:37-54), 12 floats; may be NULL
 *               when the scene holds none of the demo's SDF objects */
typedef struct {
    uint32_t n_prims;
    const uint32_t* kind;
    const float* fwd;
    const float* inv;
    const float* params;
    const uint32_t* material;
    uint32_t n_materials;
    const float* materials;
    const float* noise_transform;
} mipt_scene_desc;

/* Camera (include/Camera.hpp:25-30): m_transform, aspect ratio, field of view in degrees (used as a
 * half angle, src/Camera.cpp:76), focal distance, focal blur radius. */
typedef struct {
    float transform[12];
    float aspect;
    float fov_degrees;
    float focal_distance;
    float blur_radius;
} mipt_camera;

enum {
    MIPT_FLAG_NO_JITTER = 1u    /* pixel-centre samples: sx = x/(W-1), sy = y/(H-1) (parity maps) */
};

/* RenderSettings (include/RenderSettings.hpp:18-22) plus what the reference keeps implicit:
 * the random seed and, for sample sharding across GPUs, which global sample indices to draw. */
typedef struct {
    uint32_t width;
    uint32_t height;
    uint32_t bounces;        /* RenderSettings::numBounces: ray casts per path (src/Scene.cpp:21) */
    uint32_t spp;            /* samples this call draws per pixel */
    uint64_t seed;
    uint32_t first_sample;   /* global index of this call's first sample (0 on one GPU) */
    uint32_t flags;
} mipt_settings;

typedef struct {
    uint64_t paths;          /* (pixel, sample) pairs started */
    uint64_t rays;           /* closest-hit queries (Tree::hit equivalents) */
    uint64_t sdf_rays;       /* rays that entered an SDF primitive's bounding box */
    uint64_t sdf_shades;     /* hits shaded on an SDF primitive (4 SDF evaluations each) */
    /* work done by the SDF march: evaluations answered by the far-field early-out
     * (PathTracerDemo.cpp:364-366), evaluations that ran the near-field body, and the noise() calls
     * (PathTracerDemo.cpp:56-70) those made */
    uint64_t sdf_far_evals;
    uint64_t sdf_near_evals;
    uint64_t sdf_noise_calls;
    /* the same for the normals of SDF hits (4 evaluations per shaded hit) */
    uint64_t shade_evals;
    uint64_t shade_noise_calls;
    double   ms_device;      /* stream time of the call, HIP events */
    /* Per-stage stream time and launch counts, filled only while mipt_set_profiling(ctx, 1) is in
     * force (an event pair brackets every launch): generate, extend, sdf_march, shade, shade_sdf,
     * accumulate. */
    double   stage_ms[6];
    uint64_t stage_launches[6];
    /* Primary rays that were actually traced.  `rays` counts every closest-hit query the frame asks, as
     * the reference would make them; the queries of pixels that lie outside the screen bounds of every
     * object are answered (miss) without a ray. */
    uint64_t primary_traced;
    /* Samples per pixel that entered the result: settings->spp, or fewer when mipt_cancel arrived (the sample batches
     * that had been accumulated by then; a frame that fits one batch reports 0). */
    uint64_t samples_done;
    /* Filled only while mipt_set_counting(ctx, 1) is in force: BVH nodes visited and primitives tested (Object::hitRay
     * calls) by the closest-hit queries of the call; the reference tests every object for every ray
     * (src/ObjectTree.cpp:22-63), rays * objects. */
    uint64_t node_visits;
    uint64_t prim_tests;
    /* The slow path of scenes with MIPT_HOST_OBJECT primitives: Object::hitRay and Object::deflectRay calls made on the host. */
    uint64_t host_rays;
    uint64_t host_shades;
    /* Also from the counting build, for the closest-hit kernels of the bounces after the first: rounds of the wave
     * state machine (one node visit per lane that can take one), leaf steps among them (Object::hitRay for every lane
     * with a pending leaf), and the times the second ray of a lane started over because the first needed their shared
     * stack column.  node_visits / (64 trav_rounds) is the share of lanes at work in a round. */
    uint64_t trav_rounds;
    uint64_t trav_leaf_steps;
    uint64_t trav_restarts;
    /* SDF marches (include/Object.hpp:135-167) that one bounce's launch handed to the next one's instead of finishing
     * them: the longest marches of a launch would otherwise hold the whole chip for a few lanes' sake.  Such a path
     * runs one launch behind the others from then on; its samples and random numbers are its own, so the frame does
     * not change.  MIPT_CARRY_LAG=0 turns it off. */
    uint64_t marches_carried;
} mipt_stats;

enum { MIPT_STAGE_GENERATE = 0, MIPT_STAGE_EXTEND = 1, MIPT_STAGE_SDF_MARCH = 2, MIPT_STAGE_SHADE = 3,
       MIPT_STAGE_SHADE_SDF = 4, MIPT_STAGE_ACCUMULATE = 5 };

/* ---- context ------------------------------------------------------------------------------ */
/* Renderer::Renderer / startThreadPool (src/Renderer.cpp:9-35): binds a GPU instead of starting
 * CPU workers.  device < 0 uses the current HIP device.  stream is a hipStream_t to launch on, or
 * NULL for a stream the context creates. */
int  mipt_init(int device, void* stream, mipt_ctx** out);
/* Renderer::~Renderer / stopThreadPool (src/Renderer.cpp:14-16, 37-52). */
void mipt_shutdown(mipt_ctx*);
/* Renderer::cancelRender (src/Renderer.cpp:174-176): callable from any thread, lock-free.  The
 * running render call drains (the kernels it still has queued return at once, within one bounce)
 * and returns MIPT_ERR_CANCELLED; the sample batch in flight stays out of the result.  What comes back is
 * the mean over the sample batches accumulated before the cancel (mipt_stats::samples_done of them per pixel;
 * every pixel at fewer samples, where the reference returns some pixels finished and the rest black); a frame
 * that was one batch comes back black with samples_done = 0.  mipt_render_accumulate adds those batches'
 * unnormalised sum, so the caller divides by samples_done. */
int  mipt_cancel(mipt_ctx*);
int  mipt_last_hip_error(const mipt_ctx*);
/* No counterpart in the reference (its working set is the host Image).  A context keeps its working set on the GPU
 * between calls: the path arena (sized by the largest sample batch rendered so far; by default at most 80 % of the
 * memory that was free and at most 48 GiB, MIPT_BATCH_PATHS = path slots overrides), the frame and staging buffers.
 * mipt_trim frees them all (they are allocated again on demand); mipt_shutdown frees them too. */
int  mipt_trim(mipt_ctx*);
/* Measurement aid (no counterpart in the reference): when on, render calls bracket every kernel launch
 * with HIP events on the context's stream and report per-stage times in mipt_stats. */
int  mipt_set_profiling(mipt_ctx*, int on);
/* Measurement aid (no counterpart in the reference): when on, render calls run the counting build of the closest-hit
 * kernels -- the same traversal in the same order, plus two counters per lane -- and report mipt_stats::node_visits
 * and prim_tests.  Frames are unchanged; the counting build is a few per cent slower. */
int  mipt_set_counting(mipt_ctx*, int on);
const char* mipt_strerror(int status);

/* ---- scene -------------------------------------------------------------------------------- */
/* Scene::updateGeometry (src/Scene.cpp:80-87) + ObjectTree::Tree::Tree (src/ObjectTree.cpp:80-83):
 * packs primitives and materials into HBM and builds the linear BVH (a scalable builder replaces
 * makeTree, src/ObjectTree.cpp:90-184; any topology returns the same closest hit because the
 * reference's tree never culls).  Device memory: 128 bytes per primitive, 64 per tree node and material, and per
 * MIPT_CRATERBALL 30.7 MB of tables (the hashed lattice cells of its small and tiny craters, which depend on the
 * ball's offset alone; scenes with more than 64 balls -- MIPT_CELL_TABLE_BALLS -- go without and compute the cells
 * per evaluation, same values).  MIPT_ERR_TOO_LARGE from 2^20 SDF primitives. */
int  mipt_scene_upload(mipt_ctx*, const mipt_scene_desc*, mipt_scene** out);
/* The same with build options.  MIPT_UPLOAD_DEVICE_BVH builds the tree on the GPU (Morton order + radix tree, a few
 * launches) instead of on the host (binned surface-area heuristic): for scenes whose objects move every frame, where
 * the host build (0.5 s at 100,000 objects) would dominate.  Frames are identical; rays cost somewhat more in the
 * GPU-built tree.  MIPT_DEVICE_BVH=1 in the environment makes it the default of mipt_scene_upload.  A scene with fewer
 * than two analytic primitives has no tree to build and is uploaded as without the flag; unknown flag bits are refused
 * (MIPT_ERR_ARGUMENT), more than 60 tree levels (only a great many coincident objects get there) with MIPT_ERR_TOO_LARGE. */
enum { MIPT_UPLOAD_DEVICE_BVH = 1u };
int  mipt_scene_upload_ex(mipt_ctx*, const mipt_scene_desc*, uint32_t flags, mipt_scene** out);
/* Scenes that hold MIPT_HOST_OBJECT primitives.  The callbacks stand for the object's non-virtual wrappers
 * Object::hitRay (src/Object.cpp:22-33: world ray in; returns nonzero and the world hit position on a hit) and
 * Object::deflectRay (src/Object.cpp:35-44: ray_pos = the hit position, ray_dir = the incoming direction; out9 = emitted
 * r,g,b, attenuation r,g,b, new direction in world space), `prim` being the primitive's index in the scene description.
 * They are called from several threads of the library at once while a render call runs (the reference calls the same
 * virtuals from its worker threads) and must stay valid until mipt_scene_free.  This is the slow path: every bounce
 * of a frame costs two round trips to the host; mipt_stats::host_rays / host_shades report how much went that way.
 * Everything else about the frame -- closest hit over all objects, ties to the lower index, Scene::trace's
 * accumulation -- is as for device primitives. */
typedef struct {
    int  (*hit)(void* user, uint32_t prim, const float ray_pos[3], const float ray_dir[3], float out_hit_pos[3]);
    void (*deflect)(void* user, uint32_t prim, const float ray_pos[3], const float ray_dir[3], float out9[9]);
    void* user;
} mipt_host_callbacks;
int  mipt_scene_upload_host(mipt_ctx*, const mipt_scene_desc*, uint32_t flags, const mipt_host_callbacks*, mipt_scene** out);
void mipt_scene_free(mipt_scene*);
/* BVH facts for tests and reports: out4 = node count, depth, analytic primitives, SDF primitives. */
int  mipt_scene_info(const mipt_scene*, uint32_t out4[4]);

/* ---- rendering ---------------------------------------------------------------------------- */
/* Renderer::render (src/Renderer.cpp:139-172): one frame of settings->spp samples per pixel;
 * out_rgb (host, width*height*3 floats, row-major like Image, src/Image.cpp:83-87) receives the
 * per-pixel mean. */
int  mipt_render(mipt_ctx*, const mipt_scene*, const mipt_camera*, const mipt_settings*,
                 float* out_rgb, mipt_stats* stats /* nullable */);
/* Closest hits are the reference's (ObjectTree::Tree::hit, src/ObjectTree.cpp:22-88: closest of ALL objects, t and
 * position bit for bit, exact ties to the lower primitive index where the reference takes its tree's first leaf) with ONE
 * bounded deviation, which every call that casts rays shares (mipt_render*, mipt_primary_hits, mipt_trace_rays): the
 * reference's sphere test (src/Geometry.cpp:173-201) loses about half of float's digits when the ray starts far from a
 * small sphere (b^2 - 4ac with b^2 ~ 4 |o - c|^2), so it reports "hits" up to about |o - c|^2 * 2^-23 / (2 r) OUTSIDE the
 * sphere -- 1e-3 for r = 0.05 seen from 30 units -- and where such a point also lies outside the object's padded
 * bounding box (Object::getBoundingBox pads by 1e-3, src/Object.cpp:124-133) a traversal that culls by boxes does not
 * find it, while the reference, which tests every object, does.  Measured: 6 of 58.7 million rays on 20,000 objects of
 * radius 0.02 - 0.12 seen from 30 units, about 1e-7 of the rays on the 100,000-object configuration, none on scenes
 * of larger or nearer objects (tests/test_gpu_traversal.py holds the rate at <= 1e-5 against a never-culling scan through
 * the same kernels, MIPT_NO_CULL=1).  Padding every sphere's box by a proven cancellation bound would cost several times
 * the radius for the smallest spheres; the deviation is kept and stated instead. */
/* The same samples ADDED, unnormalised, into a device buffer (width*height*3 floats): the per-GPU
 * radiance sum of a sample-sharded render.  The caller zeroes the buffer, all-reduces it across
 * ranks and scales by 1/total_spp (mipt_scale_device).  Work is queued on the context's stream;
 * the call returns after the stream has drained. */
int  mipt_render_accumulate(mipt_ctx*, const mipt_scene*, const mipt_camera*, const mipt_settings*,
                            float* device_sum, mipt_stats* stats /* nullable */);
/* device_out[i] = device_in[i] * k  (the `acc * sppInv` of src/Renderer.cpp:102-105); in-place allowed. */
int  mipt_scale_device(mipt_ctx*, const float* device_in, float k, size_t n_floats, float* device_out);

/* device_dst[i] += device_src[i] (the `acc += rendered` of PathTracerDemo.cpp:963-964 for buffers on one device). */
int  mipt_add_device(mipt_ctx*, float* device_dst, const float* device_src, size_t n_floats);

/* Device buffers of floats for hosts without a HIP toolchain of their own (the C++ host library builds with plain
 * g++): the accumulator of a progressive display loop.  No counterpart in the reference, whose accumulator is a host
 * Image (PathTracerDemo.cpp:934-968).  alloc returns n zero-filled floats; read copies p[i] * k to host memory. */
int  mipt_device_floats_alloc(mipt_ctx*, size_t n, float** out);
int  mipt_device_floats_zero(mipt_ctx*, float* p, size_t n);
int  mipt_device_floats_read(mipt_ctx*, const float* p, float k, size_t n, float* host_out);
/* host -> device: puts a saved accumulator back (resume from an Image file, src/Image.cpp:62-81) */
int  mipt_device_floats_write(mipt_ctx*, float* p, size_t n, const float* host_in);
void mipt_device_floats_free(mipt_ctx*, float* p);
/* The display step of the demo's progressive loop (PathTracerDemo.cpp:963-971: `acc += rendered`,
 * scaleImage(acc, 1 / count), tone map, copyToSFImage :524-545) for an accumulator that stays on the
 * device: out = pack(tonemap(device_sum * scale)), 4 bytes per pixel R, G, B, 255 with copyToSFImage's
 * clamp-to-[0,1] * 255 truncation, row-major, to a HOST buffer of width*height*4 bytes.  tonemap: -1 none
 * (what the demo displays), 0 Reinhard, 1 filmic.  Moves 4 bytes per pixel over PCIe instead of 12. */
int  mipt_present_rgba8(mipt_ctx*, const float* device_sum, float scale, int tonemap, uint8_t* out_rgba,
                        size_t width, size_t height);

/* Parity probes (no counterpart in the reference's API; they expose stages of the hot path).
 * Pixel-centre primary rays through Camera::getViewRay and the closest hit of each:
 * out_prim = primitive index or -1, out_t = (hit - ray.pos) . ray.dir (src/ObjectTree.cpp:43-44). */
int  mipt_primary_hits(mipt_ctx*, const mipt_scene*, const mipt_camera*, uint32_t width, uint32_t height,
                       int32_t* out_prim, float* out_t);
/* Scene::trace (src/Scene.cpp:18-78) for n explicit rays; ray i draws from the random stream of
 * path (pixel = i, sample = 0).  out_first_prim / out_first_t (nullable) report the first hit. */
int  mipt_trace_rays(mipt_ctx*, const mipt_scene*, uint32_t n, const float* ray_pos3, const float* ray_dir3,
                     uint32_t bounces, uint64_t seed, float* out_rgb, int32_t* out_first_prim, float* out_first_t);

/* The signed distance of SDF primitive `prim` (kinds MIPT_CRATERBALL to MIPT_WEIRDWALL) at n points of the object's LOCAL frame,
 * evaluated by the device code the march uses: CraterBallObject::signedDistance (PathTracerDemo.cpp:361-466),
 * FractalObject::signedDistance (src/Object.cpp:198-252), RoughBoxObject (:237-246), RoughSphereObject (:165-182),
 * WeirdWallObject (:281-328).  pos3 = n x 3 floats, out = n floats (host). */
int  mipt_sdf_eval(mipt_ctx*, const mipt_scene*, uint32_t prim, uint32_t n, const float* pos3, float* out);
/* The material a hit at each of n LOCAL points of SDF primitive `prim` is shaded with, by the device code shading uses:
 * what the object's deflectLocalRay makes of its material there -- RoughSphereObject (PathTracerDemo.cpp:184-199),
 * RoughBoxObject (:248-259), WeirdWallObject (:330-341); the primitive's own material row for CraterBallObject (:468-472)
 * and FractalObject (src/Object.cpp:177-189).  out9 = n x 9 floats {reflected absorption r g b, emitted luminance r g b,
 * diffuse, specular, sharpness} (host).  Both probes refuse primitives that are not SDF objects with a device form
 * (MIPT_ERR_ARGUMENT), MIPT_HOST_OBJECT included. */
int  mipt_sdf_material_eval(mipt_ctx*, const mipt_scene*, uint32_t prim, uint32_t n, const float* pos3, float* out9);

/* ---- one process, several GPUs -------------------------------------------------------------------------
 * Renderer::render (src/Renderer.cpp:139-172) across the GPUs of a node, where the reference spreads 8-pixel tasks
 * over CPU threads (src/Renderer.cpp:54-137): the frame's samples per pixel are divided into one contiguous block per
 * device (random numbers are keyed by the global sample index, so the set of samples does not depend on the number
 * of devices), every device adds its block into a full-frame radiance-sum buffer of its own, ONE RCCL all-reduce
 * (ncclAllReduce, sum, f32, width*height*3 elements) over xGMI combines them, and the first device scales by
 * 1 / spp, tone-maps and copies the frame to the host.  RCCL is loaded when mipt_init_all is called (librccl.so.1), not
 * before.
 *   mipt_init_all(n, devices, flags, out): devices = NULL -> devices 0 .. n-1 (n = 0: all of the node's).
 *   MIPT_NODE_SIMULATE: no RCCL; `devices` may then name one device several times, every entry being one shard with a
 *   context of its own, and the shards' buffers are added on that device -- what k GPUs compute, on one (for tests).
 * One process per GPU with the caller's own collective (mipt_render_accumulate, above) remains the other way. */
typedef struct mipt_node mipt_node;
typedef struct mipt_node_scene mipt_node_scene;
enum { MIPT_NODE_SIMULATE = 1u };
int  mipt_init_all(int n_devices, const int* devices, uint32_t flags, mipt_node** out);
void mipt_node_shutdown(mipt_node*);
int  mipt_node_size(const mipt_node*);
mipt_ctx* mipt_node_ctx(mipt_node*, int shard);
int  mipt_node_cancel(mipt_node*);                                   /* Renderer::cancelRender for every shard */
int  mipt_node_scene_upload(mipt_node*, const mipt_scene_desc*, mipt_node_scene** out);   /* Scene::updateGeometry on every device */
/* The same for scenes that hold MIPT_HOST_OBJECT primitives (mipt_scene_upload_host on every device): every shard hands the
 * rays that reach such an object to the callbacks from host threads of its own, so the callbacks run from all shards
 * at once (they must be re-entrant anyway, include/Object.hpp:45-52).  mipt_stats::host_rays / host_shades are summed. */
int  mipt_node_scene_upload_host(mipt_node*, const mipt_scene_desc*, const mipt_host_callbacks*, mipt_node_scene** out);
void mipt_node_scene_free(mipt_node*, mipt_node_scene*);
/* tonemap: -1 none, 0 Reinhard, 1 filmic (src/ToneMapper.cpp), applied on the GPU after the all-reduce */
int  mipt_node_render(mipt_node*, const mipt_node_scene*, const mipt_camera*, const mipt_settings*, int tonemap, float* out_rgb,
                      mipt_stats* stats /* nullable; summed over the shards */);

/* ---- tone mapping ------------------------------------------------------------------------- */
/* ReinhardToneMapper / FilmicToneMapper::operator() (src/ToneMapper.cpp:6-72): kind 0 Reinhard,
 * 1 filmic.  Host image in, host image out. */
int  mipt_tonemap(mipt_ctx*, int kind, const float* in_rgb, float* out_rgb, size_t width, size_t height);
/* The same on device buffers (after the all-reduce of a sharded render); in-place allowed. */
int  mipt_tonemap_device(mipt_ctx*, int kind, const float* device_in, float* device_out, size_t width, size_t height);

/* No counterpart in the reference: mipt_tonemap_device for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Reinhard's partial sums, which
 * mipt_tonemap_device keeps in a buffer the context allocated, live in a file-scope device array here (12 KiB on
 * every GPU the library runs on, held until the process ends); the call allocates and frees nothing, enqueues its
 * kernels on the context's stream and returns WITHOUT waiting for them.  Results are bit-identical to
 * mipt_tonemap_device.  width * height above MIPT_NOALLOC_MAX_PIXELS is refused with MIPT_ERR_TOO_LARGE.  All calls
 * on one GPU share that GPU's array: calls that overlap on one GPU (two contexts or streams) race on it. */
#define MIPT_NOALLOC_MAX_PIXELS ((size_t)1 << 26)
int  mipt_tonemap_device_noalloc(mipt_ctx*, int kind, const float* device_in, float* device_out, size_t width, size_t height);

/* ---- host-side helpers (no GPU needed) ------------------------------------------------------ */
/* Affine::inverse as Object::setTransformation applies it (src/LinearAlgebra.cpp:207-264,
 * src/Object.cpp:15-20), for n transforms.  Returns MIPT_ERR_NOT_INVERTIBLE if any determinant is
 * below 1e-6 (the reference asserts). */
int  mipt_affine_inverse(size_t n, const float* fwd12, float* inv12);
/* Linear::RotationX/Y/Z (src/LinearAlgebra.cpp:90-118); axis 'x', 'y' or 'z'; 9 floats each. */
int  mipt_rotation(char axis, size_t n, const float* angles, float* out9);
/* Affine * Affine (src/LinearAlgebra.cpp:413-415), elementwise over n pairs. */
int  mipt_affine_mul(size_t n, const float* a12, const float* b12, float* out12);
/* Translation(t) * RotationX(r[0]) * RotationY(r[1]) * RotationZ(r[2]) (PathTracerDemo.cpp:48-51). */
int  mipt_noise_transform(const float t3[3], const float r3[3], float out12[12]);
/* Object::getBoundingBox (src/Object.cpp:46-80): out6 = min xyz, max xyz of the padded world box. */
int  mipt_world_bounds(uint32_t kind, const float fwd12[12], const float params9[9], float out6[6]);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */
