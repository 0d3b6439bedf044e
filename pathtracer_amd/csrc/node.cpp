// One process, several GPUs: the frame's samples per pixel are divided among the devices of a node, every
// device adds its share into a full-frame radiance-sum buffer of its own, one RCCL all-reduce (sum, f32,
// width * height * 3 elements) over xGMI combines the buffers, and the first device scales, tone-maps and hands the
// frame to the host.  The multi-GPU form of Renderer::render (src/Renderer.cpp:139-172), whose own parallelism is a
// pool of CPU threads over 8-pixel tasks (src/Renderer.cpp:54-137).
//
// RCCL is loaded at run time (dlopen of librccl.so.1, the soname both ROCm's and PyTorch's copy carry, so a process
// that already has one loaded shares it) and only by mipt_init_all: single-GPU users never touch it.
#include "../../include/mipt.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

// The handful of RCCL entry points this file uses (rccl.h: ncclCommInitAll, ncclAllReduce, ncclGroupStart / End,
// ncclCommDestroy), by their C signatures.
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*CommAbort)(void* comm) = nullptr;   // optional: drops what is queued on a communicator whose collective cannot complete
    int (*AllReduce)(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool load() {
        if (lib) return true;
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        return CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd;
    }
};
constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // ncclDataType_t / ncclRedOp_t values of rccl.h

// (first sample, count) of shard `i` of `n`: contiguous blocks of the global sample range, the remainder to the low shards
void sample_range(uint32_t spp, int n, int i, uint32_t& first, uint32_t& count) {
    const uint32_t base = spp / static_cast<uint32_t>(n), rem = spp % static_cast<uint32_t>(n);
    first = static_cast<uint32_t>(i) * base + std::min(static_cast<uint32_t>(i), rem);
    count = base + (static_cast<uint32_t>(i) < rem ? 1u : 0u);
}

}   // namespace

struct mipt_node {
    std::vector<int> devices;          // device of every shard (repeats only in simulated mode)
    std::vector<mipt_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<float*> sum;           // per shard: width * height * 3 radiance sums
    std::vector<void*> comm;           // RCCL communicators, one per shard (empty in simulated mode)
    size_t sum_floats = 0;
    float* mapped = nullptr;           // on the first device: the tone-mapped frame
    size_t mapped_floats = 0;
    bool simulated = false;
    bool broken = false;               // a collective failed half-way: the communicators cannot be used again
    Rccl rccl;
    std::mutex mutex;
};

struct mipt_node_scene {
    std::vector<mipt_scene*> scene;    // one upload per distinct shard (shards on one device share the first one's)
};

extern "C" {

int mipt_init_all(int n_devices, const int* devices, uint32_t flags, mipt_node** out) {
    if (!out || n_devices < 0 || (flags & ~static_cast<uint32_t>(MIPT_NODE_SIMULATE))) return MIPT_ERR_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return MIPT_ERR_NO_DEVICE;
    auto* node = new (std::nothrow) mipt_node;
    if (!node) return MIPT_ERR_OUT_OF_MEMORY;
    if (devices) {
        if (n_devices == 0) {
            delete node;
            return MIPT_ERR_ARGUMENT;
        }
        node->devices.assign(devices, devices + n_devices);
    } else {
        const int n = n_devices == 0 ? count : n_devices;
        for (int i = 0; i < n; ++i) node->devices.push_back(i);
    }
    node->simulated = (flags & MIPT_NODE_SIMULATE) != 0;
    for (size_t i = 0; i < node->devices.size(); ++i) {
        const int d = node->devices[i];
        bool repeated = false;
        for (size_t j = 0; j < i; ++j) repeated |= node->devices[j] == d;
        if (d < 0 || d >= count || (repeated && !node->simulated)) {   // RCCL has one rank per device
            delete node;
            return MIPT_ERR_ARGUMENT;
        }
    }
    const size_t n = node->devices.size();
    node->ctx.assign(n, nullptr);
    node->stream.assign(n, nullptr);
    node->sum.assign(n, nullptr);
    int status = MIPT_OK;
    for (size_t i = 0; i < n && status == MIPT_OK; ++i) {
        if (hipSetDevice(node->devices[i]) != hipSuccess || hipStreamCreateWithFlags(&node->stream[i], hipStreamNonBlocking) != hipSuccess) {
            status = MIPT_ERR_NO_DEVICE;
            break;
        }
        status = mipt_init(node->devices[i], node->stream[i], &node->ctx[i]);
    }
    if (status == MIPT_OK && !node->simulated) {
        if (!node->rccl.load()) status = MIPT_ERR_NO_DEVICE;
        else {
            node->comm.assign(n, nullptr);
            if (node->rccl.CommInitAll(node->comm.data(), static_cast<int>(n), node->devices.data()) != 0) {
                node->comm.clear();
                status = MIPT_ERR_HIP;
            }
        }
    }
    if (status != MIPT_OK) {
        mipt_node_shutdown(node);
        return status;
    }
    *out = node;
    return MIPT_OK;
}

void mipt_node_shutdown(mipt_node* node) {
    if (!node) return;
    for (void* c : node->comm)
        if (c) (void)((node->broken && node->rccl.CommAbort) ? node->rccl.CommAbort(c) : node->rccl.CommDestroy(c));
    for (size_t i = 0; i < node->ctx.size(); ++i) {
        (void)hipSetDevice(node->devices[i]);
        if (node->sum[i]) (void)hipFree(node->sum[i]);
        if (i == 0 && node->mapped) (void)hipFree(node->mapped);
        if (node->ctx[i]) mipt_shutdown(node->ctx[i]);
        if (node->stream[i]) (void)hipStreamDestroy(node->stream[i]);
    }
    delete node;
}

int mipt_node_size(const mipt_node* node) { return node ? static_cast<int>(node->devices.size()) : 0; }

mipt_ctx* mipt_node_ctx(mipt_node* node, int shard) {
    return node && shard >= 0 && static_cast<size_t>(shard) < node->ctx.size() ? node->ctx[static_cast<size_t>(shard)] : nullptr;
}

int mipt_node_cancel(mipt_node* node) {
    if (!node) return MIPT_ERR_ARGUMENT;
    for (mipt_ctx* c : node->ctx) mipt_cancel(c);
    return MIPT_OK;
}

int mipt_node_scene_upload(mipt_node* node, const mipt_scene_desc* desc, mipt_node_scene** out) {
    return mipt_node_scene_upload_host(node, desc, nullptr, out);
}

int mipt_node_scene_upload_host(mipt_node* node, const mipt_scene_desc* desc, const mipt_host_callbacks* cb, mipt_node_scene** out) {
    if (!node || !desc || !out) return MIPT_ERR_ARGUMENT;
    *out = nullptr;
    auto* ns = new (std::nothrow) mipt_node_scene;
    if (!ns) return MIPT_ERR_OUT_OF_MEMORY;
    ns->scene.assign(node->ctx.size(), nullptr);
    for (size_t i = 0; i < node->ctx.size(); ++i) {
        size_t first = i;
        for (size_t j = 0; j < i; ++j)
            if (node->devices[j] == node->devices[i]) {
                first = j;
                break;
            }
        if (first != i) {
            ns->scene[i] = ns->scene[first];   // shards of one device read the same upload
            continue;
        }
        // (objects without a device form are evaluated by the caller's callbacks from every shard's host threads at once)
        const int rc = cb ? mipt_scene_upload_host(node->ctx[i], desc, 0u, cb, &ns->scene[i]) : mipt_scene_upload(node->ctx[i], desc, &ns->scene[i]);
        if (rc != MIPT_OK) {
            mipt_node_scene_free(node, ns);
            return rc;
        }
    }
    *out = ns;
    return MIPT_OK;
}

void mipt_node_scene_free(mipt_node* node, mipt_node_scene* ns) {
    if (!ns) return;
    for (size_t i = 0; i < ns->scene.size(); ++i) {
        bool shared = false;
        for (size_t j = 0; j < i; ++j) shared |= ns->scene[j] == ns->scene[i];
        if (ns->scene[i] && !shared) mipt_scene_free(ns->scene[i]);
    }
    (void)node;
    delete ns;
}

int mipt_node_render(mipt_node* node, const mipt_node_scene* ns, const mipt_camera* cam, const mipt_settings* st, int tonemap, float* out_rgb,
                     mipt_stats* stats) {
    if (!node || !ns || !cam || !st || !out_rgb || tonemap < -1 || tonemap > 1 || ns->scene.size() != node->ctx.size()) return MIPT_ERR_ARGUMENT;
    if (st->width < 2 || st->height < 2 || st->spp == 0) return MIPT_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lock(node->mutex);
    if (node->broken) return MIPT_ERR_HIP;
    const size_t n = node->ctx.size();
    const size_t floats = static_cast<size_t>(st->width) * st->height * 3;
    // per-shard sum buffers (and the mapped frame on the first device), grown on demand
    if (floats > node->sum_floats) {
        for (size_t i = 0; i < n; ++i) {
            if (hipSetDevice(node->devices[i]) != hipSuccess) return MIPT_ERR_HIP;
            if (node->sum[i]) (void)hipFree(node->sum[i]);
            node->sum[i] = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&node->sum[i]), floats * sizeof(float)) != hipSuccess) {
                node->sum_floats = 0;
                return MIPT_ERR_OUT_OF_MEMORY;
            }
        }
        node->sum_floats = floats;
    }
    if (floats > node->mapped_floats) {
        if (hipSetDevice(node->devices[0]) != hipSuccess) return MIPT_ERR_HIP;
        if (node->mapped) (void)hipFree(node->mapped);
        node->mapped = nullptr;
        node->mapped_floats = 0;
        if (hipMalloc(reinterpret_cast<void**>(&node->mapped), floats * sizeof(float)) != hipSuccess) return MIPT_ERR_OUT_OF_MEMORY;
        node->mapped_floats = floats;
    }
    // every shard renders its block of the frame's global sample range on a host thread of its own (the render call
    // returns when the shard's stream has drained)
    std::vector<int> rc(n, MIPT_OK);
    std::vector<mipt_stats> sst(n);
    std::vector<std::thread> workers;
    for (size_t i = 0; i < n; ++i)
        workers.emplace_back([&, i] {
            uint32_t first, count;
            sample_range(st->spp, static_cast<int>(n), static_cast<int>(i), first, count);
            std::memset(&sst[i], 0, sizeof(mipt_stats));
            if (hipSetDevice(node->devices[i]) != hipSuccess) {
                rc[i] = MIPT_ERR_HIP;
                return;
            }
            rc[i] = mipt_device_floats_zero(node->ctx[i], node->sum[i], floats);
            if (rc[i] != MIPT_OK || count == 0) return;
            mipt_settings mine = *st;
            mine.first_sample = st->first_sample + first;
            mine.spp = count;
            rc[i] = mipt_render_accumulate(node->ctx[i], ns->scene[i], cam, &mine, node->sum[i], &sst[i]);
        });
    for (auto& w : workers) w.join();
    int status = MIPT_OK;
    for (size_t i = 0; i < n; ++i)
        if (rc[i] != MIPT_OK && (status == MIPT_OK || status == MIPT_ERR_CANCELLED)) status = rc[i];
    if (status != MIPT_OK && status != MIPT_ERR_CANCELLED) return status;
    // one all-reduce over the node; in simulated mode (several shards on one device) the buffers are added there instead
    uint64_t samples_in = 0;
    for (size_t i = 0; i < n; ++i) samples_in += sst[i].samples_done;
    if (!node->simulated) {
        if (node->rccl.GroupStart() != 0) return MIPT_ERR_HIP;
        // The group is closed on every way out.  An all-reduce that only some shards joined cannot complete, so after
        // a failure inside the bracket the node refuses further frames (mipt_node_shutdown aborts what is queued).
        bool ok = true;
        for (size_t i = 0; i < n && ok; ++i)
            ok = hipSetDevice(node->devices[i]) == hipSuccess &&
                 node->rccl.AllReduce(node->sum[i], node->sum[i], floats, kNcclFloat32, kNcclSum, node->comm[i], node->stream[i]) == 0;
        if (node->rccl.GroupEnd() != 0) ok = false;
        if (!ok) {
            node->broken = true;
            return MIPT_ERR_HIP;
        }
        for (size_t i = 0; i < n; ++i) {
            if (hipSetDevice(node->devices[i]) != hipSuccess || hipStreamSynchronize(node->stream[i]) != hipSuccess) return MIPT_ERR_HIP;
        }
    } else {
        for (size_t i = 1; i < n; ++i) {
            const int ra = mipt_add_device(node->ctx[0], node->sum[0], node->sum[i], floats);
            if (ra != MIPT_OK) return ra;
        }
    }
    // acc * sppInv (src/Renderer.cpp:102-105) over the samples that entered, tone map, to the host -- on the first device
    if (hipSetDevice(node->devices[0]) != hipSuccess) return MIPT_ERR_HIP;
    const float k = 1.0f / static_cast<float>(std::max<uint64_t>(samples_in, 1));
    int r = mipt_scale_device(node->ctx[0], node->sum[0], k, floats, node->mapped);
    if (r == MIPT_OK && tonemap >= 0) r = mipt_tonemap_device(node->ctx[0], tonemap, node->mapped, node->mapped, st->width, st->height);
    if (r == MIPT_OK) r = mipt_device_floats_read(node->ctx[0], node->mapped, 1.0f, floats, out_rgb);
    if (r != MIPT_OK) return r;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (size_t i = 0; i < n; ++i) {
            stats->paths += sst[i].paths;
            stats->rays += sst[i].rays;
            stats->sdf_rays += sst[i].sdf_rays;
            stats->sdf_shades += sst[i].sdf_shades;
            stats->sdf_far_evals += sst[i].sdf_far_evals;
            stats->sdf_near_evals += sst[i].sdf_near_evals;
            stats->sdf_noise_calls += sst[i].sdf_noise_calls;
            stats->shade_evals += sst[i].shade_evals;
            stats->shade_noise_calls += sst[i].shade_noise_calls;
            stats->primary_traced += sst[i].primary_traced;
            stats->samples_done += sst[i].samples_done;
            stats->host_rays += sst[i].host_rays;
            stats->host_shades += sst[i].host_shades;
            stats->node_visits += sst[i].node_visits;   // counting build (mipt_set_counting on the shards' contexts)
            stats->prim_tests += sst[i].prim_tests;
            stats->trav_rounds += sst[i].trav_rounds;
            stats->trav_leaf_steps += sst[i].trav_leaf_steps;
            stats->trav_restarts += sst[i].trav_restarts;
            stats->marches_carried += sst[i].marches_carried;
            stats->ms_device = std::max(stats->ms_device, sst[i].ms_device);   // the shards run side by side
        }
    }
    return status;
}

}   // extern "C"
