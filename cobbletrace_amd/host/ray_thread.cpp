// ray_thread.cpp -- RayThread(env, scene): the drop-in for the reference's Boss/Worker
// ray tracer (raythread.h:5, raythread.cpp:641-667), driving the HIP library through the
// C ABI of include/ctgpu.h.
//
// Kept from the reference's boss: lazy one-time set-up on the first call (triangle
// gather + BVH build, raythread.cpp:647-654), keyboard camera control (:388-434), the
// "re-render only when something changed" rule (:546-562), the yaw/pitch/roll matrix in
// binary32 (:564-572), and always returning true.  Replaced: the worker pool.  By default a
// frame is one synchronous GPU dispatch, so a call that has work to do returns with the
// pixels already in env->bitmap->memory; with ct_raythread_options.async the call only
// QUEUES the frame (ctgpu_render_begin) and later calls look whether it has finished
// (ctgpu_render_poll), the way the reference's boss looks at its workers' status
// (raythread.cpp:557-562) -- the caller's loop (event pump, Blit, 33 ms sleep) never waits.
#include "cobbletrace_host.h"
#include "ctgpu.h"
#include "scene_pack.h"
#include "host_internal.h"

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

namespace {

// ---- the C ABI, resolved at run time so this layer builds and loads without ROCm
struct GpuApi {
    void *lib = nullptr;
    decltype(&ctgpu_abi_version) abi_version = nullptr;
    decltype(&ctgpu_last_error) last_error = nullptr;
    decltype(&ctgpu_create) create = nullptr;
    decltype(&ctgpu_destroy) destroy = nullptr;
    decltype(&ctgpu_upload_scene) upload_scene = nullptr;
    decltype(&ctgpu_share_scene) share_scene = nullptr;
    decltype(&ctgpu_render) render = nullptr;
    decltype(&ctgpu_render_multi) render_multi = nullptr;
    decltype(&ctgpu_render_begin) render_begin = nullptr;
    decltype(&ctgpu_render_poll) render_poll = nullptr;
    decltype(&ctgpu_keep_framebuffer_pinned) keep_pinned = nullptr;
    decltype(&ctgpu_device_count) device_count = nullptr;
    decltype(&ctgpu_build_bvh) build_bvh = nullptr;
    decltype(&ctgpu_trace_rays_packet) trace_rays_packet = nullptr;
};

[[noreturn]] void die(const char *what, const char *detail) {
    std::fprintf(stderr, "cobbletrace: %s: %s\n", what, detail ? detail : "");
    std::exit(2);     // the reference's fatal path (raythread.cpp:608)
}

std::string library_beside_this_one() {
    Dl_info info;
    if (!dladdr((void *)&library_beside_this_one, &info) || !info.dli_fname) return "libctgpu.so";
    std::string here(info.dli_fname);
    const size_t slash = here.find_last_of('/');
    const std::string dir = (slash == std::string::npos) ? std::string(".") : here.substr(0, slash);
    return dir + "/../csrc/libctgpu.so";
}

GpuApi &gpu() {
    static GpuApi api;
    if (api.lib) return api;
    const char *override_path = std::getenv("CTGPU_LIBRARY");
    const std::string path = override_path ? std::string(override_path) : library_beside_this_one();
    api.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) die("cannot load the HIP ray-trace library", dlerror());
#define RESOLVE(field, symbol)                                                   \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, symbol));   \
    if (!api.field) die("missing symbol in libctgpu.so", symbol)
    RESOLVE(abi_version, "ctgpu_abi_version");
    RESOLVE(last_error, "ctgpu_last_error");
    RESOLVE(create, "ctgpu_create");
    RESOLVE(destroy, "ctgpu_destroy");
    RESOLVE(upload_scene, "ctgpu_upload_scene");
    RESOLVE(share_scene, "ctgpu_share_scene");
    RESOLVE(render, "ctgpu_render");
    RESOLVE(render_multi, "ctgpu_render_multi");
    RESOLVE(render_begin, "ctgpu_render_begin");
    RESOLVE(render_poll, "ctgpu_render_poll");
    RESOLVE(keep_pinned, "ctgpu_keep_framebuffer_pinned");
    RESOLVE(device_count, "ctgpu_device_count");
    RESOLVE(build_bvh, "ctgpu_build_bvh");
    RESOLVE(trace_rays_packet, "ctgpu_trace_rays_packet");
#undef RESOLVE
    if (api.abi_version() != CTGPU_ABI_VERSION) die("libctgpu.so ABI mismatch", path.c_str());
    return api;
}

struct BossState {
    bool initialised = false;
    bool changes_pending = true;          // the first call renders (raythread.cpp:547)
    float yaw = 0, pitch = 0, roll = 0;
    ct::PackedScene *packed = nullptr;
    std::vector<ctgpu_ctx *> ctxs;        // one per GPU the frame is split over
    uint64_t rand_consumed = 0;           // LIBC jitter mode: rand() calls used by earlier frames
    ct_raythread_options options{CTGPU_RNG_HASH, 0, 0, 1, 0, 0u, 0};
    ctgpu_stats last_stats{};
    bool in_flight = false;               // async: a frame has been queued and not yet seen finished
    uint64_t frames_finished = 0;
};

BossState &boss() {
    static BossState s;
    return s;
}

// raythread.cpp:388-434
bool drain_keyboard(environment_t *env, BossState &b, v3_t *pos) {
    bool changed = false;
    const float step = (float)(M_PI_4 / 4);
    if (!env->events.queue) return false;
    for (event_t *e = ReadEvent(&env->events); e; e = ReadEvent(&env->events)) {
        if (e->type != ET_KEY_DOWN) continue;
        switch (e->value) {
            case 'w': pos->y += 0.1; changed = true; break;
            case 's': pos->y -= 0.1; changed = true; break;
            case 'd': pos->x += 0.1; changed = true; break;
            case 'a': pos->x -= 0.1; changed = true; break;
            case 'i': pos->z += 0.1; changed = true; break;
            case 'o': pos->z -= 0.1; changed = true; break;
            case 'r': b.roll = (float)((double)b.roll + (M_PI_4 / 4)); changed = true; break;
            case 'y': b.yaw = (float)((double)b.yaw + (M_PI_4 / 4)); changed = true; break;
            case 'p': b.pitch = (float)((double)b.pitch + (M_PI_4 / 4)); changed = true; break;
            case 'c': *pos = v3_t{0, 0, 0}; b.yaw = b.pitch = b.roll = 0; changed = true; break;
            case 'm':
                std::fprintf(stderr, "camera position: %f %f %f\n", pos->x, pos->y, pos->z);
                changed = true;
                break;
            default: break;
        }
    }
    (void)step;
    return changed;
}

// raythread.cpp:564-572: products of binary32 cos/sin, stored to double
void camera_matrix(const BossState &b, m3x3_t *m) {
    const float cy = std::cos(b.yaw), sy = std::sin(b.yaw);
    const float cp = std::cos(b.pitch), sp = std::sin(b.pitch);
    const float cr = std::cos(b.roll), sr = std::sin(b.roll);
    m->data[0][0] = cy * cp;
    m->data[0][1] = cy * sp * sr - sy * cr;
    m->data[0][2] = cy * sp * cr + sy * sr;
    m->data[1][0] = sy * cp;
    m->data[1][1] = sy * sp * sr + cy * cr;
    m->data[1][2] = sy * sp * cr - cy * sr;
    m->data[2][0] = -sy;
    m->data[2][1] = cp * sr;
    m->data[2][2] = cp * cr;
}

}  // namespace

uint32_t ct::build_bvh_on_device(int device, const double *vertices, uint32_t n, void *nodes, uint32_t *indexes) {
    uint32_t used = 0;
    if (gpu().build_bvh(device, vertices, n, (ctgpu_node *)nodes, indexes, &used, nullptr) != CTGPU_OK)
        die("ctgpu_build_bvh", gpu().last_error());
    return used;
}

void RayThreadSetOptions(const ct_raythread_options *opt) {
    if (opt) boss().options = *opt;
}

namespace { void release_query_tree(); }

void RayThreadReset(void) {
    release_query_tree();
    BossState &b = boss();
    if (b.in_flight && !b.ctxs.empty()) (void)gpu().render_poll(b.ctxs[0], 1, nullptr, nullptr);
    for (ctgpu_ctx *c : b.ctxs) gpu().destroy(c);
    delete b.packed;
    const ct_raythread_options keep = b.options;
    b = BossState();
    b.options = keep;
}

extern "C" void cth_last_frame_stats(ctgpu_stats *out) { if (out) *out = boss().last_stats; }

bool RayThreadBusy(void) { return boss().in_flight; }
uint64_t RayThreadFramesFinished(void) { return boss().frames_finished; }

bool RayThread(environment_t *env, scene_t *scene) {
    BossState &b = boss();
    GpuApi &api = gpu();

    if (!b.initialised) {
        b.packed = new ct::PackedScene();
        // BuildBVH (raythread.cpp:647-652) runs on the GPU that will walk the tree and the tree stays there
        // (ctgpu_upload_scene without nodes: vertices in, a few milliseconds for the dragon instead of the host's
        // 60-150 and no half gigabyte of copies).  CT_BVH_DEVICE=0: the host builder; CT_BVH_DEVICE=copy: built on
        // the device, arrays brought back and uploaded (round 2's path).
        const char *where = std::getenv("CT_BVH_DEVICE");
        ct::PackedScene::TreeSource tree = ct::PackedScene::kNoTree;
        if (where && std::strcmp(where, "0") == 0) tree = ct::PackedScene::kHostTree;
        else if (where && std::strcmp(where, "copy") == 0) tree = ct::PackedScene::kDeviceTreeCopy;
        b.packed->build_from(scene, tree, b.options.device);
        // one context per GPU, the scene replicated on each (it is small against 288 GB): the GPUs
        // take the place of the reference's worker threads (raythread.cpp:596-618)
        int visible = 0;
        if (api.device_count(&visible) != CTGPU_OK || visible < 1) die("ctgpu_device_count", api.last_error());
        int wanted = b.options.n_devices > 1 ? b.options.n_devices : 1;
        if (const char *env = std::getenv("CT_RAYTHREAD_DEVICES")) wanted = std::max(1, std::atoi(env));
        ctgpu_scene_desc desc = b.packed->desc();
        std::vector<int> device_of;
        for (int i = 0; i < wanted; i++) {
            ctgpu_ctx *ctx = nullptr;
            const int device = (b.options.device + i) % visible;
            if (api.create(device, &ctx) != CTGPU_OK) die("ctgpu_create", api.last_error());
            // contexts that land on a device which already holds the scene take that copy
            ctgpu_ctx *holder = nullptr;
            for (size_t k = 0; k < b.ctxs.size(); k++)
                if (device_of[k] == device) { holder = b.ctxs[k]; break; }
            if (holder) {
                if (api.share_scene(ctx, holder) != CTGPU_OK) die("ctgpu_share_scene", api.last_error());
            } else if (api.upload_scene(ctx, &desc) != CTGPU_OK) {
                if (desc.nodes) die("ctgpu_upload_scene", api.last_error());
                // the device build refuses what it is not proven for (a NaN vertex): the host builder takes over
                std::fprintf(stderr, "cobbletrace: %s -- building the tree on the host\n", api.last_error());
                b.packed->build_tree_on_host();
                desc = b.packed->desc();
                if (api.upload_scene(ctx, &desc) != CTGPU_OK) die("ctgpu_upload_scene", api.last_error());
            }
            // RayThread gets env on every call and cannot know how long a bitmap lives (a harness may pass a fresh array
            // per frame): the page-lock lasts one frame.  An application whose bitmap is one allocation for its whole
            // life (cobbletrace.cpp:57-75; cobbletrace_headless, cobbletrace_viewer) says so and keeps it between frames.
            if (b.options.keep_bitmap_pinned && api.keep_pinned(ctx, 1) != CTGPU_OK) die("ctgpu_keep_framebuffer_pinned", api.last_error());
            b.ctxs.push_back(ctx);
            device_of.push_back(device);
        }
        b.initialised = true;
    }

    // HandleUpdates (raythread.cpp:546-593): keys are drained on every call; a new frame is dispatched only
    // when the previous one has finished AND something changed
    b.changes_pending = drain_keyboard(env, b, &scene->camera.position) || b.changes_pending;
    if (b.in_flight) {
        int done = 0;
        ctgpu_stats stats;
        if (api.render_poll(b.ctxs[0], 0, &done, &stats) != CTGPU_OK) die("ctgpu_render_poll", api.last_error());
        if (!done) return true;
        b.in_flight = false;
        b.last_stats = stats;
        b.frames_finished++;
    }
    if (!b.changes_pending) return true;
    b.changes_pending = false;
    camera_matrix(b, &scene->camera.rotation);

    ctgpu_frame_desc frame;
    std::memset(&frame, 0, sizeof frame);
    frame.width = env->bitmap->width;
    frame.height = env->bitmap->height;
    frame.num_threads = scene->settings.numberOfThreads;
    frame.flags = (scene->settings.supersampling ? CTGPU_F_SUPERSAMPLE : 0u) |
                  (scene->settings.subsampling ? CTGPU_F_SUBSAMPLE : 0u) |
                  (b.options.frame_flags & (CTGPU_F_SPHERES | CTGPU_F_REFLECT_FIX));
    frame.cam_pos[0] = scene->camera.position.x;
    frame.cam_pos[1] = scene->camera.position.y;
    frame.cam_pos[2] = scene->camera.position.z;
    std::memcpy(frame.cam_rot, scene->camera.rotation.data, sizeof frame.cam_rot);
    frame.rng_mode = b.options.rng_mode;
    frame.rng_seed = (b.options.rng_mode == CTGPU_RNG_LIBC && b.options.rng_seed == 0) ? 1 : b.options.rng_seed;
    frame.rng_offset = b.rand_consumed;
    frame.max_depth = 10;                 // TraceRay(..., 10, ...) raythread.cpp:474,508

    if (b.ctxs.size() == 1) {
        // (begin / poll also when the caller wants the frame before RayThread returns: the bitmap is page-locked for
        //  the frame and the pixels are copied straight into it; ctgpu_render would stage a pageable bitmap and copy it over)
        if (api.render_begin(b.ctxs[0], &frame, (uint32_t *)env->bitmap->memory) != CTGPU_OK)
            die("ctgpu_render_begin", api.last_error());
        b.in_flight = true;
        if (!b.options.async) {
            int done = 0;
            ctgpu_stats stats;
            if (api.render_poll(b.ctxs[0], 1, &done, &stats) != CTGPU_OK) die("ctgpu_render_poll", api.last_error());
            b.in_flight = false;
            b.last_stats = stats;
            b.frames_finished++;
        }
    } else {
        ctgpu_stats stats;
        if (api.render_multi(b.ctxs.data(), (int)b.ctxs.size(), &frame, (uint32_t *)env->bitmap->memory, &stats) != CTGPU_OK)
            die("ctgpu_render_multi", api.last_error());
        b.last_stats = stats;
        b.frames_finished++;
    }
    if (scene->settings.supersampling) {
        // one worker's rand() calls per frame: every column of the square x every traced row
        // (the off-bitmap row included) x 16 samples x (rx, ry)   raythread.cpp:455-471
        const uint64_t half = (uint64_t)(frame.height / 2), q = (uint64_t)(frame.height / frame.num_threads);
        uint64_t traced_per_band = q;
        if (scene->settings.subsampling) traced_per_band = (q + 1) / 2 + ((q % 2 == 0 && q >= 2) ? 1 : 0);
        b.rand_consumed += 2 * half * traced_per_band * (uint64_t)frame.num_threads * 32u;
    }
    return true;
}

// ---- the operator-level boundary: bvh.h:31-32
namespace {

// the tree of the bvh_state_t queried last, resident on the GPU (one context, re-uploaded when another tree is asked about)
struct QueryTree {
    ctgpu_ctx *ctx = nullptr;
    const bvh_node_t *nodes = nullptr;
    const triangle_t *triangles = nullptr;
    uint32_t nodes_used = 0, size = 0;
    uint64_t fingerprint = 0;             // of the arrays' CONTENT, all of it (tree_fingerprint)
};

// A hash of EVERYTHING the walk reads -- every node's box, links and count, every index entry, every triangle's three
// vertices -- so that a tree edited, refitted or rebuilt in place, or another tree at a freed one's addresses, is uploaded again
// (the reference walks whatever is in memory at call time, bvh.cpp:181-222).  Round 4 hashed a sample of ~257 elements and
// answered from the old copy when an edit missed all of them; a word-at-a-time multiply-xorshift over the whole tree is
// ~0.1 ms per megabyte on one core -- nothing for the trees this one-ray-a-call boundary is used with (BvhTest's 256
// triangles: bvh.cpp:223-310), 20 ms a call at the dragon's size, which is what being right costs there.
uint64_t tree_fingerprint(const bvh_state_t *state) {
    uint64_t h = 0x9ae16a3b2f90404full;
    auto mix = [&h](uint64_t w) { h = (h ^ w) * 0x9e3779b97f4a7c15ull; h ^= h >> 29; };
    auto word = [](const void *p) { uint64_t w; std::memcpy(&w, p, 8); return w; };
    for (uint32_t i = 0; i < state->nodesUsed; i++) {                // (the fields, not the struct's padding)
        const bvh_node_t &n = state->bvhNodes[i];
        const double box[6] = {n.aabbMin.x, n.aabbMin.y, n.aabbMin.z, n.aabbMax.x, n.aabbMax.y, n.aabbMax.z};
        for (int k = 0; k < 6; k++) mix(word(&box[k]));
        mix((uint64_t)n.leftNode | ((uint64_t)n.firstTriangleIndex << 32));
        mix((uint64_t)n.triangleCount);
    }
    for (uint32_t i = 0; i < state->triangles.size; i++) {
        const triangle_t &t = state->triangles.data[i];
        const double v[9] = {t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z, t.p3.x, t.p3.y, t.p3.z};
        for (int k = 0; k < 9; k++) mix(word(&v[k]));
        mix((uint64_t)state->triangles.indexes[i]);
    }
    return h;
}

QueryTree &query_tree() {
    static QueryTree q;
    return q;
}

void release_query_tree() {
    QueryTree &q = query_tree();
    if (q.ctx) gpu().destroy(q.ctx);
    q = QueryTree();
}

ctgpu_ctx *resident(bvh_state_t *state) {
    QueryTree &q = query_tree();
    GpuApi &api = gpu();
    const uint64_t print = tree_fingerprint(state);
    if (q.ctx && q.nodes == state->bvhNodes && q.triangles == state->triangles.data && q.nodes_used == state->nodesUsed &&
        q.size == state->triangles.size && q.fingerprint == print)
        return q.ctx;
    if (!q.ctx && api.create(boss().options.device, &q.ctx) != CTGPU_OK) die("ctgpu_create", api.last_error());
    const uint32_t n = state->triangles.size;
    std::vector<double> vertices((size_t)n * 9);
    for (uint32_t i = 0; i < n; i++) {
        const triangle_t &t = state->triangles.data[i];
        const double v[9] = {t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z, t.p3.x, t.p3.y, t.p3.z};
        std::memcpy(&vertices[(size_t)i * 9], v, sizeof v);
    }
    std::vector<ctgpu_material> materials(n, ctgpu_material{0x00ffffffu, -1, 0.0f});
    ctgpu_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_triangles = n;
    d.n_nodes = state->nodesUsed;
    d.vertices = vertices.data();
    d.materials = materials.data();
    d.nodes = reinterpret_cast<const ctgpu_node *>(state->bvhNodes);
    d.indexes = state->triangles.indexes;
    if (api.upload_scene(q.ctx, &d) != CTGPU_OK) die("ctgpu_upload_scene", api.last_error());
    q.nodes = state->bvhNodes; q.triangles = state->triangles.data; q.nodes_used = state->nodesUsed; q.size = n;
    q.fingerprint = print;
    return q.ctx;
}

ctgpu_hit one_ray(ray_t *ray, uint32_t node_index, bvh_state_t *state) {
    if (!ray || !state || !state->bvhNodes || state->triangles.size == 0) die("IntersectBVH", "null ray or empty tree");
    if (node_index != state->rootNodeIdx || state->rootNodeIdx != 0)
        die("IntersectBVH", "only whole-tree queries (nodeIdx = rootNodeIdx = 0) are supported by the GPU walk");
    ctgpu_ray r;
    std::memset(&r, 0, sizeof r);
    r.origin[0] = ray->origin.x; r.origin[1] = ray->origin.y; r.origin[2] = ray->origin.z;
    r.direction[0] = ray->direction.x; r.direction[1] = ray->direction.y; r.direction[2] = ray->direction.z;
    r.t = ray->t;
    ctgpu_hit h;
    if (gpu().trace_rays_packet(resident(state), &r, 1, 0u, &h) != CTGPU_OK) die("ctgpu_trace_rays_packet", gpu().last_error());
    return h;
}

}  // namespace

void IntersectBVH(ray_t *ray, uint32_t nodeIdx, bvh_state_t *bvhState) {
    ray->t = one_ray(ray, nodeIdx, bvhState).t;                  // every IntersectTriangle's effect on ray.t (bvh.cpp:181-196)
}

void IntersectBVHClosest(ray_t *ray, uint32_t nodeIdx, bvh_state_t *bvhState, float *tclosest, uint32_t *closestIndex) {
    if (!tclosest || !closestIndex) die("IntersectBVHClosest", "null output");
    if (*tclosest != 4294967296.0f) die("IntersectBVHClosest", "*tclosest must be 4294967296.f on entry, as ClosestIntersection sets it");
    const ctgpu_hit h = one_ray(ray, nodeIdx, bvhState);
    ray->t = h.t;
    if (h.tclosest != 4294967296.0f) {                           // a triangle was recorded (bvh.cpp:212-215)
        *tclosest = h.tclosest;
        *closestIndex = h.triangle;
    }
}
