// capi.cpp -- flat C entry points over the C++ host layer, for the Python test and
// bench harness (ctypes).  Thin: every function forwards to the reference-shaped API
// declared in include/cobbletrace_host.h.
#include "cobbletrace_host.h"
#include "host_internal.h"
#include "scene_pack.h"
#include "ct_tiles.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Handle {
    scene_t scene;
    bool owns_scene = false;
    ct::PackedScene packed;
    environment_t env;
    bitmapSettings_t bitmap;
    std::vector<event_t> events;
};

Handle *H(void *p) { return static_cast<Handle *>(p); }

void ensure_packed(Handle *h) {
    // the harness builds on the host unless told otherwise (CT_BVH_DEVICE=1: ctgpu_build_bvh on GPU 0)
    const char *where = std::getenv("CT_BVH_DEVICE");
    if (!h->packed.built)
        h->packed.build_from(&h->scene, (where && std::atoi(where) != 0) ? ct::PackedScene::kDeviceTreeCopy : ct::PackedScene::kHostTree, 0);
}

}  // namespace

#include "build_id.h"
extern "C" {

// which sources libcthost.so was made from (see include/ctgpu.h: ctgpu_build_id)
const char *cth_build_id(void) { return CTH_BUILD_ID; }

void *cth_load_scene(const char *path) {
    Handle *h = new Handle();
    std::memset(&h->scene, 0, sizeof h->scene);
    InitSceneData(&h->scene);
    std::vector<char> p(path, path + std::strlen(path) + 1);
    ParseSceneFile(p.data(), &h->scene);
    h->owns_scene = true;
    return h;
}

void *cth_scene_from_vertices(const double *verts, uint32_t n) {
    Handle *h = new Handle();
    std::memset(&h->scene, 0, sizeof h->scene);
    h->packed.build_from_vertices(verts, n);
    return h;
}

void cth_free_scene(void *p) {
    Handle *h = H(p);
    if (!h) return;
    if (h->owns_scene) {
        std::free(h->scene.lightStack.lights);
        std::free(h->scene.objectStack.objects);
        std::free(h->scene.triangleLookup.indexes);
    }
    delete h;
}

scene_t *cth_scene_ptr(void *p) { return &H(p)->scene; }

int cth_object_count(void *p) { return H(p)->scene.objectStack.index; }
int cth_light_count(void *p) { return H(p)->scene.lightStack.index; }

void cth_get_settings(void *p, int *threads, int *subsampling, int *wireframe, int *supersampling) {
    const settings_t &s = H(p)->scene.settings;
    *threads = s.numberOfThreads; *subsampling = s.subsampling; *wireframe = s.wireframe;
    *supersampling = s.supersampling;
}

void cth_set_settings(void *p, int threads, int subsampling, int supersampling) {
    settings_t &s = H(p)->scene.settings;
    if (threads > 0) s.numberOfThreads = threads;
    if (subsampling >= 0) s.subsampling = subsampling != 0;
    if (supersampling >= 0) s.supersampling = supersampling != 0;
}

void cth_get_camera(void *p, double pos[3]) {
    const v3_t &c = H(p)->scene.camera.position;
    pos[0] = c.x; pos[1] = c.y; pos[2] = c.z;
}

int cth_scene_desc(void *p, ctgpu_scene_desc *out) {
    Handle *h = H(p);
    ensure_packed(h);
    *out = h->packed.desc();
    return 0;
}

uint32_t cth_tree_depth(void *p) { ensure_packed(H(p)); return H(p)->packed.tree_depth(); }

// Known-answer hashes, defined as in SURVEY.md 8(c).
uint64_t cth_triangle_hash(void *p) {
    Handle *h = H(p);
    ct::Fnv64 f;
    for (int i = 0; i < h->scene.objectStack.index; i++) {
        const scene_object_t &o = h->scene.objectStack.objects[i];
        if (o.type != OT_TRIANGLE) continue;
        f.bytes(&o.triangle.p1, 24); f.bytes(&o.triangle.p2, 24); f.bytes(&o.triangle.p3, 24);
        f.bytes(&o.material.color, 4); f.bytes(&o.material.specular, 4); f.bytes(&o.material.reflection, 4);
    }
    return f.h;
}

uint64_t cth_node_hash(void *p) {
    Handle *h = H(p);
    ensure_packed(h);
    ct::Fnv64 f;
    for (uint32_t i = 0; i < h->packed.bvh.nodesUsed; i++) {
        const bvh_node_t &n = h->packed.bvh.bvhNodes[i];
        f.bytes(&n.aabbMin, 24); f.bytes(&n.aabbMax, 24);
        f.bytes(&n.leftNode, 4); f.bytes(&n.firstTriangleIndex, 4); f.bytes(&n.triangleCount, 4);
    }
    return f.h;
}

uint64_t cth_index_hash(void *p) {
    Handle *h = H(p);
    ensure_packed(h);
    ct::Fnv64 f;
    f.bytes(h->packed.bvh.triangles.indexes, (size_t)h->packed.bvh.triangles.size * 4);
    return f.h;
}

// ---- RayThread driven from the harness: the caller's buffer becomes env->bitmap->memory
void cth_raythread_options(int rng_mode, uint64_t seed, int device, int n_devices) {
    const ct_raythread_options o{rng_mode, seed, device, n_devices, 0, 0u, 0};
    RayThreadSetOptions(&o);
}

void cth_raythread_options_async(int rng_mode, uint64_t seed, int device, int async) {
    const ct_raythread_options o{rng_mode, seed, device, 1, async, 0u, 0};
    RayThreadSetOptions(&o);
}

void cth_raythread_options_ex(int rng_mode, uint64_t seed, int device, int n_devices, int async, unsigned frame_flags) {
    const ct_raythread_options o{rng_mode, seed, device, n_devices, async, frame_flags, 0};
    RayThreadSetOptions(&o);
}

// The application loop of cobbletrace.cpp:90-120 without the window: per iteration deliver at most one key,
// call RayThread, show the bitmap to Blit, until every key has been delivered and no frame is in flight.
// Returns the number of iterations; *polls = iterations in which RayThread found its frame still in flight.
int cth_app_loop(void *p, int width, int height, uint32_t *fb, const char *keys, int *polls) {
    Handle *h = H(p);
    h->bitmap = bitmapSettings_t{fb, width, height, nullptr};
    h->events.assign(64, event_t{});
    std::memset(&h->env, 0, sizeof h->env);
    h->env.bitmap = &h->bitmap;
    h->env.clientWidth = width * 2;
    h->env.clientHeight = height * 2;
    h->env.events.queue = h->events.data();
    h->env.events.capacity = (uint32_t)h->events.size();
    int iterations = 0, waiting = 0;
    const char *k = keys;
    for (;;) {
        const bool was_busy = RayThreadBusy();
        if (!was_busy && iterations > 0 && k && *k) AddEvent(&h->env.events, event_t{ET_KEY_DOWN, EM_NONE, v2_t{0, 0}, (uint32_t)(unsigned char)*k++});
        RayThread(&h->env, &h->scene);
        Blit(&h->env);
        iterations++;
        if (was_busy && RayThreadBusy()) waiting++;
        if (!RayThreadBusy() && !(k && *k) && iterations > 1) break;
        if (iterations > 100000000) break;
    }
    if (polls) *polls = waiting;
    return iterations;
}

uint64_t cth_blit_count(void) { return BlitFrameCount(); }
uint64_t cth_frames_finished(void) { return RayThreadFramesFinished(); }

void cth_blit(uint32_t *fb, int width, int height) {
    bitmapSettings_t bitmap{fb, width, height, nullptr};
    environment_t env{};
    env.bitmap = &bitmap;
    Blit(&env);
}

void cth_raythread_reset(void) { RayThreadReset(); }

// IntersectBVHClosest (closest != 0) / IntersectBVH on n rays of the scene's own tree, one call each, as
// ClosestIntersection makes it (raythread.cpp:204-227): rays / hits are ctgpu_ray / ctgpu_hit records
int cth_intersect_bvh(void *p, const void *rays_in, uint32_t n, int closest, void *hits_out) {
    Handle *h = H(p);
    ensure_packed(h);
    const ctgpu_ray *rays = static_cast<const ctgpu_ray *>(rays_in);
    ctgpu_hit *hits = static_cast<ctgpu_hit *>(hits_out);
    for (uint32_t i = 0; i < n; i++) {
        ray_t r;
        r.origin = v3_t{rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]};
        r.direction = v3_t{rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]};
        r.t = rays[i].t;
        float tclosest = 4294967296.0f;
        uint32_t index = 0;
        if (closest) IntersectBVHClosest(&r, h->packed.bvh.rootNodeIdx, &h->packed.bvh, &tclosest, &index);
        else IntersectBVH(&r, h->packed.bvh.rootNodeIdx, &h->packed.bvh);
        hits[i].t = r.t; hits[i].tclosest = tclosest; hits[i].triangle = index; hits[i].found = r.t != 1e30f;
    }
    return 0;
}

// The scene's own tree REBUILT IN PLACE after moving every triangle by (dx, 0, 0): same arrays, same counts, other
// content -- what a caller of the operator-level boundary may do between two queries (bvh.h:28-32 keeps no copy)
int cth_move_tree_in_place(void *p, double dx) {
    Handle *h = H(p);
    ensure_packed(h);
    bvh_state_t &b = h->packed.bvh;
    for (uint32_t i = 0; i < b.triangles.size; i++) {
        triangle_t &t = b.triangles.data[i];
        t.p1.x += dx; t.p2.x += dx; t.p3.x += dx;
        b.triangles.indexes[i] = i;
    }
    b.nodesUsed = 1;
    BuildBVH(&b);
    return (int)b.nodesUsed;
}

// ONE triangle of the scene's own tree moved by (0, 0, dz) where it lies -- no rebuild, no refit, nothing else touched:
// the smallest edit a caller of the operator-level boundary can make between two queries
int cth_move_one_triangle(void *p, uint32_t triangle, double dz) {
    Handle *h = H(p);
    ensure_packed(h);
    bvh_state_t &b = h->packed.bvh;
    if (triangle >= b.triangles.size) return -1;
    triangle_t &t = b.triangles.data[triangle];
    t.p1.z += dz; t.p2.z += dz; t.p3.z += dz;
    return 0;
}

// First call renders; each byte of `keys` is then delivered as one key-down event followed
// by one more RayThread call (one frame per key).  Returns the number of RayThread calls.
int cth_raythread_frames(void *p, int width, int height, uint32_t *fb, const char *keys) {
    Handle *h = H(p);
    h->bitmap = bitmapSettings_t{fb, width, height, nullptr};
    h->events.assign(64, event_t{});
    std::memset(&h->env, 0, sizeof h->env);
    h->env.bitmap = &h->bitmap;
    h->env.clientWidth = width * 2;
    h->env.clientHeight = height * 2;
    h->env.events.queue = h->events.data();
    h->env.events.capacity = (uint32_t)h->events.size();
    int calls = 0;
    RayThread(&h->env, &h->scene);
    calls++;
    for (const char *k = keys; k && *k; k++) {
        AddEvent(&h->env.events, event_t{ET_KEY_DOWN, EM_NONE, v2_t{0, 0}, (uint32_t)(unsigned char)*k});
        RayThread(&h->env, &h->scene);
        calls++;
    }
    return calls;
}

// Host-side counterpart of the ct_assemble kernel, for tile buffers that arrive in HOST memory (ranks on
// different nodes gathering over MPI / gloo): `gathered` holds the part_count partitions' compact buffers back
// to back, as ctgpu_render_partition leaves them; fb gets exactly the pixels ct_assemble would write.
// The layout arithmetic is the one definition both share, include/ct_tiles.h.
int cth_assemble_tiles(const ctgpu_frame_desc *frame, int tile_rows, int part_count, const uint32_t *gathered, uint32_t *fb) {
    if (!frame || !gathered || !fb || tile_rows < 1 || part_count < 1 || frame->width <= 0 || frame->height <= 1 ||
        frame->num_threads <= 0 || frame->num_threads > frame->height)
        return -1;
    ct_tile_layout L;
    L.width = frame->width; L.height = frame->height; L.half = frame->height / 2;
    L.square = (uint32_t)(2 * L.half);
    L.rows_per_band = (uint32_t)(frame->height / frame->num_threads);
    L.rendered_rows = (uint32_t)frame->num_threads * L.rows_per_band;
    L.subsample = (frame->flags & CTGPU_F_SUBSAMPLE) ? 1u : 0u;
    L.tile_rows = (uint32_t)tile_rows; L.part_count = (uint32_t)part_count;
    L.local_rows = ct_tiles_local_rows(L.rendered_rows, L.tile_rows, L.part_count);
    for (uint32_t yi = 0; yi < L.rendered_rows; yi++)
        for (uint32_t xi = 0; xi < L.square; xi++) {
            uint64_t slot;
            if (ct_tiles_bitmap_slot(&L, xi, yi, &slot)) fb[slot] = ct_tiles_pixel(&L, gathered, xi, yi);
        }
    return 0;
}

// The same script of ring operations tests run on the reference's eventQueue.cpp (oracle/ref_headers_driver.cpp):
// op >= 0 -> AddEvent of a key-down with that value; op == -1 -> ReadEvent, value (0xffffffff for NULL) appended.
uint32_t cth_event_queue_script(uint32_t capacity, const int32_t *ops, uint32_t n, uint32_t *out) {
    std::vector<event_t> ring(capacity);
    eventManager_t em{};
    em.queue = ring.data();
    em.capacity = capacity;
    uint32_t written = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (ops[i] >= 0) {
            AddEvent(&em, event_t{ET_KEY_DOWN, EM_NONE, v2_t{0, 0}, (uint32_t)ops[i]});
        } else {
            event_t *e = ReadEvent(&em);
            out[written++] = e ? e->value : 0xffffffffu;
        }
    }
    return written;
}

void cth_get_rotation(void *p, double rot[9]) { std::memcpy(rot, H(p)->scene.camera.rotation.data, 72); }

uint64_t cth_fnv1a64(const void *data, uint64_t n) {
    ct::Fnv64 f;
    f.bytes(data, (size_t)n);
    return f.h;
}

}  // extern "C"
