// ctgpu_api.hip -- host side of the C ABI declared in include/ctgpu.h.
//
// Owns the device copies of the scene (re-laid-out for coalesced 128-byte node lines and
// leaf-ordered triangle records, see ct_device.h), the per-frame scratch buffers, and the
// launches of ct_kernels.hip.  No compute happens on the host: without a HIP device every
// entry point that needs one returns an error.
#include "ctgpu_internal.h"
#include "ct_tiles.h"
#include "build_id.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace ctd;

// sizes the bindings (cobbletrace_amd/bindings.py) rely on
static_assert(sizeof(ctgpu_node) == 64 && sizeof(ctgpu_light) == 56 && sizeof(ctgpu_material) == 12, "scene records");
static_assert(sizeof(ctgpu_scene_desc) == 88 && sizeof(ctgpu_sphere) == 48 && sizeof(ctgpu_frame_desc) == 136 && sizeof(ctgpu_stats) == 112, "descriptors");
static_assert(sizeof(ctgpu_ray) == 56 && sizeof(ctgpu_hit) == 16 && sizeof(ctgpu_partition) == 12, "test-hook records");

// ------------------------------------------------------------------ error plumbing
static thread_local std::string g_last_error;

int ctgpu_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return ctgpu_fail(CTGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

static void ctgpu_own_pin_forget(const void *p);     // (the registry of this library's own page-locks, further down)

// ------------------------------------------------------------------ context
template <typename T>
struct DeviceArray {
    T *ptr = nullptr;
    size_t count = 0;
    hipError_t resize(size_t n) {
        if (n <= count && ptr) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; count = 0;
        hipError_t e = hipMalloc((void **)&ptr, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) count = n;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; count = 0; }
    void adopt(T *p, size_t n) { release(); ptr = p; count = n; }      // takes over a hipMalloc'ed block
};

// A scene resident in one device's HBM.  Immutable once packed, so the contexts of one device (frames in flight,
// ranks of a test on one GPU) can share ONE copy (ctgpu_share_scene): the dragon is ~0.5 GB per copy.
struct SceneStore {
    int device = 0;
    bool has_reflection = false;                // some material reflects (triangles' or spheres')
    bool triangles_reflect = false, spheres_reflect = false;
    uint32_t n_triangles = 0, n_lights = 0, tree_depth = 0, n_nodes = 0;
    uint32_t root_ref = 0, slot_of_triangle0 = 0;
    double root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};

    DeviceArray<WideNode> nodes;
    DeviceArray<QuadNode> quads, quads_flat;
    DeviceArray<double> quad_boxes;             // [quad][entry][6]: the folded tree's entries' binary64 boxes
    DeviceArray<uint32_t> quad_node_of_flat;    // [quad][entry]: the unfolded tree's entries' reference nodes (boxes: ref_nodes)
    uint32_t quad_root = 0, quad_root_flat = 0, quad_pending = 1;
    uint32_t n_quads = 0, n_quads_flat = 0;
    float plane_bound[3] = {0, 0, 0};
    DeviceArray<TriRecord> tris;
    // cull nodes (ct_device.h): a second copy of the folded tree with the cull nodes behind it, from cull_first on (what
    // supersampled frames walk; empty when the scene has none); the scene's magnitudes for the gate
    DeviceArray<QuadNode> quads_cull;
    uint32_t quad_root_cull = 0;
    uint32_t cull_first = kNoNode, n_cull = 0, n_big_leaves = 0, cull_levels = 0;
    float edge_bound = 0, vertex_bound = 0;
    bool cull_usable = false;
    // the shadow tree (ct_device.h, ct_shadow_tree.hip): what the frame's shadow rays walk instead of `quads`
    DeviceArray<QuadNode> stree;
    DeviceArray<TriRecord> stree_tris;          // the TriRecords once more, in that tree's own leaf order
    DeviceArray<uint32_t> leaf_of_slot;         // slot of stree_tris -> reference leaf node (the verdict of the leaf's box)
    ShadowTreeLimits stree_lim = {0, 0, 0};
    float stree_tau = 0.0f;
    uint32_t stree_pending = 0;
    double stree_gamma = 0.0, stree_ms = 0.0;
    bool stree_usable = false;
    DeviceArray<Material> materials;
    DeviceArray<Light> lights;
    DeviceArray<ctgpu_sphere> spheres;          // CTGPU_F_SPHERES
    DeviceArray<ctgpu_node> sphere_nodes;
    DeviceArray<uint32_t> sphere_indexes;
    uint32_t n_spheres = 0, sphere_tree_depth = 0;
    DeviceArray<uint32_t> shadow_lights;
    uint32_t n_shadow_lights = 0;
    // the reference's own arrays, kept on the device when the tree was built there (ctgpu_download_tree)
    DeviceArray<ctgpu_node> ref_nodes;
    DeviceArray<uint32_t> ref_indexes;
    double build_ms = 0.0, pack_ms = 0.0;       // device time of BuildBVH and of the packing kernels

    ~SceneStore() {
        (void)hipSetDevice(device);
        nodes.release(); quads.release(); quads_flat.release(); quad_boxes.release(); quad_node_of_flat.release();
        tris.release(); quads_cull.release(); stree.release(); stree_tris.release(); leaf_of_slot.release(); materials.release(); lights.release(); spheres.release(); sphere_nodes.release();
        sphere_indexes.release(); shadow_lights.release(); ref_nodes.release(); ref_indexes.release();
    }
};

struct ctgpu_ctx {
    int device = 0;
    ctgpu_config cfg;                           // what the context was created with (ctgpu_get_config); the fields below are set from it
    std::shared_ptr<SceneStore> scene;          // null until ctgpu_upload_scene / ctgpu_share_scene

    DeviceArray<uint32_t> tiles;       // compact tile buffer of a whole-frame render
    DeviceArray<uint32_t> fb;          // device framebuffer of ctgpu_render
    DeviceArray<uint32_t> gathered;    // ctgpu_render_multi: every device's tile buffer, back to back
    DeviceArray<unsigned long long> counters;   // 16 words
    DeviceArray<uint32_t> overflow;    // 1 word
    DeviceArray<int32_t> rand_table;   // LIBC jitter stream
    DeviceArray<int32_t> row_ordinal;
    uint64_t rand_key[4] = {~0ull, ~0ull, ~0ull, ~0ull};   // (seed, offset, count, rows) of the cached table
    std::vector<int32_t> rand_host, ordinal_host;          // host copies the asynchronous uploads read from

    // wavefront pipeline (ct_pipeline.hip)
    DeviceArray<float2> coords;
    DeviceArray<uint2> hits;
    DeviceArray<uint8_t> shadow;
    DeviceArray<unsigned long long> queue;      // 1 + 32 ray counters, then the hit count
    DeviceArray<unsigned long long> spill;      // stack entries beyond the LDS part
    DeviceArray<unsigned long long> walk_work;  // [64][8] fetch counters of the packet walks (ct_params.h)
    bool kernel_timing = false;                 // ctgpu_kernel_timing: events around every pipeline kernel
    std::vector<hipEvent_t> marks;              // (kPipelineKernels + 1) per timed frame
    size_t marks_used = 0;
    std::vector<hipEvent_t> phase_marks;        // ctgpu_render_sharded: 4 per timed frame (start, traced, gathered, assembled)
    size_t phase_marks_used = 0;
    int cu_count = 256;
    int walk_thresh[3] = {64, 32, 24};          // retire/fetch, inner, leaf (measured best on MI355X)
    int lds_depth = 8;                          // 6 KB of LDS per wave; deeper entries spill to HBM
    int blocks_per_cu = 0;                      // 0 = occupancy query
    int chunk = 0;                              // 0 = sized from the frame
    int chunk_shadow = 0;                       // 0 = same as chunk
    int packet_lds_levels = 14;                 // per-lane tmin levels kept in LDS by the packet walk
    int packet_budget = 0, packet_lanes = 12;   // CTGPU_PACKET_BUDGET: hand-over of incoherent packets to the per-lane walk (PipelineParams); 0 = off, the default
    DeviceArray<unsigned long long> handed_ids[2];
    // reflection bounces on the pipeline (BounceParams)
    DeviceArray<ctgpu_ray> b_rays, b_shadow_rays;
    DeviceArray<ctgpu_hit> b_hits, b_shadow_hits;
    DeviceArray<uint8_t> b_live, b_shadow_live, b_pending;
    DeviceArray<uint32_t> b_local, b_terminal, b_level_live;
    DeviceArray<float> b_refl;
    bool bounce_on_pipeline = true;             // CTGPU_BOUNCE=0: reflective frames take the general path (one kernel), =1: the bounce levels
    // Unset, reflective frames are measured on both and the faster is kept, per frame kind ([0] single sample, [1]
    // supersampled): the bounce levels win where there is work per level (cube scene at 16 spp: 1.2 against 4.8 ms),
    // the one kernel on small frames whose ten levels are mostly launches (scene_file_1 at 1 spp: 0.4 against 1.6 ms).
    // Same pixels either way.  Order: two frames on the pipeline, two on the general path, the second of each counts.
    bool bounce_auto = true;
    struct BounceChoice {
        uint32_t frames = 0;                    // reflective frames of this kind that reported a time
        bool use_general = false;               // the settled choice
        bool ran_general = false;               // what the frame in flight runs on
        double pipeline_ms = 0.0, general_ms = 0.0;
    } bounce_choice[2];
    bool frame_is_reflective = false;           // ... the frame in flight
    unsigned long long handed_over[2] = {0, 0}; // rays handed over since the last ctgpu_read_walk_work
    bool keep_shadow_rays = false;              // CTGPU_KEEP_SHADOW_RAYS=1: trace the shadow rays whose answer cannot matter, too
    bool packet_walk = true;                    // ct_walk_packet (CTGPU_PACKET=0: per-lane walk)
    // Which walk traces a frame kind ([0] single sample, [1] supersampled) when CTGPU_PACKET does not say: the shared
    // walk, unless this scene's packets turn out to share little (many fetches per packet) AND a frame traced per lane
    // was measured faster.  Both walks give the same pixels, so the choice is invisible; it is re-examined every 256
    // frames with one frame on the other walk.  Only frames that report their time (ctgpu_render, _begin / _poll) feed it.
    bool walk_auto = true;
    struct WalkChoice {
        uint32_t frames = 0;                    // frames of this kind finished
        bool use_lane = false, probe = false;   // the settled choice; the NEXT frame runs on the other walk, as a probe
        bool ran_lane = false, ran_probe = false;   // what the frame in flight runs on
        double packet_ms = 0.0, lane_ms = 0.0;  // last measured kernel time on either walk
        unsigned long long packets = 0, fetches = 0;        // walk_work totals at the last look (they only grow between reads)
    } choice[2];
    bool force_megakernel = false;
    // Heavy chunks first (OrderParams, ct_params.h): what the packets of each chunk of ray ids cost in the frame before
    // decides the order -- and, in frames with few packets per wave, the size -- in which this frame hands them out.
    // CTGPU_ORDER=0 turns it off; CTGPU_ORDER_HEAVY="a,b,floor": class 0 / 1 from a / b times the mean packet's fetches.
    bool order_on = true, order_always = false;         // (CTGPU_ORDER=2: also frames whose heavy chunks are not cut up)
    bool order_pipelined = false;                       // (CTGPU_ORDER=3: also frames the caller pipelines -- an experiment)
    float order_heavy[kOrderClasses] = {8.0f, 4.0f, 2.0f, 1.0f};
    int order_floor = 64;
    int order_parts[kOrderClasses] = {4, 2, 1, 1};      // parts a heavy chunk of class k is cut into (frames with few packets per wave)
    int order_split_below = 8;                  // chunks per wave of the grid below which heavy chunks are cut into parts
    struct OrderHistory {
        uint64_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the frame geometry the costs belong to
        bool valid = false, last_split = false, last_ordered = false;    // (last_*: what the context's latest frame did)
        int parity = 0;                         // cost[parity] is written by the next frame, cost[parity ^ 1] read
        DeviceArray<uint32_t> cost[2][2];       // [parity][walk]
        DeviceArray<unsigned long long> totals[2];      // [parity]: [walk][64][2]
        DeviceArray<uint8_t> listed[2];
        DeviceArray<uint4> list[2];
        DeviceArray<uint32_t> count;            // [walk][2]
        void release() {
            for (int a = 0; a < 2; a++) {
                for (int b = 0; b < 2; b++) cost[a][b].release();
                totals[a].release(); listed[a].release(); list[a].release();
            }
            count.release();
            valid = false;
        }
    } order;

    ctgpu_comm_state *comm = nullptr;           // ctgpu_comm_init: this rank's RCCL communicator
    ctgpu_comm_stamp comm_stamp;                // which of the communicator's frames this context issues (ctgpu_comm.hip)
    DeviceArray<unsigned long long> comm_tags;  // [2 + 2 * world]: my stamp, then (root) every rank's
    std::vector<void *> multi_comms;            // ctgpu_render_multi: one communicator per device, kept by the root context
    std::vector<int> multi_devices;
    bool multi_comm_failed = false;             // RCCL could not serve ctgpu_render_multi once: peer copies from then on

    struct Pending {                            // ctgpu_render_begin .. ctgpu_render_poll
        bool active = false;
        std::chrono::steady_clock::time_point t0;
        uint64_t pixels = 0;
        uint32_t *deliver_to = nullptr;         // the caller's pageable framebuffer the staged rectangle goes to at the end
        int width = 0, r0 = 0, r1 = -1, c0 = 0, c1 = -1;       // ... and that rectangle
        int kind = -1;                          // 0 / 1: a single-sample / supersampled frame on the fast path (walk choice)
        int reflective_kind = -1;               // 0 / 1: a reflective single-sample / supersampled frame (bounce choice)
    } pending;
    // The closest-hit walk of the shadow tree's nodes leaves a ray whose winner sits on the rim of its leaf's box to the walk of the
    // reference's tree; in scenes of axis-aligned flat geometry (a floor, the faces of the rounded cubes) that is every hit on such
    // a face, and the frame pays both walks.  A context that sees more than a sixteenth of a frame's primary rays left keeps the
    // reference's tree for its primary rays from then on (same pixels either way; a new scene starts over) -- and looks again
    // every kClosestRetry frames: the camera may have turned from the floor to the statue (one frame in 256 at the worse rate).
    bool closest_off = false;
    uint32_t closest_off_frames = 0;
    static constexpr uint32_t kClosestRetry = 256;
    struct Path { uint32_t walk_kind = 0, bounce_kind = 0, ordered = 0, cull_tree = 0, shadow_tree = 0, closest_tree = 0; } path;     // what the latest frame ran on (ctgpu_stats)
    int big_leaf_min = 8;                       // CTGPU_BIG_LEAF: leaves of at least this many triangles go through the triangle boxes (0: off)
    int big_leaf_share = 20;                    // CTGPU_BIG_LEAF_SHARE: ... and only in scenes where at least this percentage of all triangles sits in such leaves
    int probe_above = 100;                      // CTGPU_PROBE_ABOVE: fetches per packet beyond which the per-lane walk gets a trial frame
    // A framebuffer the caller did not page-lock is never the target of a copy queued on a stream: such a frame lands
    // in `stage` (page-locked, the context's own) and is copied into the caller's memory by the CPU once the frame
    // is over.  (Letting the runtime pin pageable memory behind an asynchronous copy goes wrong when that memory sits
    // in the C heap: the heap's top is given back to the system and grown again as arrays come and go, the runtime's
    // mapping of the old pages stays, and a later copy to the same addresses ends in "Memory access fault by GPU".)
    uint32_t *stage = nullptr;
    size_t stage_count = 0;
    uint32_t *pinned_fb = nullptr;              // the caller's framebuffer, page-locked by begin_frame for the asynchronous copy
    size_t pinned_bytes = 0;
    bool keep_pin = false;                      // ctgpu_keep_framebuffer_pinned: the caller promises one long-lived framebuffer

    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    int thresh[3] = {40, 20, 12};
};

// ------------------------------------------------------------------ glibc rand() stream
// The reference draws its supersample jitter from libc rand() (raythread.cpp:470-471).
// glibc's default generator is the TYPE_3 additive-feedback generator of random_r.c
// (glibc 2.35): state r[0..33] seeded by the Lehmer sequence 16807*x mod (2^31-1),
// 310 outputs discarded, then o_k = (r[k-31] + r[k-3]) >> 1 on 32-bit words.
extern "C" int ctgpu_rand_stream(uint32_t seed, uint64_t offset, uint64_t n, int32_t *out) {
    if (!out && n) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_rand_stream: null output");
    if (seed == 0) seed = 1;
    uint32_t r[34];
    r[0] = seed;
    for (int i = 1; i < 31; i++) {
        const int32_t prev = (int32_t)r[i - 1];
        const int32_t hi = prev / 127773, lo = prev % 127773;
        int32_t word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    // ring of the last 34 words
    uint32_t ring[34];
    for (int i = 0; i < 34; i++) ring[i] = r[i];
    uint64_t produced = 34;
    auto next = [&]() -> uint32_t {
        const uint32_t v = ring[(produced - 31) % 34] + ring[(produced - 3) % 34];
        ring[produced % 34] = v;
        produced++;
        return v;
    };
    for (int i = 34; i < 344; i++) (void)next();
    for (uint64_t i = 0; i < offset; i++) (void)next();
    for (uint64_t i = 0; i < n; i++) out[i] = (int32_t)(next() >> 1);
    return CTGPU_OK;
}

// ------------------------------------------------------------------ basic entry points
extern "C" int ctgpu_abi_version(void) { return CTGPU_ABI_VERSION; }
#ifndef CTGPU_BUILD_VARIANT
#define CTGPU_BUILD_VARIANT ""           // (diagnostic builds: a suffix that sets their id apart from the product's, Makefile)
#endif
extern "C" const char *ctgpu_build_id(void) { return CTGPU_BUILD_ID CTGPU_BUILD_VARIANT; }
extern "C" const char *ctgpu_last_error(void) { return g_last_error.c_str(); }

extern "C" int ctgpu_device_count(int *count) {
    if (!count) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return ctgpu_fail(CTGPU_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return CTGPU_OK;
}

static int create_resources(ctgpu_ctx *ctx);

// ------------------------------------------------------------------ configuration
static_assert(kOrderClasses == 4, "ctgpu_config lists four order classes");
extern "C" int ctgpu_default_config(ctgpu_config *c) {
    if (!c) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_default_config: null argument");
    std::memset(c, 0, sizeof *c);
    c->packet_walk = -1; c->bounce = -1;
    c->lds_depth = 8;                            // 6 KB of LDS per wave; deeper entries spill to HBM
    c->packet_lds_levels = 14;
    c->probe_above = 100;
    c->packet_lanes = 12;
    c->order = 1;
    const float heavy[4] = {8.0f, 4.0f, 2.0f, 1.0f};
    const int parts[4] = {4, 2, 1, 1};
    for (int k = 0; k < 4; k++) { c->order_heavy[k] = heavy[k]; c->order_parts[k] = parts[k]; }
    c->order_floor = 64; c->order_split_below = 8;
    c->big_leaf_min = 8; c->big_leaf_share = 20;
    c->shadow_tree = 2; c->shadow_tree_build = 0; c->shadow_tree_leaf = 2; c->shadow_tree_leaf_fill = 50;      // (2: every frame -- single-sample frames' shadow walks gain 3-15 % too, profiles/r05_single_sample.log)
    c->closest_tree = 1;
    c->shadow_split_gap = -1.0f;
    const int thresh[3] = {40, 20, 12}, walk_thresh[3] = {64, 32, 24};       // (measured best on MI355X)
    for (int k = 0; k < 3; k++) { c->thresh[k] = thresh[k]; c->walk_thresh[k] = walk_thresh[k]; }
    return CTGPU_OK;
}

extern "C" int ctgpu_config_from_environment(ctgpu_config *c) {
    int rc = ctgpu_default_config(c); if (rc) return rc;
    const auto number = [](const char *name, int32_t &field, int lo, int hi) {
        if (const char *t = std::getenv(name)) field = std::min(hi, std::max(lo, std::atoi(t)));
    };
    const auto triple = [](const char *name, int32_t field[3]) {
        int a, b, d;
        if (const char *t = std::getenv(name))
            if (std::sscanf(t, "%d,%d,%d", &a, &b, &d) == 3) { field[0] = a; field[1] = b; field[2] = d; }
    };
    const int big = 1 << 30;
    triple("CTGPU_THRESH", c->thresh);
    triple("CTGPU_WALK_THRESH", c->walk_thresh);
    number("CTGPU_LDS_DEPTH", c->lds_depth, 1, big);
    number("CTGPU_BLOCKS_PER_CU", c->blocks_per_cu, 0, big);
    number("CTGPU_CHUNK", c->chunk, 0, big);
    number("CTGPU_CHUNK_SHADOW", c->chunk_shadow, 0, big);
    if (const char *t = std::getenv("CTGPU_PACKET")) c->packet_walk = std::atoi(t) != 0 ? 1 : 0;
    if (const char *t = std::getenv("CTGPU_BOUNCE")) c->bounce = std::atoi(t) != 0 ? 1 : 0;
    if (const char *t = std::getenv("CTGPU_KEEP_SHADOW_RAYS")) c->keep_shadow_rays = std::atoi(t) != 0 ? 1 : 0;
    if (const char *t = std::getenv("CTGPU_FORCE_MEGAKERNEL")) c->force_megakernel = std::atoi(t) != 0 ? 1 : 0;
    number("CTGPU_PROBE_ABOVE", c->probe_above, 0, big);
    if (const char *t = std::getenv("CTGPU_SHADOW_SPLIT")) { float v; if (std::sscanf(t, "%f", &v) == 1 && v == v) c->shadow_split_gap = v; }
    number("CTGPU_BIG_LEAF", c->big_leaf_min, 0, big);
    number("CTGPU_BIG_LEAF_SHARE", c->big_leaf_share, 0, 100);
    number("CTGPU_SHADOW_TREE", c->shadow_tree, 0, 2);
    number("CTGPU_SHADOW_TREE_BUILD", c->shadow_tree_build, 0, 2);
    number("CTGPU_CLOSEST_TREE", c->closest_tree, 0, 2);
    number("CTGPU_SHADOW_TREE_LEAF", c->shadow_tree_leaf, 1, 4);
    number("CTGPU_SHADOW_TREE_LEAF_FILL", c->shadow_tree_leaf_fill, 0, 100);
    number("CTGPU_PACKET_BUDGET", c->packet_budget, 0, big);
    number("CTGPU_PACKET_LANES", c->packet_lanes, 1, 64);
    number("CTGPU_PACKET_LDS_LEVELS", c->packet_lds_levels, 1, big);
    number("CTGPU_ORDER", c->order, 0, 3);
    if (const char *t = std::getenv("CTGPU_ORDER_HEAVY")) {
        float v[4];
        if (std::sscanf(t, "%f,%f,%f,%f", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] >= v[1] && v[1] >= v[2] && v[2] >= v[3] && v[3] > 0.0f)
            for (int k = 0; k < 4; k++) c->order_heavy[k] = v[k];
    }
    number("CTGPU_ORDER_FLOOR", c->order_floor, 0, big);
    number("CTGPU_ORDER_SPLIT_BELOW", c->order_split_below, 0, big);
    if (const char *t = std::getenv("CTGPU_ORDER_PARTS")) {
        int v[4];
        const auto ok = [](int x) { return x == 1 || x == 2 || x == 4 || x == 8 || x == 16; };
        if (std::sscanf(t, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && ok(v[0]) && ok(v[1]) && ok(v[2]) && ok(v[3]))
            for (int k = 0; k < 4; k++) c->order_parts[k] = v[k];
    }
    if (const char *t = std::getenv("CTGPU_PACK")) c->pack_on_host = std::strcmp(t, "host") == 0 ? 1 : 0;
    c->debug = (std::getenv("CTGPU_DEBUG_PHASES") ? CTGPU_DEBUG_PHASES : 0u) | (std::getenv("CTGPU_DEBUG_TRI") ? CTGPU_DEBUG_TRI : 0u) |
               (std::getenv("CTGPU_DEBUG_REDO") ? CTGPU_DEBUG_REDO : 0u) | (std::getenv("CTGPU_DEBUG_PACKET") ? CTGPU_DEBUG_PACKET : 0u);
    return CTGPU_OK;
}

extern "C" int ctgpu_get_config(const ctgpu_ctx *ctx, ctgpu_config *out) {
    if (!ctx || !out) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_get_config: null argument");
    *out = ctx->cfg;
    return CTGPU_OK;
}

extern "C" int ctgpu_create_with_config(int device, const ctgpu_config *config, ctgpu_ctx **out) {
    if (!out || !config) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_create: null argument");
    *out = nullptr;
    const ctgpu_config &c = *config;
    const auto part_ok = [](int x) { return x == 1 || x == 2 || x == 4 || x == 8 || x == 16; };
    if (c.lds_depth < 1 || c.packet_lds_levels < 1 || c.packet_lanes < 1 || c.packet_lanes > 64 || c.blocks_per_cu < 0 || c.chunk < 0 ||
        c.chunk_shadow < 0 || c.order < 0 || c.order > 3 || c.big_leaf_min < 0 || c.big_leaf_share < 0 || c.big_leaf_share > 100 ||
        c.shadow_tree < 0 || c.shadow_tree > 2 || c.shadow_tree_build < 0 || c.shadow_tree_build > 2 ||
        c.closest_tree < 0 || c.closest_tree > 2 || c.shadow_tree_leaf < 1 || c.shadow_tree_leaf > 4 || c.shadow_tree_leaf_fill < 0 || c.shadow_tree_leaf_fill > 100 ||
        c.packet_budget < 0 || c.order_floor < 0 || c.order_split_below < 0 || c.probe_above < 0 ||
        !(c.order_heavy[0] >= c.order_heavy[1] && c.order_heavy[1] >= c.order_heavy[2] && c.order_heavy[2] >= c.order_heavy[3] && c.order_heavy[3] > 0.0f) ||
        !part_ok(c.order_parts[0]) || !part_ok(c.order_parts[1]) || !part_ok(c.order_parts[2]) || !part_ok(c.order_parts[3]))
        return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_create_with_config: a field is out of range (ctgpu_default_config gives valid values)");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ctgpu_fail(CTGPU_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= n) return ctgpu_fail(CTGPU_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    ctgpu_ctx *ctx = new ctgpu_ctx();
    ctx->device = device;
    ctx->cfg = c;
    const int rc = create_resources(ctx);
    if (rc != CTGPU_OK) { ctgpu_destroy(ctx); return rc; }      // (the message of the failing call stands)
    *out = ctx;
    return CTGPU_OK;
}

extern "C" int ctgpu_create(int device, ctgpu_ctx **out) {
    ctgpu_config c;
    int rc = ctgpu_config_from_environment(&c); if (rc) return rc;
    return ctgpu_create_with_config(device, &c, out);
}

static int create_resources(ctgpu_ctx *ctx) {
    const int device = ctx->device;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ctx->ev0));
    HIP_TRY(hipEventCreate(&ctx->ev1));
    HIP_TRY(hipEventCreate(&ctx->ev_done));
    HIP_TRY(ctx->counters.resize(kCounterWords));
    HIP_TRY(ctx->overflow.resize(1));
    HIP_TRY(hipMemset(ctx->counters.ptr, 0, kCounterWords * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(ctx->overflow.ptr, 0, sizeof(uint32_t)));
    {   // the context's working fields, from its configuration
        const ctgpu_config &c = ctx->cfg;
        for (int k = 0; k < 3; k++) { ctx->thresh[k] = c.thresh[k]; ctx->walk_thresh[k] = c.walk_thresh[k]; }
        ctx->lds_depth = c.lds_depth; ctx->blocks_per_cu = c.blocks_per_cu; ctx->chunk = c.chunk; ctx->chunk_shadow = c.chunk_shadow;
        ctx->walk_auto = c.packet_walk < 0; ctx->packet_walk = c.packet_walk != 0;
        ctx->keep_shadow_rays = c.keep_shadow_rays != 0; ctx->probe_above = c.probe_above;
        ctx->big_leaf_min = c.big_leaf_min; ctx->big_leaf_share = c.big_leaf_share;
        ctx->packet_budget = c.packet_budget; ctx->packet_lanes = c.packet_lanes; ctx->packet_lds_levels = c.packet_lds_levels;
        ctx->force_megakernel = c.force_megakernel != 0;
        ctx->bounce_auto = c.bounce < 0; ctx->bounce_on_pipeline = c.bounce != 0;
        ctx->order_on = c.order != 0; ctx->order_always = c.order == 2; ctx->order_pipelined = c.order == 3;
        for (int k = 0; k < kOrderClasses; k++) { ctx->order_heavy[k] = c.order_heavy[k]; ctx->order_parts[k] = c.order_parts[k]; }
        ctx->order_floor = c.order_floor; ctx->order_split_below = c.order_split_below;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->cu_count = prop.multiProcessorCount;
    HIP_TRY(ctx->queue.resize(queue_block_words(32)));       // (grows with the scene's light count, fill_pipeline)
    HIP_TRY(ctx->walk_work.resize(64 * 8));
    HIP_TRY(hipMemset(ctx->walk_work.ptr, 0, 64 * 8 * sizeof(unsigned long long)));
    return CTGPU_OK;
}

ctgpu_comm_state *&ctgpu_comm_slot(ctgpu_ctx *ctx) { return ctx->comm; }
ctgpu_comm_stamp &ctgpu_comm_stamp_of(ctgpu_ctx *ctx) { return ctx->comm_stamp; }
int ctgpu_comm_stamp_buffers(ctgpu_ctx *ctx, int world, bool root, unsigned long long **mine, unsigned long long **all, uint32_t **flag) {
    const size_t need = 2u + (root ? 2u * (size_t)world : 0u);
    if (ctx->comm_tags.count < need) HIP_TRY(ctx->comm_tags.resize(need));
    *mine = ctx->comm_tags.ptr;
    *all = root ? ctx->comm_tags.ptr + 2 : nullptr;
    *flag = ctx->overflow.ptr;
    return CTGPU_OK;
}
int ctgpu_device_of(const ctgpu_ctx *ctx) { return ctx->device; }
// one mark of a sharded frame's phases (ctgpu_read_phase_times); nothing happens unless timing is on
int ctgpu_phase_mark(ctgpu_ctx *ctx, int which, hipStream_t stream) {
    if (!ctx->kernel_timing) return CTGPU_OK;
    if (which == 0) {
        while (ctx->phase_marks.size() < ctx->phase_marks_used + 4) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            ctx->phase_marks.push_back(e);
        }
        ctx->phase_marks_used += 4;
    }
    if (ctx->phase_marks_used < 4) return CTGPU_OK;          // timing was switched on in mid-frame
    HIP_TRY(hipEventRecord(ctx->phase_marks[ctx->phase_marks_used - 4 + (size_t)which], stream));
    return CTGPU_OK;
}

int ctgpu_shard_buffers(ctgpu_ctx *ctx, uint64_t n_tiles, uint64_t n_all, void **tiles, void **all) {
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->tiles.resize(n_tiles));
    if (n_all) HIP_TRY(ctx->gathered.resize(n_all));
    *tiles = ctx->tiles.ptr;
    *all = n_all ? ctx->gathered.ptr : nullptr;
    return CTGPU_OK;
}

extern "C" void ctgpu_destroy(ctgpu_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)ctgpu_comm_destroy(ctx);
    ctgpu_multi_comms_release(ctx->multi_comms);
    ctx->scene.reset();
    ctx->tiles.release(); ctx->fb.release(); ctx->gathered.release(); ctx->counters.release(); ctx->overflow.release();
    ctx->rand_table.release(); ctx->row_ordinal.release();
    ctx->coords.release(); ctx->hits.release(); ctx->shadow.release(); ctx->queue.release(); ctx->spill.release();
    ctx->walk_work.release(); ctx->handed_ids[0].release(); ctx->handed_ids[1].release();
    ctx->b_rays.release(); ctx->b_shadow_rays.release(); ctx->b_hits.release(); ctx->b_shadow_hits.release();
    ctx->b_live.release(); ctx->b_shadow_live.release(); ctx->b_pending.release(); ctx->b_local.release();
    ctx->b_terminal.release(); ctx->b_refl.release(); ctx->b_level_live.release();
    ctx->order.release();
    for (hipEvent_t e : ctx->marks) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->phase_marks) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
    if (ctx->pinned_fb) { (void)hipHostUnregister(ctx->pinned_fb); ctgpu_own_pin_forget(ctx->pinned_fb); }
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ------------------------------------------------------------------ scene upload
namespace {

// A caller's tree is checked on the host before the device trusts it: child links and leaf ranges in range, no
// cycle, depth within the stack bound.  (A tree this library built itself, ctgpu_build_bvh's arrays, needs none of it.)
int validate_tree(const ctgpu_scene_desc *s, uint32_t &depth) {
    const uint32_t N = s->n_triangles, M = s->n_nodes;
    if (M > 2u * N) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_upload_scene: more nodes than 2N-1");
    for (uint32_t i = 0; i < M; i++) {
        const ctgpu_node &n = s->nodes[i];
        if (n.count == 0) {
            if (n.left == 0 || (uint64_t)n.left + 1 >= M) return ctgpu_fail(CTGPU_ERR_ARG, "node %u: child link out of range", i);
        } else if ((uint64_t)n.first + n.count > N) {
            return ctgpu_fail(CTGPU_ERR_ARG, "node %u: leaf range out of range", i);
        }
    }
    for (uint32_t i = 0; i < N; i++)
        if (s->indexes[i] >= N) return ctgpu_fail(CTGPU_ERR_ARG, "indexes[%u] out of range", i);
    depth = 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo;
    todo.push_back({0u, 1u});
    size_t visited = 0;
    while (!todo.empty()) {
        const auto [id, d] = todo.back();
        todo.pop_back();
        if (++visited > M) return ctgpu_fail(CTGPU_ERR_ARG, "node links form a cycle");
        depth = std::max(depth, d);
        const ctgpu_node &n = s->nodes[id];
        if (n.count == 0) { todo.push_back({n.left, d + 1}); todo.push_back({n.left + 1, d + 1}); }
    }
    if (depth > CTGPU_MAX_TREE_DEPTH) return ctgpu_fail(CTGPU_ERR_STACK, "tree depth %u exceeds %d", depth, CTGPU_MAX_TREE_DEPTH);
    return CTGPU_OK;
}

// The packing on the HOST (CTGPU_PACK=host): the serial restatement the device kernels of ct_pack.hip are compared
// with, array for array (tests/test_gpu_scene_pack.py).  Not the default: 0.34 s for the dragon against a few ms.
int pack_on_host(SceneStore &S, const ctgpu_scene_desc *s) {
    const uint32_t N = s->n_triangles, M = s->n_nodes;
    std::vector<uint32_t> wide_of(M, kNoNode);
    uint32_t n_wide = 0;
    for (uint32_t i = 0; i < M; i++)
        if (s->nodes[i].count == 0) wide_of[i] = n_wide++;
    auto child_ref = [&](uint32_t id) -> uint32_t {
        const ctgpu_node &c = s->nodes[id];
        return c.count ? (kLeafBit | c.first) : wide_of[id];
    };
    std::vector<WideNode> wide(std::max<uint32_t>(n_wide, 1));
    std::memset(wide.data(), 0, wide.size() * sizeof(WideNode));
    for (uint32_t i = 0; i < M; i++) {
        const ctgpu_node &n = s->nodes[i];
        if (n.count) continue;
        WideNode &w = wide[wide_of[i]];
        const ctgpu_node &l = s->nodes[n.left], &r = s->nodes[n.left + 1];
        std::memcpy(w.lmin, l.bmin, 24); std::memcpy(w.lmax, l.bmax, 24);
        std::memcpy(w.rmin, r.bmin, 24); std::memcpy(w.rmax, r.bmax, 24);
        w.lref = child_ref(n.left);
        w.rref = child_ref(n.left + 1);
    }

    // ---- the folded tree of the packet walk (QuadNode, ct_device.h) and the bound its filter needs
    double bound[3] = {0, 0, 0};
    bool bound_ok = true;
    for (uint32_t i = 0; i < M; i++) {
        const ctgpu_node &n = s->nodes[i];
        for (int c = 0; c < 3; c++) {
            const double lo = std::fabs(n.bmin[c]), hi = std::fabs(n.bmax[c]);
            if (!(lo <= 1e30) || !(hi <= 1e30)) bound_ok = false;      // also catches NaN
            bound[c] = std::max(bound[c], std::max(lo, hi));
        }
    }
    for (int c = 0; c < 3; c++)
        S.plane_bound[c] = bound_ok ? std::nextafterf((float)bound[c], INFINITY) : INFINITY;

    // an inner node may be folded into its parent's QuadNode when its children's boxes lie inside its own
    auto folds = [&](uint32_t id) {
        const ctgpu_node &c = s->nodes[id];
        if (c.count) return false;
        for (uint32_t side = 0; side < 2; side++) {
            const ctgpu_node &g = s->nodes[c.left + side];
            for (int a = 0; a < 3; a++)
                if (!(g.bmin[a] >= c.bmin[a] && g.bmax[a] <= c.bmax[a])) return false;
        }
        return true;
    };
    struct Folded { std::vector<QuadNode> quads; std::vector<double> boxes; std::vector<uint32_t> node_of; uint32_t root = kLeafBit | 0u, pending = 1; };
    // fold = true: the tree the packet walk uses; fold = false: the same structure with nothing folded
    // (two entries per node), walked a second time by rays folding is not proven for (DESIGN B2)
    auto build_quads = [&](bool fold, Folded &out) -> bool {
        std::vector<uint32_t> quad_of(M, kNoNode), quad_head;       // reference node heading each QuadNode
        auto quad_for = [&](uint32_t id) {
            if (quad_of[id] == kNoNode) {
                quad_of[id] = (uint32_t)quad_head.size();
                quad_head.push_back(id);
            }
            return quad_of[id];
        };
        if (s->nodes[0].count == 0) out.root = quad_for(0);
        for (size_t q = 0; q < quad_head.size(); q++) {             // quad_head grows while we go
            const uint32_t head = quad_head[q];
            QuadNode node;
            std::memset(&node, 0, sizeof node);
            double boxes[4][6] = {};
            uint32_t ids[4] = {0, 0, 0, 0};
            auto add = [&](uint32_t id) {
                const ctgpu_node &e = s->nodes[id];
                const uint32_t k = node.count++;
                ids[k] = id;
                for (int a = 0; a < 3; a++) { node.plane[k][a][0] = (float)e.bmin[a]; node.plane[k][a][1] = (float)e.bmax[a]; }
                node.ref[k] = e.count ? (kLeafBit | e.first) : quad_for(id);
                for (int a = 0; a < 3; a++) { boxes[k][a] = e.bmin[a]; boxes[k][3 + a] = e.bmax[a]; }
            };
            for (uint32_t side = 0; side < 2; side++) {
                const uint32_t c = s->nodes[head].left + side;
                if (fold && folds(c)) { add(s->nodes[c].left); add(s->nodes[c].left + 1); }
                else add(c);
            }
            // absent entries: a one-point box at the corner of the plane bound (never entered, see ct_walk_packet)
            for (uint32_t k = node.count; k < 4; k++)
                for (int a = 0; a < 3; a++) node.plane[k][a][0] = node.plane[k][a][1] = S.plane_bound[a];
            out.quads.push_back(node);
            out.boxes.insert(out.boxes.end(), &boxes[0][0], &boxes[0][0] + 24);
            out.node_of.insert(out.node_of.end(), ids, ids + 4);
        }
        // most entries ever waiting on the packet walk's stack: along a path, every QuadNode leaves count-1 behind
        if (!out.quads.empty()) {
            std::vector<std::pair<uint32_t, uint32_t>> todo;
            todo.push_back({out.root, 0u});
            size_t steps = 0;
            while (!todo.empty()) {
                const auto [q, before] = todo.back();
                todo.pop_back();
                if (++steps > out.quads.size()) return false;
                const QuadNode &node = out.quads[q];
                out.pending = std::max(out.pending, before + node.count - 1u);
                for (uint32_t k = 0; k < node.count; k++)
                    if (!(node.ref[k] & kLeafBit)) todo.push_back({node.ref[k], before + node.count - 1u - k});
            }
        }
        if (out.quads.empty()) { out.quads.emplace_back(); std::memset(&out.quads[0], 0, sizeof(QuadNode)); out.boxes.assign(24, 0.0); out.node_of.assign(4, 0u); }
        return true;
    };
    Folded folded, flat;
    if (!build_quads(true, folded) || !build_quads(false, flat)) return ctgpu_fail(CTGPU_ERR_ARG, "node links form a cycle");

    // leaf-ordered triangle records
    std::vector<TriRecord> recs(N);
    std::vector<uint32_t> remaining(N, 1);
    for (uint32_t i = 0; i < M; i++) {
        const ctgpu_node &n = s->nodes[i];
        for (uint32_t k = 0; k < n.count; k++) remaining[n.first + k] = n.count - k;
    }
    uint32_t slot0 = 0;
    for (uint32_t slot = 0; slot < N; slot++) {
        const uint32_t tri = s->indexes[slot];
        const double *v = s->vertices + (size_t)tri * 9;
        TriRecord &t = recs[slot];
        for (int c = 0; c < 3; c++) {
            t.p1[c] = v[c];
            t.e1[c] = v[3 + c] - v[c];      // p2 - p1, as bvh.cpp:148
            t.e2[c] = v[6 + c] - v[c];      // p3 - p1, as bvh.cpp:149
        }
        t.triangle = tri;
        t.remaining = remaining[slot];
        if (tri == 0) slot0 = slot;
    }

    HIP_TRY(S.nodes.resize(wide.size()));
    HIP_TRY(S.tris.resize(N));
    HIP_TRY(hipMemcpy(S.nodes.ptr, wide.data(), wide.size() * sizeof(WideNode), hipMemcpyHostToDevice));
    HIP_TRY(S.quads.resize(folded.quads.size()));
    HIP_TRY(hipMemcpy(S.quads.ptr, folded.quads.data(), folded.quads.size() * sizeof(QuadNode), hipMemcpyHostToDevice));
    HIP_TRY(S.quad_boxes.resize(folded.boxes.size()));
    HIP_TRY(hipMemcpy(S.quad_boxes.ptr, folded.boxes.data(), folded.boxes.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(S.quads_flat.resize(flat.quads.size()));
    HIP_TRY(hipMemcpy(S.quads_flat.ptr, flat.quads.data(), flat.quads.size() * sizeof(QuadNode), hipMemcpyHostToDevice));
    HIP_TRY(S.quad_node_of_flat.resize(flat.node_of.size()));
    HIP_TRY(hipMemcpy(S.quad_node_of_flat.ptr, flat.node_of.data(), flat.node_of.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    // (the unfolded tree's boxes are read from the reference's own node array)
    HIP_TRY(S.ref_nodes.resize(M));
    HIP_TRY(hipMemcpy(S.ref_nodes.ptr, s->nodes, (size_t)M * sizeof(ctgpu_node), hipMemcpyHostToDevice));
    S.quad_root = folded.root;
    S.quad_root_flat = flat.root;
    S.n_quads = (uint32_t)folded.quads.size();
    S.n_quads_flat = (uint32_t)flat.quads.size();
    S.quad_pending = std::max(folded.pending, flat.pending);
    HIP_TRY(hipMemcpy(S.tris.ptr, recs.data(), (size_t)N * sizeof(TriRecord), hipMemcpyHostToDevice));
    S.root_ref = s->nodes[0].count ? (kLeafBit | s->nodes[0].first) : wide_of[0];
    S.slot_of_triangle0 = slot0;
    return CTGPU_OK;
}

}  // namespace

// Scene -> HBM.  With a tree in the descriptor (nodes / indexes: the reference's arrays, from the host builder or
// from anyone else) the tree is validated on the host, copied as it is (64 B per node, 4 B per index, 72 B per
// triangle) and re-laid-out for the walks ON THE DEVICE (ct_pack.hip).  With nodes == indexes == NULL the tree is
// BUILT on the device as well (ct_build.hip: BuildBVH, bvh.cpp:16-121, the reference's arrays byte for byte) and never
// leaves it: vertices in, first frame out (ctgpu_download_tree hands the arrays back to a host that wants them).
extern "C" int ctgpu_upload_scene(ctgpu_ctx *ctx, const ctgpu_scene_desc *s) {
    if (!ctx || !s) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_upload_scene: null argument");
    const bool build_here = !s->nodes && !s->indexes && s->n_nodes == 0;
    if (s->n_triangles == 0 || !s->vertices || !s->materials || (!build_here && (s->n_nodes == 0 || !s->nodes || !s->indexes)))
        return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_upload_scene: empty scene or null array");
    if (s->n_lights && !s->lights) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_upload_scene: null light array");
    // (leaf references carry two flag bits above the TriRecord slot: kLeafBit, kOneBit -- kSlotMask, ct_device.h)
    static_assert(kSlotMask == 0x3fffffffu && (kLeafBit | kOneBit | kSlotMask) == 0xffffffffu, "slots and node indices stay below 2^30");
    if (s->n_triangles >= (1u << 30) || s->n_nodes >= (1u << 31))
        return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_upload_scene: %u triangles / %u nodes: the walks' references hold slots below 2^30", s->n_triangles, s->n_nodes);
    if (ctx->pending.active) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_upload_scene: a frame is in flight on this context (ctgpu_render_poll)");
    HIP_TRY(hipSetDevice(ctx->device));
    // (frames of this context that may still be running read the old scene: it stays alive until they are done)
    HIP_TRY(hipStreamSynchronize(ctx->stream));

    const uint32_t N = s->n_triangles;
    auto store = std::make_shared<SceneStore>();
    SceneStore &S = *store;
    S.device = ctx->device;
    const bool host_pack = ctx->cfg.pack_on_host != 0;

    uint32_t depth = 0, M = s->n_nodes;
    if (!build_here) { int rc = validate_tree(s, depth); if (rc) return rc; }
    if (host_pack && !build_here) {
        int rc = pack_on_host(S, s); if (rc) return rc;
        std::memcpy(S.root_min, s->nodes[0].bmin, 24);
        std::memcpy(S.root_max, s->nodes[0].bmax, 24);
    } else {
        DeviceArray<double> d_verts;
        struct Free { DeviceArray<double> &a; ~Free() { a.release(); } } free_verts{d_verts};
        HIP_TRY(d_verts.resize((size_t)N * 9));
        HIP_TRY(hipMemcpy(d_verts.ptr, s->vertices, (size_t)N * 9 * sizeof(double), hipMemcpyHostToDevice));
        if (build_here) {
            ctgpu_node *d_nodes = nullptr;
            uint32_t *d_idx = nullptr;
            int rc = ctgpu_build_bvh_resident(ctx->device, d_verts.ptr, N, &d_nodes, &d_idx, &M, &depth, &S.build_ms); if (rc) return rc;
            S.ref_nodes.adopt(d_nodes, M);
            S.ref_indexes.adopt(d_idx, N);
        } else {
            HIP_TRY(S.ref_nodes.resize(M));
            HIP_TRY(S.ref_indexes.resize(N));
            HIP_TRY(hipMemcpy(S.ref_nodes.ptr, s->nodes, (size_t)M * sizeof(ctgpu_node), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(S.ref_indexes.ptr, s->indexes, (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        ctgpu_packed_scene packed;
        int rc = ctgpu_pack_on_device(ctx->device, S.ref_nodes.ptr, M, S.ref_indexes.ptr, d_verts.ptr, N, &packed); if (rc) return rc;
        S.nodes.adopt((WideNode *)packed.wide, packed.n_wide);
        S.quads.adopt((QuadNode *)packed.quads, packed.n_quads);
        S.quad_boxes.adopt(packed.quad_boxes, (size_t)packed.n_quads * 24);
        S.quads_flat.adopt((QuadNode *)packed.quads_flat, packed.n_quads_flat);
        S.quad_node_of_flat.adopt(packed.quad_node_of_flat, (size_t)packed.n_quads_flat * 4);
        S.tris.adopt((TriRecord *)packed.tris, N);
        S.quad_root = packed.quad_root; S.quad_root_flat = packed.quad_root_flat;
        S.n_quads = packed.n_quads; S.n_quads_flat = packed.n_quads_flat;
        S.quad_pending = packed.quad_pending;
        std::memcpy(S.plane_bound, packed.plane_bound, sizeof S.plane_bound);
        S.root_ref = packed.root_ref;
        S.slot_of_triangle0 = packed.slot_of_triangle0;
        std::memcpy(S.root_min, packed.root_min, 24);
        std::memcpy(S.root_max, packed.root_max, 24);
        S.pack_ms = packed.device_ms;
        if (!build_here) S.ref_indexes.release();       // (the caller holds these; the node array stays: the unfolded tree's boxes are read from it)
    }
    if (depth > CTGPU_MAX_TREE_DEPTH) return ctgpu_fail(CTGPU_ERR_STACK, "tree depth %u exceeds %d", depth, CTGPU_MAX_TREE_DEPTH);

    {   // cull nodes for the folded tree's big leaves (whichever way the tree was packed): a second QuadNode array
        ctgpu_cull_scene cull;
        // (the walk's stack holds its levels in the lanes of a register: a scene whose pending entries plus three per cull
        //  level would pass 64 gets no cull nodes rather than lose the shared walk)
        const uint32_t big_min = (uint32_t)ctx->big_leaf_min;
        int rc = ctgpu_pack_cull_on_device(ctx->device, S.tris.ptr, N, S.quads.ptr, S.n_quads, S.quad_root,
                                           S.quad_pending + 3u * 6u <= 64u ? big_min : 0u, (uint32_t)ctx->big_leaf_share, S.plane_bound, &cull);
        if (rc) return rc;
        S.edge_bound = cull.edge_bound; S.vertex_bound = cull.vertex_bound;
        if (cull.quads) {
            S.quads_cull.adopt((QuadNode *)cull.quads, (size_t)S.n_quads + cull.n_cull);
            S.quad_root_cull = cull.quad_root;
            S.cull_first = cull.cull_first; S.n_cull = cull.n_cull; S.n_big_leaves = cull.n_big_leaves; S.cull_levels = cull.levels;
            S.cull_usable = true;
            S.quad_pending += 3u * cull.levels;                  // every cull level leaves up to three entries waiting
            // the binary32 filter's bound covers the triangle boxes' planes too
            for (int c = 0; c < 3; c++) S.plane_bound[c] = std::max(S.plane_bound[c], std::nextafterf(cull.plane_bound[c], INFINITY));
        }
    }

    if (ctx->cfg.shadow_tree != 0) {   // the shadow tree (the scenes that cannot have one simply keep walking the reference's)
        ctgpu_shadow_tree st;
        int rc = ctgpu_build_shadow_tree(ctx->device, S.tris.ptr, N, S.ref_nodes.ptr, M, ctx->cfg.shadow_tree_build,
                                         (uint32_t)ctx->cfg.shadow_tree_leaf, (float)ctx->cfg.shadow_tree_leaf_fill * 0.01f, &st);
        if (rc) return rc;
        if (st.usable) {
            S.stree.adopt((QuadNode *)st.nodes, st.n_nodes);
            S.stree_tris.adopt((TriRecord *)st.records, N);
            S.leaf_of_slot.adopt(st.leaf_of_slot, N);
            S.stree_lim.d_min = st.d_min; S.stree_lim.d_max = st.d_max; S.stree_lim.o_max = st.o_max;
            S.stree_pending = st.pending; S.stree_gamma = st.gamma; S.stree_ms = st.build_ms; S.stree_tau = st.tau;
            S.stree_usable = true;
        }
    }

    bool reflective = false;
    for (uint32_t i = 0; i < N; i++) reflective = reflective || (s->materials[i].reflection > 0);
    S.triangles_reflect = reflective;
    static_assert(sizeof(Material) == sizeof(ctgpu_material), "material layout");
    static_assert(sizeof(Light) == sizeof(ctgpu_light), "light layout");
    HIP_TRY(S.materials.resize(N));
    HIP_TRY(S.lights.resize(std::max<uint32_t>(s->n_lights, 1)));
    HIP_TRY(hipMemcpy(S.materials.ptr, s->materials, (size_t)N * sizeof(Material), hipMemcpyHostToDevice));
    if (s->n_lights)
        HIP_TRY(hipMemcpy(S.lights.ptr, s->lights, (size_t)s->n_lights * sizeof(Light), hipMemcpyHostToDevice));

    // non-ambient lights, in file order: each gets one shadow ray per hit (raythread.cpp:289-307)
    std::vector<uint32_t> shadow_lights;
    for (uint32_t i = 0; i < s->n_lights; i++)
        if (s->lights[i].type == CTGPU_LIGHT_POINT || s->lights[i].type == CTGPU_LIGHT_DIRECTIONAL) shadow_lights.push_back(i);
    HIP_TRY(S.shadow_lights.resize(std::max<size_t>(shadow_lights.size(), 1)));
    if (!shadow_lights.empty())
        HIP_TRY(hipMemcpy(S.shadow_lights.ptr, shadow_lights.data(), shadow_lights.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    S.n_shadow_lights = (uint32_t)shadow_lights.size();

    // ---- spheres and their tree (optional; validated like the triangle tree)
    if (s->n_spheres) {
        if (!s->spheres || !s->sphere_nodes || !s->sphere_indexes || s->n_sphere_nodes == 0 || s->n_sphere_nodes > 2u * s->n_spheres)
            return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_upload_scene: sphere arrays missing or inconsistent");
        for (uint32_t i = 0; i < s->n_sphere_nodes; i++) {
            const ctgpu_node &n = s->sphere_nodes[i];
            if (n.count == 0) {
                if (n.left == 0 || (uint64_t)n.left + 1 >= s->n_sphere_nodes) return ctgpu_fail(CTGPU_ERR_ARG, "sphere node %u: child link out of range", i);
            } else if ((uint64_t)n.first + n.count > s->n_spheres) {
                return ctgpu_fail(CTGPU_ERR_ARG, "sphere node %u: leaf range out of range", i);
            }
        }
        for (uint32_t i = 0; i < s->n_spheres; i++)
            if (s->sphere_indexes[i] >= s->n_spheres) return ctgpu_fail(CTGPU_ERR_ARG, "sphere_indexes[%u] out of range", i);
        std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 1u}};
        size_t visited = 0;
        uint32_t sphere_depth = 0;
        while (!todo.empty()) {
            const auto [id, d] = todo.back();
            todo.pop_back();
            if (++visited > s->n_sphere_nodes) return ctgpu_fail(CTGPU_ERR_ARG, "sphere node links form a cycle");
            sphere_depth = std::max(sphere_depth, d);
            if (s->sphere_nodes[id].count == 0) { todo.push_back({s->sphere_nodes[id].left, d + 1}); todo.push_back({s->sphere_nodes[id].left + 1, d + 1}); }
        }
        if (sphere_depth > CTGPU_MAX_TREE_DEPTH) return ctgpu_fail(CTGPU_ERR_STACK, "sphere tree depth %u exceeds %d", sphere_depth, CTGPU_MAX_TREE_DEPTH);
        HIP_TRY(S.spheres.resize(s->n_spheres));
        HIP_TRY(S.sphere_nodes.resize(s->n_sphere_nodes));
        HIP_TRY(S.sphere_indexes.resize(s->n_spheres));
        HIP_TRY(hipMemcpy(S.spheres.ptr, s->spheres, (size_t)s->n_spheres * sizeof(ctgpu_sphere), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(S.sphere_nodes.ptr, s->sphere_nodes, (size_t)s->n_sphere_nodes * sizeof(ctgpu_node), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(S.sphere_indexes.ptr, s->sphere_indexes, (size_t)s->n_spheres * sizeof(uint32_t), hipMemcpyHostToDevice));
        S.n_spheres = s->n_spheres;
        S.sphere_tree_depth = sphere_depth;
        for (uint32_t i = 0; i < s->n_spheres; i++) S.spheres_reflect = S.spheres_reflect || (s->spheres[i].material.reflection > 0);
        reflective = reflective || S.spheres_reflect;
    }

    S.n_triangles = N;
    S.n_nodes = M;
    S.n_lights = s->n_lights;
    S.tree_depth = depth;
    S.has_reflection = reflective;
    ctx->scene = store;
    ctx->closest_off = false;
    ctx->choice[0] = ctx->choice[1] = ctgpu_ctx::WalkChoice();       // another scene: the measured choices start over
    ctx->bounce_choice[0] = ctx->bounce_choice[1] = ctgpu_ctx::BounceChoice();
    return CTGPU_OK;
}

// Another context of the same device takes the owner's scene as its own: one copy in HBM, however many frames are
// in flight.  The scene lives as long as any context holds it.
extern "C" int ctgpu_share_scene(ctgpu_ctx *ctx, ctgpu_ctx *owner) {
    if (!ctx || !owner) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_share_scene: null argument");
    if (!owner->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "ctgpu_share_scene: the other context has no scene");
    if (owner->device != ctx->device) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_share_scene: the contexts sit on different devices");
    if (ctx->pending.active) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_share_scene: a frame is in flight on this context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->scene = owner->scene;
    ctx->closest_off = false;
    ctx->choice[0] = ctx->choice[1] = ctgpu_ctx::WalkChoice();
    ctx->bounce_choice[0] = ctx->bounce_choice[1] = ctgpu_ctx::BounceChoice();
    return CTGPU_OK;
}

// The reference's arrays of a tree that was built on the device (bvhNodes[] and triangles.indexes): nodes_out has
// room for 2N - 1 records, indexes_out for N.
extern "C" int ctgpu_download_tree(ctgpu_ctx *ctx, ctgpu_node *nodes_out, uint32_t *indexes_out, uint32_t *n_nodes) {
    if (!ctx || !n_nodes) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_download_tree: null argument");
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    const SceneStore &S = *ctx->scene;
    if (!S.ref_indexes.ptr) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_download_tree: this scene's tree came from the caller");
    HIP_TRY(hipSetDevice(ctx->device));
    *n_nodes = S.n_nodes;
    if (nodes_out) HIP_TRY(hipMemcpy(nodes_out, S.ref_nodes.ptr, (size_t)S.n_nodes * sizeof(ctgpu_node), hipMemcpyDeviceToHost));
    if (indexes_out) HIP_TRY(hipMemcpy(indexes_out, S.ref_indexes.ptr, (size_t)S.n_triangles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CTGPU_OK;
}

// What the upload cost on the device and what it left in HBM (measurement / tests).
extern "C" int ctgpu_scene_info(ctgpu_ctx *ctx, ctgpu_scene_info_t *out) {
    if (!ctx || !out) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_scene_info: null argument");
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    const SceneStore &S = *ctx->scene;
    std::memset(out, 0, sizeof *out);
    out->n_triangles = S.n_triangles; out->n_nodes = S.n_nodes; out->tree_depth = S.tree_depth;
    out->n_wide = (uint32_t)S.nodes.count; out->n_quads = S.n_quads; out->n_quads_flat = S.n_quads_flat;
    out->quad_root = S.quad_root; out->quad_root_flat = S.quad_root_flat; out->quad_pending = S.quad_pending;
    out->root_ref = S.root_ref; out->slot_of_triangle0 = S.slot_of_triangle0;
    std::memcpy(out->plane_bound, S.plane_bound, sizeof out->plane_bound);
    out->build_ms = S.build_ms; out->pack_ms = S.pack_ms;
    out->shared_by = (uint32_t)ctx->scene.use_count();
    out->hbm_bytes = S.nodes.count * sizeof(WideNode) + (S.quads.count + S.quads_flat.count + S.quads_cull.count) * sizeof(QuadNode) +
                     S.quad_boxes.count * sizeof(double) + S.quad_node_of_flat.count * sizeof(uint32_t) + S.tris.count * sizeof(TriRecord) +
                     S.materials.count * sizeof(Material) + S.ref_nodes.count * sizeof(ctgpu_node) + S.ref_indexes.count * 4 +
                     S.stree.count * sizeof(QuadNode) + S.stree_tris.count * sizeof(TriRecord) + S.leaf_of_slot.count * 4;
    out->shadow_tree_nodes = (uint32_t)S.stree.count; out->shadow_tree_pending = S.stree_pending;
    out->shadow_tree_ms = S.stree_ms; out->shadow_tree_gamma = S.stree_gamma;
    return CTGPU_OK;
}

// The packed arrays themselves, for comparing the device packing with the host's (which: 0 WideNodes, 1 QuadNodes,
// 2 their boxes, 3 unfolded QuadNodes, 4 their entries' reference nodes, 5 TriRecords, 6 the QuadNodes with the cull nodes behind them).  bytes = room in out; *needed = size of the array.
extern "C" int ctgpu_debug_scene_array(ctgpu_ctx *ctx, int which, void *out, uint64_t bytes, uint64_t *needed) {
    if (!ctx || !needed) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_debug_scene_array: null argument");
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    const SceneStore &S = *ctx->scene;
    const void *src = nullptr;
    uint64_t size = 0;
    switch (which) {
        case 0: src = S.nodes.ptr; size = (uint64_t)S.nodes.count * sizeof(WideNode); break;
        case 1: src = S.quads.ptr; size = (uint64_t)S.n_quads * sizeof(QuadNode); break;
        case 2: src = S.quad_boxes.ptr; size = (uint64_t)S.n_quads * 24 * sizeof(double); break;
        case 3: src = S.quads_flat.ptr; size = (uint64_t)S.n_quads_flat * sizeof(QuadNode); break;
        case 4: src = S.quad_node_of_flat.ptr; size = (uint64_t)S.n_quads_flat * 4 * sizeof(uint32_t); break;
        case 5: src = S.tris.ptr; size = (uint64_t)S.n_triangles * sizeof(TriRecord); break;
        case 6: src = S.quads_cull.ptr; size = (uint64_t)S.quads_cull.count * sizeof(QuadNode); break;
        case 7: src = S.stree.ptr; size = (uint64_t)S.stree.count * sizeof(QuadNode); break;
        case 8: src = S.leaf_of_slot.ptr; size = (uint64_t)S.leaf_of_slot.count * sizeof(uint32_t); break;
        case 9: src = S.stree_tris.ptr; size = (uint64_t)S.stree_tris.count * sizeof(TriRecord); break;
        default: return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_debug_scene_array: unknown array %d", which);
    }
    *needed = size;
    if (!out || bytes < size) return CTGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(out, src, size, hipMemcpyDeviceToHost));
    return CTGPU_OK;
}

// ------------------------------------------------------------------ frame geometry
namespace {

struct Geometry {
    int32_t half;
    uint32_t square, rows_per_band, rendered_rows;
};

int check_frame(const ctgpu_frame_desc *f) {
    if (!f) return ctgpu_fail(CTGPU_ERR_ARG, "null frame descriptor");
    if (f->width <= 0 || f->height <= 1 || f->width > 65536 || f->height > 65536)
        return ctgpu_fail(CTGPU_ERR_ARG, "bitmap size %dx%d not supported", f->width, f->height);
    if (f->num_threads <= 0 || f->num_threads > f->height)
        return ctgpu_fail(CTGPU_ERR_ARG, "num_threads %d must be in 1..height", f->num_threads);
    if (f->rng_mode != CTGPU_RNG_HASH && f->rng_mode != CTGPU_RNG_LIBC) return ctgpu_fail(CTGPU_ERR_ARG, "unknown rng_mode");
    if (f->max_depth < 0 || f->max_depth > 10) return ctgpu_fail(CTGPU_ERR_ARG, "max_depth must be 0..10");
    return CTGPU_OK;
}

Geometry geometry(const ctgpu_frame_desc *f) {
    Geometry g;
    g.half = f->height / 2;
    g.square = (uint32_t)(2 * g.half);
    g.rows_per_band = (uint32_t)(f->height / f->num_threads);
    g.rendered_rows = (uint32_t)f->num_threads * g.rows_per_band;
    return g;
}

int check_partition(const ctgpu_partition *p) {
    if (!p) return ctgpu_fail(CTGPU_ERR_ARG, "null partition");
    if (p->tile_rows < 1 || p->part_count < 1 || p->part_index < 0 || p->part_index >= p->part_count)
        return ctgpu_fail(CTGPU_ERR_ARG, "bad partition (tile_rows %d, part %d of %d)", p->tile_rows, p->part_index, p->part_count);
    return CTGPU_OK;
}

uint32_t local_rows_of(const Geometry &g, const ctgpu_partition *p) {
    return ct_tiles_local_rows(g.rendered_rows, (uint32_t)p->tile_rows, (uint32_t)p->part_count);      // include/ct_tiles.h
}

// rows a frame traces, in single-worker order (for the LIBC jitter stream)
void traced_row_ordinals(const ctgpu_frame_desc *f, const Geometry &g, std::vector<int32_t> &ordinal, uint32_t &n) {
    ordinal.assign(std::max<uint32_t>(g.rendered_rows, 1), -1);
    n = 0;
    const bool sub = (f->flags & CTGPU_F_SUBSAMPLE) != 0;
    for (uint32_t y = 0; y < g.rendered_rows; y++) {
        const uint32_t off = y % g.rows_per_band;
        const bool traced = !sub || ((off & 1u) == 0u) || (off == g.rows_per_band - 1u);
        if (traced) ordinal[y] = (int32_t)n++;
    }
}

// Which rays may skip entries of cull nodes, and from which ray.t on a closest hit prunes them (DESIGN appendix D.3):
// with K >= every |e1|, |e2| component, V >= every |vertex coordinate|, the eye at `cam`,
//   o_lim = max |cam_i| + 2 V     (the eye, every surface point and anything as near pass),   S = o_lim + V >= |o - p1|,
//   d_lim = min(2^20 / (K max(K, S)), 2^12 / K^2)
//                                 (S D K <= 2^20: the reference's rounding errors in u, v stay below 2^-13, far inside the
//                                  2^-9 the grown boxes guarantee; K K D <= 2^12: its error in a stays below 2^-22 of a,
//                                  which "beyond ray.t" needs),
//   tau   = 2^-12 K K S           (a closest hit prunes boxes beyond ray.t only while ray.t >= tau),
// and no cull at all unless K K S <= 2^20 (then a plane hit behind the origin computes t <= 0.0001f).
CullLimits cull_limits(const SceneStore &S, const double cam[3]) {
    CullLimits c;
    c.d_lim = 0.0f; c.o_lim = 0.0f; c.tau = INFINITY;
    if (!S.cull_usable) return c;
    const double K = S.edge_bound, V = S.vertex_bound;
    const double eye = std::max(std::fabs(cam[0]), std::max(std::fabs(cam[1]), std::fabs(cam[2])));
    if (!(eye <= 1e30)) return c;
    const double o_lim = eye + 2.0 * V, s_lim = (o_lim + V) * (1.0 + 0x1p-40);
    if (!(K * K * s_lim <= 0x1p20)) return c;
    const double denom = K * std::max(K, s_lim);
    const double d_lim = denom > 0.0 ? std::min(std::min(0x1p20 / denom, 0x1p12 / (K * K)), 0x1p80) : 0x1p80;
    c.d_lim = std::nextafterf((float)d_lim, 0.0f);          // (rounded towards zero: the comparison is made in binary32)
    c.o_lim = std::nextafterf((float)o_lim, INFINITY);      // (up: the eye itself must pass; S above has the slack)
    c.tau = std::nextafterf((float)(0x1p-12 * K * K * s_lim), INFINITY);
    return c;
}

// `stream`: the stream the frame's kernels will run on -- the LIBC jitter table is uploaded in that stream's
// order (a frame still running there may be reading the previous table).  One context serves one frame at a
// time; frames in flight on different streams use a context each.
int fill_params(ctgpu_ctx *ctx, const ctgpu_frame_desc *f, const ctgpu_partition *part, bool count_mode,
                FrameParams &P, hipStream_t stream, bool use_cull_always = false) {
    const bool use_cull = use_cull_always || (f->flags & CTGPU_F_SUPERSAMPLE) != 0;
    const Geometry g = geometry(f);
    std::memset(&P, 0, sizeof P);
    P.nodes = ctx->scene->nodes.ptr; P.quads = ctx->scene->quads.ptr; P.quad_boxes = ctx->scene->quad_boxes.ptr; P.quad_root = ctx->scene->quad_root;
    P.quads_flat = ctx->scene->quads_flat.ptr; P.quad_node_of_flat = ctx->scene->quad_node_of_flat.ptr; P.ref_nodes = ctx->scene->ref_nodes.ptr; P.quad_root_flat = ctx->scene->quad_root_flat; P.tris = ctx->scene->tris.ptr; P.materials = ctx->scene->materials.ptr; P.lights = ctx->scene->lights.ptr;
    P.n_lights = ctx->scene->n_lights;
    if ((f->flags & CTGPU_F_SPHERES) && ctx->scene->n_spheres) {
        P.spheres = ctx->scene->spheres.ptr; P.sphere_nodes = ctx->scene->sphere_nodes.ptr; P.sphere_indexes = ctx->scene->sphere_indexes.ptr;
        P.n_spheres = ctx->scene->n_spheres;
    }
    P.reflect_fix = (f->flags & CTGPU_F_REFLECT_FIX) ? 1u : 0u;
    P.root_ref = ctx->scene->root_ref;
    P.slot_of_triangle0 = ctx->scene->slot_of_triangle0;
    // pending right siblings never exceed the tree depth; COUNT mode also parks rejected ones
    P.stack_depth = std::max<uint32_t>(ctx->scene->tree_depth, 2);
    (void)count_mode;
    std::memcpy(P.plane_bound, ctx->scene->plane_bound, 12);
    // The tree with cull nodes serves packets whose rays lie close together: a supersampled frame's (four pixels' samples;
    // scene_import.json 1080p x 16: primary walk 1.73 -> 0.90 ms, shadow 1.26 -> 0.87) and the callers' ray lists.  A
    // single-sample frame's packet is an 8x8-pixel tile, wider than the triangles of such leaves, enters most of the boxes
    // and only pays for them (same scene at 1 spp: primary walk 0.32 -> 0.45 ms): it walks the plain tree.
    P.cull_first = kNoNode;
    if (use_cull && ctx->scene->cull_usable) {
        P.quads = ctx->scene->quads_cull.ptr; P.quad_root = ctx->scene->quad_root_cull;
        P.cull_first = ctx->scene->cull_first;   // (the nodes are part of this tree whether or not the frame's rays may skip: d_lim = 0 -> nobody skips)
        P.cull = cull_limits(*ctx->scene, f->cam_pos);
    }
    // the shadow tree: supersampled frames' shadow rays (CTGPU_SHADOW_TREE=2: every frame's)
    if (ctx->scene->stree_usable && (ctx->cfg.shadow_tree == 2 || (ctx->cfg.shadow_tree == 1 && (f->flags & CTGPU_F_SUPERSAMPLE) != 0))) {
        P.stree = ctx->scene->stree.ptr; P.stree_tris = ctx->scene->stree_tris.ptr; P.leaf_of_slot = ctx->scene->leaf_of_slot.ptr;
        P.stree_lim = ctx->scene->stree_lim;
        // ... and the primary rays (CTGPU_CLOSEST_TREE; appendix F): the eye must lie where the boxes' growth covers the
        // binary32 arithmetic, |eye_i| <= o_max like a shadow ray's origin (a far-away camera keeps the reference's tree)
        const bool eye_inside = std::fabs(f->cam_pos[0]) <= ctx->scene->stree_lim.o_max && std::fabs(f->cam_pos[1]) <= ctx->scene->stree_lim.o_max &&
                                std::fabs(f->cam_pos[2]) <= ctx->scene->stree_lim.o_max;
        const bool wanted = eye_inside && (ctx->cfg.closest_tree == 2 || (ctx->cfg.closest_tree == 1 && (f->flags & CTGPU_F_SUPERSAMPLE) != 0));
        if (wanted && ctx->closest_off && ++ctx->closest_off_frames >= ctgpu_ctx::kClosestRetry) ctx->closest_off = false;
        if (wanted && !ctx->closest_off) {
            P.closest_tree = 1u;
            P.closest_tau = ctx->scene->stree_tau;
        }
    }
    std::memcpy(P.root_min, ctx->scene->root_min, 24);
    std::memcpy(P.root_max, ctx->scene->root_max, 24);
    P.width = f->width; P.height = f->height; P.half = g.half;
    P.square = g.square; P.rendered_rows = g.rendered_rows; P.rows_per_band = g.rows_per_band;
    P.subsample = (f->flags & CTGPU_F_SUBSAMPLE) ? 1u : 0u;
    P.inv_height = 1.0f / (float)f->height;
    P.max_depth = f->max_depth;
    std::memcpy(P.cam, f->cam_pos, 24);
    std::memcpy(P.rot, f->cam_rot, 72);
    P.rng_mode = f->rng_mode;
    P.rng_seed = f->rng_seed;
    P.tile_rows = (uint32_t)part->tile_rows; P.part_index = (uint32_t)part->part_index; P.part_count = (uint32_t)part->part_count;
    P.local_rows = local_rows_of(g, part);
    P.thresh_a = ctx->thresh[0]; P.thresh_b = ctx->thresh[1]; P.thresh_c = ctx->thresh[2];
    P.counters = ctx->counters.ptr;
    P.overflow = ctx->overflow.ptr;

    if ((f->flags & CTGPU_F_SUPERSAMPLE) && f->rng_mode == CTGPU_RNG_LIBC) {
        std::vector<int32_t> ordinal;
        uint32_t n_rows = 0;
        traced_row_ordinals(f, g, ordinal, n_rows);
        const uint64_t n = (uint64_t)g.square * n_rows * 32u;
        const uint64_t key[4] = {f->rng_seed, f->rng_offset, n, ((uint64_t)n_rows << 32) | g.rendered_rows | ((uint64_t)P.subsample << 63)};
        if (std::memcmp(key, ctx->rand_key, sizeof key) != 0) {
            // the staging copies live in the context until the next table replaces them
            HIP_TRY(hipStreamSynchronize(stream));          // nobody may still be reading the old table (or a buffer resize frees)
            ctx->rand_host.assign(std::max<uint64_t>(n, 1), 0);
            int rc = ctgpu_rand_stream((uint32_t)f->rng_seed, f->rng_offset, n, ctx->rand_host.data());
            if (rc) return rc;
            ctx->ordinal_host = ordinal;
            HIP_TRY(ctx->rand_table.resize(ctx->rand_host.size()));
            HIP_TRY(ctx->row_ordinal.resize(ctx->ordinal_host.size()));
            // (blocking copies: the tables sit in pageable C-heap memory, which no copy queued on a stream may touch --
            //  ctgpu_ctx::stage; the stream is idle here, and the table is built once per seed and frame geometry)
            HIP_TRY(hipMemcpy(ctx->rand_table.ptr, ctx->rand_host.data(), ctx->rand_host.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(ctx->row_ordinal.ptr, ctx->ordinal_host.data(), ctx->ordinal_host.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            std::memcpy(ctx->rand_key, key, sizeof key);
        }
        P.rand_table = ctx->rand_table.ptr;
        P.row_ordinal = ctx->row_ordinal.ptr;
        P.traced_rows = n_rows;
    }
    return CTGPU_OK;
}

uint32_t tile_count(const FrameParams &P) { return ((P.square + 7u) / 8u) * ((P.local_rows + 7u) / 8u); }

int check_overflow(ctgpu_ctx *ctx) {
    uint32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, ctx->overflow.ptr, sizeof flag, hipMemcpyDeviceToHost));
    if (flag & 4u) {
        HIP_TRY(hipMemset(ctx->overflow.ptr, 0, sizeof flag));
        return ctgpu_fail(CTGPU_ERR_COMM, "a gather brought together tiles of DIFFERENT frames: contexts that share a communicator (ctgpu_comm_share) "
                          "must issue their sharded frames in the same order on every rank -- the framebuffers since the last check are not to be trusted");
    }
    if (flag == 2u) {
        HIP_TRY(hipMemset(ctx->overflow.ptr, 0, sizeof flag));
        return ctgpu_fail(CTGPU_ERR_HIP, "the shadow walk found its kernel arguments laid out differently from what it re-reads (ct_pipeline.hip, kernarg_again): nothing was traced");
    }
    if (flag) {
        HIP_TRY(hipMemset(ctx->overflow.ptr, 0, sizeof flag));
        return ctgpu_fail(CTGPU_ERR_STACK, "traversal stack overflow (tree depth %u)", ctx->scene->tree_depth);
    }
    return CTGPU_OK;
}

int read_counters(ctgpu_ctx *ctx, ctgpu_stats *st) {
    const uint32_t debug = ctx->cfg.debug;
    unsigned long long c[kCounterWords];
    HIP_TRY(hipMemcpy(c, ctx->counters.ptr, sizeof c, hipMemcpyDeviceToHost));
    if ((debug & CTGPU_DEBUG_PHASES) && c[5])
        std::fprintf(stderr, "phases: A %llu x %.1f lanes (%.0f cyc each), B %llu x %.1f lanes (%.0f cyc), C %llu x %.1f lanes (%.0f cyc)\n", c[5],
                     (double)c[6] / c[5], (double)c[11] / c[5], c[7], c[7] ? (double)c[8] / c[7] : 0.0, c[7] ? (double)c[12] / c[7] : 0.0,
                     c[9], c[9] ? (double)c[10] / c[9] : 0.0, c[9] ? (double)c[13] / c[9] : 0.0);
    if ((debug & CTGPU_DEBUG_TRI) && c[19])         // diagnostic build -DCT_TRI_STATS (tools/tri_stats.py)
        std::fprintf(stderr, "triangle records tested by a wave: %llu; the wave's furthest lane stops at |a| %.2f%%, at u %.2f%%, at v / u+v %.2f%%, "
                     "passes (a hit candidate) %.2f%%\n", c[19], 100.0 * c[20] / c[19], 100.0 * c[21] / c[19], 100.0 * c[22] / c[19], 100.0 * c[23] / c[19]);
    if ((debug & CTGPU_DEBUG_PACKET) && c[14]) {          // diagnostic build -DCT_WALK_STATS (tools/lane_stats.py): "hist <what> b0 .. b7"
        const char *what[3] = {"node visits", "triangle tests", "packets"};
        for (int h = 0; h < 3; h++) {
            std::fprintf(stderr, "hist %s by rays taking part (1-8 .. 57-64):", what[h]);
            for (int b = 0; b < 8; b++) std::fprintf(stderr, " %llu", c[24 + h * 8 + b]);
            std::fprintf(stderr, "\n");
        }
    }
    if (debug & CTGPU_DEBUG_REDO)
        std::fprintf(stderr, "packet walk: %llu rays walked again unfolded (NaN box parameter)\n", c[15]);
    if ((debug & CTGPU_DEBUG_PACKET) && c[14])
        std::fprintf(stderr, "packet walk: inner %llu (%.1f lanes, %.0f cyc each; level 1 on %.3f%%), leaf triangles %llu (%.1f lanes; %.0f cyc each), "
                     "pops %llu (%.0f cyc each); wave cycles %llu: inner %.1f%% leaf %.1f%% pop %.1f%% other %.1f%%\n",
                     c[5], (double)c[6] / c[5], (double)c[11] / c[5], 100.0 * c[9] / c[5], c[7], c[7] ? (double)c[8] / c[7] : 0.0,
                     c[7] ? (double)c[12] / c[7] : 0.0, c[10], c[10] ? (double)c[13] / c[10] : 0.0, c[14], 100.0 * c[11] / c[14],
                     100.0 * c[12] / c[14], 100.0 * c[13] / c[14], 100.0 * (c[14] - c[11] - c[12] - c[13]) / c[14]);
    else if ((debug & CTGPU_DEBUG_PHASES) && c[14])
        std::fprintf(stderr, "wave lifetime cycles total %llu: A %.1f%% B %.1f%% R %.1f%% other %.1f%%\n", c[14],
                     100.0 * c[11] / c[14], 100.0 * c[12] / c[14], 100.0 * c[13] / c[14],
                     100.0 * (c[14] - c[11] - c[12] - c[13]) / c[14]);
    HIP_TRY(hipMemset(ctx->counters.ptr, 0, sizeof c));
    ctx->handed_over[0] += c[17];                 // reported by ctgpu_read_walk_work
    ctx->handed_over[1] += c[18];
    if (st) {
        st->rays_primary = c[0]; st->rays_shadow = c[1]; st->rays_reflect = c[2];
        st->node_visits = c[3]; st->triangle_tests = c[4];
        st->rays_shadow_elided = c[kCounterElided];
        st->rays_shadow_left = c[kCounterLeft];
        st->rays_primary_left = c[kCounterPrimaryLeft];
        st->closest_tree = ctx->path.closest_tree;
        if (ctx->path.closest_tree && c[kCounterPrimaryLeft] * 16ull > c[0]) { ctx->closest_off = true; ctx->closest_off_frames = 0; }
        st->walk_kind = ctx->path.walk_kind; st->bounce_kind = ctx->path.bounce_kind; st->ordered = ctx->path.ordered;
        st->cull_tree = ctx->path.cull_tree; st->shadow_tree = ctx->path.shadow_tree;
    }
    return CTGPU_OK;
}

}  // namespace

// ------------------------------------------------------------------ one partition's trace
namespace {

// The wavefront pipeline covers every frame without reflection bounces (any number of lights, spheres included);
// reflective materials take the general path, one kernel that follows TraceRay's recursion.
// does TraceRay's recursion bounce in this frame: some material the frame draws reflects, and depth is allowed
bool frame_reflects(const ctgpu_ctx *ctx, const ctgpu_frame_desc *f) {
    const bool spheres = (f->flags & CTGPU_F_SPHERES) && ctx->scene->n_spheres;
    // (a sphere's material counts only in frames that draw spheres)
    return f->max_depth > 0 && (ctx->scene->triangles_reflect || (spheres && ctx->scene->spheres_reflect));
}

bool pipeline_eligible(const ctgpu_ctx *ctx, const ctgpu_frame_desc *f) {
    if (ctx->force_megakernel) return false;
    // reflective frames: the bounce levels ride on the packet walk's ray-list kernels, whose stack levels are lanes
    // of a register (trees that leave more than 64 entries pending take the general path, like CTGPU_BOUNCE=0)
    if (frame_reflects(ctx, f)) {
        if (!ctx->bounce_on_pipeline || ctx->scene->quad_pending > 64u) return false;
        if (!ctx->bounce_auto) return true;
        // measured choice (BounceChoice): which path does the NEXT reflective frame of this kind take
        const ctgpu_ctx::BounceChoice &c = ctx->bounce_choice[(f->flags & CTGPU_F_SUPERSAMPLE) ? 1 : 0];
        const uint32_t phase = c.frames % 256u;
        if (c.frames < 4u) return c.frames < 2u;                   // frames 0, 1: pipeline; 2, 3: general path
        if (phase == 0u) return c.use_general;                     // every 256 frames: one frame on the other path
        return !c.use_general;
    }
    return true;          // (any number of lights: the shadow walk's queue has a head-word set per light)
}

int fill_pipeline(ctgpu_ctx *ctx, uint64_t n_samples, PipelineParams &W, int kind = -1);

// Heavy chunks first: buffers, history and the arguments of ct_order_build for this frame (W.ordered = 1), or nothing
// (W.ordered = 0: per-lane walk, hand-over experiment, a chunk size that is no power of two, CTGPU_ORDER=0).
// The costs belong to one frame geometry on one partition; any other frame starts from "nothing is heavy".
int setup_order(ctgpu_ctx *ctx, const ctgpu_frame_desc *f, const FrameParams &P, PipelineParams &W, OrderBuild order[2], hipStream_t stream,
                hipEvent_t begin) {
    W.ordered = 0;
    const auto pow2 = [](uint32_t v) { return v != 0 && (v & (v - 1)) == 0; };
    if (!ctx->order_on || !W.packet || W.packet_budget || !pow2(W.chunk) || !pow2(W.chunk_shadow)) return CTGPU_OK;
    ctgpu_ctx::OrderHistory &H = ctx->order;
    const uint32_t chunk[2] = {W.chunk, W.chunk_shadow};
    uint32_t per_light[2], chunks[2];
    for (int w = 0; w < 2; w++) {
        per_light[w] = (uint32_t)((W.n_samples + chunk[w] - 1) / chunk[w]);
        const uint64_t all = (uint64_t)per_light[w] * (w == 0 ? 1u : std::max<uint32_t>(W.n_shadow_lights, 1));
        if (all >= (1ull << 31)) return CTGPU_OK;
        chunks[w] = (uint32_t)all;
    }
    const uint64_t key[8] = {W.n_samples, ((uint64_t)W.chunk << 32) | W.chunk_shadow, W.n_shadow_lights,
                             ((uint64_t)P.part_index << 32) | P.part_count, ((uint64_t)(uint32_t)f->width << 32) | (uint32_t)f->height,
                             ((uint64_t)f->flags << 32) | P.tile_rows, (uint64_t)(uintptr_t)ctx->scene.get(), P.rows_per_band};
    // few packets per wave of the grid: the launch is as long as its heaviest packet -- cut the heavy ones up
    const bool split = (uint64_t)chunks[0] < (uint64_t)ctx->order_split_below * W.walk_blocks;
    H.last_split = split;
    if (!split && !ctx->order_always) return CTGPU_OK;          // (a frame with many packets per wave: measured, the order alone only costs)
    const bool same = H.valid && std::memcmp(key, H.key, sizeof key) == 0;
    for (int w = 0; w < 2; w++) {
        for (int par = 0; par < 2; par++) HIP_TRY(H.cost[par][w].resize(chunks[w]));
        HIP_TRY(H.listed[w].resize(chunks[w]));
        HIP_TRY(H.list[w].resize((size_t)kOrderClasses * std::max<uint32_t>(chunks[w], 4096u)));
    }
    for (int par = 0; par < 2; par++) HIP_TRY(H.totals[par].resize(2 * 64 * 2));
    HIP_TRY(H.count.resize(2 * 2 * kOrderClasses));             // [parity][walk][class]
    const int cur = H.parity, prev = H.parity ^ 1;
    if (begin) HIP_TRY(hipEventRecord(begin, stream));           // (the frame's work starts here: see launch_frame)
    if (!same) {          // nothing is known about this frame: no chunk is heavy, and no build has zeroed this frame's counters
        for (int w = 0; w < 2; w++) HIP_TRY(hipMemsetAsync(H.cost[prev][w].ptr, 0, (size_t)chunks[w] * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(H.totals[prev].ptr, 0, 2 * 64 * 2 * sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(H.count.ptr, 0, 2 * 2 * kOrderClasses * sizeof(uint32_t), stream));
        std::memcpy(H.key, key, sizeof key);
        H.valid = true;
    }
    for (int w = 0; w < 2; w++) {
        uint32_t shift = 0;
        while ((1u << shift) < chunk[w]) shift++;
        const uint32_t cap = std::max<uint32_t>(chunks[w], 4096u);
        OrderBuild &B = order[w];
        std::memset(&B, 0, sizeof B);
        B.prev_cost = H.cost[prev][w].ptr;
        B.prev_totals = H.totals[prev].ptr + (size_t)w * 128;
        B.listed = H.listed[w].ptr;
        B.list = H.list[w].ptr;
        B.count = H.count.ptr + (size_t)(cur * 2 + w) * kOrderClasses;
        B.next_count = H.count.ptr + (size_t)(prev * 2 + w) * kOrderClasses;
        B.cost_out = H.cost[cur][w].ptr;
        B.totals_out = H.totals[cur].ptr + (size_t)w * 128;
        B.chunks = chunks[w]; B.cap = cap; B.chunk_shift = shift; B.chunks_per_light = per_light[w];
        for (int k = 0; k < kOrderClasses; k++) {
            B.parts[k] = split ? (uint32_t)ctx->order_parts[k] : 1u;
            B.heavy[k] = ctx->order_heavy[k];
        }
        B.floor = (uint32_t)ctx->order_floor;
        B.n_samples = W.n_samples;
        OrderParams &O = W.order[w];
        O.cost_out = H.cost[cur][w].ptr;
        O.listed = B.listed; O.list = B.list; O.count = B.count;
        O.totals_out = H.totals[cur].ptr + (size_t)w * 128;
        O.cap = cap; O.chunk_shift = shift; O.chunks_per_light = per_light[w];
    }
    W.ordered = 1;
    return CTGPU_OK;
}

// `begin` (optional): recorded on the stream right before the frame's first launch -- after the buffers have been
// sized, so that a context's first frame of a size does not count its hipMallocs as kernel time
// `alone`: the frame has the device to itself (ctgpu_render, _begin, _multi), so its launches are as long as their
// slowest packets and heavy chunks go first; frames a caller pipelines on streams of its own (ctgpu_render_partition,
// _sharded) fill each other's tails, and the order only costs there (rank-share rehearsal: 0.88 -> 1.03 ms at four ranks).
int launch_frame(ctgpu_ctx *ctx, const ctgpu_frame_desc *f, FrameParams &P, hipStream_t stream, hipEvent_t begin = nullptr, bool alone = false) {
    const bool ss = (f->flags & CTGPU_F_SUPERSAMPLE) != 0;
    if (!pipeline_eligible(ctx, f)) {
        if (begin) HIP_TRY(hipEventRecord(begin, stream));
        launch_trace_tiles(P, tile_count(P), ss, ctx->scene->has_reflection && f->max_depth > 0, false, stream);
        HIP_TRY(hipGetLastError());
        ctx->order.last_ordered = false;
        ctx->path = ctgpu_ctx::Path();
        ctx->path.walk_kind = CTGPU_WALK_GENERAL;
        ctx->path.bounce_kind = frame_reflects(ctx, f) ? CTGPU_BOUNCE_GENERAL : CTGPU_BOUNCE_NONE;
        return CTGPU_OK;
    }
    PipelineParams W;
    int rc = fill_pipeline(ctx, pipeline_sample_count(P, ss), W, ss ? 1 : 0); if (rc) return rc;

    if (ss) HIP_TRY(ctx->coords.resize(W.n_samples));
    HIP_TRY(ctx->hits.resize(W.n_samples));
    HIP_TRY(ctx->shadow.resize(W.n_samples * std::max<uint32_t>(W.n_shadow_lights, 1)));
    W.coords = ctx->coords.ptr; W.hits = ctx->hits.ptr; W.shadow = ctx->shadow.ptr;
    BounceParams B;
    const bool bounces = frame_reflects(ctx, f);
    if (bounces) {
        // per sample: the level's reflected ray, its hit, its shadow rays and their hits, and the colours parked on the
        // way down (TraceRay's stack frames, one array per level)
        const size_t n = (size_t)W.n_samples, lights = std::max<uint32_t>(W.n_shadow_lights, 1), levels = (size_t)f->max_depth;
        HIP_TRY(ctx->b_rays.resize(n));           HIP_TRY(ctx->b_hits.resize(n));           HIP_TRY(ctx->b_live.resize(n));
        HIP_TRY(ctx->b_shadow_rays.resize(n * lights)); HIP_TRY(ctx->b_shadow_hits.resize(n * lights)); HIP_TRY(ctx->b_shadow_live.resize(n * lights));
        HIP_TRY(ctx->b_local.resize(n * levels)); HIP_TRY(ctx->b_refl.resize(n * levels));
        HIP_TRY(ctx->b_terminal.resize(n));       HIP_TRY(ctx->b_pending.resize(n));
        HIP_TRY(ctx->b_level_live.resize(levels + 2));
        std::memset(&B, 0, sizeof B);
        B.rays = ctx->b_rays.ptr; B.live = ctx->b_live.ptr; B.hits = ctx->b_hits.ptr;
        B.shadow_rays = ctx->b_shadow_rays.ptr; B.shadow_live = ctx->b_shadow_live.ptr; B.shadow_hits = ctx->b_shadow_hits.ptr;
        B.local = ctx->b_local.ptr; B.refl = ctx->b_refl.ptr; B.terminal = ctx->b_terminal.ptr; B.pending = ctx->b_pending.ptr;
        B.level_live = ctx->b_level_live.ptr;
        B.max_depth = (uint32_t)f->max_depth;
        B.bounce_t = (f->flags & CTGPU_F_REFLECT_FIX) ? kNoHit : 0.0f;
    }
    OrderBuild order[2];
    W.ordered = 0;
    if (alone || ctx->order_pipelined) { rc = setup_order(ctx, f, P, W, order, stream, begin); if (rc) return rc; }
    if (begin && !W.ordered) HIP_TRY(hipEventRecord(begin, stream));
    HIP_TRY(hipMemsetAsync(ctx->queue.ptr, 0, queue_block_words(W.n_shadow_lights) * sizeof(unsigned long long), stream));
    hipEvent_t *marks = nullptr;
    if (ctx->kernel_timing) {
        const size_t need = ctx->marks_used + kPipelineKernels + 1;
        while (ctx->marks.size() < need) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            ctx->marks.push_back(e);
        }
        marks = ctx->marks.data() + ctx->marks_used;
        ctx->marks_used = need;
    }
    launch_pipeline(P, W, ss, stream, marks, bounces ? &B : nullptr, W.ordered ? order : nullptr);
    HIP_TRY(hipGetLastError());
    if (W.ordered) ctx->order.parity ^= 1;
    ctx->order.last_ordered = W.ordered != 0u;
    ctx->path.walk_kind = W.packet ? CTGPU_WALK_PACKET : CTGPU_WALK_PER_LANE;
    ctx->path.bounce_kind = bounces ? CTGPU_BOUNCE_LEVELS : CTGPU_BOUNCE_NONE;
    ctx->path.ordered = W.ordered;
    ctx->path.cull_tree = (W.packet && P.cull_first != kNoNode) ? 1u : 0u;
    // (the same condition launch_pipeline takes the shadow tree under)
    ctx->path.shadow_tree = (W.packet && W.n_shadow_lights != 0u && P.stree && W.packet_budget == 0u) ? 1u : 0u;
    ctx->path.closest_tree = (W.packet && P.closest_tree && P.stree && W.packet_budget == 0u && !W.ordered) ? 1u : 0u;
    return CTGPU_OK;
}

// grid, queue chunk and stack buffers of the walk kernels for n_samples ray ids
// kind: 0 single-sample frame, 1 supersampled frame, -1 not a frame (caller's ray list)
int fill_pipeline(ctgpu_ctx *ctx, uint64_t n_samples, PipelineParams &W, int kind) {
    std::memset(&W, 0, sizeof W);
    W.n_samples = n_samples;
    W.n_shadow_lights = ctx->scene->n_shadow_lights;
    W.keep_every_shadow_ray = ctx->keep_shadow_rays ? 1u : 0u;
    W.shadow_split_gap = ctx->cfg.shadow_split_gap;
    W.lds_depth = std::min<uint32_t>((uint32_t)ctx->lds_depth, std::max<uint32_t>(ctx->scene->tree_depth, 1));
    W.spill_depth = std::max<uint32_t>(ctx->scene->tree_depth, 1) - W.lds_depth;
    const size_t lds_bytes = (size_t)W.lds_depth * 64 * 12;
    W.packet = (ctx->packet_walk && ctx->scene->quad_pending <= 64u) ? 1u : 0u;           // stack levels are lanes of a register
    if (kind >= 0 && ctx->walk_auto && W.packet) {
        ctgpu_ctx::WalkChoice &c = ctx->choice[kind];
        c.ran_probe = c.probe;
        c.ran_lane = c.use_lane != c.probe;               // a probe frame runs on the walk that is NOT the settled one
        c.probe = false;
        if (c.ran_lane) W.packet = 0u;
    }
    W.packet_depth = ctx->scene->quad_pending;
    W.packet_lds_levels = std::min<uint32_t>((uint32_t)ctx->packet_lds_levels, W.packet_depth);
    int per_cu = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu
                                        : (W.packet ? pipeline_packet_occupancy(W.packet_depth, W.packet_lds_levels)
                                                    : pipeline_walk_occupancy(lds_bytes));
    if (per_cu < 1) per_cu = 1;
    const uint64_t waves_needed = (W.n_samples + 63) / 64;
    W.walk_blocks = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * ctx->cu_count, std::max<uint64_t>(waves_needed, 1));
    {   // the per-lane launches behind the packet launches: never more waves than the packet grid (they share its spill area)
        const int lane_per_cu = std::max(1, pipeline_walk_occupancy(lds_bytes));
        W.listed_blocks = std::max<uint32_t>(1u, std::min<uint32_t>(W.walk_blocks, (uint32_t)lane_per_cu * (uint32_t)ctx->cu_count));
    }
    // chunk: small enough that every wave gets many (dynamic balance), large enough that the
    // queue's head word sees few atomics
    // (measured on MI355X, dragon stand-in: 128 for 18M-sample frames, 64 for 1M-sample frames)
    uint64_t chunk = ctx->chunk > 0 ? (uint64_t)ctx->chunk : (W.n_samples >= (4u << 20) ? 128u : 64u);
    chunk = std::min<uint64_t>(std::max<uint64_t>((chunk + 63) / 64 * 64, 64), 4096);
    W.chunk = (uint32_t)chunk;
    uint64_t chunk_shadow = ctx->chunk_shadow > 0 ? (uint64_t)ctx->chunk_shadow : chunk;
    W.chunk_shadow = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((chunk_shadow + 63) / 64 * 64, 64), 4096);
    W.thresh_r = ctx->walk_thresh[0]; W.thresh_a = ctx->walk_thresh[1]; W.thresh_b = ctx->walk_thresh[2];
    if (W.n_samples >= 0xfffffff0ull) return ctgpu_fail(CTGPU_ERR_ARG, "frame too large: %llu samples", (unsigned long long)W.n_samples);
    HIP_TRY(ctx->spill.resize((size_t)W.walk_blocks * (size_t)std::max(W.lds_depth + W.spill_depth, W.packet_depth) * 64 * 12 / 8 + 64));
    W.queue = ctx->queue.ptr; W.spill = ctx->spill.ptr; W.shadow_lights = ctx->scene->shadow_lights.ptr;
    W.walk_work = ctx->walk_work.ptr;
    // hand-over lists: room for half of the frame's primary rays and half of its possible shadow rays
    W.packet_budget = W.packet ? (uint32_t)ctx->packet_budget : 0u;
    W.packet_lanes = (uint32_t)ctx->packet_lanes;
    HIP_TRY(ctx->queue.resize(queue_block_words(W.n_shadow_lights)));
    W.queue = ctx->queue.ptr;
    W.handed = ctx->queue.ptr + queue_words(W.n_shadow_lights);
    W.shadow_left = ctx->queue.ptr + queue_left_word(W.n_shadow_lights);
    {
        const int stree_per_cu = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : std::max(1, pipeline_stree_occupancy());
        W.stree_blocks = (uint32_t)std::min<uint64_t>((uint64_t)stree_per_cu * ctx->cu_count, std::max<uint64_t>(waves_needed, 1));
    }
    W.handed_room = 0;
    if (W.packet_budget) {
        W.handed_room = std::max<uint64_t>(n_samples / 2u, 4096u);
        HIP_TRY(ctx->handed_ids[0].resize(W.handed_room));
        HIP_TRY(ctx->handed_ids[1].resize(W.handed_room));
        W.handed_ids[0] = ctx->handed_ids[0].ptr;
        W.handed_ids[1] = ctx->handed_ids[1].ptr;
    }
    return CTGPU_OK;
}

}  // namespace

// ------------------------------------------------------------------ multi-GPU building blocks
extern "C" int ctgpu_partition_pixels(const ctgpu_frame_desc *frame, const ctgpu_partition *part, uint64_t *count) {
    if (!count) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_partition_pixels: null argument");
    int rc = check_frame(frame); if (rc) return rc;
    rc = check_partition(part); if (rc) return rc;
    const Geometry g = geometry(frame);
    *count = (uint64_t)local_rows_of(g, part) * g.square;
    return CTGPU_OK;
}

extern "C" int ctgpu_render_partition(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, const ctgpu_partition *part,
                                      void *d_tiles, void *stream) {
    if (!ctx || !d_tiles) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_render_partition: null argument");
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    int rc = check_frame(frame); if (rc) return rc;
    rc = check_partition(part); if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    FrameParams P;
    rc = fill_params(ctx, frame, part, false, P, (hipStream_t)stream); if (rc) return rc;
    P.out = (uint32_t *)d_tiles;
    return launch_frame(ctx, frame, P, (hipStream_t)stream);
}

extern "C" int ctgpu_assemble(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, const ctgpu_partition *part,
                              const void *d_all_tiles, void *d_fb, void *stream) {
    if (!ctx || !d_all_tiles || !d_fb) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_assemble: null argument");
    int rc = check_frame(frame); if (rc) return rc;
    rc = check_partition(part); if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    FrameParams P;
    ctgpu_frame_desc plain = *frame;
    plain.flags &= ~CTGPU_F_SUPERSAMPLE;            // the jitter table is not needed here
    rc = fill_params(ctx, &plain, part, false, P, (hipStream_t)stream); if (rc) return rc;
    launch_assemble(P, (const uint32_t *)d_all_tiles, (uint32_t *)d_fb, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return CTGPU_OK;
}

namespace { bool page_locked(const void *p, size_t bytes); }

// the rectangle of the bitmap the reference writes (rendered rows x square columns, clipped: raythread.cpp:176-184,
// draw2d.h:8-20), device framebuffer -> host framebuffer, asynchronous on `stream`
extern "C" int ctgpu_download_frame(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, const void *d_fb, uint32_t *host_fb, void *stream) {
    if (!ctx || !d_fb || !host_fb) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_download_frame: null argument");
    int rc = check_frame(frame); if (rc) return rc;
    if (!page_locked(host_fb, (size_t)frame->width * frame->height * sizeof(uint32_t)))
        return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_download_frame: the host framebuffer must be page-locked (hipHostMalloc / hipHostRegister): the copy is queued on the caller's stream");
    HIP_TRY(hipSetDevice(ctx->device));
    const Geometry g = geometry(frame);
    const int row_hi = frame->height / 2 + g.half, row_lo = row_hi - (int)g.rendered_rows + 1;
    const int r0 = std::max(row_lo, 0), r1 = std::min(row_hi, frame->height - 1);
    const int c0 = std::max(frame->width / 2 - g.half, 0), c1 = std::min(frame->width / 2 + g.half - 1, frame->width - 1);
    if (r1 < r0 || c1 < c0) return CTGPU_OK;
    const size_t off = (size_t)r0 * frame->width + c0;
    HIP_TRY(hipMemcpy2DAsync(host_fb + off, (size_t)frame->width * 4, (const uint32_t *)d_fb + off, (size_t)frame->width * 4,
                             (size_t)(c1 - c0 + 1) * 4, (size_t)(r1 - r0 + 1), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return CTGPU_OK;
}

extern "C" int ctgpu_read_counters(ctgpu_ctx *ctx, ctgpu_stats *stats) {
    if (!ctx || !stats) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_read_counters: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    std::memset(stats, 0, sizeof *stats);
    int rc = read_counters(ctx, stats); if (rc) return rc;
    return check_overflow(ctx);
}

// ------------------------------------------------------------------ whole frame on one GPU
// begin: queue the whole frame -- trace, assemble, copy of the written rectangle into the caller's framebuffer -- on
// the context's stream and return; poll: has it finished (or wait for it), then the counters.  ctgpu_render is the
// two back to back.  The reference's RayThread is non-blocking the same way: it dispatches its workers and looks
// at their status on later calls (raythread.cpp:546-593).
namespace {
// Page-locks this LIBRARY put on callers' memory (begin_frame, any context of the process).  A registration can outlive
// the memory it covers -- a keep_pin context whose framebuffer was freed -- and the allocator may hand the same
// addresses out again, as part of a larger array even; the runtime would still call the first byte "host memory".  So
// a pointer inside one of these ranges counts as page-locked only for the context that holds exactly that range.
std::mutex g_own_pins_lock;
std::map<uintptr_t, size_t> g_own_pins;          // start -> bytes
void own_pin_add(const void *p, size_t bytes) { std::lock_guard<std::mutex> hold(g_own_pins_lock); g_own_pins[(uintptr_t)p] = bytes; }
void own_pin_remove(const void *p) { std::lock_guard<std::mutex> hold(g_own_pins_lock); g_own_pins.erase((uintptr_t)p); }
bool overlaps_own_pin(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> hold(g_own_pins_lock);
    const uintptr_t lo = (uintptr_t)p, hi = lo + std::max<size_t>(bytes, 1);
    auto it = g_own_pins.lower_bound(hi);        // first range starting at or beyond our end
    while (it != g_own_pins.begin()) {
        --it;
        if (it->first + it->second <= lo) break; // (ranges do not nest: hipHostRegister refuses overlapping ones)
        return true;
    }
    return false;
}

bool host_type(const void *p) {
    hipPointerAttribute_t known;
    const bool yes = hipPointerGetAttributes(&known, p) == hipSuccess && known.type == hipMemoryTypeHost;
    (void)hipGetLastError();                     // (plain pageable memory makes the query fail: that is the answer)
    return yes;
}
// Did the CALLER page-lock [p, p + bytes)?  Both ends must be host memory the runtime knows, and none of it may be a
// registration of this library's own (see above): anything else is staged.
bool page_locked(const void *p, size_t bytes) {
    if (overlaps_own_pin(p, bytes)) return false;
    return host_type(p) && host_type((const char *)p + (bytes ? bytes - 1 : 0));
}

void release_pin(ctgpu_ctx *ctx) {
    if (ctx->pinned_fb) { (void)hipHostUnregister(ctx->pinned_fb); own_pin_remove(ctx->pinned_fb); }
    ctx->pinned_fb = nullptr;
    ctx->pinned_bytes = 0;
}

int stage_reserve(ctgpu_ctx *ctx, size_t count) {
    if (count <= ctx->stage_count && ctx->stage) return CTGPU_OK;
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    ctx->stage = nullptr; ctx->stage_count = 0;
    HIP_TRY(hipHostMalloc((void **)&ctx->stage, std::max<size_t>(count, 1) * sizeof(uint32_t), hipHostMallocDefault));
    ctx->stage_count = count;
    return CTGPU_OK;
}

// the written rectangle, staged copy -> the caller's framebuffer
void stage_deliver(const uint32_t *stage, uint32_t *host_fb, int width, int r0, int r1, int c0, int c1) {
    for (int r = r0; r <= r1; r++)
        std::memcpy(host_fb + (size_t)r * width + c0, stage + (size_t)r * width + c0, (size_t)(c1 - c0 + 1) * sizeof(uint32_t));
}

int begin_frame(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb, bool pin) {
    if (!ctx || !host_fb) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_render: null argument");
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    if (ctx->pending.active) return ctgpu_fail(CTGPU_ERR_ARG, "a frame is already in flight on this context (ctgpu_render_poll)");
    int rc = check_frame(frame); if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->pending.t0 = std::chrono::steady_clock::now();

    const ctgpu_partition whole = {8, 0, 1};
    const Geometry g = geometry(frame);
    FrameParams P;
    rc = fill_params(ctx, frame, &whole, false, P, ctx->stream); if (rc) return rc;
    HIP_TRY(ctx->tiles.resize((size_t)P.local_rows * P.square));
    HIP_TRY(ctx->fb.resize((size_t)frame->width * frame->height));
    P.out = ctx->tiles.ptr;

    // an asynchronous copy needs page-locked memory: the caller's framebuffer is registered for the frame (and stays
    // so between frames only under ctgpu_keep_framebuffer_pinned).  The copy goes straight into host_fb only when this
    // context holds a registration of exactly [host_fb, host_fb + fb_bytes) or the caller page-locked all of it.
    const size_t fb_bytes = (size_t)frame->width * frame->height * sizeof(uint32_t);
    if (pin && (ctx->pinned_fb != host_fb || ctx->pinned_bytes != fb_bytes)) {
        release_pin(ctx);
        // (caller-pinned memory: nothing to do, nothing to undo later.  Memory under another context's registration --
        //  a stale one, or a second context drawing into the same bitmap -- or memory the runtime will not register: staged)
        if (!page_locked(host_fb, fb_bytes) && !overlaps_own_pin(host_fb, fb_bytes)) {
            if (hipHostRegister(host_fb, fb_bytes, hipHostRegisterDefault) == hipSuccess) {
                ctx->pinned_fb = host_fb;
                ctx->pinned_bytes = fb_bytes;
                own_pin_add(host_fb, fb_bytes);
            } else (void)hipGetLastError();
        }
    }
    // (whatever fails from here on: a page-lock taken for this frame does not outlive the call)
    struct PinGuard {
        ctgpu_ctx *ctx; bool armed;
        ~PinGuard() { if (armed && !ctx->keep_pin) release_pin(ctx); }
    } pin_guard{ctx, true};

    ctx->pending.kind = pipeline_eligible(ctx, frame) ? ((frame->flags & CTGPU_F_SUPERSAMPLE) ? 1 : 0) : -1;
    ctx->frame_is_reflective = frame_reflects(ctx, frame);
    if (ctx->frame_is_reflective) {
        ctx->bounce_choice[(frame->flags & CTGPU_F_SUPERSAMPLE) ? 1 : 0].ran_general = ctx->pending.kind < 0;
        ctx->pending.reflective_kind = (frame->flags & CTGPU_F_SUPERSAMPLE) ? 1 : 0;
        ctx->pending.kind = -1;                  // (the walk choice is about frames without bounces)
    } else ctx->pending.reflective_kind = -1;
    rc = launch_frame(ctx, frame, P, ctx->stream, ctx->ev0, true); if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    launch_assemble(P, ctx->tiles.ptr, ctx->fb.ptr, ctx->stream);
    HIP_TRY(hipGetLastError());

    // copy back exactly the rectangle the reference writes: rendered rows x square columns,
    // clipped to the bitmap (raythread.cpp:176-184, draw2d.h:8-20)
    const int row_hi = frame->height / 2 + g.half;                        // y = -half
    const int row_lo = frame->height / 2 + g.half - (int)g.rendered_rows + 1;
    const int r0 = std::max(row_lo, 0), r1 = std::min(row_hi, frame->height - 1);
    const int c0 = std::max(frame->width / 2 - g.half, 0), c1 = std::min(frame->width / 2 + g.half - 1, frame->width - 1);
    // (ctgpu_render_begin page-locked the framebuffer above; ctgpu_render's may be pageable: staged)
    uint32_t *target = host_fb;
    ctx->pending.deliver_to = nullptr;
    const bool own_pin = ctx->pinned_fb == host_fb && ctx->pinned_bytes == fb_bytes;
    if (!own_pin && !page_locked(host_fb, fb_bytes)) {
        rc = stage_reserve(ctx, (size_t)frame->width * frame->height); if (rc) return rc;
        target = ctx->stage;
        ctx->pending.deliver_to = host_fb;
        ctx->pending.width = frame->width; ctx->pending.r0 = r0; ctx->pending.r1 = r1; ctx->pending.c0 = c0; ctx->pending.c1 = c1;
    }
    if (r1 >= r0 && c1 >= c0) {
        const size_t off = (size_t)r0 * frame->width + c0;
        HIP_TRY(hipMemcpy2DAsync(target + off, (size_t)frame->width * 4, ctx->fb.ptr + off, (size_t)frame->width * 4,
                                 (size_t)(c1 - c0 + 1) * 4, (size_t)(r1 - r0 + 1), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev_done, ctx->stream));
    ctx->pending.pixels = (r1 >= r0 && c1 >= c0) ? (uint64_t)(r1 - r0 + 1) * (uint64_t)(c1 - c0 + 1) : 0;
    ctx->pending.active = true;
    pin_guard.armed = false;                     // the frame is in flight: finish_frame ends the page-lock
    return CTGPU_OK;
}

// the page-lock begin_frame put on the caller's framebuffer ends with the frame: the caller may free that memory
// (and the allocator may hand the same address out again) as soon as the frame has been reported finished
// One finished frame's kernel time -> the walk choice of its kind (ctgpu_ctx::WalkChoice).
int walk_feedback(ctgpu_ctx *ctx, int kind, double kernel_ms) {
    ctgpu_ctx::WalkChoice &c = ctx->choice[kind];
    c.frames += 1;
    (c.ran_lane ? c.lane_ms : c.packet_ms) = kernel_ms;
    if (c.ran_probe) {                          // both times are fresh: settle (the per-lane walk has to win clearly)
        c.use_lane = c.lane_ms < 0.9 * c.packet_ms;
        return CTGPU_OK;
    }
    if (c.frames % 256u != 3u) return CTGPU_OK;             // (the third frame: the first two carry one-off costs)
    if (c.use_lane) { c.probe = true; return CTGPU_OK; }    // is the shared walk still the slower one?
    // on the shared walk: worth a probe only if this scene's packets share little -- fetches per packet, from the
    // walks' own counters (totals since the last ctgpu_read_walk_work; read, not cleared)
    unsigned long long w[64 * 8];
    HIP_TRY(hipMemcpy(w, ctx->walk_work.ptr, sizeof w, hipMemcpyDeviceToHost));
    unsigned long long packets = 0, fetches = 0;
    for (int slot = 0; slot < 64; slot++)
        for (int k = 0; k < 2; k++) {
            packets += w[slot * 8 + k * 3];
            fetches += w[slot * 8 + k * 3 + 1] + w[slot * 8 + k * 3 + 2];
        }
    const bool grew = packets > c.packets && fetches >= c.fetches;
    const double per_packet = grew ? (double)(fetches - c.fetches) / (double)(packets - c.packets) : 0.0;
    c.packets = packets; c.fetches = fetches;
    if (per_packet > (double)ctx->probe_above) c.probe = true;
    return CTGPU_OK;
}

int finish_frame(ctgpu_ctx *ctx, bool wait, int *done, ctgpu_stats *stats) {
    if (!ctx->pending.active) return ctgpu_fail(CTGPU_ERR_ARG, "no frame is in flight on this context");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!wait && hipEventQuery(ctx->ev_done) == hipErrorNotReady) { if (done) *done = 0; return CTGPU_OK; }
    // from here on the frame is over, whatever is reported: the context must accept the next one
    const hipError_t waited = hipEventSynchronize(ctx->ev_done);
    if (waited == hipSuccess && ctx->pending.deliver_to)
        stage_deliver(ctx->stage, ctx->pending.deliver_to, ctx->pending.width, ctx->pending.r0, ctx->pending.r1, ctx->pending.c0, ctx->pending.c1);
    ctx->pending.deliver_to = nullptr;
    const auto t1 = std::chrono::steady_clock::now();
    ctx->pending.active = false;
    if (!ctx->keep_pin) release_pin(ctx);
    if (waited != hipSuccess) return ctgpu_fail(CTGPU_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(waited));
    if (done) *done = 1;
    ctgpu_stats st;
    std::memset(&st, 0, sizeof st);
    int rc = read_counters(ctx, &st); if (rc) return rc;
    rc = check_overflow(ctx); if (rc) return rc;
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    st.kernel_ms = ms;
    st.total_ms = std::chrono::duration<double, std::milli>(t1 - ctx->pending.t0).count();
    st.pixels_written = ctx->pending.pixels;
    if (stats) *stats = st;
    if (ctx->walk_auto && ctx->pending.kind >= 0) {
        int rc2 = walk_feedback(ctx, ctx->pending.kind, ms); if (rc2) return rc2;
    }
    if (ctx->bounce_auto && ctx->pending.reflective_kind >= 0) {
        ctgpu_ctx::BounceChoice &c = ctx->bounce_choice[ctx->pending.reflective_kind];
        (c.ran_general ? c.general_ms : c.pipeline_ms) = ms;
        c.frames += 1;
        // settled after the fourth frame (frames 1 and 3 are the second on each path), re-examined after each trial frame
        if (c.frames == 4u || (c.frames > 4u && c.frames % 256u == 1u)) c.use_general = c.general_ms < c.pipeline_ms;
    }
    return CTGPU_OK;
}
}  // namespace

static void ctgpu_own_pin_forget(const void *p) { own_pin_remove(p); }

extern "C" int ctgpu_render(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb, ctgpu_stats *stats) {
    int rc = begin_frame(ctx, frame, host_fb, false); if (rc) return rc;
    return finish_frame(ctx, true, nullptr, stats);
}

extern "C" int ctgpu_render_begin(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb) {
    return begin_frame(ctx, frame, host_fb, true);
}

extern "C" int ctgpu_render_poll(ctgpu_ctx *ctx, int wait, int *done, ctgpu_stats *stats) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_render_poll: null argument");
    return finish_frame(ctx, wait != 0, done, stats);
}

extern "C" int ctgpu_keep_framebuffer_pinned(ctgpu_ctx *ctx, int keep) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_keep_framebuffer_pinned: null argument");
    if (ctx->pending.active) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_keep_framebuffer_pinned: a frame is in flight");
    ctx->keep_pin = keep != 0;
    if (!ctx->keep_pin) { HIP_TRY(hipSetDevice(ctx->device)); release_pin(ctx); }
    return CTGPU_OK;
}

// ------------------------------------------------------------------ whole frame on several GPUs, one process
// The in-process counterpart of bench.py's one-process-per-GPU loop, for hosts like RayThread
// that own all devices: every context traces its interleaved row tiles on its own device and
// stream, the tile buffers travel to the first context's device with peer-to-peer copies over
// xGMI (the gather), and that device assembles and hands the frame to the host.
static int render_multi_queued(ctgpu_ctx *const *ctxs, int n_ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb, ctgpu_stats *stats);

extern "C" int ctgpu_render_multi(ctgpu_ctx *const *ctxs, int n_ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb,
                                  ctgpu_stats *stats) {
    if (!ctxs || n_ctx < 1 || !host_fb) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_render_multi: bad argument");
    if (n_ctx == 1) return ctgpu_render(ctxs[0], frame, host_fb, stats);
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i]) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_render_multi: null context %d", i);
        if (!ctxs[i]->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "context %d has no scene", i);
        if (ctxs[i]->pending.active) return ctgpu_fail(CTGPU_ERR_ARG, "context %d has a frame in flight (ctgpu_render_poll)", i);
    }
    int rc = check_frame(frame); if (rc) return rc;
    rc = render_multi_queued(ctxs, n_ctx, frame, host_fb, stats);
    if (rc != CTGPU_OK) {
        // Work may already sit on some of the streams (kernels, peer copies, a partly queued gather): nothing of it may
        // still be running when the caller hears of the failure and frees the framebuffer or reuses the contexts.  A
        // gather some devices never joined would wait for good, so the communicators go first (ncclCommAbort).
        const std::string message = g_last_error;
        ctgpu_ctx *root = ctxs[0];
        if (!root->multi_comms.empty()) {
            ctgpu_multi_comms_release(root->multi_comms, true);
            root->multi_devices.clear();
        }
        for (int i = 0; i < n_ctx; i++) {
            (void)hipSetDevice(ctxs[i]->device);
            (void)hipStreamSynchronize(ctxs[i]->stream);
        }
        (void)hipGetLastError();
        g_last_error = message;
    }
    return rc;
}

static int render_multi_queued(ctgpu_ctx *const *ctxs, int n_ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb, ctgpu_stats *stats) {
    int rc = CTGPU_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const Geometry g = geometry(frame);
    ctgpu_ctx *root = ctxs[0];
    const ctgpu_partition part0 = {8, 0, n_ctx};
    const size_t per_part = (size_t)local_rows_of(g, &part0) * g.square;

    HIP_TRY(hipSetDevice(root->device));
    HIP_TRY(root->gathered.resize(per_part * (size_t)n_ctx));
    HIP_TRY(root->fb.resize((size_t)frame->width * frame->height));

    // The gather is RCCL's (one communicator per device, made once and kept by the root context) whenever
    // every context sits on a device of its own; RCCL cannot address one device twice, so contexts that share
    // a device (a test on a one-GPU box) hand their tiles over with plain device copies instead.
    std::vector<int> devices;
    bool distinct = true;
    for (int i = 0; i < n_ctx; i++) {
        for (int d : devices) distinct = distinct && d != ctxs[i]->device;
        devices.push_back(ctxs[i]->device);
    }
    // RCCL that cannot come up (no librccl, ncclCommInitAll failing) is remembered, not retried every frame: the
    // peer-copy hand-over below serves any set of devices
    bool use_rccl = distinct && !root->multi_comm_failed;
    if (use_rccl && root->multi_devices != devices) {
        ctgpu_multi_comms_release(root->multi_comms);
        root->multi_devices.clear();
        if (ctgpu_multi_comms(devices.data(), n_ctx, root->multi_comms) == CTGPU_OK) root->multi_devices = devices;
        else { use_rccl = false; root->multi_comm_failed = true; }
    }

    // trace: each device on its own stream
    std::vector<void *> tile_ptrs;
    std::vector<hipStream_t> streams;
    for (int i = 0; i < n_ctx; i++) {
        ctgpu_ctx *c = ctxs[i];
        HIP_TRY(hipSetDevice(c->device));
        const ctgpu_partition part = {8, i, n_ctx};
        FrameParams P;
        rc = fill_params(c, frame, &part, false, P, c->stream); if (rc) return rc;
        HIP_TRY(c->tiles.resize(per_part));
        P.out = c->tiles.ptr;
        rc = launch_frame(c, frame, P, c->stream, i == 0 ? root->ev0 : nullptr, true); if (rc) return rc;
        if (i == 0) HIP_TRY(hipEventRecord(root->ev1, root->stream));
        tile_ptrs.push_back(c->tiles.ptr);
        streams.push_back(c->stream);
    }
    // gather: ncclGather from every device's stream (the root's completes when all tiles have arrived) ...
    if (use_rccl && ctgpu_multi_gather(root->multi_comms, devices.data(), tile_ptrs.data(), per_part, root->gathered.ptr,
                                       streams.data()) != CTGPU_OK) {
        use_rccl = false;
        root->multi_comm_failed = true;
        ctgpu_multi_comms_release(root->multi_comms);
        root->multi_devices.clear();
    }
    // ... or peer-to-peer copies over xGMI, each on its device's stream, the root's stream waiting for all of them
    if (!use_rccl) {
        for (int i = 0; i < n_ctx; i++) {
            ctgpu_ctx *c = ctxs[i];
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipMemcpyPeerAsync(root->gathered.ptr + per_part * (size_t)i, root->device, c->tiles.ptr, c->device,
                                       per_part * sizeof(uint32_t), c->stream));
            if (i != 0) {
                HIP_TRY(hipEventRecord(c->ev1, c->stream));
                HIP_TRY(hipSetDevice(root->device));
                HIP_TRY(hipStreamWaitEvent(root->stream, c->ev1, 0));
            }
        }
    }

    // assemble on the root device and copy back the rectangle the reference writes
    HIP_TRY(hipSetDevice(root->device));
    FrameParams P0;
    rc = fill_params(root, frame, &part0, false, P0, root->stream); if (rc) return rc;
    launch_assemble(P0, root->gathered.ptr, root->fb.ptr, root->stream);
    HIP_TRY(hipGetLastError());
    const int row_hi = frame->height / 2 + g.half, row_lo = row_hi - (int)g.rendered_rows + 1;
    const int r0 = std::max(row_lo, 0), r1 = std::min(row_hi, frame->height - 1);
    const int c0 = std::max(frame->width / 2 - g.half, 0), c1 = std::min(frame->width / 2 + g.half - 1, frame->width - 1);
    uint32_t *target = host_fb;
    const bool staged = !page_locked(host_fb, (size_t)frame->width * frame->height * sizeof(uint32_t));            // (see ctgpu_ctx::stage)
    if (staged) { rc = stage_reserve(root, (size_t)frame->width * frame->height); if (rc) return rc; target = root->stage; }
    if (r1 >= r0 && c1 >= c0) {
        const size_t off = (size_t)r0 * frame->width + c0;
        HIP_TRY(hipMemcpy2DAsync(target + off, (size_t)frame->width * 4, root->fb.ptr + off, (size_t)frame->width * 4,
                                 (size_t)(c1 - c0 + 1) * 4, (size_t)(r1 - r0 + 1), hipMemcpyDeviceToHost, root->stream));
    }
    HIP_TRY(hipStreamSynchronize(root->stream));
    if (staged) stage_deliver(root->stage, host_fb, frame->width, r0, r1, c0, c1);
    const auto t1 = std::chrono::steady_clock::now();

    ctgpu_stats total;
    std::memset(&total, 0, sizeof total);
    for (int i = 0; i < n_ctx; i++) {
        ctgpu_ctx *c = ctxs[i];
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        ctgpu_stats st;
        std::memset(&st, 0, sizeof st);
        rc = read_counters(c, &st); if (rc) return rc;
        rc = check_overflow(c); if (rc) return rc;
        total.rays_primary += st.rays_primary; total.rays_shadow += st.rays_shadow; total.rays_reflect += st.rays_reflect;
        total.rays_shadow_elided += st.rays_shadow_elided;
    }
    HIP_TRY(hipSetDevice(root->device));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, root->ev0, root->ev1));
    total.kernel_ms = ms;                       // the root device's share of the tracing
    total.total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    total.pixels_written = (r1 >= r0 && c1 >= c0) ? (uint64_t)(r1 - r0 + 1) * (uint64_t)(c1 - c0 + 1) : 0;
    if (stats) *stats = total;
    return CTGPU_OK;
}

extern "C" int ctgpu_count_work(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, ctgpu_stats *stats) {
    if (!ctx || !stats) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_count_work: null argument");
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    int rc = check_frame(frame); if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const ctgpu_partition whole = {8, 0, 1};
    FrameParams P;
    rc = fill_params(ctx, frame, &whole, true, P, ctx->stream); if (rc) return rc;
    HIP_TRY(ctx->tiles.resize((size_t)P.local_rows * P.square));
    P.out = ctx->tiles.ptr;
    HIP_TRY(hipMemsetAsync(ctx->counters.ptr, 0, kCounterWords * sizeof(unsigned long long), ctx->stream));
    const bool ss = (frame->flags & CTGPU_F_SUPERSAMPLE) != 0;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    launch_trace_tiles(P, tile_count(P), ss, ctx->scene->has_reflection && frame->max_depth > 0, true, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::memset(stats, 0, sizeof *stats);
    rc = read_counters(ctx, stats); if (rc) return rc;
    rc = check_overflow(ctx); if (rc) return rc;
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    stats->kernel_ms = ms;
    return CTGPU_OK;
}

// ------------------------------------------------------------------ test hook
extern "C" int ctgpu_trace_rays(ctgpu_ctx *ctx, const ctgpu_ray *rays, uint64_t n, ctgpu_hit *hits) {
    if (!ctx || (n && (!rays || !hits))) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_trace_rays: null argument");
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return CTGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    ctgpu_frame_desc f;
    std::memset(&f, 0, sizeof f);
    f.width = 8; f.height = 8; f.num_threads = 1; f.max_depth = 0;
    f.cam_rot[0] = f.cam_rot[4] = f.cam_rot[8] = 1.0;
    const ctgpu_partition whole = {8, 0, 1};
    FrameParams P;
    int rc = fill_params(ctx, &f, &whole, false, P, ctx->stream); if (rc) return rc;
    ctgpu_ray *d_rays = nullptr;
    ctgpu_hit *d_hits = nullptr;
    HIP_TRY(hipMalloc((void **)&d_rays, n * sizeof(ctgpu_ray)));
    hipError_t e = hipMalloc((void **)&d_hits, n * sizeof(ctgpu_hit));
    if (e != hipSuccess) { (void)hipFree(d_rays); return ctgpu_fail(CTGPU_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpy(d_rays, rays, n * sizeof(ctgpu_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_trace_ray_list(P, d_rays, n, d_hits, ctx->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(hits, d_hits, n * sizeof(ctgpu_hit), hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    if (e != hipSuccess) return ctgpu_fail(CTGPU_ERR_HIP, "ctgpu_trace_rays: %s", hipGetErrorString(e));
    return check_overflow(ctx);
}

// The same hook through the kernel the frames are traced with: the caller's rays, 64 consecutive
// ones a packet, walked by ct_walk_packet (folded tree, binary32 filter, exact levels, unfolded
// second pass) exactly as the samples of a frame are.
extern "C" int ctgpu_trace_rays_packet(ctgpu_ctx *ctx, const ctgpu_ray *rays, uint64_t n, uint32_t flags, ctgpu_hit *hits) {
    if (!ctx || (n && (!rays || !hits))) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_trace_rays_packet: null argument");
    if (flags & ~(uint32_t)CTGPU_TRACE_ANY_HIT) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_trace_rays_packet: unknown flags %u", flags);
    if (!ctx->scene) return ctgpu_fail(CTGPU_ERR_NO_SCENE, "no scene uploaded");
    if (ctx->scene->quad_pending > 64u) return ctgpu_fail(CTGPU_ERR_STACK, "the packet walk does not serve this tree (%u entries pending)", ctx->scene->quad_pending);
    if (n == 0) return CTGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    ctgpu_frame_desc f;
    std::memset(&f, 0, sizeof f);
    f.width = 8; f.height = 8; f.num_threads = 1; f.max_depth = 0;
    f.cam_rot[0] = f.cam_rot[4] = f.cam_rot[8] = 1.0;
    const ctgpu_partition whole = {8, 0, 1};
    FrameParams P;
    int rc = fill_params(ctx, &f, &whole, false, P, ctx->stream, true); if (rc) return rc;       // (the tree with cull nodes, when the scene has them)
    PipelineParams W;
    rc = fill_pipeline(ctx, n, W); if (rc) return rc;
    W.packet = 1;
    ctgpu_ray *d_rays = nullptr;
    ctgpu_hit *d_hits = nullptr;
    HIP_TRY(hipMalloc((void **)&d_rays, n * sizeof(ctgpu_ray)));
    hipError_t e = hipMalloc((void **)&d_hits, n * sizeof(ctgpu_hit));
    if (e != hipSuccess) { (void)hipFree(d_rays); return ctgpu_fail(CTGPU_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpy(d_rays, rays, n * sizeof(ctgpu_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->queue.ptr, 0, queue_block_words(W.n_shadow_lights) * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        W.ray_list = d_rays;
        W.ray_hits = d_hits;
        launch_packet_ray_list(P, W, (flags & CTGPU_TRACE_ANY_HIT) != 0, ctx->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(hits, d_hits, n * sizeof(ctgpu_hit), hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    if (e != hipSuccess) return ctgpu_fail(CTGPU_ERR_HIP, "ctgpu_trace_rays_packet: %s", hipGetErrorString(e));
    return check_overflow(ctx);
}

// ------------------------------------------------------------------ measurement hooks
extern "C" int ctgpu_read_walk_work(ctgpu_ctx *ctx, ctgpu_walk_work *out) {
    if (!ctx || !out) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_read_walk_work: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long w[64 * 8];
    HIP_TRY(hipMemcpy(w, ctx->walk_work.ptr, sizeof w, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ctx->walk_work.ptr, 0, sizeof w));
    std::memset(out, 0, sizeof *out);
    for (int slot = 0; slot < 64; slot++)
        for (int k = 0; k < 2; k++) {
            out->packets[k] += w[slot * 8 + k * 3 + 0];
            out->quad_nodes[k] += w[slot * 8 + k * 3 + 1];
            out->triangle_records[k] += w[slot * 8 + k * 3 + 2];
        }
    for (int slot = 0; slot < 64; slot++) {
        out->shadow_rays_blocked += w[slot * 8 + 6];
        out->shadow_packets_all_blocked += w[slot * 8 + 7];
    }
    // (counted in the frame counters, words 17 / 18: what earlier counter reads collected + what is there now)
    unsigned long long handed[2];
    HIP_TRY(hipMemcpy(handed, ctx->counters.ptr + 17, sizeof handed, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ctx->counters.ptr + 17, 0, sizeof handed));
    out->rays_handed_over[0] = ctx->handed_over[0] + handed[0];
    out->rays_handed_over[1] = ctx->handed_over[1] + handed[1];
    ctx->handed_over[0] = ctx->handed_over[1] = 0;
    return CTGPU_OK;
}

extern "C" int ctgpu_walk_choice(ctgpu_ctx *ctx, int supersampled, int *per_lane, double *packet_ms, double *lane_ms) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_walk_choice: null argument");
    const ctgpu_ctx::WalkChoice &c = ctx->choice[supersampled ? 1 : 0];
    if (per_lane) *per_lane = ctx->walk_auto ? (c.use_lane ? 1 : 0) : (ctx->packet_walk ? 0 : 1);
    if (packet_ms) *packet_ms = c.packet_ms;
    if (lane_ms) *lane_ms = c.lane_ms;
    return CTGPU_OK;
}

extern "C" int ctgpu_order_info(ctgpu_ctx *ctx, uint32_t entries[8], int *split) {
    if (!ctx || !entries || !split) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_order_info: null argument");
    static_assert(kOrderClasses == 4, "ctgpu_order_info reports four classes per walk");
    for (int k = 0; k < 8; k++) entries[k] = 0;
    *split = (ctx->order.last_ordered && ctx->order.last_split) ? 1 : 0;
    if (!ctx->order.last_ordered || !ctx->order.count.ptr) return CTGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(entries, ctx->order.count.ptr + (size_t)(ctx->order.parity ^ 1) * 2 * kOrderClasses, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CTGPU_OK;
}

extern "C" int ctgpu_bounce_choice(ctgpu_ctx *ctx, int supersampled, int *general_path, double *pipeline_ms, double *general_ms) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_bounce_choice: null argument");
    const ctgpu_ctx::BounceChoice &c = ctx->bounce_choice[supersampled ? 1 : 0];
    if (general_path) *general_path = ctx->bounce_auto ? (c.use_general ? 1 : 0) : (ctx->bounce_on_pipeline ? 0 : 1);
    if (pipeline_ms) *pipeline_ms = c.pipeline_ms;
    if (general_ms) *general_ms = c.general_ms;
    return CTGPU_OK;
}

extern "C" int ctgpu_kernel_timing(ctgpu_ctx *ctx, int enable) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_kernel_timing: null argument");
    ctx->kernel_timing = enable != 0;
    ctx->marks_used = 0;
    ctx->phase_marks_used = 0;
    return CTGPU_OK;
}

extern "C" int ctgpu_read_phase_times(ctgpu_ctx *ctx, ctgpu_phase_times *out) {
    if (!ctx || !out) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_read_phase_times: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof *out);
    for (size_t f = 0; f + 4 <= ctx->phase_marks_used; f += 4) {
        HIP_TRY(hipEventSynchronize(ctx->phase_marks[f + 3]));
        float ms[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) HIP_TRY(hipEventElapsedTime(&ms[k], ctx->phase_marks[f + k], ctx->phase_marks[f + k + 1]));
        out->trace_ms += ms[0]; out->gather_ms += ms[1]; out->assemble_ms += ms[2];
        out->frames += 1;
    }
    ctx->phase_marks_used = 0;
    return CTGPU_OK;
}

extern "C" int ctgpu_read_kernel_times(ctgpu_ctx *ctx, ctgpu_kernel_times *out) {
    if (!ctx || !out) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_read_kernel_times: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof *out);
    const size_t per_frame = kPipelineKernels + 1;
    static_assert(kPipelineKernels == CTGPU_PIPELINE_KERNELS, "ctgpu_kernel_times layout");
    for (size_t f = 0; f + per_frame <= ctx->marks_used; f += per_frame) {
        HIP_TRY(hipEventSynchronize(ctx->marks[f + kPipelineKernels]));
        for (int k = 0; k < kPipelineKernels; k++) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, ctx->marks[f + k], ctx->marks[f + k + 1]));
            out->total_ms[k] += ms;
            out->launches[k] += 1;
        }
    }
    ctx->marks_used = 0;
    return CTGPU_OK;
}
