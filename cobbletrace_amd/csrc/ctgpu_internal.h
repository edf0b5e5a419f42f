// ctgpu_internal.h -- what ctgpu_api.hip and ctgpu_comm.hip share (not part of the C ABI).
#ifndef CTGPU_INTERNAL_H
#define CTGPU_INTERNAL_H

#include "ct_params.h"

#include <string>
#include <vector>

#define CTGPU_HIDDEN __attribute__((visibility("hidden")))

struct ctgpu_comm_state;                                        // ctgpu_comm.hip

// records the message ctgpu_last_error() returns, gives back `code`
CTGPU_HIDDEN int ctgpu_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
CTGPU_HIDDEN ctgpu_comm_state *&ctgpu_comm_slot(ctgpu_ctx *ctx);
CTGPU_HIDDEN int ctgpu_device_of(const ctgpu_ctx *ctx);
// the context's own tile buffer (n_tiles uint32) and, where n_all > 0, its gather buffer
CTGPU_HIDDEN int ctgpu_shard_buffers(ctgpu_ctx *ctx, uint64_t n_tiles, uint64_t n_all, void **tiles, void **all);

// What tells one frame's gather from another's when several contexts share a communicator (ctgpu_comm_share): the context's
// place among the sharers and the number of sharded frames it has issued.  Every rank stamps its gather with the pair; the
// root compares what arrived (ctgpu_comm.hip, "the order of the gathers").
struct ctgpu_comm_stamp {
    uint32_t index = 0;          // 0: the context that made the communicator, 1, 2, ...: the sharers in the order they joined
    uint64_t frames = 0;         // sharded frames this context has issued
    int64_t skew = 0;            // ctgpu_comm_debug_skew: added to what is stamped (never to what is expected)
};
CTGPU_HIDDEN ctgpu_comm_stamp &ctgpu_comm_stamp_of(ctgpu_ctx *ctx);
// device words for the stamps: mine[2], and on the root all[world][2]; flag: the context's error word (4: stamps differ)
CTGPU_HIDDEN int ctgpu_comm_stamp_buffers(ctgpu_ctx *ctx, int world, bool root, unsigned long long **mine, unsigned long long **all, uint32_t **flag);

// ctgpu_render_sharded's phase marks: which = 0 start, 1 traced, 2 gathered, 3 assembled (no-op unless timing is on)
CTGPU_HIDDEN int ctgpu_phase_mark(ctgpu_ctx *ctx, int which, hipStream_t stream);

// ---- scene on the device (ct_build.hip, ct_pack.hip)
// BuildBVH on the device from vertices that are already there; the finished arrays STAY there: *d_nodes (n_nodes
// records) and *d_indexes (n) are hipMalloc'ed blocks the caller owns.  depth = levels of the tree (root = 1).
CTGPU_HIDDEN int ctgpu_build_bvh_resident(int device, const double *d_vertices, uint32_t n, ctgpu_node **d_nodes,
                                          uint32_t **d_indexes, uint32_t *n_nodes, uint32_t *depth, double *device_ms);
// what the walks read, produced on the device from the reference's arrays (all pointers: hipMalloc'ed, caller owns)
struct ctgpu_packed_scene {
    void *wide = nullptr;              // WideNode[n_wide]
    void *quads = nullptr, *quads_flat = nullptr;          // QuadNode[n_quads], QuadNode[n_quads_flat]
    double *quad_boxes = nullptr;      // [n_quads][4][6] the folded tree's entries' binary64 boxes
    uint32_t *quad_node_of_flat = nullptr;                  // [n_quads_flat][4] the unfolded tree's entries' reference nodes (their boxes: the reference's node array)
    void *tris = nullptr;              // TriRecord[n_triangles]
    uint32_t n_wide = 0, n_quads = 0, n_quads_flat = 0;
    uint32_t quad_root = 0, quad_root_flat = 0, quad_pending = 1;
    uint32_t root_ref = 0, slot_of_triangle0 = 0;
    float plane_bound[3] = {0, 0, 0};
    double root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    double device_ms = 0.0;
};
CTGPU_HIDDEN int ctgpu_pack_on_device(int device, const ctgpu_node *d_nodes, uint32_t n_nodes, const uint32_t *d_indexes,
                                      const double *d_vertices, uint32_t n_triangles, ctgpu_packed_scene *out);

// the cull nodes of the folded tree's big leaves (ct_device.h "cull nodes"), made from packed TriRecords / QuadNodes that
// are already on the device: a second QuadNode array (the tree with cull nodes behind it) beside the plain one
struct ctgpu_cull_scene {
    void *quads = nullptr;                                 // QuadNode[cull_first + n_cull], hipMalloc'ed, caller owns (NULL: no cull nodes)
    uint32_t quad_root = 0;                                // root reference of that array's tree
    uint32_t cull_first = 0xffffffffu;                     // first cull node in it (kNoNode: none)
    uint32_t n_cull = 0, n_big_leaves = 0, levels = 0;     // cull nodes, leaves that got them, levels of the tallest such tree
    uint32_t triangles_in_big_leaves = 0;
    float edge_bound = 0, vertex_bound = 0;                // >= every |e1|, |e2| component; >= every |vertex coordinate|
    float plane_bound[3] = {0, 0, 0};                      // >= |plane| of every triangle box, per axis
    bool usable = false;                                   // false: some coordinate is not finite -- no cull for this scene
};
CTGPU_HIDDEN int ctgpu_pack_cull_on_device(int device, const void *d_tris, uint32_t n_triangles, const void *d_quads, uint32_t n_quads,
                                           uint32_t quad_root, uint32_t big_min, uint32_t share_percent, const float plane_bound[3],
                                           ctgpu_cull_scene *out);

// (ct_pack.hip) the triangles' grown boxes (appendix D.1) and the scene's magnitudes: d_box[n][6] = {lo xyz, hi xyz} per
// TriRecord slot; stats: 0 edge bound, 1 vertex bound, 2..4 plane bound per axis (float bits), 5 "something is not finite"
CTGPU_HIDDEN int ctgpu_triangle_boxes_on_device(const void *d_tris, uint32_t n_triangles, float *d_box, uint32_t stats[8]);

// the shadow tree (ct_device.h "the shadow tree", ct_shadow_tree.hip): made from the packed TriRecords and the reference's
// node array, both already on the device.  strategy: 0 the binary tree clustered on the device (the host's binned builder
// takes over should the clustering not finish), 1 binned surface-area heuristic on host threads, 2 object medians (host);
// leaf_max / leaf_rho: a leaf holds up to leaf_max triangles where their boxes fill the leaf's (ct_shadow_tree_build.h).
struct ctgpu_shadow_tree {
    void *nodes = nullptr;                 // QuadNode[n_nodes], root = node 0; hipMalloc'ed, caller owns (NULL: the scene gets none)
    void *records = nullptr;               // TriRecord[n_triangles] in the tree's own order (a leaf is a run); hipMalloc'ed, caller owns
    uint32_t *leaf_of_slot = nullptr;      // [n_triangles] slot of `records` -> reference leaf node; hipMalloc'ed, caller owns
    uint32_t n_nodes = 0;
    uint32_t pending = 0;                  // most entries ever waiting on the walk's stack (<= 63)
    float d_min = 0, d_max = 0, o_max = 0; // the gate (ShadowTreeLimits)
    double gamma = 0;                      // what every stored plane was moved outward by
    float tau = 0;                         // closest-hit walks prune entries beyond ray.t only while ray.t >= tau
    double build_ms = 0;                   // wall time of the build, copies included
    bool usable = false;
};
CTGPU_HIDDEN int ctgpu_build_shadow_tree(int device, const void *d_tris, uint32_t n_triangles, const ctgpu_node *d_ref_nodes,
                                         uint32_t n_nodes, int strategy, uint32_t leaf_max, float leaf_rho, ctgpu_shadow_tree *out);

// one process, several devices (ctgpu_render_multi)
CTGPU_HIDDEN int ctgpu_multi_comms(const int *devices, int n, std::vector<void *> &comms);
CTGPU_HIDDEN void ctgpu_multi_comms_release(std::vector<void *> &comms, bool abort = false);
CTGPU_HIDDEN int ctgpu_multi_gather(const std::vector<void *> &comms, const int *devices, void *const *d_tiles, uint64_t n,
                                    void *d_all, hipStream_t const *streams);

#endif
