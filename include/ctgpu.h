/*
 * ctgpu.h -- C ABI of the MI355X ray-trace library (libctgpu.so).
 *
 * This is the boundary a host application binds to in place of CobbleTrace's
 * Boss/Worker pool.  Plain C: opaque handle, POD descriptors, pointers + sizes,
 * int status (0 = ok, <0 = error; text via ctgpu_last_error()).
 *
 *   entry point                 replaces (reference file:line)
 *   --------------------------  ----------------------------------------------------------
 *   ctgpu_upload_scene          GetSceneTriangles + InitializeBVHState hand-over of the
 *                               triangle / node / index arrays   raythread.cpp:621-654, bvh.cpp:16-28
 *   ctgpu_render                HandleUpdates dispatch + all RayTracePartition workers for one
 *                               frame, i.e. the pixel loop -> TraceRay -> ClosestIntersection ->
 *                               IntersectBVHClosest / ComputeLighting -> CanvasPutPixel
 *                               raythread.cpp:437-543, 353-386, 197-228, 275-350, 176-184;
 *                               bvh.cpp:147-222; color.h:18-126; draw2d.h:8-20
 *   ctgpu_render_multi          the same with the row bands of HandleUpdates (raythread.cpp:574-588)
 *                               dealt to several GPUs of one process (RCCL gather between them)
 *   ctgpu_comm_* /              the same for one process per GPU: communicator set-up, the gather, and one
 *   ctgpu_gather_tiles /        rank's whole share of a frame (trace, gather, assemble on rank 0)
 *   ctgpu_render_sharded
 *   ctgpu_render_partition /    the row-band split of HandleUpdates (raythread.cpp:574-588)
 *   ctgpu_assemble              re-cut into interleaved row tiles for N GPUs, and the merge of
 *                               the per-GPU tiles into the caller's framebuffer
 *   ctgpu_trace_rays /          ClosestIntersection on caller-supplied rays (test / KAT hook),
 *   ctgpu_trace_rays_packet     through the general kernel / through the frame kernel
 *                               raythread.cpp:197-228, 303-307
 *   ctgpu_build_bvh             InitializeBVHState + BuildBVH (bvh.cpp:16-121) on the device, identical arrays
 *   ctgpu_count_work            instrumented run returning node-visit / triangle-test counts
 *                               (the reference has gprof only)
 *
 * Pixel format: uint32 0x00BBGGRR, row-major, top-down, pitch 4*width  (color.h:77-80,
 * draw2d.h:8-20).  Pixels the reference never writes are never written here either.
 */
#ifndef CTGPU_H
#define CTGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTGPU_ABI_VERSION 4

typedef struct ctgpu_ctx ctgpu_ctx;

/* ---- status codes ---- */
#define CTGPU_OK              0
#define CTGPU_ERR_ARG        -1   /* bad argument / descriptor                      */
#define CTGPU_ERR_HIP        -2   /* a HIP runtime call failed                      */
#define CTGPU_ERR_NO_SCENE   -3   /* render before upload                           */
#define CTGPU_ERR_STACK      -4   /* traversal stack overflow (tree deeper than the
                                     library's capacity, CTGPU_MAX_TREE_DEPTH)      */
#define CTGPU_ERR_NO_DEVICE  -5
#define CTGPU_ERR_COMM       -6   /* RCCL: library missing, no communicator, or a failed call */

#define CTGPU_MAX_TREE_DEPTH 64

/* ---- scene (all arrays are copied; the caller keeps ownership) ---- */
typedef struct ctgpu_material {      /* scenefile.h:25-29 */
    uint32_t color;                  /* 0x00BBGGRR */
    int32_t  specular;               /* -1 = no specular term */
    float    reflection;
} ctgpu_material;

typedef struct ctgpu_node {          /* bvh.h:5-11, 64 bytes */
    double   bmin[3], bmax[3];
    uint32_t left;                   /* left child; right child = left + 1  */
    uint32_t first;                  /* first slot in indexes[]             */
    uint32_t count;                  /* >0 = leaf                           */
    uint32_t pad_;
} ctgpu_node;

#define CTGPU_LIGHT_POINT       0    /* scenefile.h:13 */
#define CTGPU_LIGHT_DIRECTIONAL 1
#define CTGPU_LIGHT_AMBIENT     2
typedef struct ctgpu_light {         /* scenefile.h:61-66, 56 bytes */
    int32_t type;
    float   intensity;
    double  position[3];
    double  direction[3];
} ctgpu_light;

/* A sphere object of the scene file (scenefile.h:31-34, scenefile.cpp:58-71).  The reference parses spheres
 * but no longer renders them ("TODO add sphere's to the bvh", raythread.cpp:207; README.md:20); this build can,
 * on request (CTGPU_F_SPHERES), from a second tree over them. */
typedef struct ctgpu_sphere {
    double center[3];
    float  radius;
    uint32_t pad_;
    ctgpu_material material;
    uint32_t pad2_;
} ctgpu_sphere;                      /* 48 bytes */

typedef struct ctgpu_scene_desc {
    uint32_t n_triangles;
    uint32_t n_nodes;                /* nodesUsed */
    uint32_t n_lights;
    uint32_t pad_;
    const double         *vertices;  /* [n_triangles][9]: p1 p2 p3 in scene-triangle order */
    const ctgpu_material *materials; /* [n_triangles] */
    const ctgpu_node     *nodes;     /* [n_nodes], node 0 = root */
    const uint32_t       *indexes;   /* [n_triangles] leaf slot -> triangle */
    const ctgpu_light    *lights;    /* [n_lights], file order */
    /* optional (all zero / NULL: none): the scene's spheres in object order and a tree over them in the
     * same node format (leaf slots index sphere_indexes[]), built like the triangle tree (bvh.cpp:16-121) over
     * each sphere's bounding cube */
    uint32_t n_spheres, n_sphere_nodes;
    const ctgpu_sphere   *spheres;
    const ctgpu_node     *sphere_nodes;
    const uint32_t       *sphere_indexes;
} ctgpu_scene_desc;

/* ---- frame ---- */
#define CTGPU_F_SUPERSAMPLE 1u       /* settings.supersampling, raythread.cpp:460-505 */
#define CTGPU_F_SUBSAMPLE   2u       /* settings.subsampling,   raythread.cpp:512-531 */
/* Two things the reference has in its source but not in its pictures; both off unless asked for, so that
 * the default frame stays the reference's:
 * SPHERES: render the scene's sphere objects too -- IntersectRaySphere (raythread.cpp:158-174) under the
 *   acceptance rule of the reference's own pre-BVH loop (the commented block raythread.cpp:230-246:
 *   tmin <= t && t <= tmax && t <= tclosest for t1, then t2), tmin / tmax as TraceRay and ComputeLighting
 *   pass them (1 / 0.001 .. FINF, shadows 0.001 .. 1 for a point light), the sphere normal of
 *   NormalOfSceneObject (raythread.cpp:331-334); spheres are visited after the triangles, through their tree.
 * REFLECT_FIX: a reflected ray starts with ray.t = 1e30f like every other ray instead of the 0 of
 *   raythread.cpp:373, which makes IntersectBVHClosest reject everything in front of it. */
#define CTGPU_F_SPHERES      4u
#define CTGPU_F_REFLECT_FIX  8u

#define CTGPU_RNG_HASH 0             /* counter-based jitter, independent of thread count */
#define CTGPU_RNG_LIBC 1             /* glibc rand() stream in the call order of ONE worker */

typedef struct ctgpu_frame_desc {
    int32_t  width, height;          /* bitmap size */
    int32_t  num_threads;            /* T: rows rendered = T * (height / T)  (raythread.cpp:576) */
    uint32_t flags;                  /* CTGPU_F_* */
    double   cam_pos[3];             /* camera.position */
    double   cam_rot[9];             /* camera.rotation, row-major */
    int32_t  rng_mode;               /* CTGPU_RNG_* (used when supersampling) */
    int32_t  max_depth;              /* reflection recursion limit; the reference passes 10 */
    uint64_t rng_seed;               /* HASH: seed.  LIBC: srand() seed (reference default 1) */
    uint64_t rng_offset;             /* LIBC: rand() calls already consumed by earlier frames */
} ctgpu_frame_desc;

typedef struct ctgpu_stats {
    uint64_t rays_primary, rays_shadow, rays_reflect;   /* BVH traversals started */
    uint64_t pixels_written;
    uint64_t node_visits, triangle_tests;               /* only from ctgpu_count_work */
    double   kernel_ms;                                 /* HIP-event time of the trace kernel(s) */
    double   total_ms;                                  /* dispatch -> pixels in the destination */
    uint64_t rays_shadow_elided;     /* shadow rays ComputeLighting would start (raythread.cpp:303) whose answer
                                        cannot change the pixel (n.l <= 0 and no specular reach) and which are
                                        therefore not traced: rays_shadow + rays_shadow_elided is the reference's count */
    /* Which code rendered the context's latest frame (several paths give the same pixels; a context picks by scene, by frame
     * kind and -- where ctgpu_config says "measured" -- by its own timings, so a caller who measures wants to know): */
    uint32_t walk_kind;              /* CTGPU_WALK_*                                                                    */
    uint32_t bounce_kind;            /* CTGPU_BOUNCE_*: how TraceRay's recursion ran (frames whose materials reflect)   */
    uint32_t ordered;                /* 1: heavy chunks were handed out first (ctgpu_order_info)                        */
    uint32_t cull_tree;              /* 1: the tree the frame walked carries cull nodes in place of big leaves          */
    uint32_t shadow_tree;            /* 1: the shadow rays walked the shadow tree (ctgpu_config.shadow_tree)            */
    uint32_t closest_tree;           /* 1: the primary rays walked it too (ctgpu_config.closest_tree)                   */
    uint64_t rays_shadow_left;       /* ... and this many of them it left to the walk of the reference's tree (rays outside
                                        its gate; counted in rays_shadow like the others)                               */
    uint64_t rays_primary_left;      /* the same for the primary rays: outside the gate, or a closest hit on the rim of its
                                        leaf's box (counted in rays_primary like the others)                             */
} ctgpu_stats;
#define CTGPU_WALK_GENERAL   0       /* the general path's one kernel (reflective frames it was chosen for, CTGPU_FORCE_MEGAKERNEL) */
#define CTGPU_WALK_PER_LANE  1       /* wavefront pipeline, every lane its own walk (ct_walk)                          */
#define CTGPU_WALK_PACKET    2       /* wavefront pipeline, the shared walk (ct_walk_packet)                           */
#define CTGPU_BOUNCE_NONE    0
#define CTGPU_BOUNCE_LEVELS  1       /* bounce levels on the pipeline                                                  */
#define CTGPU_BOUNCE_GENERAL 2       /* inside the general path's kernel                                               */

/* ---- partition of one frame over several GPUs ---- */
typedef struct ctgpu_partition {
    int32_t tile_rows;               /* rows per tile (>=1)                      */
    int32_t part_index, part_count;  /* this GPU renders tiles t with t % count == index */
} ctgpu_partition;

/* ---- test hook ---- */
typedef struct ctgpu_ray { double origin[3]; double direction[3]; float t; uint32_t pad_; } ctgpu_ray;
typedef struct ctgpu_hit {
    float    t;          /* ray.t after traversal (1e30f = nothing)   */
    float    tclosest;   /* 4294967296.f when nothing was recorded    */
    uint32_t triangle;   /* closestIndex (0 when nothing was recorded)*/
    uint32_t found;      /* ray.t != 1e30f                            */
} ctgpu_hit;

/* ---- entry points ---- */
int  ctgpu_abi_version(void);
/* Which sources this binary was made from: the first 16 hex digits of the sha256 over the .hip and .h files of cobbletrace_amd/csrc
 * and the two public headers, in name order, taken by the Makefile when the library was linked.  Whoever runs a prebuilt
 * library against a source tree (the tests, smoke(), bench.py) compares it with the same hash of the files on disk
 * (cobbletrace_amd.bindings.source_build_id) and rebuilds or refuses when they differ. */
const char *ctgpu_build_id(void);
const char *ctgpu_last_error(void);
int  ctgpu_device_count(int *count);

/* ---- configuration ----
 * Every knob of a context in one place.  None changes a pixel (tests/test_gpu_parity.py::
 * test_pipeline_knobs_do_not_change_pixels; the GPU suite is run under the main ones); they trade speed between
 * workloads and switch measurement / experiment code.  ctgpu_default_config gives the built-in values,
 * ctgpu_config_from_environment those overridden by the environment variable named beside each field (malformed values
 * are ignored), ctgpu_create uses the latter, ctgpu_create_with_config what the caller hands it, ctgpu_get_config reports
 * what a context runs with.  -1 in a "measured" field: the context tries both and keeps the faster (same pixels). */
typedef struct ctgpu_config {
    int32_t packet_walk;          /* CTGPU_PACKET         -1 measured per scene (ctgpu_walk_choice), 1 the shared (packet) walk, 0 the per-lane walk */
    int32_t bounce;               /* CTGPU_BOUNCE         reflective frames: -1 measured (ctgpu_bounce_choice), 1 bounce levels, 0 the general path's one kernel */
    int32_t force_megakernel;     /* CTGPU_FORCE_MEGAKERNEL  1: the general path for every frame */
    int32_t keep_shadow_rays;     /* CTGPU_KEEP_SHADOW_RAYS  1: trace also the shadow rays whose answer cannot change a pixel (ct_light_gate) */
    int32_t blocks_per_cu;        /* CTGPU_BLOCKS_PER_CU  persistent waves per CU of the walk launches (0: what fits; fewer suits several frames in flight) */
    int32_t chunk, chunk_shadow;  /* CTGPU_CHUNK, CTGPU_CHUNK_SHADOW  ray ids a wave takes per queue atomic (multiple of 64; 0: by frame size / = chunk) */
    int32_t lds_depth;            /* CTGPU_LDS_DEPTH      per-lane walk: stack levels kept in LDS */
    int32_t packet_lds_levels;    /* CTGPU_PACKET_LDS_LEVELS  shared walk: per-lane tmin levels kept in LDS */
    int32_t probe_above;          /* CTGPU_PROBE_ABOVE    fetches per packet beyond which the per-lane walk gets a trial frame */
    int32_t packet_budget, packet_lanes;  /* CTGPU_PACKET_BUDGET, CTGPU_PACKET_LANES  hand-over experiment (0: off, the default) */
    int32_t order;                /* CTGPU_ORDER          heavy chunks first: 0 off, 1 frames with few packets per wave (default), 2 every frame, 3 pipelined frames too */
    float   order_heavy[4];       /* CTGPU_ORDER_HEAVY    "a,b,c,d": a chunk is of class k from this many times the mean packet's fetches */
    int32_t order_floor;          /* CTGPU_ORDER_FLOOR    ... and never below this many fetches */
    int32_t order_split_below;    /* CTGPU_ORDER_SPLIT_BELOW  chunks per wave of the grid below which heavy chunks are cut into parts */
    int32_t order_parts[4];       /* CTGPU_ORDER_PARTS    "a,b,c,d": parts a chunk of class k is cut into (1, 2, 4, 8, 16) */
    float   shadow_split_gap;     /* CTGPU_SHADOW_SPLIT   shadow packets: the hit samples of a window (two 2x2-pixel blocks) walk as two packets when their
                                     primary hits' depths fall into two ranges this far apart, relative to the middle depth (< 0: never) */
    int32_t big_leaf_min;         /* CTGPU_BIG_LEAF       leaves of at least this many triangles get cull nodes (0: none) */
    int32_t big_leaf_share;       /* CTGPU_BIG_LEAF_SHARE ... in scenes where at least this percentage of the triangles sits in such leaves */
    int32_t shadow_tree;          /* CTGPU_SHADOW_TREE    the frame's shadow rays walk a tree of this library's own over the same triangles (order-free
                                     any-hit query, raythread.cpp:303-307; DESIGN appendix E): 0 never, 1 supersampled frames only, 2 every frame (default) */
    int32_t closest_tree;         /* CTGPU_CLOSEST_TREE   the frame's PRIMARY rays walk that tree too (the least t over every test the reference
                                     could make, the winner checked against the reference's own leaf box; DESIGN appendix F): 0 never, 1
                                     supersampled frames (default), 2 every frame.  Needs shadow_tree != 0 (it is the same tree) */
    int32_t shadow_tree_build;    /* CTGPU_SHADOW_TREE_BUILD  how that tree is made: 0 clustered on the device (default; nearest neighbours along a
                                     space-filling curve merge round by round), 1 binned surface-area heuristic on host threads, 2 object medians (host) */
    int32_t shadow_tree_leaf;     /* CTGPU_SHADOW_TREE_LEAF   most triangles one of its leaves holds (1..4, default 2) ... */
    int32_t shadow_tree_leaf_fill;/* CTGPU_SHADOW_TREE_LEAF_FILL  ... and c > 1 of them only where c - sum(box area of each) / (box area of all) <= this
                                     percentage of c (default 50): the triangles' boxes nearly fill the leaf's, a box test each would exclude little */
    int32_t pack_on_host;         /* CTGPU_PACK=host      1: re-lay-out an uploaded tree on the host (the restatement the device packing is tested against) */
    int32_t thresh[3];            /* CTGPU_THRESH         "a,b,c": general path, lanes that make a phase worth running */
    int32_t walk_thresh[3];       /* CTGPU_WALK_THRESH    "r,a,b": per-lane walk, the same (diagnostic builds: the middle number picks the walk they count) */
    uint32_t debug;               /* CTGPU_DEBUG_PHASES (1), CTGPU_DEBUG_TRI (2), CTGPU_DEBUG_REDO (4), CTGPU_DEBUG_PACKET (8): diagnostics on stderr */
} ctgpu_config;
#define CTGPU_DEBUG_PHASES 1u
#define CTGPU_DEBUG_TRI    2u
#define CTGPU_DEBUG_REDO   4u
#define CTGPU_DEBUG_PACKET 8u
int  ctgpu_default_config(ctgpu_config *out);
int  ctgpu_config_from_environment(ctgpu_config *out);
/* (not in the struct because it is not per context: CTGPU_RCCL_LIBRARY, the librccl to bind -- default: one the process
 *  already holds, else ROCm's) */

/* A context is bound to one GPU.  ctgpu_create: configuration from the environment, read here, once. */
int  ctgpu_create(int device, ctgpu_ctx **out);
int  ctgpu_create_with_config(int device, const ctgpu_config *config, ctgpu_ctx **out);
int  ctgpu_get_config(const ctgpu_ctx *ctx, ctgpu_config *out);
void ctgpu_destroy(ctgpu_ctx *ctx);

/* Scene -> HBM (raythread.cpp:641-654: GetSceneTriangles + InitializeBVHState + BuildBVH at RayThread's first call).
 *   scene->nodes / indexes given: the reference's arrays from any builder (this repository's host builder, the
 *     reference's own BuildBVH).  Validated on the host, copied as they are, re-laid-out for the walks by kernels
 *     (ct_pack.hip; CTGPU_PACK=host does that packing on the host instead -- the restatement the kernels are tested
 *     against, two orders of magnitude slower).
 *   scene->nodes == indexes == NULL, n_nodes == 0: BuildBVH runs on the device too (ct_build.hip, bvh.cpp:16-121: the
 *     same arrays byte for byte) and the tree never leaves it -- vertices in, frames out.  ctgpu_download_tree hands
 *     bvhNodes[] / indexes[] back to a host that wants them (nodes_out: room for 2N-1, indexes_out: N; either may
 *     be NULL).
 * ctgpu_share_scene: a second context of the SAME device takes the owner's scene (one copy in HBM for all frames
 * in flight; it lives as long as any context holds it).  ctgpu_scene_info: sizes, device time of build and packing. */
int  ctgpu_upload_scene(ctgpu_ctx *ctx, const ctgpu_scene_desc *scene);
int  ctgpu_share_scene(ctgpu_ctx *ctx, ctgpu_ctx *owner);
int  ctgpu_download_tree(ctgpu_ctx *ctx, ctgpu_node *nodes_out, uint32_t *indexes_out, uint32_t *n_nodes);
typedef struct ctgpu_scene_info_t {
    uint32_t n_triangles, n_nodes, tree_depth, n_wide, n_quads, n_quads_flat;
    uint32_t quad_root, quad_root_flat, quad_pending, root_ref, slot_of_triangle0, shared_by;
    float    plane_bound[3];
    uint32_t pad_;
    double   build_ms, pack_ms;      /* device time of BuildBVH (0 when the caller brought the tree) and of the packing */
    uint64_t hbm_bytes;              /* what the scene occupies */
    uint32_t shadow_tree_nodes;      /* the shadow tree: its QuadNodes (0: the scene has none), */
    uint32_t shadow_tree_pending;    /*   the most entries its walk ever has waiting, */
    double   shadow_tree_ms;         /*   wall time of its build, */
    double   shadow_tree_gamma;      /*   what its planes were moved outward by */
} ctgpu_scene_info_t;
int  ctgpu_scene_info(ctgpu_ctx *ctx, ctgpu_scene_info_t *out);
/* test hook: one packed array as bytes (0 WideNodes, 1 QuadNodes, 2 their boxes, 3 unfolded QuadNodes, 4 their entries'
 * reference nodes, 5 TriRecords, 6 the QuadNodes with the cull nodes behind them, 7 the shadow tree's QuadNodes, 8 that tree's
 * record slot -> reference leaf node, 9 its TriRecords, in its own order); *needed = its size, copied when `bytes` is enough */
int  ctgpu_debug_scene_array(ctgpu_ctx *ctx, int which, void *out, uint64_t bytes, uint64_t *needed);

/* One whole frame on this context's GPU into a caller-owned host framebuffer
 * (width*height uint32).  Synchronous.  stats may be NULL.  host_fb may be ordinary pageable memory: the frame then
 * arrives in a page-locked buffer of the context's own and is copied over by the CPU (no copy queued on a stream ever
 * targets pageable memory: the runtime's own pinning of C-heap memory goes stale as the heap's top is returned to the
 * system and grown again, and ends in "Memory access fault by GPU"); page-locked memory is written directly. */
int  ctgpu_render(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb, ctgpu_stats *stats);

/* The same frame without blocking the caller (raythread.cpp:546-593: the reference's RayThread dispatches its
 * workers and looks at their status on later calls).  begin queues trace + assemble + the copy into host_fb on
 * the context's stream (host_fb is page-locked for the frame and must stay valid until poll reports it done; one
 * frame in flight per context); poll says whether it has finished -- or waits for it when `wait` is non-zero --
 * and then fills `stats`.  Once poll has reported the frame done (or failed) the page-lock is gone and the context
 * takes the next frame.  A host that hands over the SAME long-lived framebuffer every frame (the reference's
 * env->bitmap) calls ctgpu_keep_framebuffer_pinned(ctx, 1) to keep the page-lock between frames, and (ctx, 0) --
 * or ctgpu_destroy -- before it frees that memory.  ctgpu_upload_scene and ctgpu_render_multi refuse a context
 * with a frame in flight. */
int  ctgpu_render_begin(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb);
int  ctgpu_render_poll(ctgpu_ctx *ctx, int wait, int *done, ctgpu_stats *stats);
int  ctgpu_keep_framebuffer_pinned(ctgpu_ctx *ctx, int keep);

/* The same frame split over several contexts of THIS process (one per GPU): interleaved row tiles, an RCCL
 * gather (one communicator per device, ncclGather in one group) to the first context's device, assembly
 * there.  All contexts must hold the same scene.  Contexts that share a device are allowed (RCCL cannot
 * address a device twice: their tiles travel as plain device copies instead; the same peer copies over xGMI
 * take over, for the rest of the context's life, when RCCL cannot be loaded or its communicators fail).  This is what RayThread uses when
 * it is given more than one device (raythread.cpp:574-618 had one worker thread per band). */
int  ctgpu_render_multi(ctgpu_ctx *const *ctxs, int n_ctx, const ctgpu_frame_desc *frame, uint32_t *host_fb,
                        ctgpu_stats *stats);

/* Multi-GPU building blocks (device pointers; `stream` is a hipStream_t or NULL).
 * ctgpu_partition_pixels: number of uint32 a partition's compact tile buffer holds
 *   (same for every part_index, so that a gather has equal counts).
 * ctgpu_render_partition: trace this partition's tiles into d_tiles (asynchronous).
 * ctgpu_assemble: merge part_count tile buffers laid out back to back in d_all_tiles
 *   (as an RCCL gather / all-gather leaves them) into the device framebuffer d_fb
 *   (width*height uint32, pre-filled by the caller with whatever unwritten pixels should show). */
int  ctgpu_partition_pixels(const ctgpu_frame_desc *frame, const ctgpu_partition *part, uint64_t *count);
int  ctgpu_render_partition(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, const ctgpu_partition *part,
                            void *d_tiles, void *stream);
int  ctgpu_assemble(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, const ctgpu_partition *part,
                    const void *d_all_tiles, void *d_fb, void *stream);
/* One process per GPU (raythread.cpp:574-618 had one worker thread per band; here one rank per device):
 * the frame's only collective -- an RCCL gather (ncclGather) of the ranks' tile buffers to rank 0 over xGMI --
 * issued by this library on the frame's stream.
 * ctgpu_comm_unique_id: rank 0 makes the id (ncclGetUniqueId) and hands its CTGPU_COMM_ID_BYTES bytes to the
 *   other ranks by whatever channel the host application has (bench.py: a torch.distributed broadcast).
 * ctgpu_comm_init: every rank, collectively (ncclCommInitRank on the context's device).
 * ctgpu_comm_share: another context of this process on the same device (a second frame in flight) uses `owner`'s
 *   communicator instead of one of its own; RCCL then runs all the process's gathers in the order they were issued.
 * ctgpu_gather_tiles: n uint32 from every rank, back to back into d_all on `root` (d_all ignored elsewhere).
 * ctgpu_render_sharded: this rank's share of one frame -- trace my tiles, gather to rank 0, assemble there into
 *   d_fb (rank 0's device framebuffer, width*height uint32) -- all queued on `stream`, nothing synchronises. */
#define CTGPU_COMM_ID_BYTES 128
int  ctgpu_comm_unique_id(void *id);
int  ctgpu_comm_init(ctgpu_ctx *ctx, const void *id, int rank, int world);
int  ctgpu_comm_share(ctgpu_ctx *ctx, ctgpu_ctx *owner);
int  ctgpu_comm_rank(ctgpu_ctx *ctx, int *rank, int *world);
int  ctgpu_comm_destroy(ctgpu_ctx *ctx);
/* Contexts that share a communicator must issue their sharded frames in the same order on every rank (RCCL matches the
 * gathers of a communicator by issue order).  ctgpu_render_sharded checks it: every frame's gather carries {the context's
 * place among the sharers, the frames it has issued}, the root compares all ranks' pairs with its own, and a mismatch makes
 * the next ctgpu_read_counters fail with CTGPU_ERR_COMM.  ctgpu_comm_debug_skew (test hook): stamp `delta` frames off. */
int  ctgpu_comm_debug_skew(ctgpu_ctx *ctx, int delta);
int  ctgpu_gather_tiles(ctgpu_ctx *ctx, const void *d_tiles, uint64_t n, void *d_all, int root, void *stream);
int  ctgpu_render_sharded(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, int tile_rows, void *d_fb, void *stream);

/* ctgpu_download_frame: the part of the device framebuffer d_fb the reference ever writes (rendered rows x the
 *   centred square, clipped to the bitmap) -> the same pixels of the caller's host framebuffer, asynchronous on
 *   `stream`.  host_fb must be page-locked (hipHostMalloc / hipHostRegister): CTGPU_ERR_ARG otherwise.  What
 *   ctgpu_render does after assembling. */
int  ctgpu_download_frame(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, const void *d_fb, uint32_t *host_fb, void *stream);

/* Ray counters accumulated by render calls since the last read (resets them). Synchronises. */
int  ctgpu_read_counters(ctgpu_ctx *ctx, ctgpu_stats *stats);

int  ctgpu_trace_rays(ctgpu_ctx *ctx, const ctgpu_ray *rays, uint64_t n, ctgpu_hit *hits);

/* The same per-ray hook through the kernel that traces the frames (the shared "packet" walk,
 * ct_walk_packet): 64 consecutive rays of the list form one packet.  flags = 0: closest hit,
 * results as ctgpu_trace_rays gives them (IntersectBVHClosest, bvh.cpp:198-222, under
 * ClosestIntersection, raythread.cpp:197-228).  CTGPU_TRACE_ANY_HIT: the shadow query of
 * ComputeLighting (raythread.cpp:303-307) -- every ray starts at t = 1e30f whatever `t` says and
 * leaves at its first ray.t update; only `found` is comparable with a full traversal. */
#define CTGPU_TRACE_ANY_HIT 1u
int  ctgpu_trace_rays_packet(ctgpu_ctx *ctx, const ctgpu_ray *rays, uint64_t n, uint32_t flags, ctgpu_hit *hits);

/* Instrumented render of the same frame (slower; no framebuffer): fills node_visits and
 * triangle_tests besides the ray counts.  These are the V_r / Q_r sums of the roofline model. */
int  ctgpu_count_work(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, ctgpu_stats *stats);

/* BuildBVH on the device (bvh.cpp:16-121): the node and index arrays the reference's recursion produces for these
 * triangles -- same boxes (the a<b?a:b folds included), same split planes (binary32 midpoints), same index
 * permutation (the swap loop of bvh.cpp:82-93, in closed form), same node numbering (children allocated when their
 * parent is subdivided, depth first).  vertices: [n][9] doubles, p1 p2 p3 per triangle in scene-triangle order;
 * nodes_out: room for 2n - 1 nodes; indexes_out: n entries.  *device_ms (may be NULL): the build itself, copies
 * excluded.  The arrays go straight into ctgpu_scene_desc.nodes / .indexes. */
int  ctgpu_build_bvh(int device, const double *vertices, uint32_t n, ctgpu_node *nodes_out, uint32_t *indexes_out,
                     uint32_t *n_nodes_out, double *device_ms);

/* ---- measurement hooks (bench.py's roofline object; no effect on pixels) ----
 * ctgpu_read_walk_work: what the two traversal launches (the shared "packet" walk of the fast path) asked the
 *   memory system for since the last read -- packets walked, QuadNodes (128 B) and triangle records (80 B)
 *   fetched; [0] primary walk, [1] shadow walk.  Counted by the production kernels themselves.  Synchronises, resets.
 * ctgpu_kernel_timing / ctgpu_read_kernel_times: HIP events on the frame's own stream around every kernel of the
 *   fast path, for every frame rendered while enabled; read = synchronise on them, sum, reset. */
#define CTGPU_PIPELINE_KERNELS 5       /* ct_sample_grid, primary walk, ct_light_gate, shadow walk, ct_shade */
typedef struct ctgpu_walk_work {
    uint64_t packets[2], quad_nodes[2], triangle_records[2];
    uint64_t shadow_rays_blocked, shadow_packets_all_blocked;      /* shadow walk: rays that found a blocker; packets whose every ray did */
    uint64_t rays_handed_over[2];      /* rays of packets that had stopped sharing their walk, traced by the per-lane kernel instead */
} ctgpu_walk_work;
typedef struct ctgpu_kernel_times { uint64_t launches[CTGPU_PIPELINE_KERNELS]; double total_ms[CTGPU_PIPELINE_KERNELS]; } ctgpu_kernel_times;
/* ctgpu_render_sharded's phases on this rank, HIP events on the frame's stream while ctgpu_kernel_timing is on:
 * trace (this rank's tiles), gather (ncclGather: on the root it ends when the slowest rank's tiles have arrived),
 * assemble (root only). */
typedef struct ctgpu_phase_times { uint64_t frames; double trace_ms, gather_ms, assemble_ms; } ctgpu_phase_times;
int  ctgpu_read_phase_times(ctgpu_ctx *ctx, ctgpu_phase_times *out);
int  ctgpu_read_walk_work(ctgpu_ctx *ctx, ctgpu_walk_work *out);
/* Which walk traces this context's frames of a kind (supersampled or not) at the moment -- the shared walk or the
 * per-lane one -- and the kernel times last measured on either.  Without CTGPU_PACKET in the environment the context
 * chooses by measurement: a scene whose packets share little (more than CTGPU_PROBE_ABOVE fetches per packet, default
 * 100) gets one frame on the per-lane walk, and whichever was faster is kept (same pixels either way); every 256
 * frames the other walk gets one frame again. */
int  ctgpu_walk_choice(ctgpu_ctx *ctx, int supersampled, int *per_lane, double *packet_ms, double *lane_ms);
/* The same for frames whose materials reflect: TraceRay's recursion runs either as bounce levels on the wavefront
 * pipeline (level 0 through the frame walks, levels 1..depth through the packet walk's ray-list kernels) or in the
 * general path's one kernel; without CTGPU_BOUNCE in the environment the context renders two frames of a kind on each
 * and keeps the faster (same pixels), trying the other again every 256 frames. */
int  ctgpu_bounce_choice(ctgpu_ctx *ctx, int supersampled, int *general_path, double *pipeline_ms, double *general_ms);
/* Heavy chunks first.  The launch of a packet walk ends when its slowest packet does, so each frame notes what the
 * packets of every chunk of ray ids cost (QuadNode + triangle-record fetches), and the NEXT frame of the same geometry
 * hands the chunks that were heavy out first -- whole, or, when the frame has few packets per wave (single-sample
 * frames), cut into 4 or 16 parts that different waves walk.  Same pixels and ray counts whatever the order (who walks
 * with whom never changes a ray's result); a frame of another size, partition or scene starts from "nothing is heavy".
 * Only frames that have the device to themselves are ordered (ctgpu_render, _begin / _poll, _multi): frames a caller
 * pipelines on its own streams (ctgpu_render_partition, _sharded) fill each other's tails and are left as they are
 * (CTGPU_ORDER=2: order also frames with many packets per wave, whole chunks; =3: also pipelined frames -- both
 * measured, neither pays: profiles/r03_order_sweep.log).
 * entries[walk * 2 + class]: list entries the last frame built (walk 0 primary / 1 shadow, class 0 the heaviest);
 * *split: whether that frame cut its heavy chunks into parts.  Knobs: CTGPU_ORDER=0 off; CTGPU_ORDER_HEAVY="a,b,floor"
 * class 0 / 1 from a / b times the mean packet's fetches (default 6, 2.5) and never below `floor` fetches (64);
 * CTGPU_ORDER_SPLIT_BELOW chunks per wave of the grid below which heavy chunks are cut up (8). */
int  ctgpu_order_info(ctgpu_ctx *ctx, uint32_t entries[8], int *split);
int  ctgpu_kernel_timing(ctgpu_ctx *ctx, int enable);
int  ctgpu_read_kernel_times(ctgpu_ctx *ctx, ctgpu_kernel_times *out);

/* The jitter stream CTGPU_RNG_LIBC uses: glibc 2.35 rand() after srand(seed), skipping the
 * first `offset` values (host-side; exported so a test can compare it with libc itself). */
int  ctgpu_rand_stream(uint32_t seed, uint64_t offset, uint64_t n, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
