// ct_params.h -- kernel argument block and launcher prototypes shared by ct_kernels.hip
// (device) and ctgpu_api.hip (host side of the C ABI).
#ifndef CT_PARAMS_H
#define CT_PARAMS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ctgpu.h"
#include "ct_device.h"

namespace ctd {

// words of FrameParams::counters: 0 primary rays, 1 shadow rays, 2 reflected rays, 3 node visits, 4 triangle
// tests, 5..14 diagnostic builds, 15 rays the packet walk walked again unfolded, 16 shadow rays not needed,
// 17 / 18 primary / shadow rays the packet walk handed over to the per-lane walk; 19..23 triangle-test stages (CT_TRI_STATS);
// 24..47 histograms of the CT_WALK_STATS build, eight bins of eight lanes each: node visits, triangle tests and packets by
// the number of rays taking part (tools/lane_stats.py)
// 48 shadow rays ct_walk_shadow_tree left to the second shadow launch
constexpr int kCounterWords = 50;
constexpr int kCounterLeft = 48;
constexpr int kCounterPrimaryLeft = 49;
// ray-queue head words: walk 0 primary rays, walks 1..L the shadow rays of the scene's L shadow-casting lights, walks
// L+1 / L+2 the handed-over primary / shadow rays, walks L+3..2L+2 the same lights' shadow rays once more (the second
// shadow launch of a frame whose shadow rays walk the shadow tree: the rays that tree does not take, ct_walk_shadow_tree);
// 8 shards each, one 128-byte line per head word; then 32 words for the two hand-over counters (everything cleared
// together, per frame)
// walk 2L+3 the primary rays once more (the launch behind ct_walk_closest_tree: the rays it left);
__host__ __device__ constexpr uint32_t queue_walks(uint32_t n_shadow_lights) { return 2u * n_shadow_lights + 4u; }
__host__ __device__ constexpr uint32_t queue_second_shadow_walk(uint32_t n_shadow_lights) { return n_shadow_lights + 3u; }
__host__ __device__ constexpr uint32_t queue_second_primary_walk(uint32_t n_shadow_lights) { return 2u * n_shadow_lights + 3u; }
__host__ __device__ constexpr size_t queue_words(uint32_t n_shadow_lights) { return (size_t)queue_walks(n_shadow_lights) * 8u * 16u; }
// ... and behind them: the two hand-over counters (32 words), then the head words of the heavy-chunk lists (ORDERED
// walks: class 0 / 1 of the primary walk, class 0 / 1 of the shadow walk, 16 words apart).  One block, cleared per frame.
__host__ __device__ constexpr size_t queue_order_heads(uint32_t n_shadow_lights) { return queue_words(n_shadow_lights) + 32u; }
// ... and one more line: word 0 counts the shadow rays the shadow tree's walk left to the second shadow launch, word 1 the
// primary rays ct_walk_closest_tree left to the second primary launch
__host__ __device__ constexpr size_t queue_left_word(uint32_t n_shadow_lights) { return queue_words(n_shadow_lights) + 32u + 64u; }
__host__ __device__ constexpr size_t queue_block_words(uint32_t n_shadow_lights) { return queue_words(n_shadow_lights) + 32u + 64u + 16u; }
constexpr int kCounterElided = 16;

struct FrameParams {
    // scene (device pointers)
    const WideNode  *nodes;
    const QuadNode  *quads;       // the tree folded to four-wide nodes, binary32 planes (packet walk)
    const double    *quad_boxes;  // [quad][entry][6] the entries' binary64 boxes (min xyz, max xyz): levels 1 and 2
    uint32_t quad_root;           // root reference of the folded tree (a leaf reference for a one-leaf tree)
    const QuadNode  *quads_flat;  // the same tree with nothing folded (two entries per node): second pass of
    const uint32_t  *quad_node_of_flat;  // the packet walk, for rays folding is not proven for; [quad][entry] -> reference node,
    const ctgpu_node *ref_nodes;         // whose box (bmin, bmax: the node's first six doubles) is that entry's
    uint32_t quad_root_flat;
    const TriRecord *tris;
    const Material  *materials;
    const Light     *lights;
    uint32_t n_lights;
    // spheres (CTGPU_F_SPHERES, general path only): the scene's sphere objects, the tree over their bounding cubes in
    // the C ABI's own node format, leaf slot -> sphere; n_spheres = 0 when the frame does not ask for them
    const ctgpu_sphere *spheres;
    const ctgpu_node   *sphere_nodes;
    const uint32_t     *sphere_indexes;
    uint32_t n_spheres;
    uint32_t reflect_fix;         // CTGPU_F_REFLECT_FIX: reflected rays start at ray.t = 1e30f, not 0
    uint32_t root_ref;            // WideNode index, or kLeafBit|0 when the whole tree is one leaf
    uint32_t slot_of_triangle0;   // TriRecord slot holding scene triangle 0
    uint32_t stack_depth;         // LDS stack entries per lane
    double   root_min[3], root_max[3];
    float    plane_bound[3];      // >= |plane| of every box in the tree AND of every triangle box, per axis (+inf: no filter)
    uint32_t cull_first;          // QuadNodes from this index on are cull nodes (ct_device.h; kNoNode: the scene has none)
    CullLimits cull;              // which rays may skip their entries (d_lim = 0: none), from the scene's magnitudes and the eye
    // the shadow-only tree (ct_device.h "shadow tree", DESIGN appendix E; stree == NULL: the frame's shadow rays walk `quads`)
    const QuadNode *stree;        // its nodes, root = node 0
    const TriRecord *stree_tris;  // the TriRecords in its own leaf order (a leaf entry: kLeafBit | first slot, `remaining` as in tris[])
    const uint32_t *leaf_of_slot; // slot of stree_tris -> the reference node (a leaf of ref_nodes[]) that holds that triangle
    ShadowTreeLimits stree_lim;   // which shadow rays may walk it
    uint32_t closest_tree;        // 1: the frame's PRIMARY rays walk it too (ct_walk_closest_tree, DESIGN appendix F)
    float    closest_tau;         // ... pruning entries beyond ray.t only while ray.t >= this (appendix D.2's tau, doubled)

    // frame
    int32_t  width, height, half; // bitmap size; half = height/2
    uint32_t square;              // 2*half: columns (and at most rows) of the rendered square
    uint32_t rendered_rows;       // T * (height/T)
    uint32_t rows_per_band;       // height/T
    uint32_t subsample;
    float    inv_height;          // 1.f/(float)height
    int32_t  max_depth;
    double   cam[3], rot[9];
    int32_t  rng_mode;
    uint32_t traced_rows;         // LIBC: number of traced rows (row ordinals)
    uint64_t rng_seed;
    const int32_t *rand_table;    // LIBC: rand() values in single-worker call order
    const int32_t *row_ordinal;   // LIBC: traced-row ordinal per frame row

    // partition
    uint32_t tile_rows, part_index, part_count;
    uint32_t local_rows;          // rows of one partition's compact buffer (equal for all partitions)

    // divergence policy: run phase A/B/C once this many lanes wait for it
    int32_t thresh_a, thresh_b, thresh_c;

    // outputs
    uint32_t *out;                         // [local_rows][square]
    unsigned long long *counters;          // [kCounterWords] primary, shadow, reflect, visits, tests, diagnostics ...
    uint32_t *overflow;                    // set to 1 on traversal-stack overflow
};

// Heavy chunks first (ct_walk_packet<.., ORDERED>).  A frame's launch ends when its slowest packet does, and a packet is
// one dependent chain of fetches: the frame before tells which chunks of ray ids (W.chunk / W.chunk_shadow consecutive
// ids: the rays of one packet, or two) were heavy, and those are handed out FIRST -- and, in frames with few packets
// per wave slot, cut into parts that different waves walk.  Who walks with whom never changes a ray's result, so the
// pixels do not depend on any of this.  One set per walk (0 primary, 1 shadow).
constexpr int kOrderClasses = 4;
struct OrderParams {
    uint32_t *cost_out;               // [chunks] fetches (QuadNodes + triangle records) this frame's packets of a chunk make
    const uint8_t *listed;            // [chunks] 1: the chunk is on a heavy list, the run over all chunks skips it
    const uint4 *list;                // [kOrderClasses][cap] the heavy lists, class 0 (heaviest) first: {first id within
                                      //          the light, ids, light (shadow walk), chunk}; x = ~0: an entry nobody wrote
    const uint32_t *count;            // [kOrderClasses] entries per class (may run past cap: the reader clamps)
    unsigned long long *totals_out;   // [64 slots][2] {fetches, packets} of this frame, summed by the next frame's ct_order_build
    uint32_t cap;
    uint32_t chunk_shift;             // log2 of the walk's chunk (ORDERED needs a power of two)
    uint32_t chunks_per_light;
};

// Buffers and knobs of the wavefront pipeline (ct_pipeline.hip); all pointers are device memory.
struct PipelineParams {
    float2   *coords;            // [n_samples] jittered canvas position (supersampled frames)
    uint2    *hits;              // [n_samples] {ray.t bits, closest TriRecord slot | kNoNode | invalid}
    uint8_t  *shadow;            // [n_shadow_lights][n_samples] 1 = that light is blocked
    unsigned long long *queue;   // [queue_walks(n_shadow_lights)][8 shards][16] head words (128-byte stride) of the ray queues
    void     *spill;             // [walk_blocks][spill_depth][64] stack entries beyond the LDS part
    const uint32_t *shadow_lights;   // [n_shadow_lights] index into lights[]
    uint64_t n_samples;
    uint32_t n_shadow_lights;
    uint32_t lds_depth, spill_depth, walk_blocks;
    uint32_t listed_blocks;      // grid of the per-lane launches that trace the handed-over rays
    uint32_t stree_blocks;       // grid of ct_walk_shadow_tree (0: walk_blocks)
    uint32_t chunk;              // ray ids a wave reserves per queue atomic (multiple of 64)
    uint32_t chunk_shadow;       // the same for the shadow walk (its candidates are only partly rays)
    uint32_t packet;             // 1: supersampled frames use ct_walk_packet (shared walk per wave)
    uint32_t packet_lds_levels;  // stack levels whose per-lane tmin lives in LDS (deeper ones in HBM)
    uint32_t packet_depth;       // most entries the folded tree can have pending at once
    int32_t  thresh_r, thresh_a, thresh_b;
    uint32_t keep_every_shadow_ray;   // 1: ct_light_gate passes every hit sample's shadow rays (CTGPU_KEEP_SHADOW_RAYS)
    unsigned long long *walk_work;    // [64 slots][8]: {packets, QuadNodes fetched, triangle records fetched} of the primary
                                      // walk in words 0..2, of the shadow walk in words 3..5 (packet walk only); word 6: shadow
                                      // rays that found a blocker, word 7: shadow packets whose every ray did
    // Hand-over of incoherent packets (ct_walk_packet -> ct_walk): a packet that has made `packet_budget` visits with
    // fewer than `packet_lanes` rays per visit on average is not sharing its walk any more -- it stops, its rays go on
    // the list, and the per-lane kernel traces them from the start after the packet launch.  0 = never hand over.
    uint32_t packet_budget, packet_lanes;
    unsigned long long *handed;  // [0] primary, [16] shadow: number of listed ray ids (128-byte apart); never reset in a launch
    unsigned long long *handed_ids[2];   // the ids: primary ray = sample id; shadow ray = light * n_samples + sample
    uint64_t handed_room;        // ids each list can hold (a full list: the packet simply walks on)
    const ctgpu_ray *ray_list;   // ctgpu_trace_rays_packet: the caller's rays (n_samples of them) ...
    ctgpu_hit *ray_hits;         // ... and where their results go
    // ray lists the frame makes for itself (reflection bounces, ct_pipeline.hip "bounce levels"):
    const uint8_t *ray_live;     // which list entries are rays at all (NULL: every one)
    uint32_t list_slots;         // 1: ray_hits[].triangle receives the TriRecord SLOT (kNoNode: none recorded), not the scene triangle
    uint32_t list_counter;       // frame counter word the walk adds its ray count to (0: none; 1 shadow rays, 2 reflected rays)
    const uint32_t *list_gate;   // when set and *list_gate == 0 the list holds no ray: the walk's waves leave at once
    float    shadow_split_gap;   // shadow packets: a window's hit samples go out as two groups when their primary hits' depths fall into
                                 // two ranges at least this far apart (relative to the middle depth); < 0: never
    uint32_t shadow_live;        // the shadow byte that marks a ray the shadow walk has to trace: 0 (ct_light_gate), or kShadowLeft in the
                                 // launch that traces what ct_walk_shadow_tree left (it then also clears the byte of an unblocked ray)
    unsigned long long *shadow_left;   // [0] rays ct_walk_shadow_tree left to that launch (it returns at once on 0); [1] the same for
                                       // the primary rays ct_walk_closest_tree left
    uint32_t primary_left;       // 1: this launch of the primary walk traces the samples marked kPrimaryLeft in hits[], nothing else
    uint32_t ordered;            // 1: the frame's packet walks are the ORDERED instantiations, order[] is set
    OrderParams order[2];
};

// ct_order_build's arguments for one walk: last frame's costs in, this frame's heavy lists out
struct OrderBuild {
    const uint32_t *prev_cost;                // [chunks]
    const unsigned long long *prev_totals;    // [64][2]
    uint8_t  *listed;                         // [chunks]
    uint4    *list;                           // [kOrderClasses][cap]
    uint32_t *count;                          // [kOrderClasses], zero on entry (the build of the frame before, or a memset, left it so)
    uint32_t *next_count;                     // [kOrderClasses] the next frame's, zeroed here
    uint32_t *cost_out;                       // [chunks] this frame's cost words, zeroed here
    unsigned long long *totals_out;           // [64][2] this frame's totals, zeroed here
    uint32_t chunks, cap, chunk_shift, chunks_per_light;
    uint32_t parts[kOrderClasses];            // parts a chunk of class k is cut into (1: handed out whole, but early)
    float    heavy[kOrderClasses];            // a chunk is of class k from this many times the mean packet's fetches (descending)
    uint32_t floor;                           // ... and never below this many fetches
    uint64_t n_samples;
};

// Reflection bounces on the wavefront pipeline: level 0 is the camera ray's hit, level k the k-th reflected ray's
// (TraceRay's recursion, raythread.cpp:353-386, unrolled into passes over the frame's samples).  All arrays are
// indexed by sample id; `shadow_*` by light * n_samples + sample id like PipelineParams::shadow.
struct BounceParams {
    ctgpu_ray *rays;             // the current level's reflected ray of every sample that has one
    uint8_t   *live;             // ... which samples those are
    ctgpu_hit *hits;             // ... and what the closest-hit walk found (triangle = TriRecord slot | sphere tag | kNoNode)
    ctgpu_ray *shadow_rays;      // the current level's shadow rays,
    uint8_t   *shadow_live;      // which of them exist (the light can contribute at that point),
    ctgpu_hit *shadow_hits;      // and whether something blocks them (found)
    uint32_t  *local;            // [level][sample] colour computed at a level that went on reflecting ...
    float     *refl;             // ... and its material's reflection: mixed with the deeper result on the way back
    uint32_t  *terminal;         // [sample] colour of the deepest call (its local colour, or the background)
    uint8_t   *pending;          // [sample] levels whose mix is still owed (0 .. max_depth)
    uint32_t  *level_live;       // [max_depth + 2] samples with a reflected ray entering each level: an empty level's passes leave at once
    uint32_t  level;             // the level being processed (1 .. max_depth)
    uint32_t  max_depth;         // TraceRay's recursionDepth at the camera ray (10)
    float     bounce_t;          // ray.t a reflected ray starts with: 0 (raythread.cpp:373) or 1e30f (CTGPU_F_REFLECT_FIX)
};

uint64_t pipeline_sample_count(const FrameParams &P, bool ss);
int  pipeline_walk_occupancy(size_t lds_bytes);
int  pipeline_packet_occupancy(uint32_t depth, uint32_t lds_levels);
int  pipeline_stree_occupancy();
constexpr int kPipelineKernels = 5;      // ct_sample_grid, primary walk, ct_light_gate, shadow walk, ct_shade
void launch_pipeline(const FrameParams &P, const PipelineParams &W, bool ss, hipStream_t stream, hipEvent_t *marks,
                     const BounceParams *bounce = nullptr, const OrderBuild *order = nullptr);
void launch_packet_ray_list(const FrameParams &P, const PipelineParams &W, bool any_hit, hipStream_t stream);

void launch_trace_tiles(const FrameParams &P, uint32_t n_tiles, bool ss, bool reflect, bool count, hipStream_t stream);
void launch_trace_ray_list(const FrameParams &P, const ctgpu_ray *rays, uint64_t n, ctgpu_hit *hits, hipStream_t stream);
void launch_assemble(const FrameParams &P, const uint32_t *all_tiles, uint32_t *fb, hipStream_t stream);

}  // namespace ctd
#endif
