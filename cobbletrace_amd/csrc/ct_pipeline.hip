// ct_pipeline.hip -- wavefront pipeline of lean gfx950 kernels (the fast path).
//
// The megakernel in ct_kernels.hip keeps a whole pixel's life in one lane; its shading
// code (pow, sqrt, HSV) drives the register count past 200 VGPRs, i.e. two waves per
// SIMD, and the traversal -- which is issue-bound -- starves.  Here the frame is a short
// chain of kernels over flat per-sample arrays in HBM (a few hundred MB at 1080p x 16 spp,
// a rounding error against 288 GB and 8 TB/s):
//
//   ct_sample_grid           lane = pixel   jittered sample positions (the float residue of
//                                           raythread.cpp:462-504 is sequential per pixel)
//   ct_walk_packet<PRIMARY>  lane = ray     closest-hit traversal, the wave's 64 rays sharing ONE
//                                           walk over the four-wide folded tree   -> hits[sample]
//   ct_walk_packet<SHADOW>   lane = ray     any-hit traversal per non-ambient light of every hit
//                                           sample                                -> shadow bytes
//   ct_shade                 lane = sample, then pixel   lighting + HSV, ordered supersample blend
//
// The walk kernels are PERSISTENT: one wave per workgroup, as many workgroups as fit on the chip,
// ray ids handed out in chunks from sharded queue heads (one shard per XCD).  In ct_walk_packet
// everything that does not differ between the rays of a wave is scalar: the node comes through the
// scalar cache into SGPRs, the stack holds one {entry, lane mask} per level, and since a CU has
// four vector units but ONE scalar unit, that scalar control flow is what had to be made lean
// (DESIGN.md section 4).  ct_walk<KIND,SS> below is the earlier per-lane design (every lane its own
// stack and cursor over the binary tree); it stays as CTGPU_PACKET=0, as the fallback for trees
// whose folded form could leave more than 64 entries pending, and as an independent
// implementation the tests compare against.
//
// Since round 3 every frame can take this path: any number of lights, spheres as passes of their own, reflective
// materials as bounce levels over the ray-list kinds of the packet walk (a context measures those against the
// megakernel and keeps the faster).  Round 4: in scenes whose triangles sit in many-triangle leaves the tree
// supersampled frames walk carries CULL NODES in place of those leaves (ct_device.h, DESIGN appendix D;
// ct_walk_packet<.., CULL>), and shadow packets are formed from windows of 128 ray ids listed in LDS.
// Semantics and citations: ct_device.h, ct_kernels.hip header.
#include <type_traits>
#include "ct_params.h"
#include "ct_raygen.h"
#include "ct_spheres.h"

#include <algorithm>

using namespace ctd;

namespace {

constexpr int kWave = 64;
#ifndef CT_PACKET_WAVES
#define CT_PACKET_WAVES 7    // seven waves per SIMD (72 VGPRs): dragon frame 3.24 -> 3.12 ms in bench.py (A/B, three runs each);
                             // bunny +1 %, single-sample frames +2 %; 5: +5 %, 8: +2 % (6 was round 1's choice)
#endif
#ifndef CT_WALK_WAVES
#define CT_WALK_WAVES 5      // waves per SIMD the traversal kernel is register-budgeted for
#endif
constexpr uint32_t kShards = 8;                     // ray-queue head words (one per XCD)
constexpr uint32_t kInvalidSample = 0xfffffffeu;   // hits[].slot of a sample that is not part of the frame
// hits[].slot of a sample whose closest object is a SPHERE (CTGPU_F_SPHERES): bit 30 and the sphere's index (TriRecord
// slots stay below 2^30, kNoNode / kInvalidSample have bit 31)
constexpr uint32_t kSphereTag = 0x40000000u;
__device__ __forceinline__ bool is_sphere_hit(uint32_t slot) { return (slot & 0xC0000000u) == kSphereTag; }

struct StackEntry { uint32_t ref; float tmin; };

// ------------------------------------------------------------------ sample id <-> pixel
// Supersampled frames: 16 consecutive ids are the 16 samples of a pixel, 4 consecutive
// pixels are a 2x2 block (a wave = a 2x2-pixel footprint: rays as coherent as they get).
// Single-sample frames: 64 consecutive ids are an 8x8 tile.
struct PixelOfSample { uint32_t xi, local_row, sample; };

// n / d and n % d for n < 2^24 and 1 <= d < 2^24 with one float multiply instead of the ~35
// instructions of a 32-bit integer divide: the float quotient is off by at most one.
__device__ __forceinline__ void small_divmod(uint32_t n, uint32_t d, uint32_t &q, uint32_t &r) {
    q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d));
    int32_t rem = (int32_t)(n - q * d);
    if (rem < 0) { q -= 1u; rem += (int32_t)d; }
    if (rem >= (int32_t)d) { q += 1u; rem -= (int32_t)d; }
    r = (uint32_t)rem;
}

template <bool SS>
__device__ __forceinline__ PixelOfSample pixel_of(const FrameParams &P, uint64_t id) {
    PixelOfSample r;
    uint32_t col, row;
    if (SS) {
        r.sample = (uint32_t)(id & 15u);
        const uint32_t p = (uint32_t)(id >> 4);
        const uint32_t within = p & 3u, blk = p >> 2;
        small_divmod(blk, (P.square + 1u) >> 1, row, col);        // frames up to 2^24 2x2 blocks
        r.xi = col * 2u + (within & 1u);
        r.local_row = row * 2u + (within >> 1);
    } else {
        r.sample = 0;
        // (inside the tile in Z order: 4 consecutive ids are 2x2 pixels, 16 are 4x4 -- the parts a heavy tile is cut into)
        const uint32_t within = (uint32_t)(id & 63u), tile = (uint32_t)(id >> 6);
        small_divmod(tile, (P.square + 7u) >> 3, row, col);
        r.xi = col * 8u + ((within & 1u) | ((within >> 1) & 2u) | ((within >> 2) & 4u));
        r.local_row = row * 8u + (((within >> 1) & 1u) | ((within >> 2) & 2u) | ((within >> 3) & 4u));
    }
    return r;
}

// frame row of a partition-local row (interleaved row tiles), and whether the pixel exists
__device__ __forceinline__ bool pixel_in_frame(const FrameParams &P, uint32_t xi, uint32_t local_row, uint32_t &yi) {
    uint32_t local_tile, row_in_tile;
    small_divmod(local_row, P.tile_rows, local_tile, row_in_tile);
    yi = (local_tile * P.part_count + P.part_index) * P.tile_rows + row_in_tile;
    if (xi >= P.square || local_row >= P.local_rows || yi >= P.rendered_rows) return false;
    if (P.subsample) {
        // rows a band actually traces (raythread.cpp:526-535): even offsets and the band's last row;
        // the rows between are averages made by ct_assemble
        const uint32_t off = yi % P.rows_per_band;
        if ((off & 1u) != 0u && off != P.rows_per_band - 1u) return false;
    }
    // off-bitmap pixels are computed and dropped by the reference (draw2d.h:11-14): not traced here --
    // except that with subsampling the dropped row still feeds the average stored one row above it
    const int row = P.height / 2 - ((int)yi - P.half), col = (int)xi - P.half + P.width / 2;
    return col >= 0 && col < P.width && (P.subsample || (row >= 0 && row < P.height));
}

}  // namespace

// ------------------------------------------------------------------------------------
// ct_sample_grid: jittered sample positions of every pixel (supersampled frames only).
// One lane = one pixel, walking the 4x4 stratified grid exactly as raythread.cpp:462-504
// does, including the float add/subtract residue that carries from sample to sample.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ct_sample_grid(const FrameParams P, const PipelineParams W) {
    // the 16 positions of a pixel are a sequential chain, so a lane owns a pixel; they leave through
    // LDS so that the block stores its 4096 positions as whole cache lines, not 8 bytes per line
    __shared__ float stage_x[256 * 17], stage_y[256 * 17];              // row stride 17: conflict-free both ways
    const uint64_t pixel_slot = (uint64_t)blockIdx.x * 256u + threadIdx.x;      // id >> 4
    bool live = pixel_slot * 16u < W.n_samples;
    PixelOfSample px = PixelOfSample{};
    uint32_t yi = 0;
    if (live) {
        px = pixel_of<true>(P, pixel_slot * 16u);
        live = pixel_in_frame(P, px.xi, px.local_row, yi);
    }
    if (live) {
        const float fx = (float)((int)px.xi - P.half), fy = (float)((int)yi - P.half);
        const float jitter = 0.25f / 8.0f;
        float run_x = fx;
        for (uint32_t xs = 0; xs < 4; xs++) {
            float run_y = fy;
            for (uint32_t ys = 0; ys < 4; ys++) {
                const uint32_t s = xs * 4u + ys;
                float jx = (float)jitter_int(P, px.xi, yi, s, 0) / 2147483648.0f;
                float jy = (float)jitter_int(P, px.xi, yi, s, 1) / 2147483648.0f;
                jx = (jx * (2.0f * jitter)) - jitter;
                jy = (jy * (2.0f * jitter)) - jitter;
                run_x += jx;
                run_y += jy;
                stage_x[threadIdx.x * 17u + s] = run_x;
                stage_y[threadIdx.x * 17u + s] = run_y;
                run_y -= jy;
                run_y += 0.25f;
                run_x -= jx;
            }
            run_x += 0.25f;
        }
    }
    __syncthreads();
    const uint64_t block_first = (uint64_t)blockIdx.x * 4096u;
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t local = j * 256u + threadIdx.x;                  // sample within the block
        const uint64_t id = block_first + local;
        if (id < W.n_samples) W.coords[id] = make_float2(stage_x[(local >> 4) * 17u + (local & 15u)], stage_y[(local >> 4) * 17u + (local & 15u)]);
    }
}

namespace {

// position of a sample on the canvas: from the grid buffer (supersampled) or the pixel centre
template <bool SS>
__device__ __forceinline__ bool sample_position(const FrameParams &P, const PipelineParams &W, uint64_t id, float &px,
                                                float &py) {
    const PixelOfSample s = pixel_of<SS>(P, id);
    uint32_t yi;
    if (!pixel_in_frame(P, s.xi, s.local_row, yi)) return false;
    if (SS) {
        const float2 c = W.coords[id];
        px = c.x; py = c.y;
    } else {
        px = (float)((int)s.xi - P.half);
        py = (float)((int)yi - P.half);
    }
    return true;
}

// ------------------------------------------------------------------ traversal state of one lane
struct Walker {
    d3 o, d, inv;
    float t;              // ray.t
    uint32_t slot;        // closest record (kNoNode = none); *tclosest == t whenever slot is set
    uint32_t cur;         // node / triangle slot to process next, kNoNode = traversal finished
    uint32_t sp;          // stack entries in use
    uint32_t leaf_left;   // triangles left in the current leaf (0 = cur is an inner node)
    bool fast;            // division-free slab test allowed for this direction
};

// Per-lane stack of pending right children: {child reference, approximate tmin} plus the
// WideNode that holds the child's box (needed only when the popped tmin is too close to
// ray.t to be trusted, see pop_next).  Entry e of lane l: [e][l]; the first lds_depth
// entries in LDS, deeper ones in this wave's spill area in HBM.
// (LDS and HBM parts are addressed through address-space-qualified pointers and separate
// branches: selecting between two generic pointers makes the compiler emit flat_load / flat_store
// with a double wait on every push and pop.)
typedef StackEntry __attribute__((address_space(3))) LdsEntry;
typedef uint32_t __attribute__((address_space(3))) LdsWord;
typedef StackEntry __attribute__((address_space(1))) HbmEntry;
typedef uint32_t __attribute__((address_space(1))) HbmWord;

struct WalkStack {
    LdsEntry *lds;            // [lds_depth][64]
    LdsWord *lds_parent;      // [lds_depth][64]
    HbmEntry *spill;          // [spill_depth][64]
    HbmWord *spill_parent;    // [spill_depth][64]
    uint32_t lds_depth, spill_depth, lane;
    uint32_t *overflow;

    __device__ __forceinline__ void push(Walker &w, uint32_t ref, float tmin, uint32_t parent) const {
        if (w.sp < lds_depth) {
            const uint32_t at = w.sp * kWave + lane;
            lds[at].ref = ref;
            lds[at].tmin = tmin;
            lds_parent[at] = parent;
        } else if (w.sp - lds_depth < spill_depth) {
            const uint32_t at = (w.sp - lds_depth) * kWave + lane;
            spill[at].ref = ref;
            spill[at].tmin = tmin;
            spill_parent[at] = parent;
        } else {
            atomicOr(overflow, 1u);
            return;
        }
        w.sp += 1;
    }
    // entry sp-1: child reference, approximate tmin
    __device__ __forceinline__ void top(const Walker &w, uint32_t &ref, float &tmin) const {
        const uint32_t i = w.sp - 1u;
        if (i < lds_depth) {
            const uint32_t at = i * kWave + lane;
            ref = lds[at].ref;
            tmin = lds[at].tmin;
        } else {
            const uint32_t at = (i - lds_depth) * kWave + lane;
            ref = spill[at].ref;
            tmin = spill[at].tmin;
        }
    }
    __device__ __forceinline__ uint32_t top_parent(const Walker &w) const {
        const uint32_t i = w.sp - 1u;
        if (i < lds_depth) return lds_parent[i * kWave + lane];
        return spill_parent[(i - lds_depth) * kWave + lane];
    }
};

// leaf_left doubles as the lane's mode: 0 = cur is an inner node; 1 .. N = triangles left in the
// current leaf; the values below are markers
constexpr uint32_t kLeafUnknown = 0xfffffff0u;        // leaf entered, count comes with its first record
constexpr uint32_t kExactPopPending = 0xfffffffdu;    // cur = parent whose RIGHT child must be re-tested exactly
constexpr uint32_t kExactPending = 0xfffffffeu;       // redo this inner node with real divisions

__device__ __forceinline__ void enter(Walker &w, uint32_t ref) {
    w.cur = ref;
    w.leaf_left = 0;
    if (ref != kNoNode && (ref & kLeafBit)) {
        w.cur = ref & ~kLeafBit;
        w.leaf_left = kLeafUnknown;     // taken from the first record
    }
}

// Pop until an entry survives `tmin < ray.t`, the deferred half of the right child's box test
// (bvh.cpp:205 when the reference enters node left+1).  The stored tmin is approximate (one
// float step); if it is within one step of ray.t the verdict is not certain and the child
// is parked for an exact re-test (phase E).
__device__ __forceinline__ void pop_next(Walker &w, const WalkStack &st) {
    const int32_t bt = __float_as_int(w.t);        // ray.t > 0: raw bits order like the floats do
    while (w.sp > 0) {
        uint32_t ref;
        float tmin;
        st.top(w, ref, tmin);
        if (ordinals_near(__float_as_int(tmin), bt, 2u)) {
            w.cur = st.top_parent(w);
            w.leaf_left = kExactPopPending;
            w.sp -= 1;
            return;
        }
        w.sp -= 1;
        if (tmin < w.t) { enter(w, ref); return; }
    }
    w.cur = kNoNode;
    w.leaf_left = 0;
}

// (A) open an inner node: both children's boxes from one 128-byte line, division-free.
// A lane that cannot prove all six verdicts parks the node for phase (E) instead of dividing
// here: keeping the rare path out of this block keeps its registers out, too.
__device__ __forceinline__ void step_inner(Walker &w, const WideNode *__restrict__ nodes, const WalkStack &st) {
    const WideNode *n = nodes + w.cur;
    const double4 a0 = *reinterpret_cast<const double4 *>(&n->lmin[0]);   // lmin xyz, lmax x
    const double4 a1 = *reinterpret_cast<const double4 *>(&n->lmax[1]);   // lmax yz, rmin xy
    const double4 a2 = *reinterpret_cast<const double4 *>(&n->rmin[2]);   // rmin z, rmax xyz
    const uint2 refs = *reinterpret_cast<const uint2 *>(&n->lref);
    const double lmin[3] = {a0.x, a0.y, a0.z}, lmax[3] = {a0.w, a1.x, a1.y};
    const double rmin[3] = {a1.z, a1.w, a2.x}, rmax[3] = {a2.y, a2.z, a2.w};
    float ltmin, ltmax, rtmin, rtmax;
    slab_approx(w.o, w.inv, lmin, lmax, ltmin, ltmax);
    slab_approx(w.o, w.inv, rmin, rmax, rtmin, rtmax);
    bool lpass, rpass;
    const bool sure = box_verdict_positive_t(ltmin, ltmax, w.t, lpass) & box_verdict_positive_t(rtmin, rtmax, w.t, rpass);
    if (!(sure & w.fast)) { w.leaf_left = kExactPending; return; }
    const uint32_t parent = w.cur;
    if (lpass) {
        // ray.t only shrinks while the left subtree is walked: a right child failing now fails
        // later too; survivors are re-checked against the then-current ray.t when popped
        if (rpass) st.push(w, refs.y, rtmin, parent);
        enter(w, refs.x);
    } else if (rpass) {
        enter(w, refs.y);
    } else {
        pop_next(w, st);
    }
}

// (E) the rare exact paths, with the reference's divisions (bvh.cpp:165-179)
__device__ __forceinline__ void step_exact(Walker &w, const WideNode *__restrict__ nodes, const WalkStack &st) {
    const WideNode *n = nodes + w.cur;
    const uint2 refs = *reinterpret_cast<const uint2 *>(&n->lref);
    float rtmin, rtmax;
    slab(w.o, w.d, n->rmin, n->rmax, rtmin, rtmax);
    const bool rpass = rtmax >= rtmin && rtmin < w.t && rtmax > 0;
    if (w.leaf_left == kExactPopPending) {
        // a popped right child: the whole test with the current ray.t, as the reference does it
        if (rpass) enter(w, refs.y);
        else pop_next(w, st);
        return;
    }
    float ltmin, ltmax;
    slab(w.o, w.d, n->lmin, n->lmax, ltmin, ltmax);
    const bool lpass = ltmax >= ltmin && ltmin < w.t && ltmax > 0;
    const uint32_t parent = w.cur;
    if (lpass) {
        if (rpass) st.push(w, refs.y, rtmin, parent);     // exact tmin is fine as an "approximate" one
        enter(w, refs.x);
    } else if (rpass) {
        enter(w, refs.y);
    } else {
        pop_next(w, st);
    }
}

// Two Newton steps: whatever the seed's accuracy (the ISA only promises a single-precision-class
// estimate), the result is within a few ulp_D of 1/d -- the bound Appendix A of DESIGN.md needs is
// merely "far below 2^-24".
__device__ __forceinline__ double quick_reciprocal(double d) {
    const double seed = __builtin_amdgcn_rcp(d);
    double x = seed;
    x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);     // x + x * (1 - d * x)
    x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
    return d == 0.0 ? seed : x;                              // 1 / +-0 = +-inf (the Newton step would make it a NaN)
}

__device__ __forceinline__ void begin_ray(Walker &w, const FrameParams &P) {
    w.fast = direction_allows_fast_slab(w.d);
    // The approximate slab test only needs |q - Q| far below the spacing of floats (2^29 ulp_D), so
    // the reciprocal need not be correctly rounded: v_rcp_f64 plus two Newton steps (relative error
    // about 2^-50) replaces three 35-instruction divides per ray.  Exact paths divide by d itself.
    w.inv = make_d3(quick_reciprocal(w.d.x), quick_reciprocal(w.d.y), quick_reciprocal(w.d.z));
    // root box (bvh.cpp:205 at rootNodeIdx)
    float tmin, tmax;
    bool pass;
    slab_approx(w.o, w.inv, P.root_min, P.root_max, tmin, tmax);
    if (!(box_verdict_positive_t(tmin, tmax, w.t, pass) & w.fast)) {
        slab(w.o, w.d, P.root_min, P.root_max, tmin, tmax);
        pass = tmax >= tmin && tmin < w.t && tmax > 0;
    }
    w.sp = 0;
    w.slot = kNoNode;
    enter(w, pass ? P.root_ref : kNoNode);
}

// (B) one triangle of the current leaf
__device__ __forceinline__ void step_leaf(Walker &w, const TriRecord *__restrict__ tris, const WalkStack &st,
                                          bool any_hit) {
    const TriRecord *rec = tris + w.cur;
    const double2 b0 = *reinterpret_cast<const double2 *>(&rec->p1[0]);
    const double2 b1 = *reinterpret_cast<const double2 *>(&rec->p1[2]);
    const double2 b2 = *reinterpret_cast<const double2 *>(&rec->e1[1]);
    const double2 b3 = *reinterpret_cast<const double2 *>(&rec->e2[0]);
    const double2 b4 = *reinterpret_cast<const double2 *>(&rec->e2[2]);
    const d3 p1 = make_d3(b0.x, b0.y, b1.x), e1 = make_d3(b1.y, b2.x, b2.y), e2 = make_d3(b3.x, b3.y, b4.x);
    if (w.leaf_left == kLeafUnknown) w.leaf_left = (uint32_t)(__double_as_longlong(b4.y) >> 32);

    const float before = w.t;
    const bool hit = triangle_test(w.o, w.d, w.t, p1, e1, e2);
    // bvh.cpp:212-215 with *tclosest == (slot set ? ray.t before this test : 2^32)
    const float tclosest = (w.slot != kNoNode) ? before : kFarAway;
    if (hit && w.t != kNoHit && w.t < tclosest) w.slot = w.cur;
    w.leaf_left -= 1;
    w.cur += 1;
    if (any_hit && w.t != kNoHit) {          // shadow ray decided (raythread.cpp:303-307)
        w.cur = kNoNode; w.leaf_left = 0; w.sp = 0;
        return;
    }
    if (w.leaf_left == 0) pop_next(w, st);
}

}  // namespace

// ------------------------------------------------------------------------------------
// ct_walk: persistent traversal kernel.
//   KIND 0  primary rays: ray index = sample id; result hits[id] = {ray.t, slot}
//   KIND 1  shadow rays of all non-ambient lights: ray index = light * n_samples + sample id, but
//           only samples whose primary ray found something get a ray; result: shadow byte of
//           (light, sample) set when the light is blocked
// Ray indices are handed out in chunks of W.chunk consecutive ids: one global atomic per chunk
// and wave (a single head word saturates near 90 dequeues/us on this chip, so per-ray or
// per-refill atomics would serialise the whole kernel), then the wave serves its lanes from
// the chunk with ballots only.
// ------------------------------------------------------------------------------------

// LISTED: the rays are the ones ct_walk_packet handed over (W.handed_ids[KIND], ids as above in any order): the
// queue then runs over list positions, and a shadow ray's light comes from its id.
template <int KIND, bool SS, bool LISTED = false>
__global__ __launch_bounds__(kWave, CT_WALK_WAVES) void ct_walk(const FrameParams P, const PipelineParams W) {
    const uint32_t q_chunk = LISTED ? 64u : ((KIND == 1) ? W.chunk_shadow : W.chunk);      // ray ids per queue atomic
    extern __shared__ __align__(16) unsigned char lds_raw[];
    WalkStack st;
    st.lds = (LdsEntry *)lds_raw;
    st.lds_parent = (LdsWord *)(lds_raw + (size_t)W.lds_depth * kWave * sizeof(StackEntry));
    st.lane = threadIdx.x;
    st.lds_depth = W.lds_depth;
    st.spill_depth = W.spill_depth;
    // this wave's spill area: spill_depth x 64 entries of 8 bytes, then as many parent words
    unsigned char *spill_base = reinterpret_cast<unsigned char *>(W.spill) + (size_t)blockIdx.x * W.spill_depth * kWave * 12u;
    st.spill = (HbmEntry *)spill_base;
    st.spill_parent = (HbmWord *)(spill_base + (size_t)W.spill_depth * kWave * sizeof(StackEntry));
    st.overflow = P.overflow;
    const uint32_t lane = threadIdx.x;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    // Ray ids: the primary walk's are the sample ids; the shadow walk's are light * n_samples + sample
    // (which is also the index of the ray's result byte).  The id space is cut into shards -- kShards
    // per light -- each with its own head word on its own 128-byte line (one head saturates near 90
    // dequeues/us chip-wide); a wave starts on the shard of its workgroup id (workgroups b and b+8
    // share an XCD under round-robin placement: speed only) and moves to the next when one runs dry.
    // (LISTED: "samples" below are list positions, one run of kShards shards; the queue's head words are the last two walks', ct_params.h)
    uint64_t listed = 0;
    if (LISTED) {
        listed = W.handed[KIND == 0 ? 0 : 16];
        if (listed > W.handed_room) listed = W.handed_room;
        if (listed == 0) return;
    }
    const unsigned long long *listed_ids = W.handed_ids[KIND == 0 ? 0 : 1];
    const uint64_t n_samples = LISTED ? listed : W.n_samples;
    const uint32_t n_shards = (KIND == 0 || LISTED) ? kShards : kShards * W.n_shadow_lights;
    unsigned long long *heads = W.queue + (size_t)(LISTED ? (W.n_shadow_lights + (KIND == 0 ? 1u : 2u)) * kShards : (KIND == 0 ? 0u : kShards)) * 16u;
    const uint64_t shard_span = ((n_samples + kShards - 1) / kShards + q_chunk - 1) / q_chunk * q_chunk;
    uint32_t shard = blockIdx.x % n_shards, shards_done = 0;
    uint64_t light_base = 0;                 // first ray id of the light the current chunk belongs to
    Light shadow_light = Light{};
    const d3 eye = make_d3(P.cam[0], P.cam[1], P.cam[2]);

    Walker w;
    w.o = eye; w.d = make_d3(0, 0, 1); w.inv = make_d3(0, 0, 1);
    w.t = kNoHit; w.slot = kNoNode; w.cur = kNoNode; w.sp = 0; w.leaf_left = 0; w.fast = false;
    uint64_t ray_id = 0;
    bool has_ray = false;
    // wave-uniform chunk cursor
    uint64_t chunk_next = 0, chunk_end = 0;
    bool wave_drained = false;
    uint32_t n_traced = 0;
#ifdef CT_WALK_STATS
    unsigned long long stat_cycles[3] = {0, 0, 0}, stat_exec[3] = {0, 0, 0}, stat_lanes[3] = {0, 0, 0};
    const unsigned long long stat_begin = __builtin_amdgcn_s_memtime();
#endif

    for (;;) {
        const bool walking = has_ray && w.cur != kNoNode;
        const bool finished = has_ray && w.cur == kNoNode;
        const bool want_e = walking && w.leaf_left >= kExactPopPending;
        const bool want_b = walking && w.leaf_left != 0 && !want_e;
        const bool want_a = walking && w.leaf_left == 0;
        const unsigned long long m_a = __ballot(want_a), m_b = __ballot(want_b);
        const int n_a = __popcll(m_a), n_b = __popcll(m_b);
        const int n_r = wave_drained ? __popcll(__ballot(finished)) : __popcll(__ballot(!walking));
#ifdef CT_WALK_STATS
        const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();
#define CT_STAMP(slot) do { stat_cycles[slot - 11] += __builtin_amdgcn_s_memtime() - stamp0; } while (0)
#else
#define CT_STAMP(slot) do {} while (0)
#endif
        if (__ballot(want_e) != 0ull) {       // rare (about 1e-7 per node): settle it at once
            if (want_e) step_exact(w, P.nodes, st);
            continue;
        }
        if (n_r == 0 && n_a == 0 && n_b == 0) break;
#ifdef CT_WALK_STATS
        {   // divergence statistics (diagnostic build only): executions and lanes served per phase
            const bool do_r_ = n_r >= W.thresh_r || (n_a == 0 && n_b == 0);
            const bool do_a_ = !do_r_ && (((n_a >= W.thresh_a) || (n_b < W.thresh_b && n_a >= n_b) || n_b == 0) && n_a > 0);
            const int which = do_r_ ? 2 : (do_a_ ? 0 : 1);
            stat_exec[which] += 1;
            stat_lanes[which] += (unsigned long long)(which == 2 ? n_r : (which == 0 ? n_a : n_b));
        }
#endif

        // ---------------- (R) retire finished rays, hand out new ones
        if (n_r >= W.thresh_r || (n_a == 0 && n_b == 0)) {
            if (finished) {
                if (KIND == 0) W.hits[ray_id] = make_uint2(__float_as_uint(w.t), w.slot);
                else if (w.t != kNoHit) W.shadow[ray_id] = 1;         // [light][sample] == ray id
                has_ray = false;
            }
            for (;;) {
                const unsigned long long m_i = __ballot(!has_ray);
                if (m_i == 0ull || wave_drained) break;
                if (chunk_next >= chunk_end) {
                    // (reserving one chunk ahead to hide the atomic's round trip was tried: the chunk a
                    // busy wave sits on at the end of the launch costs more than the latency it hides)
                    unsigned long long offset = 0;
                    if (lane == 0) offset = atomicAdd(heads + (size_t)shard * 16u, (unsigned long long)q_chunk);
                    offset = __shfl(offset, 0);
                    const uint32_t light = shard / kShards, part = shard % kShards;
                    const uint64_t lo = (uint64_t)part * shard_span + offset;       // sample id within the light
                    if (offset >= shard_span || lo >= n_samples) {                   // this shard is dry: next one
                        shards_done += 1;
                        shard = (shard + 1u) % n_shards;
                        if (shards_done >= n_shards) { wave_drained = true; break; }
                        continue;
                    }
                    uint64_t hi = lo + q_chunk;
                    if (hi > ((uint64_t)part + 1u) * shard_span) hi = ((uint64_t)part + 1u) * shard_span;
                    if (hi > n_samples) hi = n_samples;
                    light_base = LISTED ? 0u : (uint64_t)light * n_samples;
                    chunk_next = light_base + lo;
                    chunk_end = light_base + hi;
                    if (KIND == 1 && !LISTED) shadow_light = P.lights[W.shadow_lights[light]];
                }
                const uint64_t avail = chunk_end - chunk_next;
                const uint32_t rank = (uint32_t)__popcll(m_i & lanes_below);
                const bool take = !has_ray && rank < avail;
                uint64_t id = chunk_next + rank;
                const uint64_t wanted = (uint64_t)__popcll(m_i);
                chunk_next += (wanted < avail) ? wanted : avail;
                if (take) {
                    bool listed_ok = true;
                    if (LISTED) {
                        id = listed_ids[id];
                        // (a slot some packet reserved but could not use carries ~0; anything else out of range cannot
                        //  be there, and is not followed if it is)
                        listed_ok = id < W.n_samples * (KIND == 1 ? (uint64_t)W.n_shadow_lights : 1u);
                        if (!listed_ok) id = 0;
                        light_base = (KIND == 1) ? (id / W.n_samples) * W.n_samples : 0u;
                        if (KIND == 1) shadow_light = P.lights[W.shadow_lights[id / W.n_samples]];
                    }
                    const uint64_t sample = id - light_base;
                    float px, py;
                    bool live = sample_position<SS>(P, W, sample, px, py) && listed_ok;
                    if (KIND == 0) {
                        if (live) {
                            w.o = eye;
                            w.d = primary_direction(P, px, py);
                        } else if (!LISTED) {                             // (a listed ray's slot was marked when its packet was formed)
                            W.hits[id] = make_uint2(__float_as_uint(kNoHit), kInvalidSample);
                        }
                    } else if (live) {
                        live = W.shadow[id] == 0;                        // ct_light_gate: a hit sample whose light can contribute
                        if (live) {
                            const uint2 h = W.hits[sample];
                            // P = origin + tclosest * dir (raythread.cpp:360); tclosest = ray.t when a
                            // triangle was recorded, else 2^32
                            const float tclosest = (h.y != kNoNode) ? __uint_as_float(h.x) : kFarAway;
                            w.o = eye + scale((double)tclosest, primary_direction(P, px, py));
                            w.d = (shadow_light.type == CTGPU_LIGHT_POINT) ? load_d3(shadow_light.position) - w.o
                                                                           : load_d3(shadow_light.direction);
                        }
                    }
                    if (live) {
                        w.t = kNoHit;
                        begin_ray(w, P);
                        ray_id = id;
                        has_ray = true;
                        n_traced += 1;
                    }
                }
            }
            CT_STAMP(13);
            continue;
        }
        // ---------------- (A) / (B): the fuller queue goes first unless the other reached its threshold
        const bool do_a = (n_a >= W.thresh_a) || (n_b < W.thresh_b && n_a >= n_b) || n_b == 0;
        if (do_a && n_a > 0) {
            // tight loop: keep opening nodes while enough lanes want to, re-evaluating only that
            bool more = want_a;
            int n_more;
            do {
                if (more) step_inner(w, P.nodes, st);
                more = has_ray && w.cur != kNoNode && w.leaf_left == 0;
                n_more = __popcll(__ballot(more));
            } while (n_more >= W.thresh_a);
            CT_STAMP(11);
        } else {
            bool more = want_b;
            int n_more;
            do {
                if (more) step_leaf(w, P.tris, st, KIND == 1);
                more = has_ray && w.cur != kNoNode && w.leaf_left != 0 && w.leaf_left < kExactPopPending;
                n_more = __popcll(__ballot(more));
            } while (n_more >= W.thresh_b);
            CT_STAMP(12);
        }
    }

#ifdef CT_WALK_STATS
    if (lane == 0) {
        for (int k = 0; k < 3; k++) {
            atomicAdd(&P.counters[5 + k * 2], stat_exec[k]);
            atomicAdd(&P.counters[6 + k * 2], stat_lanes[k]);
            atomicAdd(&P.counters[11 + k], stat_cycles[k]);
        }
        atomicAdd(&P.counters[14], __builtin_amdgcn_s_memtime() - stat_begin);
    }
#endif
    // ray counters: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) n_traced += __shfl_down(n_traced, off);
    if (lane == 0) atomicAdd(&P.counters[KIND == 0 ? 0 : 1], (unsigned long long)n_traced);
}

// ------------------------------------------------------------------------------------
// ct_walk_packet: the 64 rays of a wave walk the tree TOGETHER (supersampled frames).
// The reference visits nodes in one fixed order for every ray (left subtree, then right), so a
// group of rays can share one walk over the UNION of their nodes: a lane takes part in a node
// exactly when the reference would have entered it for that lane's ray (its own box verdicts, its
// own ray.t), and sits masked out otherwise.  For the 4 pixels x 16 samples a wave holds the rays
// are nearly parallel, so the union is barely larger than one ray's set -- and everything that was
// per-lane becomes per-wave: the node is fetched once with scalar loads (no 7 KB of L1 traffic per
// step), the stack holds one {node, lane mask} entry per level instead of 64, and pushing / popping
// is scalar work.  Per lane only the deferred tmin of a pushed right child is kept (LDS).
// Semantics, verdict margins and the exact fallback are those of ct_walk.
// ------------------------------------------------------------------------------------
// One WideNode through the scalar cache into SGPRs (the address is wave-uniform).  hipcc will not
// pick s_load here on its own -- the kernel also stores to global memory, so it cannot prove the
// node array unclobbered -- and vector loads of a uniform address cost 26 VGPRs per lane.
typedef int sgpr16_t __attribute__((ext_vector_type(16)));
struct ScalarNode {
    sgpr16_t a, b;            // bytes 0..63, 64..127 of the node
    __device__ __forceinline__ double f64(int i) const {      // i-th double of the node (0..11)
        const int lo = (i < 8) ? a[2 * i] : b[2 * (i - 8)];
        const int hi = (i < 8) ? a[2 * i + 1] : b[2 * (i - 8) + 1];
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ uint32_t lref() const { return (uint32_t)b[8]; }
    __device__ __forceinline__ uint32_t rref() const { return (uint32_t)b[9]; }
};
__device__ __forceinline__ ScalarNode scalar_load_node(const WideNode *n) {
    ScalarNode r;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r.a), "=&s"(r.b)
                 : "s"(n)
                 : "memory");
    return r;
}

struct ScalarQuad {
    sgpr16_t a, b;            // bytes 0..63, 64..127 of the QuadNode
    __device__ __forceinline__ f32x2 planes(int entry, int axis) const {       // {min, max} of entry's box on axis
        const int i = entry * 3 + axis;
        f32x2 r;
        r.x = __int_as_float((i < 8) ? a[2 * i] : b[2 * (i - 8)]);
        r.y = __int_as_float((i < 8) ? a[2 * i + 1] : b[2 * (i - 8) + 1]);
        return r;
    }
    __device__ __forceinline__ uint32_t ref(int entry) const { return (uint32_t)b[8 + entry]; }
    __device__ __forceinline__ uint32_t count() const { return (uint32_t)b[12]; }
};
__device__ __forceinline__ ScalarQuad scalar_load_quad(const QuadNode *n) {
    ScalarQuad r;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r.a), "=&s"(r.b)
                 : "s"(n)
                 : "memory");
    return r;
}

typedef int sgpr4_t __attribute__((ext_vector_type(4)));
struct ScalarTri {
    sgpr16_t a;               // p1 xyz, e1 xyz, e2 xy
    sgpr4_t b;                // e2 z, triangle, remaining
    __device__ __forceinline__ uint32_t remaining() const { return (uint32_t)b[3]; }
    __device__ __forceinline__ double f64(int i) const {      // i-th double of the record (0..8)
        const int lo = (i < 8) ? a[2 * i] : b[0];
        const int hi = (i < 8) ? a[2 * i + 1] : b[1];
        return __hiloint2double(hi, lo);
    }
};
__device__ __forceinline__ ScalarTri scalar_load_tri(const TriRecord *t) {
    ScalarTri r;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r.a), "=&s"(r.b)
                 : "s"(t)
                 : "memory");
    return r;
}

// LDS / HBM part of the packet walk's stack (the wave-uniform {entry reference, lane mask} of each
// level sit in vector registers, see ct_walk_packet; at most 64 levels)
struct PacketStack {
    LdsWord *parent;       // [depth]      QuadNode * 4 + entry the box belongs to (exact re-test at pop)
    float __attribute__((address_space(3))) *tmin;        // [lds_levels][64] approximate tmin per lane
    float __attribute__((address_space(1))) *tmin_spill;  // [depth - lds_levels][64] deeper levels, in HBM
    uint32_t lds_levels;
    static __host__ __device__ size_t bytes(uint32_t depth, uint32_t lds_levels) {
        // (at least 512 bytes behind the parent words: the shadow walk lists a window's 128 ray offsets there)
        return (size_t)((depth + 3u) & ~3u) * 4u + (size_t)(lds_levels < 2u ? 2u : lds_levels) * 256u;
    }
    __device__ __forceinline__ void put_tmin(uint32_t level, uint32_t lane, float v) const {
        if (level < lds_levels) tmin[level * kWave + lane] = v;
        else tmin_spill[(level - lds_levels) * kWave + lane] = v;
    }
    __device__ __forceinline__ float get_tmin(uint32_t level, uint32_t lane) const {
        if (level < lds_levels) return tmin[level * kWave + lane];
        return tmin_spill[(level - lds_levels) * kWave + lane];
    }
};

#ifdef CT_PACKET_TRACE
// diagnostic build (tools/packet_trace.py): one record per walked packet -- who, when, how much
__device__ unsigned long long ct_packet_trace[5u << 20];
__device__ unsigned int ct_packet_trace_n;
extern "C" __attribute__((visibility("default"))) int ctgpu_debug_packet_trace(unsigned long long *out, unsigned int max_records, unsigned int *n) {
    unsigned int count = 0;
    if (hipMemcpyFromSymbol(&count, HIP_SYMBOL(ct_packet_trace_n), sizeof count) != hipSuccess) return -2;
    if (count > (1u << 20)) count = 1u << 20;
    if (count > max_records) count = max_records;
    if (count && hipMemcpyFromSymbol(out, HIP_SYMBOL(ct_packet_trace), (size_t)count * 5 * 8) != hipSuccess) return -2;
    const unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(ct_packet_trace_n), &zero, sizeof zero) != hipSuccess) return -2;
    *n = count;
    return 0;
}
#endif

// KIND 0  primary rays of the frame            (closest hit -> hits[])
// KIND 1  shadow rays of the frame             (any hit     -> shadow bytes)
// KIND 2  caller's ray list, closest hit       (-> ctgpu_hit; the per-ray test hook ctgpu_trace_rays_packet:
// KIND 3  caller's ray list, any hit               64 consecutive list entries are one packet, same walk code)
// HANDOVER: the instantiation that may give a packet's rays up to the per-lane kernel (W.packet_budget; an experiment
// that is off by default -- it is a template parameter because the bookkeeping alone, compiled in but switched off,
// cost the primary walk 12 % through register pressure: profiles/r03_handover_sweep.log).
// ORDERED: chunks the frame before found heavy come first, in parts where that frame's packets were few (ct_params.h,
// OrderParams), then every other chunk in the usual order; each packet adds its fetches to its chunk's cost word.
// CULL: the tree holds cull nodes (ct_device.h; a scene without them runs the instantiation that does not carry the code).
// The launch's arguments, read again (the frame's shadow walk).  A persistent kernel keeps its arguments in scalar
// registers for the whole launch, ~150 of them alive across the walk, and the compiler parks what does not fit in lanes of
// spare vector registers: 140 v_readlane per packet in the shadow ray's set-up alone, on the vector pipe.  The kernel
// argument segment sits in the scalar cache, so each phase of a packet takes a fresh copy of what IT needs through s_load
// (the empty asm makes the segment's address a new value: nothing can be carried over from the copy before).  Measured
// (profiles/r04_args_again.log, both builds on one box, rocprofv3 counters): shadow walk 809.1 M -> 776.9 M vector
// instructions per launch, 22.6 M -> 28.0 M scalar loads, 1.543 -> 1.523 ms.  The primary walk has little set-up to
// save (-0.5 % vector instructions) and pays for the loads (+56 % scalar loads, +2 % time, +7 % single-sample): it keeps
// its arguments as they come.
#ifndef CT_ARGS_AGAIN
#define CT_ARGS_AGAIN 1
#endif
typedef const __attribute__((address_space(4))) unsigned char *KernargBytes;
template <typename T>
__device__ __forceinline__ void kernarg_again(T &out, uint32_t byte_offset) {
    KernargBytes base = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    // (word by word through the constant address space, so that every read is a scalar load whatever the optimiser
    //  makes of a copy; the words nobody reads are never loaded)
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "kernel argument structs are made of words");
    const __attribute__((address_space(4))) uint32_t *src = (const __attribute__((address_space(4))) uint32_t *)(base + byte_offset);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&out);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 4; i++) dst[i] = src[i];
}
// A pointer that arrives as two words is a generic pointer to the compiler (flat_load, and a wait on two counters per
// use); the pointers of a launch all point into HBM, and saying so gives the global loads back.
template <typename T>
__device__ __forceinline__ void in_hbm(T *&p) { p = (T *)(__attribute__((address_space(1))) T *)(unsigned long long)p; }
__device__ __forceinline__ void args_point_into_hbm(FrameParams &P, PipelineParams &W) {
    in_hbm(P.nodes); in_hbm(P.quads); in_hbm(P.quad_boxes); in_hbm(P.quads_flat); in_hbm(P.quad_node_of_flat); in_hbm(P.ref_nodes);
    in_hbm(P.tris); in_hbm(P.materials); in_hbm(P.lights); in_hbm(P.spheres); in_hbm(P.sphere_nodes); in_hbm(P.sphere_indexes);
    in_hbm(P.rand_table); in_hbm(P.row_ordinal); in_hbm(P.out); in_hbm(P.counters); in_hbm(P.overflow);
    in_hbm(W.coords); in_hbm(W.hits); in_hbm(W.shadow); in_hbm(W.queue); in_hbm(W.shadow_lights); in_hbm(W.walk_work);
    in_hbm(W.handed); in_hbm(W.handed_ids[0]); in_hbm(W.handed_ids[1]); in_hbm(W.ray_list); in_hbm(W.ray_hits); in_hbm(W.ray_live);
    in_hbm(W.list_gate); in_hbm(W.shadow_left); in_hbm(P.stree); in_hbm(P.stree_tris); in_hbm(P.leaf_of_slot);
    for (int k = 0; k < 2; k++) {
        in_hbm(W.order[k].cost_out); in_hbm(W.order[k].listed); in_hbm(W.order[k].list); in_hbm(W.order[k].count); in_hbm(W.order[k].totals_out);
    }
}
// (arguments lie in the segment in order, each at its natural alignment)
#ifndef CT_ARGS_SKEW                 // (diagnostic builds: a wrong offset, to see the check below report it)
#define CT_ARGS_SKEW 0
#endif
constexpr uint32_t kWalkArgsW = (uint32_t)((sizeof(FrameParams) + alignof(PipelineParams) - 1) / alignof(PipelineParams) * alignof(PipelineParams)) + CT_ARGS_SKEW;

template <int KIND, bool SS, bool HANDOVER = false, bool ORDERED = false, bool CULL = false>
__global__ __launch_bounds__(kWave, CT_PACKET_WAVES) void ct_walk_packet(const FrameParams P_launch, const PipelineParams W_launch) {
    FrameParams P = P_launch;
    PipelineParams W = W_launch;
    constexpr bool AGAIN = CT_ARGS_AGAIN != 0 && KIND == 1;
#ifndef CT_ARGS_POINTS
#define CT_ARGS_POINTS 15
#endif
#define ARGS_AGAIN_AT(bit) do { if (AGAIN && ((CT_ARGS_POINTS >> (bit)) & 1)) { kernarg_again(P, 0u); kernarg_again(W, kWalkArgsW); args_point_into_hbm(P, W); } } while (0)
#define ARGS_AGAIN() do { if (AGAIN) { kernarg_again(P, 0u); kernarg_again(W, kWalkArgsW); args_point_into_hbm(P, W); } } while (0)
    ARGS_AGAIN();
    // (the layout assumption, checked once per wave on first and last members: nothing is read through a copy that does not
    //  match, and the frame ends with an error -- word 2 in the flag the host looks at when the frame is done)
    if (AGAIN) if (P.nodes != P_launch.nodes || P.rng_seed != P_launch.rng_seed || W.coords != W_launch.coords || W.n_samples != W_launch.n_samples ||
        W.order[1].totals_out != W_launch.order[1].totals_out || P.counters != P_launch.counters) {
        if (threadIdx.x == 0) *P_launch.overflow = 2u;
        return;
    }
    const uint32_t q_chunk = (KIND == 1) ? W.chunk_shadow : W.chunk;      // ray ids per queue atomic
    constexpr bool LIST = KIND >= 2;                 // rays come from W.ray_list instead of the frame's samples
    if (LIST && W.list_gate && *W.list_gate == 0u) return;      // (a bounce level nobody reached)
    constexpr bool ANY_HIT = KIND == 1 || KIND == 3; // a ray leaves the walk at its first ray.t update
    constexpr bool OCTANTS = KIND < 2 && !HANDOVER;  // the frame walks: the shared walk once per octant (box_filter_f32_oct)
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const uint32_t depth = W.packet_depth;                      // most entries the folded tree leaves pending
    PacketStack st;
    st.parent = (LdsWord *)lds_raw;
    st.tmin = (float __attribute__((address_space(3))) *)(st.parent + ((depth + 3u) & ~3u));
    st.lds_levels = W.packet_lds_levels < depth ? W.packet_lds_levels : depth;
    st.tmin_spill = (float __attribute__((address_space(1))) *)(reinterpret_cast<unsigned char *>(W.spill) +
                                                                (size_t)blockIdx.x * depth * kWave * 12u);

    const uint32_t lane = threadIdx.x;
    // (a wave mask becomes a lane predicate with inverse_ballot: the mask goes straight into EXEC)
    const uint64_t n_samples = W.n_samples;
    const uint32_t n_shards = (KIND != 1) ? kShards : kShards * W.n_shadow_lights;
    // (the frame's shadow walk drains walks 1..L; the launch behind ct_walk_shadow_tree, which traces the rays that walk
    //  left -- W.shadow_live == kShadowLeft --, a second set of head words, and leaves at once when nothing was left)
    if (KIND == 1 && W.shadow_live != 0u && *W.shadow_left == 0ull) return;
    // (the same for the frame's primary rays: behind ct_walk_closest_tree this launch traces the samples that walk marked
    //  kPrimaryLeft in hits[] and no others)
    if (KIND == 0 && W.primary_left != 0u && W.shadow_left[1] == 0ull) return;
    unsigned long long *heads = W.queue + (size_t)(KIND == 1 ? (W.shadow_live != 0u ? queue_second_shadow_walk(W.n_shadow_lights) : 1u) * kShards
                                                  : (KIND == 0 && W.primary_left != 0u) ? queue_second_primary_walk(W.n_shadow_lights) * kShards : 0u) * 16u;
    const uint64_t shard_span = ((n_samples + kShards - 1) / kShards + q_chunk - 1) / q_chunk * q_chunk;
    uint32_t shard = blockIdx.x % n_shards, shards_done = 0;
    uint64_t light_base = 0, chunk_light_base = 0;       // first ray id of the packet's / the current chunk's light
    uint32_t light_index = 0, chunk_light_index = 0;     // lights[] entry of the shadow rays in the packet / chunk
    d3 eye = make_d3(P.cam[0], P.cam[1], P.cam[2]);

    // ray origin: the eye for every primary ray (kept in scalar registers), per lane for shadow rays
    d3 shadow_o = eye, d = make_d3(0, 0, 1);
#define RAY_O (KIND == 0 ? eye : shadow_o)
    RayFilter rf = RayFilter{};
    float t = kNoHit;
    uint32_t slot = kNoNode;
    bool has_ray = false, root_pass = false;
    bool redo = false;            // this lane's ray met a NaN box parameter: walked again alone, unfolded (below)
    uint64_t ray_id = 0;          // shadow packets: per lane; primary packets: packet_base + lane
    uint64_t packet_base = 0;
    uint64_t chunk_next = 0, chunk_end = 0;
    bool wave_drained = false;
    uint32_t n_traced = 0;
    // shadow walk: the window of up to 128 ray ids the next packets come from -- the live ones' offsets, listed in LDS (the
    // per-lane tmin levels' room, which an any-hit walk does not use), near cluster first (see the fetch below)
    uint32_t win_pos = 0, win_count = 0, win_split = 0;
    uint64_t win_base = 0;
    // what this wave asked the memory system for: packets walked, QuadNodes (128 B each) and triangle
    // records (80 B each) fetched -- the kernel's own bytes of the roofline (ctgpu_read_walk_work)
    uint32_t n_packets = 0, n_quads = 0, n_records = 0;
    uint32_t n_blocked = 0, n_all_blocked = 0;     // shadow walk: rays that found a blocker, packets whose every ray did
    uint32_t n_handed = 0;                         // rays given to the per-lane walk (their packets had stopped sharing)
    bool list_full = false;
    const uint32_t budget = (HANDOVER && !LIST) ? W.packet_budget : 0u;
    // ORDERED: which heavy class is being handed out (2: none left, the run over all chunks), the current chunk's and
    // the packet's cost word, and the wave's fetch count when the packet began
    uint32_t order_phase = ORDERED ? 0u : 2u, order_at = blockIdx.x, chunk_cost_at = 0, cost_at = 0, cost_mark = 0;
#ifdef CT_WALK_STATS
    unsigned long long pk_n[6] = {0, 0, 0, 0, 0, 0}, pk_cyc[3] = {0, 0, 0};
    const unsigned long long pk_begin = __builtin_amdgcn_s_memtime();
    // (one wave per workgroup: the histograms are this wave's own until the end)
    __shared__ unsigned int pk_hist[24];
    if (lane < 24) pk_hist[lane] = 0u;
#define PK_HIST(which, lanes, n) do { if (lane == 0 && (lanes) > 0) pk_hist[(which) * 8 + ((lanes) - 1) / 8] += (unsigned int)(n); } while (0)
#define PK_COUNT(i, v) do { pk_n[i] += (unsigned long long)(v); } while (0)
#define PK_T0() const unsigned long long pk_t0 = __builtin_amdgcn_s_memtime()
#define PK_T1(i) do { pk_cyc[i] += __builtin_amdgcn_s_memtime() - pk_t0; } while (0)
#else
#define PK_HIST(which, lanes, n) do {} while (0)
#define PK_COUNT(i, v) do {} while (0)
#define PK_T0() do {} while (0)
#define PK_T1(i) do {} while (0)
#endif

    for (;;) {
        // (the eye as a value the compiler has to take as new for every packet: what it otherwise computes once per
        //  launch from it -- the root box relative to the eye, in binary64, i.e. in vector registers -- it keeps in
        //  scratch and reloads per packet, 0.3 GB of traffic per launch; six additions per packet are cheaper)
        if (KIND == 0) asm volatile("" : "+s"(eye.x), "+s"(eye.y), "+s"(eye.z));
        ARGS_AGAIN_AT(0);
        // ---------------- fetch a packet: every lane gets a ray (or stays empty at the end)
        has_ray = false;
        root_pass = false;
        redo = false;
        float px = 0.0f, py = 0.0f;
        for (;;) {
            if (KIND == 1) {
                // Shadow packets come from a WINDOW of up to 128 consecutive ids (two 2x2-pixel blocks' samples under one
                // light) whose live ids -- the hit samples whose light can contribute -- were listed in LDS when the window
                // was opened: 64 list entries a packet.  Where the window's primary hits fall into two depth ranges that
                // are clearly apart (a silhouette: the samples of the near surface and those of what lies behind it), the
                // list holds the near ones first and a packet does not cross from one to the other: rays that start on
                // different surfaces share little of their way to the light, and walking them together only makes every
                // visit of either wider.  Who walks with whom never changes a ray's result.
                if (win_pos < win_count) {
                    uint32_t upto = win_count;
                    if (win_split > win_pos && win_split < upto) upto = win_split;
                    const uint32_t n = upto - win_pos < 64u ? upto - win_pos : 64u;
                    LdsWord *win_list = (LdsWord *)st.tmin;
                    if (lane < n) { ray_id = win_base + win_list[win_pos + lane]; has_ray = true; }
                    win_pos += n;
                    break;
                }
                if (wave_drained) break;
            }
            const unsigned long long m_i = __ballot(!has_ray);
            if (m_i == 0ull || wave_drained) break;
            if (chunk_next >= chunk_end) {
                bool from_list = false;
                if (ORDERED && order_phase < 2u) {
                    // the heavy lists, heaviest class first, as one sequence of which wave b takes entries b, b + grid, ...
                    // (no queue: every wave of the launch asking one head word for a ticket costs more than the lists save)
                    const OrderParams &O = W.order[KIND == 0 ? 0 : 1];
                    uint32_t at = order_at, cls = 0;
                    for (; cls < (uint32_t)kOrderClasses; cls++) {
                        uint32_t listed_entries = (uint32_t)__builtin_amdgcn_readfirstlane((int)O.count[cls]);
                        if (listed_entries > O.cap) listed_entries = O.cap;
                        if (at < listed_entries) break;
                        at -= listed_entries;
                    }
                    if (cls == (uint32_t)kOrderClasses) { order_phase = 2u; continue; }
                    order_at += gridDim.x;
                    const uint4 e = O.list[(size_t)cls * O.cap + at];
                    const uint32_t e_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x), e_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
                    const uint32_t e_light = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z);
                    if (e_lo == ~0u || e_lo >= n_samples || e_light >= (KIND == 1 ? W.n_shadow_lights : 1u)) continue;
                    uint64_t hi = (uint64_t)e_lo + e_n;
                    if (hi > n_samples) hi = n_samples;
                    chunk_light_base = (uint64_t)e_light * n_samples;
                    chunk_next = chunk_light_base + e_lo;
                    chunk_end = chunk_light_base + hi;
                    chunk_cost_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.w);
                    if (KIND == 1) chunk_light_index = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.shadow_lights[e_light]);
                    from_list = true;
                }
                if (!from_list) {
                unsigned long long offset = 0;
                if (lane == 0) offset = atomicAdd(heads + (size_t)shard * 16u, (unsigned long long)q_chunk);
                offset = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)offset) |
                         ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(offset >> 32)) << 32);
                const uint32_t light = shard / kShards, part = shard % kShards;
                const uint64_t lo = (uint64_t)part * shard_span + offset;
                if (offset >= shard_span || lo >= n_samples) {
                    shards_done += 1;
                    shard = (shard + 1u) % n_shards;
                    if (shards_done >= n_shards) wave_drained = true;
                    continue;
                }
                if (ORDERED) {
                    const OrderParams &O = W.order[KIND == 0 ? 0 : 1];
                    chunk_cost_at = light * O.chunks_per_light + (uint32_t)(lo >> O.chunk_shift);
                    if (__builtin_amdgcn_readfirstlane((int)O.listed[chunk_cost_at]) != 0) continue;      // went out with the heavy lists
                }
                uint64_t hi = lo + q_chunk;
                if (hi > ((uint64_t)part + 1u) * shard_span) hi = ((uint64_t)part + 1u) * shard_span;
                if (hi > n_samples) hi = n_samples;
                chunk_light_base = (uint64_t)light * n_samples;
                chunk_next = chunk_light_base + lo;
                chunk_end = chunk_light_base + hi;
                if (KIND == 1) chunk_light_index = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.shadow_lights[light]);
                }
                // (a shadow packet never mixes lights -- its rays should point the same way -- nor chunks: the next chunk
                //  this wave gets lies elsewhere in the image; topping up from it was measured, 5 % slower)
            }
            if (KIND == 1) {
                // open the next window of the chunk: which of its ids are rays, and how deep their primary hits lie
                const uint32_t n_ids = chunk_end - chunk_next < 128u ? (uint32_t)(chunk_end - chunk_next) : 128u;
                const uint64_t id0 = chunk_next + lane, id1 = chunk_next + 64u + lane;
                const bool live0 = lane < n_ids && W.shadow[id0] == W.shadow_live, live1 = 64u + lane < n_ids && W.shadow[id1] == W.shadow_live;
                bool far0 = false, far1 = false;
                unsigned long long m_live0 = __ballot(live0), m_live1 = __ballot(live1);
                if (W.shadow_split_gap >= 0.0f && (m_live0 | m_live1) != 0ull) {       // (an experiment that lost: off by default)
                    float t0 = 0.0f, t1 = 0.0f;
                    if (live0) { const uint2 h = W.hits[id0 - chunk_light_base]; t0 = (h.y != kNoNode) ? __uint_as_float(h.x) : kFarAway; }
                    if (live1) { const uint2 h = W.hits[id1 - chunk_light_base]; t1 = (h.y != kNoNode) ? __uint_as_float(h.x) : kFarAway; }
                    const float inf = __builtin_inff();
                    float lo = __builtin_fminf(live0 ? t0 : inf, live1 ? t1 : inf), hi = __builtin_fmaxf(live0 ? t0 : -inf, live1 ? t1 : -inf);
                    for (int off = 32; off > 0; off >>= 1) { lo = __builtin_fminf(lo, __shfl_xor(lo, off)); hi = __builtin_fmaxf(hi, __shfl_xor(hi, off)); }
                    const float mid = 0.5f * (lo + hi);
                    float near_hi = __builtin_fmaxf(live0 && t0 <= mid ? t0 : -inf, live1 && t1 <= mid ? t1 : -inf);
                    float far_lo = __builtin_fminf(live0 && t0 > mid ? t0 : inf, live1 && t1 > mid ? t1 : inf);
                    for (int off = 32; off > 0; off >>= 1) { near_hi = __builtin_fmaxf(near_hi, __shfl_xor(near_hi, off)); far_lo = __builtin_fminf(far_lo, __shfl_xor(far_lo, off)); }
                    if (far_lo - near_hi > W.shadow_split_gap * mid) { far0 = live0 && t0 > mid; far1 = live1 && t1 > mid; }
                }
                const unsigned long long m_far0 = __ballot(far0), m_far1 = __ballot(far1);
                const unsigned long long m_near0 = m_live0 & ~m_far0, m_near1 = m_live1 & ~m_far1;
                const uint32_t n_near0 = (uint32_t)__popcll(m_near0), n_near = n_near0 + (uint32_t)__popcll(m_near1);
                const uint32_t n_far0 = (uint32_t)__popcll(m_far0), n_far = n_far0 + (uint32_t)__popcll(m_far1);
                const auto rank_in = [&](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
                LdsWord *win_list = (LdsWord *)st.tmin;
                if (live0) win_list[far0 ? n_near + rank_in(m_far0) : rank_in(m_near0)] = lane;
                if (live1) win_list[far1 ? n_near + n_far0 + rank_in(m_far1) : n_near0 + rank_in(m_near1)] = 64u + lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                win_base = chunk_next;
                win_pos = 0; win_count = n_near + n_far; win_split = n_far != 0u ? n_near : 0u;
                chunk_next += n_ids;
                light_base = chunk_light_base;                // (the packets' light)
                light_index = chunk_light_index;
                if (ORDERED) cost_at = chunk_cost_at;
                continue;
            }
            const uint64_t avail = chunk_end - chunk_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_i >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_i, 0u));
            const bool take = !has_ray && rank < avail;
            const uint64_t id = chunk_next + rank;
            if (KIND == 0) packet_base = chunk_next;      // every lane is free here: rank == lane
            light_base = chunk_light_base;                // (the packet's light: it only ever takes from chunks of one)
            light_index = chunk_light_index;
            if (ORDERED) cost_at = chunk_cost_at;
            const uint64_t wanted = (uint64_t)__popcll(m_i);
            chunk_next += (wanted < avail) ? wanted : avail;
            // phase 1 (may run several times for a shadow packet): which ids become rays.  A shadow ray
            // exists where ct_light_gate left a 0 -- a hit sample whose light could contribute.
            if (take) {
                bool live = true;
                if (KIND == 0) {
                    live = sample_position<SS>(P, W, id, px, py);
                    if (W.primary_left != 0u) live = live && W.hits[id].y == kPrimaryLeft;
                    else if (!live) W.hits[id] = make_uint2(__float_as_uint(kNoHit), kInvalidSample);
                } else if (KIND == 1) {
                    live = W.shadow[id] == W.shadow_live;
                } else if (W.ray_live) {
                    live = W.ray_live[id] != 0;               // (the frame's own bounce lists: not every sample has a ray)
                }
                if (live) {
                    if (KIND != 0) ray_id = id;
                    has_ray = true;
                }
            }
            if (KIND != 1) break;          // primary / list packets: one aligned run of 64 ids, no top-up (measured for the
                                           // frame's own bounce lists, where many ids have no ray: 5-10 % slower with it)
        }
        n_traced += (uint32_t)__popcll(__ballot(has_ray));
        if (__ballot(has_ray) == 0ull) {
            if (wave_drained) break;
            continue;
        }
        n_packets += 1;
        { const int rays_in_packet = __popcll(__ballot(has_ray)); PK_HIST(2, rays_in_packet, 1); (void)rays_in_packet; }
        if (ORDERED) cost_mark = n_quads + n_records;
#ifdef CT_PACKET_TRACE
        const unsigned long long trace_begin = wall_clock64();
        const uint32_t trace_quads = n_quads, trace_records = n_records;
        const uint32_t trace_rays = (uint32_t)__popcll(__ballot(has_ray));
#endif
        // phase 2 (once per packet, all its lanes together): the rays themselves
        ARGS_AGAIN_AT(1);
        if (has_ray) {
            float t_start = kNoHit;
            if (LIST) {
                const ctgpu_ray *ray = W.ray_list + ray_id;
                shadow_o = load_d3(ray->origin);
                d = load_d3(ray->direction);
                if (!ANY_HIT) t_start = ray->t;            // (an any-hit ray starts at 1e30f, raythread.cpp:303)
            } else if (KIND == 0) {
                d = primary_direction(P, px, py);
            } else {
                // P = origin + tclosest * dir (raythread.cpp:360); tclosest = ray.t when a triangle was
                // recorded, else 2^32
                const uint64_t sample = ray_id - light_base;
                // (a shadow ray's sample was hit, so it is a pixel of the frame: with a grid buffer its position is just read)
                if (SS) { const float2 c = W.coords[sample]; px = c.x; py = c.y; }
                else (void)sample_position<SS>(P, W, sample, px, py);
                const uint2 h = W.hits[sample];
                const float tclosest = (h.y != kNoNode) ? __uint_as_float(h.x) : kFarAway;
                shadow_o = eye + scale((double)tclosest, primary_direction(P, px, py));
                const Light &shadow_light = P.lights[light_index];
                d = (shadow_light.type == CTGPU_LIGHT_POINT) ? load_d3(shadow_light.position) - shadow_o
                                                             : load_d3(shadow_light.direction);
            }
            t = t_start;
            slot = kNoNode;
            const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
            rf = ray_filter(RAY_O, d, inv, P.plane_bound, ANY_HIT ? 0x1p60f : 0x1p80f);
            // every frame ray starts at ray.t = 1e30f and only drops to hits > 0.0001f, which the
            // approximate levels rely on; a caller's ray with another start (the reference's
            // reflected rays start at 0, raythread.cpp:373) takes the exact level throughout
            const bool positive_t = !LIST || t > 0.0f;
            if (!positive_t) { rf.inv_z_e2.y = __builtin_inff(); rf.noi_z_e.y = __builtin_inff(); }
            // root box (bvh.cpp:205 at rootNodeIdx)
            float tmin, tmax;
            const bool root_nan = slab_approx_checked(RAY_O, inv, P.root_min, P.root_max, tmin, tmax);
            if (!(box_verdict_positive_t(tmin, tmax, t, root_pass) & direction_allows_checked_slab(d) & positive_t & !root_nan)) {
                slab(RAY_O, d, P.root_min, P.root_max, tmin, tmax);
                root_pass = tmax >= tmin && tmin < t && tmax > 0;
            }
        }

        // Which of the packet's rays may skip entries of cull nodes (ct_device.h "cull nodes", DESIGN appendix D): magnitudes
        // inside the frame's gate; a caller's closest-hit ray only when it starts at 1e30f like a frame's ray (with any
        // other start a "hit" that does not lower ray.t still records its triangle, bvh.cpp:212).  A closest-hit ray
        // skips only while its ray.t is at least tau (looked at where a skip is decided).
        ARGS_AGAIN_AT(2);
        unsigned long long cull_lanes = 0ull;
        if (CULL && P.cull.d_lim > 0.0f) {
            const double d_lim = (double)P.cull.d_lim, o_lim = (double)P.cull.o_lim;
            bool ok = has_ray && __builtin_fabs(d.x) <= d_lim && __builtin_fabs(d.y) <= d_lim && __builtin_fabs(d.z) <= d_lim;
            if (KIND != 0) ok = ok && __builtin_fabs(shadow_o.x) <= o_lim && __builtin_fabs(shadow_o.y) <= o_lim && __builtin_fabs(shadow_o.z) <= o_lim;
            if (KIND == 2) ok = ok && t == kNoHit;
            cull_lanes = __ballot(ok);
        }
        // lanes still tracing (a shadow ray stops at its first hit, raythread.cpp:303-307); every
        // lane mask below is wave-uniform and stays inside it
        // Pass 0 walks the folded tree.  A ray that meets a NaN box parameter there (0/0: a zero
        // direction component and the origin exactly on a box plane -- the one case folding is not
        // proven for, DESIGN B2) leaves the packet, and pass 1 walks those rays again from the root
        // over the same tree UNFOLDED (two entries per node, the reference's own structure).
        // the packet's octant: the signs of 1/d, when all its rays agree on every axis (else 8: the filter sorts the two
        // parameters of each axis itself, as it must for rays that point different ways)
        uint32_t octant = 8u;
        if (OCTANTS) {
            const unsigned long long rays = __ballot(has_ray);
            const unsigned long long down_x = __ballot(has_ray && __float_as_int(rf.inv_xy.x) < 0);
            const unsigned long long down_y = __ballot(has_ray && __float_as_int(rf.inv_xy.y) < 0);
            const unsigned long long down_z = __ballot(has_ray && __float_as_int(rf.inv_z_e2.x) < 0);
            if ((down_x == 0ull || down_x == rays) && (down_y == 0ull || down_y == rays) && (down_z == 0ull || down_z == rays))
                octant = (down_x ? 1u : 0u) | (down_y ? 2u : 0u) | (down_z ? 4u : 0u);
        }
        uint32_t visits = 0, lane_visits = 0;         // of this packet: nodes visited, rays that took part in them
        bool hand_over = false;
        for (int walk_pass = 0; walk_pass < 2; walk_pass++) {
        const QuadNode *quads = walk_pass == 0 ? P.quads : P.quads_flat;
        // an entry's binary64 box: beside the folded tree's QuadNodes (one dependent fetch on the undecided visits); the
        // unfolded tree's, walked by one ray in a million, through the entry's reference node
        const auto entry_box = [&](uint32_t index) -> const double * {
            if (walk_pass == 0) return P.quad_boxes + (size_t)index * 6u;
            return P.ref_nodes[P.quad_node_of_flat[index]].bmin;
        };
        if (walk_pass == 1) {
            const unsigned long long again = __ballot(redo);
            if (again == 0ull) break;
            if (lane == 0) atomicAdd(&P.counters[15], (unsigned long long)__popcll(again));    // diagnostic
            // (from the ray's own start value: a caller's / a bounce level's ray may start at 0, raythread.cpp:373)
            if (redo) { t = (LIST && !ANY_HIT) ? W.ray_list[ray_id].t : kNoHit; slot = kNoNode; }
        }
        const uint32_t cull_first = (CULL && walk_pass == 0) ? P.cull_first : kNoNode;      // (the unfolded tree has no cull nodes)
        unsigned long long tracing = walk_pass == 0 ? __ballot(has_ray) : __ballot(redo);
        uint32_t cur = walk_pass == 0 ? P.quad_root : P.quad_root_flat;
        unsigned long long mask = walk_pass == 0 ? __ballot(root_pass) : tracing;
        if (walk_pass == 1) redo = false;             // nothing below sets it again: pass 1 is the last word
        uint32_t sp = 0;
        // the wave-uniform part of the stack lives in three vector registers, entry i in lane i
        // (read back with v_readlane, the level as lane select): no LDS round trip on a pop
        int stack_ref = 0, stack_mask_lo = 0, stack_mask_hi = 0;
#ifdef CT_SKIP_WALK                        // diagnostic build: time ray set-up and retirement alone
        if (KIND == CT_SKIP_WALK) asm volatile("s_mov_b64 %0, 0" : "=s"(mask));      // (opaque: the ray set-up must not be optimised away with the walk)
#endif

        // ---------------- the shared walk
        // (instantiated once per octant the packet's rays can all point into, plus once for packets whose rays
        //  disagree: box_filter_f32_oct)
        auto shared_walk = [&](auto octant_tag) {
        constexpr int OCT = decltype(octant_tag)::value;
        while (mask != 0ull) {
            // A packet whose rays have gone separate ways walks the union of 64 different paths one node at a time:
            // after `budget` visits with fewer than packet_lanes rays per visit it stops here and its rays are traced
            // again, each on its own, by the per-lane kernel (a ray's result does not depend on who walks it).
            if (HANDOVER && budget != 0u && !list_full) {
                visits += 1u;
                lane_visits += (uint32_t)__popcll(mask);
                if (visits >= budget && walk_pass == 0) {
                    if (lane_visits < visits * W.packet_lanes) {
                        // a shadow ray that has met its blocker is done; every other ray of the packet goes on the list
                        const bool listed = has_ray && (!ANY_HIT || t == kNoHit);
                        const unsigned long long m_l = __ballot(listed);
                        const uint32_t n_l = (uint32_t)__popcll(m_l);
                        unsigned long long at = 0;
                        if (lane == 0) at = atomicAdd(W.handed + (KIND == 0 ? 0 : 16), (unsigned long long)n_l);
                        at = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at) |
                             ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(at >> 32)) << 32);
                        if (at + n_l <= W.handed_room) {
                            if (listed) {
                                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_l >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_l, 0u));
                                W.handed_ids[KIND == 0 ? 0 : 1][at + rank] = (KIND == 0) ? packet_base + lane : ray_id;
                                has_ray = false;               // (the retirement below writes nothing for it)
                            }
                            n_traced -= n_l;                   // the per-lane kernel counts them
                            n_handed += n_l;
                            hand_over = true;
                            break;
                        }
                        // no room (the list is sized for a part of the frame's rays): this wave walks its packets out from
                        // now on.  A reservation that straddles the end must not leave unwritten slots behind: the
                        // per-lane kernel reads every slot below the end, so they get a mark it skips.
                        if (at < W.handed_room) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_l >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_l, 0u));
                            if (listed && at + rank < W.handed_room) W.handed_ids[KIND == 0 ? 0 : 1][at + rank] = ~0ull;
                        }
                        list_full = true;
                    }
                    visits = 0u; lane_visits = 0u;             // still sharing well: look again after another `budget`
                }
            }
            if (cur & kLeafBit) {
                // leaf: every triangle in index order, each lane keeps its own closest hit
                const uint32_t first = cur & kSlotMask;
                const bool mine = __builtin_amdgcn_inverse_ballot_w64(mask);
                PK_T0();
                uint32_t count = 1;        // the first record says how many the leaf holds (a cull node's entry: this one only)
#ifdef CT_SKIP_LEAF_KIND                   // diagnostic build: the walk alone, no triangle is loaded or tested
                if (KIND == CT_SKIP_LEAF_KIND) count = 0;
#endif
                for (uint32_t k = 0; k < count; k++) {
                    const ScalarTri tr = scalar_load_tri(P.tris + first + k);
                    if (k == 0) {
                        count = (CULL && (cur & kOneBit)) ? 1u : tr.remaining();
                        n_records += count;
                        PK_COUNT(2, count);
                        PK_COUNT(3, (uint64_t)count * __popcll(mask));
                        PK_HIST(1, __popcll(mask), count);
                    }
#ifdef CT_TRI_STATS
                    // diagnostic build (tools/tri_stats.py): how far does the WAVE get into IntersectTriangle for this
                    // record -- the test's exits are wave-level branches, so a stage costs only if some lane reaches it
                    if (KIND == W.thresh_a) {
                        const d3 p1s = make_d3(tr.f64(0), tr.f64(1), tr.f64(2)), e1s = make_d3(tr.f64(3), tr.f64(4), tr.f64(5)),
                                 e2s = make_d3(tr.f64(6), tr.f64(7), tr.f64(8));
                        const bool active = mine && (!ANY_HIT || t == kNoHit);
                        int stage = 0;                                      // 0: out at |a| < 1e-4, 1: out at u, 2: out at v / u+v, 3: passes
                        {
                            const d3 hs = cross(d, e2s);
                            const float as = dotf(e1s, hs);
                            if (!(as > -0.0001f && as < 0.0001f)) {
                                const float fs = 1.0f / as;
                                const d3 ss = RAY_O - p1s;
                                const float us = fs * dotf(ss, hs);
                                stage = 1;
                                if (!(us < 0 || us > 1)) {
                                    const d3 qs = cross(ss, e1s);
                                    const float vs = fs * dotf(d, qs);
                                    stage = 2;
                                    if (!(vs < 0 || us + vs > 1)) stage = 3;
                                }
                            }
                        }
                        const unsigned long long n_act = __ballot(active), r3 = __ballot(active && stage == 3),
                                                 r2 = __ballot(active && stage == 2), r1 = __ballot(active && stage == 1);
                        const int reach = r3 ? 3 : (r2 ? 2 : (r1 ? 1 : 0));
                        if (n_act != 0ull && lane == 0) {
                            atomicAdd(&P.counters[19], 1ull);
                            atomicAdd(&P.counters[20 + reach], 1ull);
                        }
                    }
#endif
                    if (mine && (!ANY_HIT || t == kNoHit)) {
                        const d3 p1 = make_d3(tr.f64(0), tr.f64(1), tr.f64(2)), e1 = make_d3(tr.f64(3), tr.f64(4), tr.f64(5)),
                                 e2 = make_d3(tr.f64(6), tr.f64(7), tr.f64(8));
                        const float before = t;
                        const bool hit = triangle_test(RAY_O, d, t, p1, e1, e2);
                        const float tclosest = (slot != kNoNode) ? before : kFarAway;      // bvh.cpp:212-215
                        if (hit && t != kNoHit && t < tclosest) slot = first + k;
                    }
                }
                if (ANY_HIT) tracing &= __ballot(t == kNoHit);
                mask = 0ull;               // on to the pop below
                PK_T1(1);
            } else {
                PK_T0();
                PK_COUNT(0, 1);
                n_quads += 1;
                PK_COUNT(1, __popcll(mask));
                PK_HIST(0, __popcll(mask), 1);
                const uint32_t node_id = CULL ? (cur & kSlotMask) : cur;       // (a cull node's entries carry kOneBit)
                const ScalarQuad sq = scalar_load_quad(quads + node_id);
#ifdef CT_DUMMY_VALU                       // diagnostic builds: what N more vector / scalar instructions per visit cost
                {
                    float dummy = t;
#pragma unroll
                    for (int i = 0; i < CT_DUMMY_VALU; i++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dummy));
                    asm volatile("" :: "v"(dummy));
                }
#endif
#ifdef CT_DUMMY_SALU
                {
                    uint32_t dummy = node_id;
#pragma unroll
                    for (int i = 0; i < CT_DUMMY_SALU; i++) asm volatile("s_add_u32 %0, %0, %0" : "+s"(dummy) :: "scc");
                    asm volatile("" :: "s"(dummy));
                }
#endif
                // level 0: binary32 planes, verdicts with an error bound.  Every lane computes (the
                // masked-out ones on stale rays, harmlessly) and the verdicts leave the vector unit
                // as wave masks: no per-lane flag crosses a branch.
                float tmin[4];
                unsigned long long enter[4];
                // An absent entry (a QuadNode has 2..4) holds a one-point box at a corner of the scene's
                // bound: its tmax - tmin is never positive, so it is never entered, and no scalar
                // instruction is spent on "is entry k there".  Whether ANY lane is undecided on ANY
                // entry comes from one compare: the least of |least_k| - margin_k must be positive
                // (a NaN, or the rare ray that grazes that corner point, sends the wave to the exact
                // levels, which look at the entries one by one).
                // A cull node (ct_device.h): its entries' boxes only ever EXCLUDE -- a lane stays out of an entry when the
                // filter proves that the real slab test on the stored planes fails (`least < -margin`: the bound of appendix
                // B read from the other side) and its ray is inside the gate (appendix D; a closest-hit ray only while its
                // ray.t is at least tau); undecided lanes and lanes outside the gate go in.  No exact levels: nothing here
                // is decided against the reference's arithmetic, only triangle tests with a known outcome are left out.
                const bool cull_node = CULL && node_id >= cull_first;
                if (cull_node) {
                    const unsigned long long gated = ANY_HIT ? cull_lanes : (cull_lanes & __ballot(t >= P.cull.tau));
                    const uint32_t n_entries = sq.count();
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float least, margin;
                        box_filter_f32_oct<OCT, ANY_HIT>(rf, sq.planes(k, 0), sq.planes(k, 1), sq.planes(k, 2), t, tmin[k], least, margin);
                        enter[k] = ((uint32_t)k < n_entries) ? (mask & ~(__ballot(least < -margin) & gated)) : 0ull;
                    }
                } else {
                float worst = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float least, margin;
                    box_filter_f32_oct<OCT, ANY_HIT>(rf, sq.planes(k, 0), sq.planes(k, 1), sq.planes(k, 2), t, tmin[k], least, margin);
                    enter[k] = __ballot(least > margin) & mask;
                    const float slack = __builtin_fabsf(least) - margin;           // > 0: decided
                    worst = (k == 0) ? slack : __builtin_fminf(worst, slack);
                }
                const unsigned long long unsure = __ballot(!(worst > 0.0f)) & mask;
                if (unsure != 0ull) {
                    PK_COUNT(4, 1);
                    // which entries, which lanes: the filter once more, entry by entry
                    const uint32_t n_entries = sq.count();
                    unsigned long long unsure_of[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float tmin_again, least, margin;
                        box_filter_f32<ANY_HIT>(rf, sq.planes(k, 0), sq.planes(k, 1), sq.planes(k, 2), t, tmin_again, least, margin);
                        unsure_of[k] = ((uint32_t)k < n_entries) ? (__ballot(!(__builtin_fabsf(least) > margin)) & mask) : 0ull;
                    }
                    {
                    // level 1 (some lane sits inside the bound): binary64 planes, one-step margins;
                    // level 2 (rarer still): the reference's divisions.  One box at a time, planes by
                    // vector load: this path is rare and must stay light on registers (it would
                    // otherwise set the kernel's occupancy).
                    uint32_t todo = 0u, passed = 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++) todo |= __builtin_amdgcn_inverse_ballot_w64(unsure_of[k]) ? (1u << k) : 0u;
                    if (todo != 0u) {
                        // (a zero direction component is fine here: its parameters are the reference's own
                        //  infinities, and the 0 * inf of an origin on a box plane is seen and sent on)
                        const bool fast = direction_allows_checked_slab(d) && (!LIST || t > 0.0f);
#pragma nounroll
                        for (uint32_t k = 0; k < n_entries; k++) {
                            if (((todo >> k) & 1u) == 0u) continue;
                            const double *bmin = entry_box(node_id * 4u + k), *bmax = bmin + 3;
                            const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
                            float tmin64, tmax64;
                            bool verdict;
                            const bool some_nan = slab_approx_checked(RAY_O, inv, bmin, bmax, tmin64, tmax64);
                            if (!(box_verdict_positive_t(tmin64, tmax64, t, verdict) & fast & !some_nan)) {
                                const bool nan_parameter = slab_in_sequence(RAY_O, d, bmin, bmax, tmin64, tmax64);
                                if (walk_pass == 0) redo |= nan_parameter;      // only the folded walk gives rays up
                                verdict = tmax64 >= tmin64 && tmin64 < t && tmax64 > 0;
                            }
                            passed |= verdict ? (1u << k) : 0u;
                        }
                    }
                    // a NaN parameter: folding is not proven for this ray (DESIGN B2) -- it leaves the
                    // packet here and is walked again from the root, unfolded, after the packet
                    const unsigned long long gone = __ballot(redo);
                    tracing &= ~gone;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        enter[k] = ((enter[k] & ~unsure_of[k]) | __ballot(((passed >> k) & 1u) != 0u)) & ~gone;
                    }
                }
                }   // (not a cull node)
                PK_T1(0);
                // the first entry some lane enters comes next; later ones wait on the stack (pushed last
                // first), each lane's `tmin < ray.t` re-evaluated when the entry is popped.  Going down
                // from entry 3: whenever a lower entry turns up, the one found before it is pushed.
                // (the stack cannot overflow: packet_depth is the exact maximum over the tree, set at upload)
                uint32_t next_ref = 0;
                unsigned long long next_mask = 0ull;
                float next_tmin = 0.0f;
                uint32_t next_entry = 0;
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    if (enter[k] != 0ull) {
                        if (next_mask != 0ull) {
                            const bool level = lane == sp;       // (v_writelane cannot take value and lane select from two scalar registers)
                            stack_ref = level ? (int)next_ref : stack_ref;
                            stack_mask_lo = level ? (int)(uint32_t)next_mask : stack_mask_lo;
                            stack_mask_hi = level ? (int)(uint32_t)(next_mask >> 32) : stack_mask_hi;
                            // (an any-hit ray's tmin < ray.t holds for as long as it is traced -- ray.t is the 1e30f it started
                            //  with until the ray leaves -- whichever level decided it when the entry was pushed: nothing waits
                            //  with the entry, neither a tmin nor the way to its binary64 box)
                            if (!ANY_HIT) {
                                st.parent[sp] = node_id * 4u + next_entry;            // every lane, same word: read on the exact path only
                                st.put_tmin(sp, lane, next_tmin);
                            }
                            sp += 1;
                        }
                        next_ref = sq.ref(k);
                        next_mask = enter[k];
                        next_tmin = tmin[k];
                        next_entry = (uint32_t)k;
                    }
                }
                cur = next_ref;
                mask = next_mask;
                if (mask != 0ull) continue;
            }
            // ---------------- pop: next pending entry some lane still wants
            while (mask == 0ull && sp > 0) {
                PK_T0();
                PK_COUNT(5, 1);
                sp -= 1;
                const uint32_t pushed_lo = (uint32_t)__builtin_amdgcn_readlane(stack_mask_lo, (int)sp);
                const uint32_t pushed_hi = (uint32_t)__builtin_amdgcn_readlane(stack_mask_hi, (int)sp);
                const unsigned long long pushed = ((unsigned long long)pushed_lo | ((unsigned long long)pushed_hi << 32)) & tracing;
                if (pushed == 0ull) continue;
                bool keep = false;
                if (ANY_HIT && !CULL) {
                    // an any-hit ray's deferred  tmin < ray.t  was settled when the entry was pushed (ray.t is 1e30f for as long
                    // as the ray is traced; the filter's rays have |tmin| < 2^84 -- ray_filter's e_limit --, the others were
                    // judged by the exact levels against that same 1e30f): whoever was pushed and is still traced goes in
                    mask = pushed;
                    cur = (uint32_t)__builtin_amdgcn_readlane(stack_ref, (int)sp);
                    PK_T1(2);
                    continue;
                }
                if (CULL && !ANY_HIT && !LIST && (pushed & __ballot(t != kNoHit)) == 0ull) {
                    // none of the rays the entry waits for has found a triangle yet: their ray.t is the 1e30f it was when the
                    // entry was pushed, and "tmin < ray.t" was part of the verdict that pushed it -- nothing to look up.
                    // (The instantiation with cull nodes only: packets over big leaves stay hit-free for long, C2's primary
                    //  walk -2.3 %; the dragon's packets do not, and there the extra compare per pop costs 2 %.)
                    mask = pushed;
                    cur = (uint32_t)__builtin_amdgcn_readlane(stack_ref, (int)sp);
                    PK_T1(2);
                    continue;
                }
                if (!CULL) {
                    // (a tree without cull nodes: the pop as it has always been -- every wave-level operation here is paid per pop)
                    if (__builtin_amdgcn_inverse_ballot_w64(pushed)) {
                        // the deferred  tmin < ray.t : from the filter's tmin, good to M(tmin)
                        const bool settled = deferred_verdict_f32(st.get_tmin(sp, lane), rf.noi_z_e.y, t, keep);
                        if (!settled) {
                        // too close to ray.t to trust (or a ray the filter does not take: e = inf): tmin from
                        // the binary64 box, one step off at most -- and if that is still too close, the whole
                        // test exactly
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const double *bmin = entry_box(st.parent[sp]);
                        const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
                        float etmin, etmax;
                        const bool some_nan = slab_approx_checked(RAY_O, inv, bmin, bmin + 3, etmin, etmax);
                        const bool near_t = ordinals_near(__float_as_int(etmin), __float_as_int(t), 2u);
                        keep = etmin < t;        // (tmax >= tmin and tmax > 0 held when the entry was pushed)
                        if (some_nan | near_t | !direction_allows_checked_slab(d) | (LIST && !(t > 0.0f))) {
                            (void)slab_in_sequence(RAY_O, d, bmin, bmin + 3, etmin, etmax);  // (a NaN here was seen at push time)
                            keep = etmax >= etmin && etmin < t && etmax > 0;
                        }
                        }
                    }
                    mask = __ballot(keep);
                    cur = (uint32_t)__builtin_amdgcn_readlane(stack_ref, (int)sp);
                    PK_T1(2);
                    continue;
                }
                bool unsettled = false;
                if (__builtin_amdgcn_inverse_ballot_w64(pushed)) {
                    if (ANY_HIT) keep = true;
                    else { bool settled = deferred_verdict_f32(st.get_tmin(sp, lane), rf.noi_z_e.y, t, keep); unsettled = !settled; }
                }
                cur = (uint32_t)__builtin_amdgcn_readlane(stack_ref, (int)sp);
                // An entry a cull node pushed (a lower cull node, or one triangle: kOneBit): no binary64 box stands behind
                // it; a lane stays out only when "beyond ray.t" is certain and its ray is inside the gate -- all at the level
                // of wave masks.  (The root of a leaf's cull nodes was pushed by an ordinary node: the leaf's own box, the
                // reference's test, settled exactly like any entry's.)
                if ((cur & kOneBit) != 0u) {
                    mask = (__ballot(keep) | __ballot(unsettled) | ~(ANY_HIT ? cull_lanes : (cull_lanes & __ballot(t >= P.cull.tau)))) & pushed;
                    PK_T1(2);
                    continue;
                }
                if (!ANY_HIT && __ballot(unsettled) != 0ull) {
                    if (unsettled) {
                        // too close to ray.t to trust (or a ray the filter does not take: e = inf): tmin from
                        // the binary64 box, one step off at most -- and if that is still too close, the whole
                        // test exactly
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const double *bmin = entry_box(st.parent[sp]);
                        const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
                        float etmin, etmax;
                        const bool some_nan = slab_approx_checked(RAY_O, inv, bmin, bmin + 3, etmin, etmax);
                        const bool near_t = ordinals_near(__float_as_int(etmin), __float_as_int(t), 2u);
                        keep = etmin < t;        // (tmax >= tmin and tmax > 0 held when the entry was pushed)
                        if (some_nan | near_t | !direction_allows_checked_slab(d) | (LIST && !(t > 0.0f))) {
                            (void)slab_in_sequence(RAY_O, d, bmin, bmin + 3, etmin, etmax);  // (a NaN here was seen at push time)
                            keep = etmax >= etmin && etmin < t && etmax > 0;
                        }
                    }
                }
                mask = __ballot(keep);
                PK_T1(2);
            }
        }
        };   // shared_walk
        if (OCTANTS && walk_pass == 0) {
            switch (octant) {
                case 0: shared_walk(std::integral_constant<int, 0>{}); break;
                case 1: shared_walk(std::integral_constant<int, 1>{}); break;
                case 2: shared_walk(std::integral_constant<int, 2>{}); break;
                case 3: shared_walk(std::integral_constant<int, 3>{}); break;
                case 4: shared_walk(std::integral_constant<int, 4>{}); break;
                case 5: shared_walk(std::integral_constant<int, 5>{}); break;
                case 6: shared_walk(std::integral_constant<int, 6>{}); break;
                case 7: shared_walk(std::integral_constant<int, 7>{}); break;
                default: shared_walk(std::integral_constant<int, 8>{}); break;
            }
        } else {
            shared_walk(std::integral_constant<int, 8>{});
        }

        if (hand_over) break;
        }   // passes

        // ---------------- retire the packet
        ARGS_AGAIN_AT(3);
#ifdef CT_PACKET_TRACE
        if (lane == 0 && !LIST) {
            const unsigned int at = atomicAdd(&ct_packet_trace_n, 1u);
            if (at < (1u << 20)) {
                unsigned long long *rec = ct_packet_trace + (size_t)at * 5;
                rec[0] = ((unsigned long long)KIND << 56) | ((unsigned long long)blockIdx.x << 32) | trace_rays;
                rec[1] = trace_begin;
                rec[2] = wall_clock64();
                rec[3] = ((unsigned long long)(n_quads - trace_quads) << 32) | (n_records - trace_records);
                rec[4] = (KIND == 0) ? packet_base : (unsigned long long)redo;
            }
        }
#endif
        if (ORDERED && lane == 0) atomicAdd(W.order[KIND == 0 ? 0 : 1].cost_out + cost_at, n_quads + n_records - cost_mark);
        if (KIND == 1) {
            const unsigned long long rays = __ballot(has_ray), blocked = __ballot(has_ray && t != kNoHit);
            n_blocked += (uint32_t)__popcll(blocked);
            n_all_blocked += (blocked == rays) ? 1u : 0u;
        }
        if (has_ray) {
            if (LIST) {
                ctgpu_hit h;
                h.t = t;
                h.tclosest = (slot != kNoNode) ? t : kFarAway;         // *tclosest == ray.t once a triangle is recorded
                h.triangle = W.list_slots ? slot : ((slot != kNoNode) ? P.tris[slot].triangle : 0u);
                h.found = (t != kNoHit) ? 1u : 0u;
                W.ray_hits[ray_id] = h;
            }
            else if (KIND == 0) W.hits[packet_base + lane] = make_uint2(__float_as_uint(t), slot);
            else if (t != kNoHit) W.shadow[ray_id] = 1;
            else if (W.shadow_live != 0u) W.shadow[ray_id] = 0;       // (a ray the shadow tree's walk left, and nothing blocks it)
        }
    }

#ifdef CT_WALK_STATS
    if (lane == 0 && KIND == W.thresh_a) {     // diagnostic build: CTGPU_WALK_THRESH=r,<kind>,b picks the kernel
        for (int k = 0; k < 6; k++) atomicAdd(&P.counters[5 + k], pk_n[k]);
        for (int k = 0; k < 24; k++) atomicAdd(&P.counters[24 + k], (unsigned long long)pk_hist[k]);
        for (int k = 0; k < 3; k++) atomicAdd(&P.counters[11 + k], pk_cyc[k]);
        atomicAdd(&P.counters[14], __builtin_amdgcn_s_memtime() - pk_begin);
    }
#endif
    if (lane == 0 && LIST && W.list_counter && n_traced) atomicAdd(&P.counters[W.list_counter], (unsigned long long)n_traced);
    if (lane == 0 && !LIST) {
        atomicAdd(&P.counters[KIND == 0 ? 0 : 1], (unsigned long long)n_traced);
        // (64 slots on separate cache lines: atomics on one word would queue up at ~11 ns each)
        unsigned long long *work = W.walk_work + (size_t)(blockIdx.x & 63u) * 8u + (KIND == 0 ? 0u : 3u);
        atomicAdd(work + 0, (unsigned long long)n_packets);
        atomicAdd(work + 1, (unsigned long long)n_quads);
        atomicAdd(work + 2, (unsigned long long)n_records);
        if (n_handed) atomicAdd(&P.counters[KIND == 0 ? 17 : 18], (unsigned long long)n_handed);
        if (ORDERED) {
            unsigned long long *totals = W.order[KIND == 0 ? 0 : 1].totals_out + (size_t)(blockIdx.x & 63u) * 2u;
            atomicAdd(totals + 0, (unsigned long long)(n_quads + n_records));
            atomicAdd(totals + 1, (unsigned long long)n_packets);
        }
        if (KIND == 1) {
            atomicAdd(W.walk_work + (size_t)(blockIdx.x & 63u) * 8u + 6u, (unsigned long long)n_blocked);
            atomicAdd(W.walk_work + (size_t)(blockIdx.x & 63u) * 8u + 7u, (unsigned long long)n_all_blocked);
        }
    }
}
#undef RAY_O
#undef ARGS_AGAIN
#undef ARGS_AGAIN_AT

// ------------------------------------------------------------------------------------
// ct_walk_shadow_tree: the frame's shadow rays through the SHADOW TREE (ct_device.h "the shadow tree", DESIGN appendix E)
// instead of the reference's tree.  Same packets as ct_walk_packet<1> (windows of 128 ray ids under one light, 64 live
// ids a packet), same shared walk -- one {entry, lane mask} per stack level in the lanes of three vector registers, nodes
// and triangle records through the scalar cache -- but of a tree whose shape is this build's own:
//   * an entry's box only excludes, and the test is bare: three packed fmas, max3 / min3 by the packet's octant,
//     tmax >= max(tmin, 0).  No margin, no undecided lanes, no binary64 boxes, no exact levels: the boxes were grown at
//     build time by what the binary32 arithmetic can be off for any ray inside the gate (P.stree_lim);
//   * the lowest entries are single triangles with their own boxes, so a triangle is tested by the lanes that meet ITS box;
//   * a triangle test that lowers ray.t is followed, for those lanes, by the reference's own test of the box of the leaf
//     that holds the triangle in the REFERENCE's tree (binary64, the exact levels of appendix A): only then would the
//     reference have made this test at all.  A lane whose leaf box fails carries on as if nothing had happened;
//   * rays outside the gate are left to a second launch of ct_walk_packet<1> (shadow byte kShadowLeft, W.shadow_left).
// Results are those of the reference's full traversal: raythread.cpp:303-307 asks only whether ray.t moved.
// ------------------------------------------------------------------------------------
#ifndef CT_STREE_WAVES
#define CT_STREE_WAVES 7
#endif
struct RayLine {              // t(plane) = fma(plane, 1/d, -o/d) in binary32
    f32x2 inv_xy, noi_xy;     // {1/d.x, 1/d.y}, {-o.x/d.x, -o.y/d.y}
    f32x2 z;                  // {1/d.z, -o.z/d.z}: one register pair serves both operands of the z axis' packed fma
};
template <int OCT>
__device__ __forceinline__ bool line_enters(const RayLine &r, f32x2 px, f32x2 py, f32x2 pz) {
    f32x2 tx, ty, tz;
    float tmin, tmax, from;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tx) : "s"(px), "v"(r.inv_xy), "v"(r.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(ty) : "s"(py), "v"(r.inv_xy), "v"(r.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tz) : "s"(pz), "v"(r.z), "v"(r.z));
    if (OCT == 8) {           // a packet whose rays point different ways: each axis' two parameters sorted per lane
        float nx, ny, nz, fx, fy, fz;
        asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(tx.x), "v"(tx.y));
        asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ty.x), "v"(ty.y));
        asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(tz.x), "v"(tz.y));
        asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(tx.x), "v"(tx.y));
        asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ty.x), "v"(ty.y));
        asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(tz.x), "v"(tz.y));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(nx), "v"(ny), "v"(nz));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(fx), "v"(fy), "v"(fz));
    } else {                  // (box_filter_f32_oct's selection: the near plane of an axis is known from the sign of 1/d)
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"((OCT & 1) ? tx.y : tx.x), "v"((OCT & 2) ? ty.y : ty.x), "v"((OCT & 4) ? tz.y : tz.x));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"((OCT & 1) ? tx.x : tx.y), "v"((OCT & 2) ? ty.x : ty.y), "v"((OCT & 4) ? tz.x : tz.y));
    }
    asm("v_max_f32 %0, 0, %1" : "=v"(from) : "v"(tmin));
    return tmax >= from;
}
// The four entries of a node against the packet's rays in ONE block of code (the packets whose rays share an octant:
// OCT < 8): per entry three packed fmas, the near planes' max3, the far planes' min3, max(tmin, 0), one compare -- 28
// vector instructions, the verdicts leaving as four wave masks.  Written as a single asm statement with its own seven
// registers (v56..v62, clobbered; the four masks are EARLY-CLOBBER outputs: an entry's verdict is written while later
// entries' planes are still to be read, and without the '&' the compiler may -- and, in one build of round 5, did -- give a
// mask the register pair of a later entry's planes: DESIGN appendix E.5): between separate asm statements the compiler cannot see that none of these
// instructions writes part of a register and puts a wait state in front of every dependent one -- ten s_nop per visit, each
// a slot of the CU's one scalar unit, which is what this walk is shortest of.
#define CT_LINE_ENTRY(OUT, PX, PY, PZ, NX, FX, NY, FY, NZ, FZ)                                        \
    "v_pk_fma_f32 v[56:57], " PX ", %16, %17 op_sel_hi:[1,0,0]\n\t"                                   \
    "v_pk_fma_f32 v[58:59], " PY ", %16, %17 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"                    \
    "v_pk_fma_f32 v[60:61], " PZ ", %18, %18 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                    \
    "v_max3_f32 v62, " NX ", " NY ", " NZ "\n\t"                                                     \
    "v_min3_f32 v56, " FX ", " FY ", " FZ "\n\t"                                                     \
    "v_max_f32 v62, 0, v62\n\t"                                                                      \
    "v_cmp_ge_f32_e64 " OUT ", v56, v62\n\t"
#define CT_LINE_NODE(NX, FX, NY, FY, NZ, FZ)                                                          \
    CT_LINE_ENTRY("%0", "%4", "%5", "%6", NX, FX, NY, FY, NZ, FZ)                                      \
    CT_LINE_ENTRY("%1", "%7", "%8", "%9", NX, FX, NY, FY, NZ, FZ)                                      \
    CT_LINE_ENTRY("%2", "%10", "%11", "%12", NX, FX, NY, FY, NZ, FZ)                                   \
    CT_LINE_ENTRY("%3", "%13", "%14", "%15", NX, FX, NY, FY, NZ, FZ)
template <int OCT>
__device__ __forceinline__ void node_enters(const RayLine &r, const ScalarQuad &sq, unsigned long long enter[4]) {
    static_assert(OCT >= 0 && OCT < 8, "the packets whose rays share an octant");
#define CT_LINE_ASM(TEXT)                                                                                             \
    asm volatile(TEXT : "=&s"(enter[0]), "=&s"(enter[1]), "=&s"(enter[2]), "=&s"(enter[3])                             \
                 : "s"(sq.planes(0, 0)), "s"(sq.planes(0, 1)), "s"(sq.planes(0, 2)), "s"(sq.planes(1, 0)), "s"(sq.planes(1, 1)), "s"(sq.planes(1, 2)), \
                   "s"(sq.planes(2, 0)), "s"(sq.planes(2, 1)), "s"(sq.planes(2, 2)), "s"(sq.planes(3, 0)), "s"(sq.planes(3, 1)), "s"(sq.planes(3, 2)), \
                   "v"(r.inv_xy), "v"(r.noi_xy), "v"(r.z)                                                              \
                 : "v56", "v57", "v58", "v59", "v60", "v61", "v62")
    // (the near plane of an axis: the minimum plane -- the low half of the fma's pair -- where the rays go up, else the maximum)
    if (OCT == 0) CT_LINE_ASM(CT_LINE_NODE("v56", "v57", "v58", "v59", "v60", "v61"));
    if (OCT == 1) CT_LINE_ASM(CT_LINE_NODE("v57", "v56", "v58", "v59", "v60", "v61"));
    if (OCT == 2) CT_LINE_ASM(CT_LINE_NODE("v56", "v57", "v59", "v58", "v60", "v61"));
    if (OCT == 3) CT_LINE_ASM(CT_LINE_NODE("v57", "v56", "v59", "v58", "v60", "v61"));
    if (OCT == 4) CT_LINE_ASM(CT_LINE_NODE("v56", "v57", "v58", "v59", "v61", "v60"));
    if (OCT == 5) CT_LINE_ASM(CT_LINE_NODE("v57", "v56", "v58", "v59", "v61", "v60"));
    if (OCT == 6) CT_LINE_ASM(CT_LINE_NODE("v56", "v57", "v59", "v58", "v61", "v60"));
    if (OCT == 7) CT_LINE_ASM(CT_LINE_NODE("v57", "v56", "v59", "v58", "v61", "v60"));
#undef CT_LINE_ASM
}
// A node / a triangle record through the scalar cache at base + byte offset, the offset in a register: the tree's
// references ARE byte offsets (ct_shadow_tree.hip), and the second load's base is the array's own base + 64, kept beside
// it -- no address arithmetic on the walk's chain.
__device__ __forceinline__ ScalarQuad scalar_load_quad_at(const void *base, const void *base64, uint32_t offset) {
    ScalarQuad r;
    asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx16 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r.a), "=&s"(r.b)
                 : "s"(base), "s"(base64), "s"(offset)
                 : "memory");
    return r;
}
__device__ __forceinline__ ScalarTri scalar_load_tri_at(const void *base, const void *base64, uint32_t offset) {
    ScalarTri r;
    asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx4 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r.a), "=&s"(r.b)
                 : "s"(base), "s"(base64), "s"(offset)
                 : "memory");
    return r;
}
// ORDERED: as in ct_walk_packet -- the chunks the frame before found heavy come first (in parts where its packets were few),
// then the run over all chunks skips them; every packet adds its fetches to its chunk's cost word.
template <bool SS, bool ORDERED = false>
__global__ __launch_bounds__(kWave, CT_STREE_WAVES) void ct_walk_shadow_tree(const FrameParams P_launch, const PipelineParams W_launch) {
    FrameParams P = P_launch;
    PipelineParams W = W_launch;
    // (the launch's arguments read again per phase of a packet, as in ct_walk_packet<1>: kernarg_again)
#define STREE_ARGS() do { if (CT_ARGS_AGAIN != 0) { kernarg_again(P, 0u); kernarg_again(W, kWalkArgsW); args_point_into_hbm(P, W); } } while (0)
    STREE_ARGS();
    if (CT_ARGS_AGAIN != 0) if (P.nodes != P_launch.nodes || P.rng_seed != P_launch.rng_seed || W.coords != W_launch.coords || W.n_samples != W_launch.n_samples ||
        W.order[1].totals_out != W_launch.order[1].totals_out || P.counters != P_launch.counters) {
        if (threadIdx.x == 0) *P_launch.overflow = 2u;
        return;
    }
    const uint32_t q_chunk = W.chunk_shadow;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    LdsWord *win_list = (LdsWord *)lds_raw;          // [128] the open window's live ray offsets
    const uint32_t lane = threadIdx.x;
    const uint64_t n_samples = W.n_samples;
    const uint32_t n_shards = kShards * W.n_shadow_lights;
    unsigned long long *heads = W.queue + (size_t)kShards * 16u;
    const uint64_t shard_span = ((n_samples + kShards - 1) / kShards + q_chunk - 1) / q_chunk * q_chunk;
    uint32_t shard = blockIdx.x % n_shards, shards_done = 0;
    uint64_t light_base = 0, chunk_light_base = 0;
    uint32_t light_index = 0, chunk_light_index = 0;
    d3 shadow_o = make_d3(0, 0, 0), d = make_d3(0, 0, 1);
    float t = kNoHit;
    bool has_ray = false;
    uint32_t ray_at = 0;               // the ray's id = ray_base (the packet's window: the same for all its lanes) + ray_at
    uint64_t ray_base = 0;
    uint64_t chunk_next = 0, chunk_end = 0;
    bool wave_drained = false;
    uint32_t n_traced = 0, n_left = 0;
    uint32_t win_pos = 0, win_count = 0;
    uint64_t win_base = 0;
    uint32_t n_packets = 0, n_quads = 0, n_records = 0, n_blocked = 0, n_all_blocked = 0;
    uint32_t order_phase = ORDERED ? 0u : 2u, order_at = blockIdx.x, chunk_cost_at = 0, cost_at = 0, cost_mark = 0;

    for (;;) {
        STREE_ARGS();
        // ---------------- fetch a packet (ct_walk_packet<1>'s windows)
        has_ray = false;
        for (;;) {
            if (win_pos < win_count) {
                const uint32_t n = win_count - win_pos < 64u ? win_count - win_pos : 64u;
                ray_base = win_base;
                if (lane < n) { ray_at = win_list[win_pos + lane]; has_ray = true; }
                win_pos += n;
                break;
            }
            if (wave_drained) break;
            if (chunk_next >= chunk_end) {
                bool from_list = false;
                if (ORDERED && order_phase < 2u) {
                    // the heavy lists, heaviest class first: wave b takes entries b, b + grid, ... (ct_walk_packet)
                    const OrderParams &O = W.order[1];
                    uint32_t at = order_at, cls = 0;
                    for (; cls < (uint32_t)kOrderClasses; cls++) {
                        uint32_t listed_entries = (uint32_t)__builtin_amdgcn_readfirstlane((int)O.count[cls]);
                        if (listed_entries > O.cap) listed_entries = O.cap;
                        if (at < listed_entries) break;
                        at -= listed_entries;
                    }
                    if (cls == (uint32_t)kOrderClasses) { order_phase = 2u; continue; }
                    order_at += gridDim.x;
                    const uint4 e = O.list[(size_t)cls * O.cap + at];
                    const uint32_t e_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x), e_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
                    const uint32_t e_light = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z);
                    if (e_lo == ~0u || e_lo >= n_samples || e_light >= W.n_shadow_lights) continue;
                    uint64_t hi = (uint64_t)e_lo + e_n;
                    if (hi > n_samples) hi = n_samples;
                    chunk_light_base = (uint64_t)e_light * n_samples;
                    chunk_next = chunk_light_base + e_lo;
                    chunk_end = chunk_light_base + hi;
                    chunk_cost_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.w);
                    chunk_light_index = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.shadow_lights[e_light]);
                    from_list = true;
                }
                if (!from_list) {
                unsigned long long offset = 0;
                if (lane == 0) offset = atomicAdd(heads + (size_t)shard * 16u, (unsigned long long)q_chunk);
                offset = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)offset) |
                         ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(offset >> 32)) << 32);
                const uint32_t light = shard / kShards, part = shard % kShards;
                const uint64_t lo = (uint64_t)part * shard_span + offset;
                if (offset >= shard_span || lo >= n_samples) {
                    shards_done += 1;
                    shard = (shard + 1u) % n_shards;
                    if (shards_done >= n_shards) wave_drained = true;
                    continue;
                }
                if (ORDERED) {
                    const OrderParams &O = W.order[1];
                    chunk_cost_at = light * O.chunks_per_light + (uint32_t)(lo >> O.chunk_shift);
                    if (__builtin_amdgcn_readfirstlane((int)O.listed[chunk_cost_at]) != 0) continue;      // went out with the heavy lists
                }
                uint64_t hi = lo + q_chunk;
                if (hi > ((uint64_t)part + 1u) * shard_span) hi = ((uint64_t)part + 1u) * shard_span;
                if (hi > n_samples) hi = n_samples;
                chunk_light_base = (uint64_t)light * n_samples;
                chunk_next = chunk_light_base + lo;
                chunk_end = chunk_light_base + hi;
                chunk_light_index = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.shadow_lights[light]);
                }
            }
            // open the next window of the chunk: which of its ids are rays
            const uint32_t n_ids = chunk_end - chunk_next < 128u ? (uint32_t)(chunk_end - chunk_next) : 128u;
            const uint64_t id0 = chunk_next + lane, id1 = chunk_next + 64u + lane;
            const bool live0 = lane < n_ids && W.shadow[id0] == 0, live1 = 64u + lane < n_ids && W.shadow[id1] == 0;
            const unsigned long long m_live0 = __ballot(live0), m_live1 = __ballot(live1);
            const uint32_t n_live0 = (uint32_t)__popcll(m_live0);
            const auto rank_in = [&](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
            if (live0) win_list[rank_in(m_live0)] = lane;
            if (live1) win_list[n_live0 + rank_in(m_live1)] = 64u + lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            win_base = chunk_next;
            win_pos = 0; win_count = n_live0 + (uint32_t)__popcll(m_live1);
            chunk_next += n_ids;
            light_base = chunk_light_base;
            light_index = chunk_light_index;
            if (ORDERED) cost_at = chunk_cost_at;
        }
        if (__ballot(has_ray) == 0ull) {
            if (wave_drained) break;
            continue;
        }
        if (ORDERED) cost_mark = n_quads + n_records;
        // ---------------- the rays (raythread.cpp:360, 289-303), and which of them the shadow tree takes
        STREE_ARGS();
        RayLine rl = RayLine{};                  // (per packet: nothing of it is carried from one packet to the next)
        bool left = false;
        if (has_ray) {
            const uint64_t sample = (ray_base - light_base) + ray_at;
            float px, py;
            if (SS) { const float2 c = W.coords[sample]; px = c.x; py = c.y; }
            else (void)sample_position<SS>(P, W, sample, px, py);
            const uint2 h = W.hits[sample];
            const float tclosest = (h.y != kNoNode) ? __uint_as_float(h.x) : kFarAway;
            shadow_o = make_d3(P.cam[0], P.cam[1], P.cam[2]) + scale((double)tclosest, primary_direction(P, px, py));
            const Light &shadow_light = P.lights[light_index];
            d = (shadow_light.type == CTGPU_LIGHT_POINT) ? load_d3(shadow_light.position) - shadow_o
                                                         : load_d3(shadow_light.direction);
            t = kNoHit;
            const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
            rl.inv_xy = f32x2{(float)inv.x, (float)inv.y};
            rl.noi_xy = f32x2{-(float)(shadow_o.x * inv.x), -(float)(shadow_o.y * inv.y)};
            rl.z = f32x2{(float)inv.z, -(float)(shadow_o.z * inv.z)};
            // the gate (appendix E.2): magnitudes for which the boxes' growth covers the binary32 arithmetic and appendix D's
            // lemma covers the reference's; anything else -- a zero component first of all -- is left to the second launch
            const double d_min = (double)P.stree_lim.d_min, d_max = (double)P.stree_lim.d_max, o_max = (double)P.stree_lim.o_max;
            const double ax = __builtin_fabs(d.x), ay = __builtin_fabs(d.y), az = __builtin_fabs(d.z);
            const bool inside = ax >= d_min && ax <= d_max && ay >= d_min && ay <= d_max && az >= d_min && az <= d_max &&
                                __builtin_fabs(shadow_o.x) <= o_max && __builtin_fabs(shadow_o.y) <= o_max && __builtin_fabs(shadow_o.z) <= o_max;
            left = !inside;
            if (left) { W.shadow[ray_base + ray_at] = kShadowLeft; has_ray = false; }
        }
        n_left += (uint32_t)__popcll(__ballot(left));
        unsigned long long tracing = __ballot(has_ray);
        n_traced += (uint32_t)__popcll(tracing);
        if (tracing == 0ull) continue;
        n_packets += 1;
        uint32_t octant = 8u;
        {
            const unsigned long long down_x = __ballot(has_ray && __float_as_int(rl.inv_xy.x) < 0);
            const unsigned long long down_y = __ballot(has_ray && __float_as_int(rl.inv_xy.y) < 0);
            const unsigned long long down_z = __ballot(has_ray && __float_as_int(rl.z.x) < 0);
            if ((down_x == 0ull || down_x == tracing) && (down_y == 0ull || down_y == tracing) && (down_z == 0ull || down_z == tracing))
                octant = (down_x ? 1u : 0u) | (down_y ? 2u : 0u) | (down_z ? 4u : 0u);
        }
        STREE_ARGS();
        const unsigned long long rays = tracing;
        const unsigned char *nodes_base = (const unsigned char *)P.stree, *nodes_base64 = nodes_base + 64;
        const unsigned char *tris_base = (const unsigned char *)P.stree_tris, *tris_base64 = tris_base + 64;
        uint32_t cur = 0u;                      // the root is node 0 (a tree always has one)
        unsigned long long mask = tracing;
        uint32_t sp = 0;
        int stack_ref = 0, stack_mask_lo = 0, stack_mask_hi = 0;
        uint32_t hit_at = 0u;                   // the TriRecord whose test moved this lane's ray.t

        auto shared_walk = [&](auto octant_tag) {
        constexpr int OCT = decltype(octant_tag)::value;
        while (mask != 0ull) {
            if (cur & kLeafBit) {
                // a leaf: a run of triangles (mostly one or two), tested by the lanes that met the leaf's box
                const uint32_t first = cur & ~(kLeafBit | 15u);      // (byte offset of the leaf's first record)
                const bool mine = __builtin_amdgcn_inverse_ballot_w64(mask);
                const uint32_t count = (cur & 15u) + 1u;              // (a leaf's reference says how many: ct_shadow_tree_build.h)
                n_records += count;
                for (uint32_t k = 0; k < count; k++) {
                    const ScalarTri tr = scalar_load_tri_at(tris_base, tris_base64, first + k * (uint32_t)sizeof(TriRecord));
                    if (mine && t == kNoHit) {
                        const d3 p1 = make_d3(tr.f64(0), tr.f64(1), tr.f64(2)), e1 = make_d3(tr.f64(3), tr.f64(4), tr.f64(5)),
                                 e2 = make_d3(tr.f64(6), tr.f64(7), tr.f64(8));
                        (void)triangle_test(shadow_o, d, t, p1, e1, e2);
                        if (t != kNoHit) hit_at = first + k * (uint32_t)sizeof(TriRecord);
                    }
                }
                // (a lane whose ray.t moved stops here; whether the reference would have made that test at all is looked
                //  at after the walk, once per packet: "the verdict of the leaf's box" below)
                tracing &= __ballot(t == kNoHit);
                if (tracing == 0ull) sp = 0;               // every ray of the packet has its blocker
                mask = 0ull;
            } else {
                n_quads += 1;
                const ScalarQuad sq = scalar_load_quad_at(nodes_base, nodes_base64, cur);      // (cur: the node's byte offset)
                unsigned long long enter[4];
                if constexpr (OCT < 8) {
                    node_enters<OCT>(rl, sq, enter);
#pragma unroll
                    for (int k = 0; k < 4; k++) enter[k] &= mask;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        enter[k] = __ballot(line_enters<OCT>(rl, sq.planes(k, 0), sq.planes(k, 1), sq.planes(k, 2))) & mask;
                }
                // (an entry the node does not have is a far corner point with a safe reference: ct_shadow_tree_build.h)
                uint32_t next_ref = 0;
                unsigned long long next_mask = 0ull;
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    if (enter[k] != 0ull) {
                        if (next_mask != 0ull) {
                            const bool level = lane == sp;
                            stack_ref = level ? (int)next_ref : stack_ref;
                            stack_mask_lo = level ? (int)(uint32_t)next_mask : stack_mask_lo;
                            stack_mask_hi = level ? (int)(uint32_t)(next_mask >> 32) : stack_mask_hi;
                            sp += 1;
                        }
                        next_ref = sq.ref(k);
                        next_mask = enter[k];
                    }
                }
                cur = next_ref;
                mask = next_mask;
                if (mask != 0ull) continue;
            }
            while (mask == 0ull && sp > 0) {
                sp -= 1;
                const uint32_t pushed_lo = (uint32_t)__builtin_amdgcn_readlane(stack_mask_lo, (int)sp);
                const uint32_t pushed_hi = (uint32_t)__builtin_amdgcn_readlane(stack_mask_hi, (int)sp);
                mask = ((unsigned long long)pushed_lo | ((unsigned long long)pushed_hi << 32)) & tracing;
                cur = (uint32_t)__builtin_amdgcn_readlane(stack_ref, (int)sp);
            }
        }
        };
        switch (octant) {
            case 0: shared_walk(std::integral_constant<int, 0>{}); break;
            case 1: shared_walk(std::integral_constant<int, 1>{}); break;
            case 2: shared_walk(std::integral_constant<int, 2>{}); break;
            case 3: shared_walk(std::integral_constant<int, 3>{}); break;
            case 4: shared_walk(std::integral_constant<int, 4>{}); break;
            case 5: shared_walk(std::integral_constant<int, 5>{}); break;
            case 6: shared_walk(std::integral_constant<int, 6>{}); break;
            case 7: shared_walk(std::integral_constant<int, 7>{}); break;
            default: shared_walk(std::integral_constant<int, 8>{}); break;
        }

        // ---------------- the verdict of the leaf's box
        // A triangle test moved the ray's ray.t: would the reference have made that test?  Only inside the triangle's leaf,
        // i.e. if that leaf's box passes IntersectAABB for this ray at ray.t = 1e30f (bvh.cpp:165-179, 200); its
        // ancestors' boxes then pass too -- nested boxes, and no NaN parameter since the gate keeps zero components out
        // (appendix B2).  The reference's node, its binary64 box, the exact levels of appendix A.  A ray whose box FAILS
        // (the triangle test and the box test disagree at the box's rim: one ray in millions) has stopped half-way through
        // its walk; it is left to the second launch like a ray outside the gate.
        STREE_ARGS();
        if (has_ray && t != kNoHit) {
            const ctgpu_node *leaf = P.ref_nodes + P.leaf_of_slot[hit_at / (uint32_t)sizeof(TriRecord)];
            const double4 b0 = *reinterpret_cast<const double4 *>(&leaf->bmin[0]);     // bmin xyz, bmax x
            const double2 b1 = *reinterpret_cast<const double2 *>(&leaf->bmax[1]);     // bmax yz
            const double bmin[3] = {b0.x, b0.y, b0.z}, bmax[3] = {b0.w, b1.x, b1.y};
            const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
            float tmin64, tmax64;
            bool verdict;
            slab_approx(shadow_o, inv, bmin, bmax, tmin64, tmax64);
            if (!box_verdict_positive_t(tmin64, tmax64, kNoHit, verdict)) {
                (void)slab_in_sequence(shadow_o, d, bmin, bmax, tmin64, tmax64);
                verdict = tmax64 >= tmin64 && tmin64 < kNoHit && tmax64 > 0;
            }
            if (!verdict) { W.shadow[ray_base + ray_at] = kShadowLeft; has_ray = false; left = true; }
        }
        {
            const uint32_t undone = (uint32_t)__popcll(__ballot(left && t != kNoHit));
            n_left += undone;
            n_traced -= undone;
        }
        // ---------------- retire the packet
        if (ORDERED && lane == 0) atomicAdd(W.order[1].cost_out + cost_at, n_quads + n_records - cost_mark);
        {
            const unsigned long long blocked = __ballot(has_ray && t != kNoHit);
            n_blocked += (uint32_t)__popcll(blocked);
            n_all_blocked += (blocked == rays) ? 1u : 0u;
        }
        if (has_ray && t != kNoHit) W.shadow[ray_base + ray_at] = 1;
    }

    if (lane == 0) {
        atomicAdd(&P.counters[1], (unsigned long long)n_traced);
        if (n_left) { atomicAdd(W.shadow_left, (unsigned long long)n_left); atomicAdd(&P.counters[kCounterLeft], (unsigned long long)n_left); }
        unsigned long long *work = W.walk_work + (size_t)(blockIdx.x & 63u) * 8u;
        atomicAdd(work + 3, (unsigned long long)n_packets);
        atomicAdd(work + 4, (unsigned long long)n_quads);
        atomicAdd(work + 5, (unsigned long long)n_records);
        atomicAdd(work + 6, (unsigned long long)n_blocked);
        atomicAdd(work + 7, (unsigned long long)n_all_blocked);
        if (ORDERED) {
            unsigned long long *totals = W.order[1].totals_out + (size_t)(blockIdx.x & 63u) * 2u;
            atomicAdd(totals + 0, (unsigned long long)(n_quads + n_records));
            atomicAdd(totals + 1, (unsigned long long)n_packets);
        }
    }
}
#undef STREE_ARGS

// ------------------------------------------------------------------------------------
// ct_walk_closest_tree: the frame's PRIMARY rays through the shadow tree's nodes (DESIGN appendix F).
// ClosestIntersection (raythread.cpp:197-228) returns ray.t = the least t over the triangle tests the reference makes and
// the first triangle, in its visiting order, that reached it.  Which tests it makes depends on its order only through
// pruning -- a box is skipped when its tmin is not below the ray.t found so far -- so the answer can be had from ANY walk that
// (1) cannot miss a triangle whose test could lower or equal the least t (boxes that only exclude; an entry lying wholly
// beyond ray.t (1 + 2^-18) excluded too, appendix D.2's third case with room to spare, while ray.t >= tau), (2) keeps the
// least {t, place in the reference's order} over everything it tests, and (3) makes sure, for the winner alone, that the
// reference tests it whatever it found before: its leaf's box passes the reference's own test with ray.t = the winner's t
// (tmin < t is then certain for every ray.t the reference can have carried at that moment: all of them are >= t).  A ray
// whose winner fails (3) -- the hit sits on the rim of its leaf's box; all hits on axis-aligned flat leaves do -- is left to a
// second launch of ct_walk_packet<0>, like a ray outside the gate.  Same packets as the primary walk (64 consecutive sample
// ids: four pixels' samples), same stack in the lanes of three registers; the per-lane tmin of a waiting entry sits in LDS
// for the first kClosestLdsLevels levels (deeper ones are not pruned when popped, only when visited).
// ------------------------------------------------------------------------------------
#ifndef CT_CLOSEST_WAVES
#define CT_CLOSEST_WAVES 7
#endif
#ifndef CT_CLOSEST_LDS_LEVELS
#define CT_CLOSEST_LDS_LEVELS 16
#endif
constexpr uint32_t kClosestLdsLevels = CT_CLOSEST_LDS_LEVELS;
#define CT_CLOSEST_ENTRY(OUT, TMIN, PX, PY, PZ, NX, FX, NY, FY, NZ, FZ)                                \
    "v_pk_fma_f32 v[56:57], " PX ", %20, %21 op_sel_hi:[1,0,0]\n\t"                                   \
    "v_pk_fma_f32 v[58:59], " PY ", %20, %21 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"                    \
    "v_pk_fma_f32 v[60:61], " PZ ", %22, %22 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                    \
    "v_max3_f32 " TMIN ", " NX ", " NY ", " NZ "\n\t"                                                  \
    "v_min3_f32 v56, " FX ", " FY ", " FZ "\n\t"                                                       \
    "v_min_f32 v56, v56, %23\n\t"                                                                      \
    "v_max_f32 v62, 0, " TMIN "\n\t"                                                                   \
    "v_cmp_ge_f32_e64 " OUT ", v56, v62\n\t"
#define CT_CLOSEST_NODE(NX, FX, NY, FY, NZ, FZ)                                                       \
    CT_CLOSEST_ENTRY("%0", "%4", "%8", "%9", "%10", NX, FX, NY, FY, NZ, FZ)                            \
    CT_CLOSEST_ENTRY("%1", "%5", "%11", "%12", "%13", NX, FX, NY, FY, NZ, FZ)                          \
    CT_CLOSEST_ENTRY("%2", "%6", "%14", "%15", "%16", NX, FX, NY, FY, NZ, FZ)                          \
    CT_CLOSEST_ENTRY("%3", "%7", "%17", "%18", "%19", NX, FX, NY, FY, NZ, FZ)
// the four entries against the packet's rays, each lane's "beyond ray.t" limit folded in: enter iff min(tmax, limit) >=
// max(tmin, 0); tmin[k] comes back for the entries that will wait on the stack
template <int OCT>
__device__ __forceinline__ void node_enters_closest(const RayLine &r, const ScalarQuad &sq, float limit, unsigned long long enter[4], float tmin[4]) {
    static_assert(OCT >= 0 && OCT < 8, "the packets whose rays share an octant");
#define CT_CLOSEST_ASM(TEXT)                                                                                          \
    asm volatile(TEXT : "=&s"(enter[0]), "=&s"(enter[1]), "=&s"(enter[2]), "=&s"(enter[3]), "=&v"(tmin[0]), "=&v"(tmin[1]), "=&v"(tmin[2]), "=&v"(tmin[3]) \
                 : "s"(sq.planes(0, 0)), "s"(sq.planes(0, 1)), "s"(sq.planes(0, 2)), "s"(sq.planes(1, 0)), "s"(sq.planes(1, 1)), "s"(sq.planes(1, 2)), \
                   "s"(sq.planes(2, 0)), "s"(sq.planes(2, 1)), "s"(sq.planes(2, 2)), "s"(sq.planes(3, 0)), "s"(sq.planes(3, 1)), "s"(sq.planes(3, 2)), \
                   "v"(r.inv_xy), "v"(r.noi_xy), "v"(r.z), "v"(limit)                                                  \
                 : "v56", "v57", "v58", "v59", "v60", "v61", "v62")
    if (OCT == 0) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v56", "v57", "v58", "v59", "v60", "v61"));
    if (OCT == 1) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v57", "v56", "v58", "v59", "v60", "v61"));
    if (OCT == 2) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v56", "v57", "v59", "v58", "v60", "v61"));
    if (OCT == 3) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v57", "v56", "v59", "v58", "v60", "v61"));
    if (OCT == 4) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v56", "v57", "v58", "v59", "v61", "v60"));
    if (OCT == 5) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v57", "v56", "v58", "v59", "v61", "v60"));
    if (OCT == 6) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v56", "v57", "v59", "v58", "v61", "v60"));
    if (OCT == 7) CT_CLOSEST_ASM(CT_CLOSEST_NODE("v57", "v56", "v59", "v58", "v61", "v60"));
#undef CT_CLOSEST_ASM
}
// (the packets whose rays point different ways: each axis' two parameters sorted per lane)
__device__ __forceinline__ bool line_enters_closest(const RayLine &r, f32x2 px, f32x2 py, f32x2 pz, float limit, float &tmin) {
    f32x2 tx, ty, tz;
    float tmax, from, nx, ny, nz, fx, fy, fz;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tx) : "s"(px), "v"(r.inv_xy), "v"(r.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(ty) : "s"(py), "v"(r.inv_xy), "v"(r.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tz) : "s"(pz), "v"(r.z), "v"(r.z));
    asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(tx.x), "v"(tx.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ty.x), "v"(ty.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(tz.x), "v"(tz.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(tx.x), "v"(tx.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ty.x), "v"(ty.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(tz.x), "v"(tz.y));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(nx), "v"(ny), "v"(nz));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(fx), "v"(fy), "v"(fz));
    asm("v_min_f32 %0, %1, %2" : "=v"(tmax) : "v"(tmax), "v"(limit));
    asm("v_max_f32 %0, 0, %1" : "=v"(from) : "v"(tmin));
    return tmax >= from;
}
// what "beyond ray.t" means for a lane: entries whose tmin exceeds this are skipped (appendix F.2)
__device__ __forceinline__ float closest_limit(float t, float tau) { return t >= tau ? t * (1.0f + 0x1p-18f) : __builtin_inff(); }

template <bool SS>
__global__ __launch_bounds__(kWave, CT_CLOSEST_WAVES) void ct_walk_closest_tree(const FrameParams P_launch, const PipelineParams W_launch) {
    FrameParams P = P_launch;
    PipelineParams W = W_launch;
    const uint32_t q_chunk = W.chunk;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float __attribute__((address_space(3))) *waiting_tmin = (float __attribute__((address_space(3))) *)lds_raw;      // [kClosestLdsLevels][64]
    const uint32_t lane = threadIdx.x;
    const uint64_t n_samples = W.n_samples;
    unsigned long long *heads = W.queue;
    const uint64_t shard_span = ((n_samples + kShards - 1) / kShards + q_chunk - 1) / q_chunk * q_chunk;
    uint32_t shard = blockIdx.x % kShards, shards_done = 0;
    d3 eye = make_d3(P.cam[0], P.cam[1], P.cam[2]);
    uint64_t chunk_next = 0, chunk_end = 0;
    bool wave_drained = false;
    uint32_t n_traced = 0, n_left = 0, n_packets = 0, n_quads = 0, n_records = 0;
    const unsigned char *nodes_base = (const unsigned char *)P.stree, *nodes_base64 = nodes_base + 64;
    const unsigned char *tris_base = (const unsigned char *)P.stree_tris, *tris_base64 = tris_base + 64;
    const float tau = P.closest_tau;

    for (;;) {
        asm volatile("" : "+s"(eye.x), "+s"(eye.y), "+s"(eye.z));      // (as in ct_walk_packet<0>: nothing derived from the eye is kept across packets)
        // ---------------- fetch a packet: 64 consecutive sample ids
        if (chunk_next >= chunk_end) {
            if (wave_drained) break;
            unsigned long long offset = 0;
            if (lane == 0) offset = atomicAdd(heads + (size_t)shard * 16u, (unsigned long long)q_chunk);
            offset = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)offset) |
                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(offset >> 32)) << 32);
            const uint64_t lo = (uint64_t)shard * shard_span + offset;
            if (offset >= shard_span || lo >= n_samples) {
                shards_done += 1;
                shard = (shard + 1u) % kShards;
                if (shards_done >= kShards) wave_drained = true;
                continue;
            }
            uint64_t hi = lo + q_chunk;
            if (hi > ((uint64_t)shard + 1u) * shard_span) hi = ((uint64_t)shard + 1u) * shard_span;
            if (hi > n_samples) hi = n_samples;
            chunk_next = lo; chunk_end = hi;
        }
        const uint64_t packet_base = chunk_next;
        const uint64_t avail = chunk_end - chunk_next;
        chunk_next += avail < 64u ? avail : 64u;
        const uint64_t id = packet_base + lane;
        float px = 0.0f, py = 0.0f;
        bool has_ray = lane < avail;
        if (has_ray) {
            has_ray = sample_position<SS>(P, W, id, px, py);
            if (!has_ray) W.hits[id] = make_uint2(__float_as_uint(kNoHit), kInvalidSample);
        }
        if (__ballot(has_ray) == 0ull) continue;
        // ---------------- the rays (raythread.cpp:186-194, 462-508) and the gate
        d3 d = make_d3(0, 0, 1);
        RayLine rl = RayLine{};
        bool left = false;
        if (has_ray) {
            d = primary_direction(P, px, py);
            const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
            rl.inv_xy = f32x2{(float)inv.x, (float)inv.y};
            rl.noi_xy = f32x2{-(float)(eye.x * inv.x), -(float)(eye.y * inv.y)};
            rl.z = f32x2{(float)inv.z, -(float)(eye.z * inv.z)};
            const double d_min = (double)P.stree_lim.d_min, d_max = (double)P.stree_lim.d_max;
            const double ax = __builtin_fabs(d.x), ay = __builtin_fabs(d.y), az = __builtin_fabs(d.z);
            left = !(ax >= d_min && ax <= d_max && ay >= d_min && ay <= d_max && az >= d_min && az <= d_max);      // (the eye itself: checked by the host)
        }
        float t = kNoHit, limit = __builtin_inff();
        float t_other = kNoHit;                   // the least t over every tested triangle but the winner (appendix F.1, promise 3)
        uint32_t best = kNoNode;                  // the winner's place in the reference's order (= its TriRecord slot there)
        uint32_t hit_at = 0u;                     // ... and the byte offset of its record in the tree's own order
        const unsigned long long rays = __ballot(has_ray && !left);
        n_traced += (uint32_t)__popcll(rays);
        if (rays != 0ull) {
        n_packets += 1;
        uint32_t octant = 8u;
        {
            const unsigned long long down_x = __ballot(has_ray && !left && __float_as_int(rl.inv_xy.x) < 0);
            const unsigned long long down_y = __ballot(has_ray && !left && __float_as_int(rl.inv_xy.y) < 0);
            const unsigned long long down_z = __ballot(has_ray && !left && __float_as_int(rl.z.x) < 0);
            if ((down_x == 0ull || down_x == rays) && (down_y == 0ull || down_y == rays) && (down_z == 0ull || down_z == rays))
                octant = (down_x ? 1u : 0u) | (down_y ? 2u : 0u) | (down_z ? 4u : 0u);
        }
        uint32_t cur = 0u;
        unsigned long long mask = rays;
        uint32_t sp = 0;
        int stack_ref = 0, stack_mask_lo = 0, stack_mask_hi = 0;

        auto shared_walk = [&](auto octant_tag) {
        constexpr int OCT = decltype(octant_tag)::value;
        while (mask != 0ull) {
            if (cur & kLeafBit) {
                const uint32_t first = cur & ~(kLeafBit | 15u);
                const bool mine = __builtin_amdgcn_inverse_ballot_w64(mask);
                const uint32_t count = (cur & 15u) + 1u;                     // (a leaf's reference says how many: ct_shadow_tree_build.h)
                n_records += count;
                for (uint32_t k = 0; k < count; k++) {
                    const ScalarTri tr = scalar_load_tri_at(tris_base, tris_base64, first + k * (uint32_t)sizeof(TriRecord));
                    if (mine) {
                        const d3 p1 = make_d3(tr.f64(0), tr.f64(1), tr.f64(2)), e1 = make_d3(tr.f64(3), tr.f64(4), tr.f64(5)),
                                 e2 = make_d3(tr.f64(6), tr.f64(7), tr.f64(8));
                        float found = kNoHit;           // what THIS test would make of ray.t = 1e30f: its own t where it would lower it
                        (void)triangle_test(eye, d, found, p1, e1, e2);
                        const uint32_t place = (uint32_t)tr.b[2];          // (this tree's records carry their slot in the reference's order)
                        // the least t; among equal ones the triangle the reference meets first (bvh.cpp:212: strict <)
                        // (the least over all tested but the winner: a new winner hands its t down, anything else competes
                        //  with its own -- min(t_other, max(found, t)) in both cases, and 1e30f where the test found nothing)
                        t_other = lesser(t_other, greater(found, t));
                        if (found < t || (found == t && found != kNoHit && place < best)) {
                            t = found; best = place; hit_at = first + k * (uint32_t)sizeof(TriRecord);
                            limit = closest_limit(t, tau);
                        }
                    }
                }
                mask = 0ull;
            } else {
                n_quads += 1;
                const ScalarQuad sq = scalar_load_quad_at(nodes_base, nodes_base64, cur);
                unsigned long long enter[4];
                float tmin[4];
                if constexpr (OCT < 8) {
                    node_enters_closest<OCT>(rl, sq, limit, enter, tmin);
#pragma unroll
                    for (int k = 0; k < 4; k++) enter[k] &= mask;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        enter[k] = __ballot(line_enters_closest(rl, sq.planes(k, 0), sq.planes(k, 1), sq.planes(k, 2), limit, tmin[k])) & mask;
                }
                // Front to back: the node's entries are stored in ascending order of their boxes' centres along one axis
                // (ct_shadow_tree_build.h); a packet whose rays point DOWN that axis goes through them backwards.  The entry met
                // first is entered, the others wait, nearest on top.
                uint32_t next_ref = 0;
                unsigned long long next_mask = 0ull;
                float next_tmin = 0.0f;
                const auto meet = [&](int k) {
                    if (enter[k] != 0ull) {
                        if (next_mask != 0ull) {
                            const bool level = lane == sp;
                            stack_ref = level ? (int)next_ref : stack_ref;
                            stack_mask_lo = level ? (int)(uint32_t)next_mask : stack_mask_lo;
                            stack_mask_hi = level ? (int)(uint32_t)(next_mask >> 32) : stack_mask_hi;
                            if (sp < kClosestLdsLevels) waiting_tmin[sp * kWave + lane] = next_tmin;
                            sp += 1;
                        }
                        next_ref = sq.ref(k);
                        next_mask = enter[k];
                        next_tmin = tmin[k];
                    }
                };
                const uint32_t axis = (uint32_t)sq.b[13];
                const bool backwards = OCT < 8 && axis < 3u && (((uint32_t)OCT >> axis) & 1u) != 0u;
                if (!backwards) { meet(3); meet(2); meet(1); meet(0); }
                else { meet(0); meet(1); meet(2); meet(3); }
                cur = next_ref;
                mask = next_mask;
                if (mask != 0ull) continue;
            }
            while (mask == 0ull && sp > 0) {
                sp -= 1;
                const uint32_t pushed_lo = (uint32_t)__builtin_amdgcn_readlane(stack_mask_lo, (int)sp);
                const uint32_t pushed_hi = (uint32_t)__builtin_amdgcn_readlane(stack_mask_hi, (int)sp);
                mask = (unsigned long long)pushed_lo | ((unsigned long long)pushed_hi << 32);
                cur = (uint32_t)__builtin_amdgcn_readlane(stack_ref, (int)sp);
                // a lane whose ray.t has dropped below the entry's tmin since it was pushed need not go in (the levels kept in LDS)
                if (sp < kClosestLdsLevels) mask &= __ballot(!(waiting_tmin[sp * kWave + lane] > limit));
            }
        }
        };
        switch (octant) {
            case 0: shared_walk(std::integral_constant<int, 0>{}); break;
            case 1: shared_walk(std::integral_constant<int, 1>{}); break;
            case 2: shared_walk(std::integral_constant<int, 2>{}); break;
            case 3: shared_walk(std::integral_constant<int, 3>{}); break;
            case 4: shared_walk(std::integral_constant<int, 4>{}); break;
            case 5: shared_walk(std::integral_constant<int, 5>{}); break;
            case 6: shared_walk(std::integral_constant<int, 6>{}); break;
            case 7: shared_walk(std::integral_constant<int, 7>{}); break;
            default: shared_walk(std::integral_constant<int, 8>{}); break;
        }
        // ---------------- the winner's leaf: does the reference test it, whatever it has found before?  Its leaf's box
        // (bvh.cpp:165-179): tmax >= tmin, tmax > 0, and tmin below every ray.t the reference can carry when it comes to that
        // leaf.  That ray.t is the least t_i over triangles it tested BEFORE the leaf: each of them was either tested by this
        // walk too -- t_i >= t_other, the least over all this lane tested but the winner -- or lay beyond the walk's limit when
        // it was skipped: t_i > t (1 + 2^-19) (appendix F.1).  So tmin < min(t_other, t (1 + 2^-19)) settles it; the ancestors
        // follow (nested boxes, they are met with a ray.t no smaller: appendix B2).  A hit on the rim of a flat leaf's box,
        // where tmin and t are two roundings of one number, passes unless another tested triangle sits at (nearly) the same
        // t.  Not certain: the ray is left.
        if (has_ray && !left && t != kNoHit) {
            const ctgpu_node *leaf = P.ref_nodes + P.leaf_of_slot[hit_at / (uint32_t)sizeof(TriRecord)];
            const double4 b0 = *reinterpret_cast<const double4 *>(&leaf->bmin[0]);
            const double2 b1 = *reinterpret_cast<const double2 *>(&leaf->bmax[1]);
            const double bmin[3] = {b0.x, b0.y, b0.z}, bmax[3] = {b0.w, b1.x, b1.y};
            const d3 inv = make_d3(quick_reciprocal(d.x), quick_reciprocal(d.y), quick_reciprocal(d.z));
            float tmin64, tmax64;
            bool verdict;
            const float carried = lesser(t_other, t * (1.0f + 0x1p-19f));
            slab_approx(eye, inv, bmin, bmax, tmin64, tmax64);
            if (!box_verdict_positive_t(tmin64, tmax64, carried, verdict)) {
                (void)slab_in_sequence(eye, d, bmin, bmax, tmin64, tmax64);
                verdict = tmax64 >= tmin64 && tmin64 < carried && tmax64 > 0;
            }
            left = !verdict;
        }
        }   // (a packet with rays inside the gate)
        // ---------------- retire
        {
            const uint32_t gone = (uint32_t)__popcll(__ballot(has_ray && left));
            const uint32_t walked_and_gone = (uint32_t)__popcll(__ballot(has_ray && left) & rays);
            n_left += gone;
            n_traced -= walked_and_gone;
        }
        if (has_ray) W.hits[id] = left ? make_uint2(__float_as_uint(kNoHit), kPrimaryLeft) : make_uint2(__float_as_uint(t), best);
    }
    if (lane == 0) {
        atomicAdd(&P.counters[0], (unsigned long long)n_traced);
        if (n_left) { atomicAdd(W.shadow_left + 1, (unsigned long long)n_left); atomicAdd(&P.counters[kCounterPrimaryLeft], (unsigned long long)n_left); }
        unsigned long long *work = W.walk_work + (size_t)(blockIdx.x & 63u) * 8u;
        atomicAdd(work + 0, (unsigned long long)n_packets);
        atomicAdd(work + 1, (unsigned long long)n_quads);
        atomicAdd(work + 2, (unsigned long long)n_records);
    }
}

// ------------------------------------------------------------------------------------
// ct_shade: lighting of every sample of a pixel in order, running blend, pixel store.
// raythread.cpp:353-366 (TraceRay without reflection), 275-327 (ComputeLighting with the
// shadow decisions read from the mask), 482-498 (blend), color.h.
// ------------------------------------------------------------------------------------
// The shaded point of a sample: what TraceRay derives from the closest hit (raythread.cpp:357-366, 329-350).
struct ShadedPoint {
    d3 position, normal, view;
    Material material;
};
// SPHERES: the frame draws spheres (CTGPU_F_SPHERES), so a hit may be one -- a template parameter because the sphere
// branch (a binary64 square root) cost ct_light_gate a third of its time in frames that have none
// The point a ray (origin, dir) reaches at tclosest on the object `hit` names: a TriRecord slot, kNoNode (nothing was
// recorded: scene triangle 0 at 2^32, the reference's closestIndex default) or a sphere tag.
template <bool SPHERES>
__device__ __forceinline__ ShadedPoint shaded_core(const FrameParams &P, const d3 &origin, const d3 &dir, float t_recorded, uint32_t hit) {
    const float tclosest = (hit != kNoNode) ? t_recorded : kFarAway;
    ShadedPoint s;
    s.position = origin + scale((double)tclosest, dir);
    s.view = -dir;
    if (SPHERES && is_sphere_hit(hit)) {
        // NormalOfSceneObject, OT_SPHERE (raythread.cpp:331-334): n / Magnitude(n) = (1.0f / (D)F|n|) * n
        const ctgpu_sphere &sphere = P.spheres[hit & ~kSphereTag];
        s.material.color = sphere.material.color; s.material.specular = sphere.material.specular; s.material.reflection = sphere.material.reflection;
        const d3 n = s.position - load_d3(sphere.center);
        s.normal = scale((double)1.0f / (double)lengthf(n), n);
        return s;
    }
    const uint32_t rec_slot = (hit != kNoNode) ? hit : P.slot_of_triangle0;
    const TriRecord *rec = P.tris + rec_slot;
    s.material = P.materials[rec->triangle];
    s.normal = cross(load_d3(rec->e1), load_d3(rec->e2));
    if (!(dotf(s.normal, dir) < 0)) s.normal = -s.normal;
    return s;
}
template <bool SS, bool SPHERES>
__device__ __forceinline__ ShadedPoint shaded_point(const FrameParams &P, const PipelineParams &W, uint64_t id, uint2 h,
                                                    uint32_t xi, uint32_t yi) {
    const d3 eye = make_d3(P.cam[0], P.cam[1], P.cam[2]);
    float cx, cy;
    if (SS) { const float2 c = W.coords[id]; cx = c.x; cy = c.y; }
    else { cx = (float)((int)xi - P.half); cy = (float)((int)yi - P.half); }
    return shaded_core<SPHERES>(P, eye, primary_direction(P, cx, cy), __uint_as_float(h.x), h.y);
}
// vector towards a non-ambient light (raythread.cpp:289-296)
__device__ __forceinline__ d3 towards_light(const Light &L, const d3 &position) {
    return (L.type == CTGPU_LIGHT_POINT) ? load_d3(L.position) - position : load_d3(L.direction);
}
// the two quantities ComputeLighting gates a light's contribution on (raythread.cpp:309-325): the light
// adds something only if n.l > 0, or (with a specular exponent) r.v > 0 -- whatever the shadow ray says
__device__ __forceinline__ float diffuse_reach(const ShadedPoint &s, const d3 &to_light) { return dotf(s.normal, to_light); }
__device__ __forceinline__ float specular_reach(const ShadedPoint &s, const d3 &to_light, d3 &reflected) {
    const double nl = (double)dotf(s.normal, to_light);
    reflected = scale(nl, scale(2.0, s.normal)) - to_light;
    return dotf(reflected, s.view);
}

// ComputeLighting (raythread.cpp:275-327) at a shaded point, the shadow decisions read from `skip(k)` for the k-th
// shadow-casting light (the ray found something, or the gate saw that the light adds nothing there)
template <typename Skip>
__device__ __forceinline__ float lighting(const FrameParams &P, const ShadedPoint &s, Skip skip) {
    float intensity = 0.0f;
    uint32_t shadow_slot = 0;
    for (uint32_t li = 0; li < P.n_lights; li++) {
        const Light L = P.lights[li];
        if (L.type == CTGPU_LIGHT_AMBIENT) { intensity += L.intensity; continue; }
        else if (L.type != CTGPU_LIGHT_POINT && L.type != CTGPU_LIGHT_DIRECTIONAL) continue;
        const d3 to_light = towards_light(L, s.position);
        const bool dark = skip(shadow_slot);
        shadow_slot += 1;
        if (dark) continue;
        const float n_dot_l = diffuse_reach(s, to_light);
        if (n_dot_l > 0) intensity += L.intensity * n_dot_l / (lengthf(s.normal) * lengthf(to_light));
        if (s.material.specular != -1) {
            d3 reflected;
            const float r_dot_v = specular_reach(s, to_light, reflected);
            if (r_dot_v > 0) {
                const float base = r_dot_v / (lengthf(reflected) * lengthf(s.view));
                intensity = (float)((double)intensity + (double)L.intensity * pow((double)base, (double)s.material.specular));
            }
        }
    }
    return intensity;
}

// colour of one sample from its closest hit and its shadow flags
template <bool SS, bool SPHERES>
__device__ __forceinline__ uint32_t shade_sample(const FrameParams &P, const PipelineParams &W, uint64_t id, uint32_t xi, uint32_t yi) {
    const uint2 h = W.hits[id];
    if (__uint_as_float(h.x) == kNoHit) return kBackground;
    const ShadedPoint s = shaded_point<SS, SPHERES>(P, W, id, h, xi, yi);
    // 1: the shadow ray found something; 2: ct_light_gate saw that this light adds nothing here
    const float intensity = lighting(P, s, [&](uint32_t k) { return W.shadow[(size_t)k * W.n_samples + id] != 0; });
    return shade_value(s.material.color, intensity);
}

// ------------------------------------------------------------------------------------
// ct_sphere_pass: the scene's spheres (CTGPU_F_SPHERES), after a triangle walk.  The reference renders no spheres since
// it got its BVH (raythread.cpp:207); behind the frame flag they are intersected as its pre-BVH loop did
// (raythread.cpp:230-246, ct_spheres.h) AFTER the triangles, against the closest distance those left -- so the
// triangles stay on the packet walk and the spheres are one lane per ray over their own small tree.
//   KIND 0 (after the primary walk): a sphere nearer than the closest triangle takes the sample over --
//          hits[] = {its distance, kSphereTag | sphere}; camera rays accept t in [1, inf) (raythread.cpp:480).
//   KIND 1 (after the shadow walk): a shadow ray no triangle blocked may still meet a sphere, t in [0.001, 1] towards
//          a point light, [0.001, inf) along a directional one (raythread.cpp:289-307).
// ------------------------------------------------------------------------------------
template <bool SS, int KIND>
__global__ __launch_bounds__(256) void ct_sphere_pass(const FrameParams P, const PipelineParams W) {
    const uint64_t total = (KIND == 0) ? W.n_samples : W.n_samples * (uint64_t)W.n_shadow_lights;
    const d3 eye = make_d3(P.cam[0], P.cam[1], P.cam[2]);
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t id = (KIND == 0) ? i : i % W.n_samples;
        if (KIND == 1 && W.shadow[i] != 0) continue;             // blocked by a triangle already, or no ray here
        const uint2 h = W.hits[id];
        if (h.y == kInvalidSample) continue;
        float px, py;
        if (!sample_position<SS>(P, W, id, px, py)) continue;
        const d3 dir = primary_direction(P, px, py);
        if (KIND == 0) {
            float tclosest = (h.y != kNoNode) ? __uint_as_float(h.x) : kFarAway;
            int sphere = -1;
            walk_spheres(P, eye, dir, 1.0f, kFarAway, tclosest, sphere);
            if (sphere >= 0) W.hits[id] = make_uint2(__float_as_uint(tclosest), kSphereTag | (uint32_t)sphere);
        } else {
            const float tclosest = (h.y != kNoNode) ? __uint_as_float(h.x) : kFarAway;
            const d3 position = eye + scale((double)tclosest, dir);
            const Light L = P.lights[W.shadow_lights[i / W.n_samples]];
            const d3 to_light = (L.type == CTGPU_LIGHT_POINT) ? load_d3(L.position) - position : load_d3(L.direction);
            float nearest = kFarAway;
            int sphere = -1;
            walk_spheres(P, position, to_light, 0.001f, L.type == CTGPU_LIGHT_POINT ? 1.0f : kFarAway, nearest, sphere);
            if (sphere >= 0) W.shadow[i] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------
// ct_light_gate: which shadow rays the frame needs.  ComputeLighting (raythread.cpp:275-327) sends a
// shadow ray to every non-ambient light from every hit and only then looks at n.l and r.v; when
// neither is positive the light adds nothing whether it is blocked or not, so that ray's answer is
// never used.  One lane per sample, between the primary and the shadow walk: shadow[light][sample]
// = 0 where the shadow walk has to find out, 2 where there is nothing to find out (no hit, or a
// light that cannot contribute -- the same float comparisons ct_shade makes afterwards, on the same
// operands).  The shadow walk writes 1 where a light is blocked; ct_shade skips a light on 1 and 2.
// ------------------------------------------------------------------------------------
template <bool SS, bool SPHERES>
__global__ __launch_bounds__(256) void ct_light_gate(const FrameParams P, const PipelineParams W) {
    // (a fixed grid that strides over the samples, one counter atomic per block at the end: one atomic
    // per wave on a single word would take longer than the whole kernel, ~11 ns each)
    __shared__ uint32_t block_elided[4];
    uint32_t elided = 0;
    for (uint64_t id = (uint64_t)blockIdx.x * 256u + threadIdx.x; id < W.n_samples; id += (uint64_t)gridDim.x * 256u) {
        const uint2 h = W.hits[id];
        const bool hit = __uint_as_float(h.x) != kNoHit;         // (samples outside the frame are stored as misses)
        ShadedPoint s = ShadedPoint{};
        if (hit) {
            const PixelOfSample px = pixel_of<SS>(P, id);
            uint32_t yi = 0;
            (void)pixel_in_frame(P, px.xi, px.local_row, yi);
            s = shaded_point<SS, SPHERES>(P, W, id, h, px.xi, yi);
        }
        for (uint32_t k = 0; k < W.n_shadow_lights; k++) {
            uint8_t verdict = 2;
            if (hit && W.keep_every_shadow_ray) verdict = 0;
            else if (hit) {
                const Light L = P.lights[W.shadow_lights[k]];
                const d3 to_light = towards_light(L, s.position);
                bool contributes = diffuse_reach(s, to_light) > 0;
                if (!contributes && s.material.specular != -1) {
                    d3 reflected;
                    contributes = specular_reach(s, to_light, reflected) > 0;
                }
                verdict = contributes ? 0 : 2;
                elided += contributes ? 0u : 1u;
            }
            W.shadow[(size_t)k * W.n_samples + id] = verdict;
        }
    }
    for (int off = 32; off > 0; off >>= 1) elided += __shfl_down(elided, off);
    if ((threadIdx.x & 63u) == 0) block_elided[threadIdx.x >> 6] = elided;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = block_elided[0] + block_elided[1] + block_elided[2] + block_elided[3];
        if (total) atomicAdd(&P.counters[kCounterElided], (unsigned long long)total);
    }
}

template <bool SS, bool SPHERES>
__global__ __launch_bounds__(256) void ct_shade(const FrameParams P, const PipelineParams W) {
    if (!SS) {
        const uint64_t id = (uint64_t)blockIdx.x * 256u + threadIdx.x;
        if (id >= W.n_samples) return;
        const PixelOfSample px = pixel_of<false>(P, id);
        uint32_t yi;
        if (!pixel_in_frame(P, px.xi, px.local_row, yi)) return;
        P.out[(size_t)px.local_row * P.square + px.xi] = shade_sample<false, SPHERES>(P, W, id, px.xi, yi);
        return;
    }
    // Supersampled: first one lane per SAMPLE (consecutive lanes read consecutive hit / position /
    // shadow entries -- whole cache lines), colours parked in LDS; then one lane per PIXEL folds its
    // 16 colours in order with the reference's running blend.
    __shared__ uint32_t colour[256 * 17];                               // row stride 17: conflict-free both ways
    const uint64_t block_first = (uint64_t)blockIdx.x * 4096u;
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t local = j * 256u + threadIdx.x;
        const uint64_t id = block_first + local;
        uint32_t c = kBackground;
        if (id < W.n_samples) {
            const PixelOfSample px = pixel_of<true>(P, id);
            uint32_t yi;
            if (pixel_in_frame(P, px.xi, px.local_row, yi)) c = shade_sample<true, SPHERES>(P, W, id, px.xi, yi);
        }
        colour[(local >> 4) * 17u + (local & 15u)] = c;
    }
    __syncthreads();
    const uint64_t first = block_first + (uint64_t)threadIdx.x * 16u;
    if (first >= W.n_samples) return;
    const PixelOfSample px = pixel_of<true>(P, first);
    uint32_t yi;
    if (!pixel_in_frame(P, px.xi, px.local_row, yi)) return;
    uint32_t pixel_color = colour[threadIdx.x * 17u];
    for (uint32_t s = 1; s < 16; s++) pixel_color = blend_sample(pixel_color, colour[threadIdx.x * 17u + s]);
    P.out[(size_t)px.local_row * P.square + px.xi] = pixel_color;
}

// ------------------------------------------------------------------------------------
// Bounce levels: TraceRay's recursion (raythread.cpp:353-386) as passes over the frame's samples.
//
//   level 0   the camera ray: primary walk, gate, shadow walk as for any frame; then ct_bounce_first shades the
//             sample (its LOCAL colour) and, where the material reflects and depth remains, parks that colour and the
//             reflection factor and emits the reflected ray  {P, ReflectRay(-dir, N), t = 0}  (raythread.cpp:372-373)
//   level k   the list walk (closest hit) over the reflected rays; ct_bounce_gate emits the shadow rays of the points
//             they reach; the list walk (any hit) over those; ct_bounce_shade = the same decision one level down
//   the end   ct_bounce_finish folds the parked colours back, innermost first (raythread.cpp:375-379), and blends the
//             pixel's samples.
// A sample is "live" while it has a reflected ray in flight.  Nothing is compacted: a level's passes run over all
// sample ids and skip the ones that are not live (the list walk forms its packets from 64 consecutive ids).
// With ray.t = 0 a reflected ray "finds" something always (ClosestIntersection reports ray.t != 1e30f): the first
// triangle in walk order whose u, v test passes, at distance 0 -- or scene triangle 0 at 2^32 when none does
// (SURVEY.md 9.8); the list walk takes such rays through its exact level.  CTGPU_F_REFLECT_FIX starts them at 1e30f.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ d3 reflect_ray(const ShadedPoint &s) {          // ReflectRay(-dir, normal), raythread.cpp:268-273
    const double ni = (double)dotf(s.normal, s.view);
    return scale(ni, scale(2.0, s.normal)) - s.view;
}
__device__ __forceinline__ void store_ray(ctgpu_ray *dst, const d3 &o, const d3 &d, float t) {
    ctgpu_ray r;
    r.origin[0] = o.x; r.origin[1] = o.y; r.origin[2] = o.z;
    r.direction[0] = d.x; r.direction[1] = d.y; r.direction[2] = d.z;
    r.t = t; r.pad_ = 0u;
    *dst = r;
}
// what a level does with the colour it computed: reflect on (park colour and factor, emit the ray) or end here
__device__ __forceinline__ void bounce_or_end(const BounceParams &B, uint64_t n_samples, uint64_t id, uint32_t level,
                                              const ShadedPoint &s, uint32_t local) {
    const bool reflects = level < B.max_depth && s.material.reflection > 0;
    {   // how many samples go on to the next level: one atomic per wave (the lanes that are here together)
        const unsigned long long together = __ballot(reflects);
        if (reflects && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)together) - 1))
            atomicAdd(B.level_live + level + 1u, (uint32_t)__popcll(together));
    }
    if (reflects) {
        B.local[(size_t)level * n_samples + id] = local;
        B.refl[(size_t)level * n_samples + id] = s.material.reflection;
        B.pending[id] = (uint8_t)(level + 1u);
        store_ray(B.rays + id, s.position, reflect_ray(s), B.bounce_t);
        B.live[id] = 1;
    } else {
        B.terminal[id] = local;
        B.live[id] = 0;
    }
}

template <bool SS, bool SPHERES>
__global__ __launch_bounds__(256) void ct_bounce_first(const FrameParams P, const PipelineParams W, const BounceParams B) {
    for (uint64_t id = (uint64_t)blockIdx.x * 256u + threadIdx.x; id < W.n_samples; id += (uint64_t)gridDim.x * 256u) {
        const uint2 h = W.hits[id];
        B.pending[id] = 0;
        if (__uint_as_float(h.x) == kNoHit) {                      // a miss, or a sample outside the frame
            B.terminal[id] = kBackground;
            B.live[id] = 0;
            continue;
        }
        const PixelOfSample px = pixel_of<SS>(P, id);
        uint32_t yi = 0;
        (void)pixel_in_frame(P, px.xi, px.local_row, yi);
        const ShadedPoint s = shaded_point<SS, SPHERES>(P, W, id, h, px.xi, yi);
        const float intensity = lighting(P, s, [&](uint32_t k) { return W.shadow[(size_t)k * W.n_samples + id] != 0; });
        bounce_or_end(B, W.n_samples, id, 0u, s, shade_value(s.material.color, intensity));
    }
}

// did the level's reflected ray find anything, and the point it reached
template <bool SPHERES>
__device__ __forceinline__ bool bounce_point(const FrameParams &P, const BounceParams &B, uint64_t id, ShadedPoint &s) {
    const ctgpu_hit h = B.hits[id];
    const bool found = h.t != kNoHit || (SPHERES && is_sphere_hit(h.triangle));      // ClosestIntersection, raythread.cpp:227 (+ spheres)
    if (!found) return false;
    const ctgpu_ray r = B.rays[id];
    s = shaded_core<SPHERES>(P, load_d3(r.origin), load_d3(r.direction), h.tclosest, h.triangle);
    return true;
}

// the shadow rays of the points the level's reflected rays reached (ComputeLighting starts one per non-ambient light;
// where the light cannot contribute the answer is never used: no ray, as in ct_light_gate)
template <bool SPHERES>
__global__ __launch_bounds__(256) void ct_bounce_gate(const FrameParams P, const PipelineParams W, const BounceParams B) {
    if (B.level_live[B.level] == 0u) return;
    __shared__ uint32_t block_elided[4];
    uint32_t elided = 0;
    for (uint64_t id = (uint64_t)blockIdx.x * 256u + threadIdx.x; id < W.n_samples; id += (uint64_t)gridDim.x * 256u) {
        ShadedPoint s = ShadedPoint{};
        const bool hit = B.live[id] && bounce_point<SPHERES>(P, B, id, s);
        for (uint32_t k = 0; k < W.n_shadow_lights; k++) {
            const size_t at = (size_t)k * W.n_samples + id;
            bool needed = false;
            if (hit) {
                const Light L = P.lights[W.shadow_lights[k]];
                const d3 to_light = towards_light(L, s.position);
                needed = W.keep_every_shadow_ray || diffuse_reach(s, to_light) > 0;
                if (!needed && s.material.specular != -1) {
                    d3 reflected;
                    needed = specular_reach(s, to_light, reflected) > 0;
                }
                elided += needed ? 0u : 1u;
                if (needed) {
                    store_ray(B.shadow_rays + at, s.position, to_light, kNoHit);
                    B.shadow_hits[at].found = 0u;
                }
            }
            B.shadow_live[at] = needed ? 1 : 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) elided += __shfl_down(elided, off);
    if ((threadIdx.x & 63u) == 0) block_elided[threadIdx.x >> 6] = elided;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = block_elided[0] + block_elided[1] + block_elided[2] + block_elided[3];
        if (total) atomicAdd(&P.counters[kCounterElided], (unsigned long long)total);
    }
}

template <bool SPHERES>
__global__ __launch_bounds__(256) void ct_bounce_shade(const FrameParams P, const PipelineParams W, const BounceParams B) {
    if (B.level_live[B.level] == 0u) return;
    for (uint64_t id = (uint64_t)blockIdx.x * 256u + threadIdx.x; id < W.n_samples; id += (uint64_t)gridDim.x * 256u) {
        if (!B.live[id]) continue;
        ShadedPoint s;
        if (!bounce_point<SPHERES>(P, B, id, s)) {                 // the reflected ray found nothing: BACKGROUND_COLOR
            B.terminal[id] = kBackground;
            B.live[id] = 0;
            continue;
        }
        // a light is left out when its shadow ray does not exist (it cannot contribute here) or found something
        const float intensity = lighting(P, s, [&](uint32_t k) {
            const size_t at = (size_t)k * W.n_samples + id;
            return B.shadow_live[at] == 0 || B.shadow_hits[at].found != 0u;
        });
        bounce_or_end(B, W.n_samples, id, B.level, s, shade_value(s.material.color, intensity));
    }
}

// spheres at a bounce level (CTGPU_F_SPHERES): reflected rays accept t in [0.001, inf) (raythread.cpp:373), their
// shadow rays [0.001, 1] towards a point light, [0.001, inf) along a directional one
template <int KIND>
__global__ __launch_bounds__(256) void ct_bounce_spheres(const FrameParams P, const PipelineParams W, const BounceParams B) {
    if (B.level_live[B.level] == 0u) return;
    const uint64_t total = (KIND == 0) ? W.n_samples : W.n_samples * (uint64_t)W.n_shadow_lights;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
        if (KIND == 0) {
            if (!B.live[i]) continue;
            const ctgpu_ray r = B.rays[i];
            ctgpu_hit h = B.hits[i];
            float tclosest = (h.triangle != kNoNode) ? h.tclosest : kFarAway;
            int sphere = -1;
            walk_spheres(P, load_d3(r.origin), load_d3(r.direction), 0.001f, kFarAway, tclosest, sphere);
            if (sphere >= 0) {
                h.tclosest = tclosest;
                h.triangle = kSphereTag | (uint32_t)sphere;
                B.hits[i] = h;
            }
        } else {
            if (!B.shadow_live[i] || B.shadow_hits[i].found) continue;
            const ctgpu_ray r = B.shadow_rays[i];
            const Light L = P.lights[W.shadow_lights[i / W.n_samples]];
            float nearest = kFarAway;
            int sphere = -1;
            walk_spheres(P, load_d3(r.origin), load_d3(r.direction), 0.001f, L.type == CTGPU_LIGHT_POINT ? 1.0f : kFarAway, nearest, sphere);
            if (sphere >= 0) B.shadow_hits[i].found = 1u;
        }
    }
}

// the colour TraceRay returns for a sample: the deepest call's colour mixed back through the levels that reflected
__device__ __forceinline__ uint32_t bounce_colour(const BounceParams &B, uint64_t n_samples, uint64_t id) {
    uint32_t c = B.terminal[id];
    for (uint32_t k = B.pending[id]; k > 0; k--)
        c = mix_reflection(B.local[(size_t)(k - 1u) * n_samples + id], c, B.refl[(size_t)(k - 1u) * n_samples + id]);
    return c;
}
// ct_shade's second half with the colours taken from the bounce levels
template <bool SS>
__global__ __launch_bounds__(256) void ct_bounce_finish(const FrameParams P, const PipelineParams W, const BounceParams B) {
    if (!SS) {
        const uint64_t id = (uint64_t)blockIdx.x * 256u + threadIdx.x;
        if (id >= W.n_samples) return;
        const PixelOfSample px = pixel_of<false>(P, id);
        uint32_t yi;
        if (!pixel_in_frame(P, px.xi, px.local_row, yi)) return;
        P.out[(size_t)px.local_row * P.square + px.xi] = bounce_colour(B, W.n_samples, id);
        return;
    }
    __shared__ uint32_t colour[256 * 17];
    const uint64_t block_first = (uint64_t)blockIdx.x * 4096u;
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t local = j * 256u + threadIdx.x;
        const uint64_t id = block_first + local;
        colour[(local >> 4) * 17u + (local & 15u)] = (id < W.n_samples) ? bounce_colour(B, W.n_samples, id) : kBackground;
    }
    __syncthreads();
    const uint64_t first = block_first + (uint64_t)threadIdx.x * 16u;
    if (first >= W.n_samples) return;
    const PixelOfSample px = pixel_of<true>(P, first);
    uint32_t yi;
    if (!pixel_in_frame(P, px.xi, px.local_row, yi)) return;
    uint32_t pixel_color = colour[threadIdx.x * 17u];
    for (uint32_t smp = 1; smp < 16; smp++) pixel_color = blend_sample(pixel_color, colour[threadIdx.x * 17u + smp]);
    P.out[(size_t)px.local_row * P.square + px.xi] = pixel_color;
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
namespace ctd {

uint64_t pipeline_sample_count(const FrameParams &P, bool ss) {
    if (ss) {
        const uint64_t blocks = (uint64_t)((P.square + 1u) / 2u) * ((P.local_rows + 1u) / 2u);
        return blocks * 4u * 16u;
    }
    const uint64_t tiles = (uint64_t)((P.square + 7u) / 8u) * ((P.local_rows + 7u) / 8u);
    return tiles * 64u;
}

int pipeline_walk_occupancy(size_t lds_bytes) {
    int a = 0, b = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, ct_walk<0, true>, kWave, lds_bytes);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ct_walk<1, true>, kWave, lds_bytes);
    return a < b ? a : b;
}

int pipeline_packet_occupancy(uint32_t depth, uint32_t lds_levels) {
    int a = 0, b = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, ct_walk_packet<0, true>, kWave, PacketStack::bytes(depth, lds_levels));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ct_walk_packet<1, true>, kWave, PacketStack::bytes(depth, lds_levels));
    int c = a, e = b;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, ct_walk_packet<0, true, false, false, true>, kWave, PacketStack::bytes(depth, lds_levels));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&e, ct_walk_packet<1, true, false, false, true>, kWave, PacketStack::bytes(depth, lds_levels));
    a = std::min(a, c); b = std::min(b, e);
    return a < b ? a : b;
}

int pipeline_stree_occupancy() {
    int a = 0, b = 0;
    int c = 0, e = 0, f = 0, g = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, ct_walk_closest_tree<true>, kWave, (size_t)kClosestLdsLevels * kWave * sizeof(float));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&g, ct_walk_closest_tree<false>, kWave, (size_t)kClosestLdsLevels * kWave * sizeof(float));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, ct_walk_shadow_tree<true, false>, kWave, 512);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ct_walk_shadow_tree<false, false>, kWave, 512);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, ct_walk_shadow_tree<true, true>, kWave, 512);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&e, ct_walk_shadow_tree<false, true>, kWave, 512);
    return std::min(std::min(std::min(a, b), std::min(c, e)), std::min(f, g));
}

// ct_order_build: the heavy lists of this frame's two packet walks from the costs the frame before left (blockIdx.y:
// the walk).  A chunk is heavy when its packets made `heavy[class]` times the mean packet's fetches (and at least
// `floor`); it goes on its class's list in `parts[class]` pieces and is marked, so that the run over all chunks skips it.
// A list that is full leaves the chunk where it was (and a sentinel in the slots it had drawn, if any lie below the end).
__global__ __launch_bounds__(256) void ct_order_build(const OrderBuild primary, const OrderBuild shadow) {
    const OrderBuild &A = blockIdx.y == 0 ? primary : shadow;
    // last frame's totals: [64 slots][2] -> {fetches, packets} (lanes of equal parity add up; two waves hold the 128 words)
    __shared__ unsigned long long sums[2][2];
    if (threadIdx.x < 128) {
        unsigned long long v = A.prev_totals[threadIdx.x];
        for (int off = 2; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63u) < 2u) sums[threadIdx.x >> 6][threadIdx.x & 1u] = v;
    }
    __syncthreads();
    const unsigned long long fetches = sums[0][0] + sums[1][0], packets = sums[0][1] + sums[1][1];
    // (what the walks of THIS frame add to starts from zero; so do the list counters of the next frame's build)
    if (blockIdx.x == 0 && threadIdx.x < 128) A.totals_out[threadIdx.x] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x < kOrderClasses) A.next_count[threadIdx.x] = 0u;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    const bool there = c < A.chunks;
    if (there) A.cost_out[c] = 0u;
    const float mean = packets ? (float)fetches / (float)packets : 0.0f;
    const uint32_t cost = there ? A.prev_cost[c] : 0u;
    int cls = kOrderClasses;
    if (there && mean > 0.0f && cost >= A.floor)
        for (cls = 0; cls < kOrderClasses; cls++)
            if ((float)cost >= A.heavy[cls] * mean) break;
    uint8_t listed = 0;
    // (list slots are drawn once per wave and class -- thousands of single atomics on one word would take longer than
    //  the walk -- and the four classes' draws go out together, from lanes 0..3)
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long m_mine = 0ull, m_lane = 0ull;
    for (int k = 0; k < kOrderClasses; k++) {
        const unsigned long long m = __ballot(cls == k);
        if (cls == k) m_mine = m;
        if (lane == (uint32_t)k) m_lane = m;
    }
    uint32_t drawn = 0;
    if (lane < (uint32_t)kOrderClasses && m_lane != 0ull) drawn = atomicAdd(A.count + lane, (uint32_t)__popcll(m_lane) * A.parts[lane]);
    const uint32_t base = __shfl(drawn, cls < kOrderClasses ? cls : 0);
    const uint32_t at = base + (uint32_t)__popcll(m_mine & ((1ull << lane) - 1ull)) * (cls < kOrderClasses ? A.parts[cls] : 0u);
    if (cls < kOrderClasses) {
        const uint32_t parts = A.parts[cls];
        uint4 *list = A.list + (size_t)cls * A.cap;
        if ((uint64_t)at + parts <= A.cap) {
            const uint32_t light = c / A.chunks_per_light, within = c - light * A.chunks_per_light;
            const uint32_t ids = (1u << A.chunk_shift) / parts, first = within << A.chunk_shift;
            for (uint32_t k = 0; k < parts; k++) list[at + k] = make_uint4(first + k * ids, ids, light, c);
            listed = 1;
        } else {
            for (uint32_t k = 0; k < parts && (uint64_t)at + k < A.cap; k++) list[at + k] = make_uint4(~0u, 0u, 0u, 0u);
        }
    }
    if (there) A.listed[c] = listed;
}

// the frame's packet walk of one kind: the instantiation for its sampling, its order, its tree (with or without cull nodes)
template <int KIND>
static void launch_frame_walk(bool ss, bool handover, bool ordered, bool cull, dim3 grid, dim3 wave, size_t lds, hipStream_t stream,
                              const FrameParams &P, const PipelineParams &W) {
#define CT_WALK(SS_, HO_, ORD_, CULL_) hipLaunchKernelGGL((ct_walk_packet<KIND, SS_, HO_, ORD_, CULL_>), grid, wave, lds, stream, P, W)
    if (handover) {           // (the hand-over experiment always carries the cull code: one instantiation per sampling)
        if (ss) CT_WALK(true, true, false, true); else CT_WALK(false, true, false, true);
        return;
    }
    if (ss) {
        if (ordered) { if (cull) CT_WALK(true, false, true, true); else CT_WALK(true, false, true, false); }
        else { if (cull) CT_WALK(true, false, false, true); else CT_WALK(true, false, false, false); }
    } else {
        if (ordered) { if (cull) CT_WALK(false, false, true, true); else CT_WALK(false, false, true, false); }
        else { if (cull) CT_WALK(false, false, false, true); else CT_WALK(false, false, false, false); }
    }
#undef CT_WALK
}

void launch_pipeline(const FrameParams &P, const PipelineParams &W, bool ss, hipStream_t stream, hipEvent_t *marks,
                     const BounceParams *bounce, const OrderBuild *order) {
    if (W.n_samples == 0) return;
    // marks (optional, kPipelineKernels + 1 events): one before each kernel and one after the last, so that the
    // time between marks k and k+1 is kernel k's (ctgpu_kernel_timing)
    auto mark = [&](int k) { if (marks) (void)hipEventRecord(marks[k], stream); };
    const uint32_t pixel_slots = (uint32_t)(ss ? W.n_samples / 16u : W.n_samples);
    const uint32_t pixel_blocks = (pixel_slots + 255u) / 256u;
    const uint32_t sample_blocks = (uint32_t)std::min<uint64_t>((W.n_samples + 255u) / 256u, 2048u);      // ct_light_gate strides
    const bool shadows = W.n_shadow_lights != 0;
    // the spheres' own passes (CTGPU_F_SPHERES frames of scenes that have any): their time goes to the walk before them
    const bool sph = P.n_spheres != 0;
    auto light_gate = [&]() {
        if (ss && !sph) hipLaunchKernelGGL((ct_light_gate<true, false>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
        if (!ss && !sph) hipLaunchKernelGGL((ct_light_gate<false, false>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
        if (ss && sph) hipLaunchKernelGGL((ct_light_gate<true, true>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
        if (!ss && sph) hipLaunchKernelGGL((ct_light_gate<false, true>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
    };
    auto sphere_pass = [&](int kind) {
        if (!P.n_spheres) return;
        if (kind == 0 && ss) hipLaunchKernelGGL((ct_sphere_pass<true, 0>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
        if (kind == 0 && !ss) hipLaunchKernelGGL((ct_sphere_pass<false, 0>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
        if (kind == 1 && ss) hipLaunchKernelGGL((ct_sphere_pass<true, 1>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
        if (kind == 1 && !ss) hipLaunchKernelGGL((ct_sphere_pass<false, 1>), dim3(sample_blocks), dim3(256), 0, stream, P, W);
    };
    mark(0);
    if (ss) hipLaunchKernelGGL(ct_sample_grid, dim3(pixel_blocks), dim3(256), 0, stream, P, W);
    const bool ordered = W.ordered != 0u && order != nullptr;
    if (ordered) {      // (booked on ct_sample_grid's mark)
        const uint32_t most = std::max(order[0].chunks, shadows ? order[1].chunks : 0u);
        hipLaunchKernelGGL(ct_order_build, dim3((most + 255u) / 256u, shadows ? 2u : 1u), dim3(256), 0, stream, order[0], order[1]);
    }
    mark(1);
    const dim3 grid(W.walk_blocks), wave(kWave);
    if (W.packet) {
        // shared walk: a wave holds 2x2 pixels x 16 samples (supersampled) or an 8x8 pixel tile
        const uint32_t depth_p = W.packet_depth;
        const size_t lds_p = PacketStack::bytes(depth_p, W.packet_lds_levels < depth_p ? W.packet_lds_levels : depth_p);
        // (the per-lane kernel behind each packet launch traces the rays of packets that had stopped sharing their
        //  walk: a short list, usually empty for supersampled frames -- its waves then leave at once)
        const size_t lds_l = (size_t)W.lds_depth * kWave * 12u;
        const dim3 grid_l(W.listed_blocks);
        const bool handover = W.packet_budget != 0u;
        if (P.closest_tree && P.stree && !handover && !ordered) {
            // the primary rays through the shadow tree's nodes (appendix F); what that walk leaves -- rays outside the gate, winners
            // on the rim of their leaf's box: hits[].slot = kPrimaryLeft, counted in W.shadow_left[1] -- goes through the
            // reference's tree in a launch whose waves return at once on zero
            const dim3 grid_s(W.stree_blocks ? W.stree_blocks : W.walk_blocks);
            const size_t lds_c = (size_t)kClosestLdsLevels * kWave * sizeof(float);
            if (ss) hipLaunchKernelGGL((ct_walk_closest_tree<true>), grid_s, wave, lds_c, stream, P, W);
            else hipLaunchKernelGGL((ct_walk_closest_tree<false>), grid_s, wave, lds_c, stream, P, W);
            PipelineParams Wl = W;
            Wl.primary_left = 1u;
            launch_frame_walk<0>(ss, false, false, P.cull_first != kNoNode, grid, wave, lds_p, stream, P, Wl);
        } else
        launch_frame_walk<0>(ss, handover, ordered, P.cull_first != kNoNode, grid, wave, lds_p, stream, P, W);
        if (handover && ss) hipLaunchKernelGGL((ct_walk<0, true, true>), grid_l, wave, lds_l, stream, P, W);
        if (handover && !ss) hipLaunchKernelGGL((ct_walk<0, false, true>), grid_l, wave, lds_l, stream, P, W);
        sphere_pass(0);
        mark(2);
        if (shadows) light_gate();
        mark(3);
        if (shadows && P.stree && !handover) {
            // the shadow tree takes the frame's shadow rays; what it leaves (rays outside its gate: shadow byte kShadowLeft,
            // counted in *W.shadow_left) goes through the reference's tree in a launch whose waves return at once on zero
            const dim3 grid_s(W.stree_blocks ? W.stree_blocks : W.walk_blocks);
            if (ss && ordered) hipLaunchKernelGGL((ct_walk_shadow_tree<true, true>), grid_s, wave, 512, stream, P, W);
            if (ss && !ordered) hipLaunchKernelGGL((ct_walk_shadow_tree<true, false>), grid_s, wave, 512, stream, P, W);
            if (!ss && ordered) hipLaunchKernelGGL((ct_walk_shadow_tree<false, true>), grid_s, wave, 512, stream, P, W);
            if (!ss && !ordered) hipLaunchKernelGGL((ct_walk_shadow_tree<false, false>), grid_s, wave, 512, stream, P, W);
            PipelineParams Wl = W;
            Wl.shadow_live = kShadowLeft;
            launch_frame_walk<1>(ss, false, false, P.cull_first != kNoNode, grid, wave, lds_p, stream, P, Wl);
        } else
        if (shadows) launch_frame_walk<1>(ss, handover, ordered, P.cull_first != kNoNode, grid, wave, lds_p, stream, P, W);
        if (shadows && handover && ss) hipLaunchKernelGGL((ct_walk<1, true, true>), grid_l, wave, lds_l, stream, P, W);
        if (shadows && handover && !ss) hipLaunchKernelGGL((ct_walk<1, false, true>), grid_l, wave, lds_l, stream, P, W);
    } else {
        const size_t lds = (size_t)W.lds_depth * kWave * 12u;      // entries + parent words
        if (ss) hipLaunchKernelGGL((ct_walk<0, true>), grid, wave, lds, stream, P, W);
        else hipLaunchKernelGGL((ct_walk<0, false>), grid, wave, lds, stream, P, W);
        sphere_pass(0);
        mark(2);
        if (shadows) light_gate();
        mark(3);
        if (shadows && ss) hipLaunchKernelGGL((ct_walk<1, true>), grid, wave, lds, stream, P, W);
        if (shadows && !ss) hipLaunchKernelGGL((ct_walk<1, false>), grid, wave, lds, stream, P, W);
    }
    if (shadows) sphere_pass(1);
    mark(4);
    if (!bounce) {
        if (ss && !sph) hipLaunchKernelGGL((ct_shade<true, false>), dim3(pixel_blocks), dim3(256), 0, stream, P, W);
        if (!ss && !sph) hipLaunchKernelGGL((ct_shade<false, false>), dim3(pixel_blocks), dim3(256), 0, stream, P, W);
        if (ss && sph) hipLaunchKernelGGL((ct_shade<true, true>), dim3(pixel_blocks), dim3(256), 0, stream, P, W);
        if (!ss && sph) hipLaunchKernelGGL((ct_shade<false, true>), dim3(pixel_blocks), dim3(256), 0, stream, P, W);
        mark(5);
        return;
    }

    // ---- reflective materials: the bounce levels (their time is booked on ct_shade's mark)
    BounceParams B = *bounce;
    const dim3 strided(sample_blocks), threads(256);
    (void)hipMemsetAsync(B.level_live, 0, (B.max_depth + 2u) * sizeof(uint32_t), stream);
    if (ss && !sph) hipLaunchKernelGGL((ct_bounce_first<true, false>), strided, threads, 0, stream, P, W, B);
    if (!ss && !sph) hipLaunchKernelGGL((ct_bounce_first<false, false>), strided, threads, 0, stream, P, W, B);
    if (ss && sph) hipLaunchKernelGGL((ct_bounce_first<true, true>), strided, threads, 0, stream, P, W, B);
    if (!ss && sph) hipLaunchKernelGGL((ct_bounce_first<false, true>), strided, threads, 0, stream, P, W, B);
    // the list walks: the packet walk's caller-list kinds over the frame's own ray arrays, 64 consecutive ids a packet
    const uint32_t depth_p = W.packet_depth;
    const size_t lds_p = PacketStack::bytes(depth_p, W.packet_lds_levels < depth_p ? W.packet_lds_levels : depth_p);
    const size_t queue_bytes = queue_block_words(W.n_shadow_lights) * sizeof(unsigned long long);
    PipelineParams Wc = W;                       // closest hit over the reflected rays
    Wc.ray_list = B.rays; Wc.ray_hits = B.hits; Wc.ray_live = B.live; Wc.list_slots = 1u; Wc.list_counter = 2u;
    PipelineParams Ws = W;                       // any hit over their shadow rays, [light][sample]
    Ws.ray_list = B.shadow_rays; Ws.ray_hits = B.shadow_hits; Ws.ray_live = B.shadow_live; Ws.list_slots = 1u; Ws.list_counter = 1u;
    Ws.n_samples = W.n_samples * (uint64_t)W.n_shadow_lights;
    const auto list_grid = [&](uint64_t n) { return dim3((uint32_t)std::min<uint64_t>(W.walk_blocks, std::max<uint64_t>((n + 63u) / 64u, 1u))); };
    for (uint32_t level = 1; level <= B.max_depth; level++) {
        B.level = level;
        Wc.list_gate = Ws.list_gate = B.level_live + level;
        (void)hipMemsetAsync(W.queue, 0, queue_bytes, stream);
        hipLaunchKernelGGL((ct_walk_packet<2, false, false, false, true>), list_grid(Wc.n_samples), wave, lds_p, stream, P, Wc);
        if (sph) hipLaunchKernelGGL((ct_bounce_spheres<0>), strided, threads, 0, stream, P, W, B);
        if (shadows) {
            if (sph) hipLaunchKernelGGL((ct_bounce_gate<true>), strided, threads, 0, stream, P, W, B);
            else hipLaunchKernelGGL((ct_bounce_gate<false>), strided, threads, 0, stream, P, W, B);
            (void)hipMemsetAsync(W.queue, 0, queue_bytes, stream);
            hipLaunchKernelGGL((ct_walk_packet<3, false, false, false, true>), list_grid(Ws.n_samples), wave, lds_p, stream, P, Ws);
            if (sph) hipLaunchKernelGGL((ct_bounce_spheres<1>), strided, threads, 0, stream, P, W, B);
        }
        if (sph) hipLaunchKernelGGL((ct_bounce_shade<true>), strided, threads, 0, stream, P, W, B);
        else hipLaunchKernelGGL((ct_bounce_shade<false>), strided, threads, 0, stream, P, W, B);
    }
    if (ss) hipLaunchKernelGGL((ct_bounce_finish<true>), dim3(pixel_blocks), dim3(256), 0, stream, P, W, B);
    else hipLaunchKernelGGL((ct_bounce_finish<false>), dim3(pixel_blocks), dim3(256), 0, stream, P, W, B);
    mark(5);
}

// the per-ray hook of the shared walk: W.ray_list[0 .. n_samples) -> W.ray_hits, 64 consecutive rays a packet
void launch_packet_ray_list(const FrameParams &P, const PipelineParams &W, bool any_hit, hipStream_t stream) {
    if (W.n_samples == 0) return;
    const uint32_t depth_p = W.packet_depth;
    const size_t lds_p = PacketStack::bytes(depth_p, W.packet_lds_levels < depth_p ? W.packet_lds_levels : depth_p);
    if (any_hit) hipLaunchKernelGGL((ct_walk_packet<3, false, false, false, true>), dim3(W.walk_blocks), dim3(kWave), lds_p, stream, P, W);
    else hipLaunchKernelGGL((ct_walk_packet<2, false, false, false, true>), dim3(W.walk_blocks), dim3(kWave), lds_p, stream, P, W);
}

}  // namespace ctd
