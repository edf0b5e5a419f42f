// ct_kernels.hip -- gfx950 kernels of the CobbleTrace hot path.
//
// One wave64 = one 8x8 pixel tile; one lane = one pixel (all 16 supersamples of a pixel
// stay in their lane so that the order-dependent running blend of raythread.cpp:482-498
// is sequential).  A lane is a small state machine around ONE traversal loop:
//
//     primary ray -> [hit] -> shadow ray per non-ambient light -> shade
//                 -> (reflection bounce)* -> blend sample -> next sample / next pixel
//
// so primary, shadow and reflection rays of different lanes share the same traversal
// instructions instead of serialising in three copies of the loop.  Divergence is
// handled with wave ballots: the three phases of the loop body
//     (A) open an inner node   (B) test one leaf triangle   (C) ray finished -> next ray
// run only when enough lanes wait for them (or nothing else can run), so each phase
// executes with a well-filled exec mask.
//
// The per-lane traversal stack lives in LDS ([depth][64] x {child ref, F tmin}); the
// depth is sized by the host from the actual tree.
//
// Reference semantics implemented here: raythread.cpp:437-543 (pixel loop, supersample),
// 353-386 (TraceRay), 197-228 (ClosestIntersection), 275-350 (lighting, normal),
// bvh.cpp:147-222 (triangle, slab, DFS order: left subtree before right, leaf
// triangles in index order, strict '<' for the closest hit).
#include "ct_device.h"
#include "ct_params.h"
#include "ct_tiles.h"
#include "ct_raygen.h"
#include "ct_spheres.h"

using namespace ctd;

namespace {

constexpr int kWave = 64;

// phase of a lane
enum : uint32_t { PH_CLOSEST = 0, PH_SHADOW = 1, PH_IDLE = 2 };

struct LaneRay {
    d3 o, d;
    d3 inv;           // RN(1/d) per component, for the division-free slab test
    bool fast;        // direction_allows_fast_slab(d)
    float t;          // ray.t
    float tclosest;   // *tclosest
    uint32_t slot;    // TriRecord slot of closestIndex (kNoNode = none recorded -> triangle 0)
    uint32_t cur;     // node to process next (kNoNode = none; traversal over when stack empty too)
    uint32_t sp;      // stack entries in use
    uint32_t leaf_left;   // triangles still to test in the current leaf (cur is then a slot)
};

// LDS stack: entry e of lane l at [e*64 + l]
struct StackEntry { uint32_t ref; float tmin; };

// both paths give the reference's floats; the fast one falls back when it cannot prove it
__device__ __forceinline__ void box_test(const LaneRay &r, const double *bmin, const double *bmax, float &tmin,
                                         float &tmax) {
    if (!r.fast || slab_fast(r.o, r.inv, bmin, bmax, tmin, tmax)) slab(r.o, r.d, bmin, bmax, tmin, tmax);
}

__device__ __forceinline__ void start_traversal(LaneRay &r, const FrameParams &P) {
    // (COUNT: callers book the root's box test)
    r.fast = direction_allows_fast_slab(r.d);
    r.inv = make_d3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    // root box test (bvh.cpp:205 at nodeIdx = rootNodeIdx)
    float tmin, tmax;
    box_test(r, P.root_min, P.root_max, tmin, tmax);
    const bool pass = tmax >= tmin && tmin < r.t && tmax > 0;
    r.cur = pass ? P.root_ref : kNoNode;
    r.sp = 0;
    r.leaf_left = 0;
    if (pass && (P.root_ref & kLeafBit)) {      // a tree that is a single leaf (<= 2 triangles, or unsplittable)
        r.cur = P.root_ref & ~kLeafBit;
        r.leaf_left = 0xffffffffu;               // filled from the first record
    }
    r.tclosest = kFarAway;
    r.slot = kNoNode;
}

template <bool COUNT>
struct Counters {
    uint32_t primary = 0, shadow = 0, reflect = 0, visits = 0, tests = 0;
};

// (A) open one inner node: both child boxes, DFS order left then right.
// COUNT mode books one "visit" per box test at the point where the reference performs it
// (left: now; right: now if the left child is skipped, otherwise when it is popped), which
// makes the totals equal to the oracle's with shadow_early_exit = 1.
template <bool COUNT>
__device__ __forceinline__ void step_inner(LaneRay &r, const WideNode *__restrict__ nodes, StackEntry *stack,
                                           uint32_t lane, uint32_t stack_depth, uint32_t *overflow,
                                           Counters<COUNT> &cnt) {
    const WideNode *n = nodes + r.cur;
    // one 128-byte line: 3 x 32 B of boxes + 8 B of child references
    const double4 a0 = *reinterpret_cast<const double4 *>(&n->lmin[0]);   // lmin xyz, lmax x
    const double4 a1 = *reinterpret_cast<const double4 *>(&n->lmax[1]);   // lmax yz, rmin xy
    const double4 a2 = *reinterpret_cast<const double4 *>(&n->rmin[2]);   // rmin z, rmax xyz
    const uint2 refs = *reinterpret_cast<const uint2 *>(&n->lref);
    const double lmin[3] = {a0.x, a0.y, a0.z}, lmax[3] = {a0.w, a1.x, a1.y};
    const double rmin[3] = {a1.z, a1.w, a2.x}, rmax[3] = {a2.y, a2.z, a2.w};

    float ltmin, ltmax, rtmin, rtmax;
    bool redo = !r.fast;
    if (!redo) {
        redo = slab_fast(r.o, r.inv, lmin, lmax, ltmin, ltmax);
        redo |= slab_fast(r.o, r.inv, rmin, rmax, rtmin, rtmax);
    }
    if (redo) {     // rare: a quotient too close to a binary32 rounding boundary (or an exotic direction)
        slab(r.o, r.d, lmin, lmax, ltmin, ltmax);
        slab(r.o, r.d, rmin, rmax, rtmin, rtmax);
    }
    const bool lpass = ltmax >= ltmin && ltmin < r.t && ltmax > 0;
    // ray.t can only shrink while the left subtree is walked, so a right child failing
    // `tmin < ray.t` now fails it later as well; the survivors are re-checked when popped.
    const bool rpass = rtmax >= rtmin && rtmin < r.t && rtmax > 0;
    if (COUNT) cnt.visits += 1;           // left child's box test

    if (lpass) {
        r.cur = refs.x;
        if (rpass || COUNT) {
            if (r.sp < stack_depth) {
                // COUNT keeps a failed right child on the stack with tmin = +inf so that its
                // box test is booked at the point the reference reaches it.
                stack[r.sp * kWave + lane] = StackEntry{refs.y, rpass ? rtmin : __builtin_inff()};
                r.sp += 1;
            } else {
                atomicOr(overflow, 1u);
            }
        }
    } else {
        if (COUNT) cnt.visits += 1;       // right child's box test happens right away
        r.cur = rpass ? refs.y : kNoNode;
    }
    if ((r.cur != kNoNode) && (r.cur & kLeafBit)) {
        r.cur &= ~kLeafBit;
        r.leaf_left = 0xffffffffu;        // filled from the first record
    }
}

// pop until an entry survives `tmin < ray.t` (the deferred part of the right child's box test)
template <bool COUNT>
__device__ __forceinline__ void pop_next(LaneRay &r, const StackEntry *stack, uint32_t lane, Counters<COUNT> &cnt) {
    r.cur = kNoNode;
    r.leaf_left = 0;
    while (r.sp > 0) {
        r.sp -= 1;
        const StackEntry e = stack[r.sp * kWave + lane];
        if (COUNT) cnt.visits += 1;       // this is where the reference runs the right child's box test
        if (e.tmin < r.t) {
            r.cur = e.ref;
            if (r.cur & kLeafBit) {
                r.cur &= ~kLeafBit;
                r.leaf_left = 0xffffffffu;
            }
            return;
        }
    }
}

// (B) one triangle of the current leaf.  ANY_HIT: shadow-ray semantics, the traversal may
// stop as soon as ray.t has moved (raythread.cpp:303-307 only asks ray.t != 1e30f).
template <bool COUNT>
__device__ __forceinline__ void step_leaf(LaneRay &r, const TriRecord *__restrict__ tris, bool any_hit,
                                          Counters<COUNT> &cnt) {
    const TriRecord *rec = tris + r.cur;
    // five 16-byte loads (the record is 16-byte aligned, 80 bytes)
    const double2 b0 = *reinterpret_cast<const double2 *>(&rec->p1[0]);   // p1 xy
    const double2 b1 = *reinterpret_cast<const double2 *>(&rec->p1[2]);   // p1 z, e1 x
    const double2 b2 = *reinterpret_cast<const double2 *>(&rec->e1[1]);   // e1 yz
    const double2 b3 = *reinterpret_cast<const double2 *>(&rec->e2[0]);   // e2 xy
    const double2 b4 = *reinterpret_cast<const double2 *>(&rec->e2[2]);   // e2 z, {triangle, remaining}
    const d3 p1 = make_d3(b0.x, b0.y, b1.x), e1 = make_d3(b1.y, b2.x, b2.y), e2 = make_d3(b3.x, b3.y, b4.x);
    const uint2 tail = make_uint2((uint32_t)__double_as_longlong(b4.y), (uint32_t)(__double_as_longlong(b4.y) >> 32));
    if (r.leaf_left == 0xffffffffu) r.leaf_left = tail.y;
    if (COUNT) cnt.tests += 1;

    const bool hit = triangle_test(r.o, r.d, r.t, p1, e1, e2);
    if (hit && r.t != kNoHit && r.t < r.tclosest) {     // bvh.cpp:212-215
        r.slot = r.cur;
        r.tclosest = r.t;
    }
    r.leaf_left -= 1;
    r.cur += 1;
    if (any_hit && r.t != kNoHit) {        // shadow ray decided: drop the rest of the walk
        r.leaf_left = 0;
        r.sp = 0;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------
// Render kernel.  SS: 16 jittered samples per pixel; REFLECT: materials with reflection>0
// exist (keeps a per-lane colour stack in scratch); COUNT: work counters for the roofline.
// ------------------------------------------------------------------------------------
template <bool SS, bool REFLECT, bool COUNT>
__global__ __launch_bounds__(kWave) void ct_trace_tiles(const FrameParams P) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    StackEntry *stack = reinterpret_cast<StackEntry *>(lds_raw);

    const uint32_t lane = threadIdx.x;
    // tile -> pixel: 8 columns x 8 local rows
    const uint32_t tiles_x = (P.square + 7u) >> 3;
    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t xi = tx * 8u + (lane & 7u);
    const uint32_t local_row = ty * 8u + (lane >> 3);
    // local row -> frame row (interleaved row tiles across partitions)
    const uint32_t local_tile = local_row / P.tile_rows;
    const uint32_t yi = (local_tile * P.part_count + P.part_index) * P.tile_rows + (local_row % P.tile_rows);

    bool alive = xi < P.square && local_row < P.local_rows && yi < P.rendered_rows;
    if (alive && P.subsample) {
        // rows a band actually traces (raythread.cpp:526-535): even offsets and the band's last row
        const uint32_t off = yi % P.rows_per_band;
        alive = ((off & 1u) == 0u) || (off == P.rows_per_band - 1u);
    }
    if (alive) {
        // a pixel that lands outside the bitmap is traced by the reference but its store is
        // dropped (draw2d.h:11-14); it has no observable effect, so it is not traced here.
        // (With subsampling the dropped row still feeds the average stored one row above it,
        // raythread.cpp:517-524, so only its columns may be clipped.)
        const int row = P.height / 2 - ((int)yi - P.half), col = (int)xi - P.half + P.width / 2;
        alive = col >= 0 && col < P.width && (P.subsample || (row >= 0 && row < P.height));
    }

    const float fx = (float)((int)xi - P.half), fy = (float)((int)yi - P.half);
    const d3 eye = make_d3(P.cam[0], P.cam[1], P.cam[2]);

    Counters<COUNT> cnt;
    LaneRay r;
    r.cur = kNoNode; r.sp = 0; r.leaf_left = 0; r.t = kNoHit; r.tclosest = kFarAway; r.slot = kNoNode;
    r.o = eye; r.d = make_d3(0, 0, 1); r.inv = make_d3(0, 0, 1); r.fast = false;

    // sample iteration state (raythread.cpp:462-504)
    uint32_t sample = 0;
    float run_x = fx, run_y = fy;      // sampleX / sampleY with their float residue
    float jx = 0.0f, jy = 0.0f;        // rx, ry of the sample in flight
    uint32_t pixel_color = 0;

    // shading state of the hit being lit
    uint32_t phase = PH_IDLE;
    d3 normal = make_d3(0, 0, 0), view = make_d3(0, 0, 0);
    float intensity = 0.0f;
    uint32_t light = 0;
    uint32_t mat_color = 0; int32_t mat_specular = -1; float mat_reflection = 0.0f;
    // reflection bookkeeping (REFLECT only)
    int32_t depth_left = 0;
    uint32_t local_stack[REFLECT ? 12 : 1];
    float refl_stack[REFLECT ? 12 : 1];
    uint32_t n_pending = 0;

    // launch the first primary ray
    auto launch_primary = [&]() {
        float px = fx, py = fy;
        if (SS) {
            const float jitter = 0.25f / 8.0f;
            jx = (float)jitter_int(P, xi, yi, sample, 0) / 2147483648.0f;
            jy = (float)jitter_int(P, xi, yi, sample, 1) / 2147483648.0f;
            jx = (jx * (2.0f * jitter)) - jitter;
            jy = (jy * (2.0f * jitter)) - jitter;
            run_x += jx;
            run_y += jy;
            px = run_x; py = run_y;
        }
        r.o = eye;
        r.d = primary_direction(P, px, py);
        r.t = kNoHit;
        start_traversal(r, P);
        if (COUNT) cnt.visits += 1;
        phase = PH_CLOSEST;
        depth_left = P.max_depth;
        n_pending = 0;
        cnt.primary += 1;
    };

    if (alive) launch_primary();

    for (;;) {
        const bool busy = alive && phase != PH_IDLE;
        const bool want_c = busy && r.cur == kNoNode;                   // traversal finished
        const bool want_b = busy && r.cur != kNoNode && r.leaf_left != 0;
        const bool want_a = busy && r.cur != kNoNode && r.leaf_left == 0;
        const unsigned long long mask_busy = __ballot(busy);
        if (mask_busy == 0ull) break;
        const unsigned long long mask_a = __ballot(want_a), mask_b = __ballot(want_b), mask_c = __ballot(want_c);
        const int n_a = __popcll(mask_a), n_b = __popcll(mask_b), n_c = __popcll(mask_c);

        // Which phase runs this iteration: the first one whose queue reached its threshold,
        // otherwise the fullest queue.  Every busy lane is in exactly one queue, so the
        // loop always makes progress.
        int pick;
        if (n_a >= P.thresh_a) pick = 0;
        else if (n_b >= P.thresh_b) pick = 1;
        else if (n_c >= P.thresh_c) pick = 2;
        else pick = (n_a >= n_b && n_a >= n_c) ? 0 : ((n_b >= n_c) ? 1 : 2);

        // ---------------- (A) inner nodes
        if (pick == 0) {
            if (want_a) {
                step_inner<COUNT>(r, P.nodes, stack, lane, P.stack_depth, P.overflow, cnt);
                if (r.cur == kNoNode) pop_next<COUNT>(r, stack, lane, cnt);
            }
            continue;
        }
        // ---------------- (B) leaf triangles
        if (pick == 1) {
            if (want_b) {
                step_leaf<COUNT>(r, P.tris, phase == PH_SHADOW, cnt);
                if (r.leaf_left == 0) pop_next<COUNT>(r, stack, lane, cnt);
            }
            continue;
        }
        // ---------------- (C) finished rays: shade / next ray
        if (want_c) {
            bool color_ready = false;
            uint32_t color = kBackground;
            bool go_lights = false;

            if (phase == PH_CLOSEST) {
                int sphere = -1;
                if (P.n_spheres)      // camera rays: TraceRay(.., 1, FINF, ..); reflected ones: (.., 0.001, FINF, ..)   raythread.cpp:480, 373
                    walk_spheres(P, r.o, r.d, n_pending > 0 ? 0.001f : 1.0f, kFarAway, r.tclosest, sphere);
                if (r.t != kNoHit || sphere >= 0) {
                    const d3 position = r.o + scale((double)r.tclosest, r.d);
                    if (sphere >= 0) {
                        // NormalOfSceneObject, OT_SPHERE (raythread.cpp:331-334): n / Magnitude(n) = (1.0f / (D)F|n|) * n
                        const ctgpu_sphere &s = P.spheres[sphere];
                        mat_color = s.material.color; mat_specular = s.material.specular; mat_reflection = s.material.reflection;
                        const d3 n = position - load_d3(s.center);
                        normal = scale((double)1.0f / (double)lengthf(n), n);
                    } else {
                        // closestScene = objects[lookup[closestIndex]] (raythread.cpp:211): slot of
                        // triangle 0 when nothing was recorded
                        const uint32_t slot = (r.slot == kNoNode) ? P.slot_of_triangle0 : r.slot;
                        const TriRecord *rec = P.tris + slot;
                        const d3 e1 = load_d3(rec->e1), e2 = load_d3(rec->e2);
                        const Material m = P.materials[rec->triangle];
                        mat_color = m.color; mat_specular = m.specular; mat_reflection = m.reflection;
                        d3 n = cross(e1, e2);                                 // raythread.cpp:337-345
                        if (!(dotf(n, r.d) < 0)) n = -n;
                        normal = n;
                    }
                    view = -r.d;
                    r.o = position;                                       // origin of every follow-up ray
                    intensity = 0.0f;
                    light = 0;
                    go_lights = true;
                } else {
                    color_ready = true;                                   // miss: background
                }
            } else {   // PH_SHADOW came back
                bool lit = r.t == kNoHit;
                if (lit && P.n_spheres) {      // ClosestIntersection(.., 0.001, tMax, ..): tMax = 1 towards a point light (raythread.cpp:289-307)
                    int sphere = -1;
                    float nearest = kFarAway;
                    walk_spheres(P, r.o, r.d, 0.001f, P.lights[light].type == CTGPU_LIGHT_POINT ? 1.0f : kFarAway, nearest, sphere);
                    lit = sphere < 0;
                }
                if (lit) {
                    // light reaches the point: diffuse + specular (raythread.cpp:309-324)
                    const Light L = P.lights[light];
                    const d3 to_light = r.d;
                    const float n_dot_l = dotf(normal, to_light);
                    if (n_dot_l > 0)
                        intensity += L.intensity * n_dot_l / (lengthf(normal) * lengthf(to_light));
                    if (mat_specular != -1) {
                        const double nl = (double)dotf(normal, to_light);
                        const d3 reflected = scale(nl, scale(2.0, normal)) - to_light;
                        const float r_dot_v = dotf(reflected, view);
                        if (r_dot_v > 0) {
                            const float base = r_dot_v / (lengthf(reflected) * lengthf(view));
                            intensity = (float)((double)intensity +
                                                (double)L.intensity * pow((double)base, (double)mat_specular));
                        }
                    }
                }
                light += 1;
                go_lights = true;
            }

            if (go_lights) {
                // next light that needs a shadow ray (raythread.cpp:277-307)
                bool shadow_started = false;
                while (light < P.n_lights) {
                    const Light L = P.lights[light];
                    if (L.type == CTGPU_LIGHT_AMBIENT) { intensity += L.intensity; light += 1; continue; }
                    if (L.type == CTGPU_LIGHT_POINT) r.d = load_d3(L.position) - r.o;
                    else if (L.type == CTGPU_LIGHT_DIRECTIONAL) r.d = load_d3(L.direction);
                    else { light += 1; continue; }
                    r.t = kNoHit;
                    start_traversal(r, P);
                    if (COUNT) cnt.visits += 1;
                    phase = PH_SHADOW;
                    cnt.shadow += 1;
                    shadow_started = true;
                    break;
                }
                if (!shadow_started) {
                    const uint32_t local = shade_value(mat_color, intensity);   // raythread.cpp:362-366
                    if (REFLECT && depth_left > 0 && mat_reflection > 0) {
                        // bounce (raythread.cpp:372-373): ReflectRay(-dir, normal), ray.t = 0 (sic)
                        local_stack[n_pending] = local;
                        refl_stack[n_pending] = mat_reflection;
                        n_pending += 1;
                        const double ni = (double)dotf(normal, view);
                        r.d = scale(ni, scale(2.0, normal)) - view;
                        r.t = P.reflect_fix ? kNoHit : 0.0f;
                        start_traversal(r, P);
                        if (COUNT) cnt.visits += 1;
                        phase = PH_CLOSEST;
                        depth_left -= 1;
                        cnt.reflect += 1;
                    } else {
                        color = local;
                        color_ready = true;
                    }
                }
            }

            if (color_ready) {
                if (REFLECT) {
                    while (n_pending > 0) {                              // unwind raythread.cpp:375-379
                        n_pending -= 1;
                        color = mix_reflection(local_stack[n_pending], color, refl_stack[n_pending]);
                    }
                }
                bool pixel_done = true;
                if (SS) {
                    pixel_color = (sample == 0) ? color : blend_sample(pixel_color, color);
                    // advance the stratified grid exactly as raythread.cpp:500-504 does
                    run_y -= jy;
                    run_y += 0.25f;
                    run_x -= jx;
                    sample += 1;
                    if ((sample & 3u) == 0u) {      // inner ys loop finished
                        run_x += 0.25f;
                        run_y = fy;
                    }
                    pixel_done = sample >= 16u;
                } else {
                    pixel_color = color;
                }
                if (pixel_done) {
                    P.out[(size_t)local_row * P.square + xi] = pixel_color;
                    phase = PH_IDLE;
                } else {
                    launch_primary();
                }
            }
        }
    }

    // counters: one atomic per wave and counter
    auto wave_sum = [](uint32_t v) {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        return v;
    };
    const uint32_t s_primary = wave_sum(cnt.primary), s_shadow = wave_sum(cnt.shadow), s_reflect = wave_sum(cnt.reflect);
    if (lane == 0) {
        atomicAdd(&P.counters[0], (unsigned long long)s_primary);
        atomicAdd(&P.counters[1], (unsigned long long)s_shadow);
        atomicAdd(&P.counters[2], (unsigned long long)s_reflect);
    }
    if (COUNT) {
        const uint32_t s_visits = wave_sum(cnt.visits), s_tests = wave_sum(cnt.tests);
        if (lane == 0) {
            atomicAdd(&P.counters[3], (unsigned long long)s_visits);
            atomicAdd(&P.counters[4], (unsigned long long)s_tests);
        }
    }
}

// ------------------------------------------------------------------------------------
// Test hook: ClosestIntersection on caller-supplied rays, same traversal code.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void ct_trace_ray_list(const FrameParams P, const ctgpu_ray *__restrict__ rays,
                                                            uint64_t n, ctgpu_hit *__restrict__ hits) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    StackEntry *stack = reinterpret_cast<StackEntry *>(lds_raw);
    const uint32_t lane = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * kWave + lane;
    Counters<false> cnt;
    LaneRay r;
    bool busy = i < n;
    r.cur = kNoNode; r.sp = 0; r.leaf_left = 0; r.t = kNoHit; r.tclosest = kFarAway; r.slot = kNoNode;
    r.o = make_d3(0, 0, 0); r.d = make_d3(0, 0, 1); r.inv = make_d3(0, 0, 1); r.fast = false;
    if (busy) {
        r.o = load_d3(rays[i].origin);
        r.d = load_d3(rays[i].direction);
        r.t = rays[i].t;
        start_traversal(r, P);
    }
    for (;;) {
        const bool walking = busy && r.cur != kNoNode;
        if (__ballot(walking) == 0ull) break;
        if (walking) {
            if (r.leaf_left == 0) {
                step_inner<false>(r, P.nodes, stack, lane, P.stack_depth, P.overflow, cnt);
                if (r.cur == kNoNode) pop_next<false>(r, stack, lane, cnt);
            } else {
                step_leaf<false>(r, P.tris, false, cnt);
                if (r.leaf_left == 0) pop_next<false>(r, stack, lane, cnt);
            }
        }
    }
    if (busy) {
        ctgpu_hit h;
        h.t = r.t;
        h.tclosest = r.tclosest;
        h.triangle = (r.slot == kNoNode) ? 0u : P.tris[r.slot].triangle;
        h.found = (r.t != kNoHit) ? 1u : 0u;
        hits[i] = h;
    }
}

// ------------------------------------------------------------------------------------
// Assemble: compact per-partition tile buffers -> framebuffer, including the subsample
// in-between rows (raythread.cpp:512-524) which need two traced rows that may live in
// different partitions.  One lane per pixel of the rendered square.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ct_assemble(const FrameParams P, const uint32_t *__restrict__ all_tiles,
                                                   uint32_t *__restrict__ fb) {
    const uint32_t xi = blockIdx.x * 256u + threadIdx.x;
    const uint32_t yi = blockIdx.y;
    if (xi >= P.square || yi >= P.rendered_rows) return;
    // the layout arithmetic is include/ct_tiles.h, shared with the host layer's cth_assemble_tiles
    ct_tile_layout L;
    L.width = P.width; L.height = P.height; L.half = P.half; L.square = P.square;
    L.rendered_rows = P.rendered_rows; L.rows_per_band = P.rows_per_band; L.subsample = P.subsample;
    L.tile_rows = P.tile_rows; L.part_count = P.part_count; L.local_rows = P.local_rows;
    uint64_t slot;
    if (ct_tiles_bitmap_slot(&L, xi, yi, &slot)) fb[slot] = ct_tiles_pixel(&L, all_tiles, xi, yi);
}

// ------------------------------------------------------------------------------------
// host-callable launchers (declared in ct_params.h)
// ------------------------------------------------------------------------------------
namespace ctd {

template <bool SS, bool REFLECT, bool COUNT>
static void launch_variant(const FrameParams &P, uint32_t n_tiles, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL((ct_trace_tiles<SS, REFLECT, COUNT>), dim3(n_tiles), dim3(kWave), lds_bytes, stream, P);
}

void launch_trace_tiles(const FrameParams &P, uint32_t n_tiles, bool ss, bool reflect, bool count,
                        hipStream_t stream) {
    const size_t lds = (size_t)P.stack_depth * kWave * sizeof(StackEntry);
    if (count) {
        if (ss) { if (reflect) launch_variant<true, true, true>(P, n_tiles, lds, stream); else launch_variant<true, false, true>(P, n_tiles, lds, stream); }
        else    { if (reflect) launch_variant<false, true, true>(P, n_tiles, lds, stream); else launch_variant<false, false, true>(P, n_tiles, lds, stream); }
    } else {
        if (ss) { if (reflect) launch_variant<true, true, false>(P, n_tiles, lds, stream); else launch_variant<true, false, false>(P, n_tiles, lds, stream); }
        else    { if (reflect) launch_variant<false, true, false>(P, n_tiles, lds, stream); else launch_variant<false, false, false>(P, n_tiles, lds, stream); }
    }
}

void launch_trace_ray_list(const FrameParams &P, const ctgpu_ray *rays, uint64_t n, ctgpu_hit *hits,
                           hipStream_t stream) {
    const size_t lds = (size_t)P.stack_depth * kWave * sizeof(StackEntry);
    const uint32_t blocks = (uint32_t)((n + kWave - 1) / kWave);
    if (blocks) hipLaunchKernelGGL(ct_trace_ray_list, dim3(blocks), dim3(kWave), lds, stream, P, rays, n, hits);
}

void launch_assemble(const FrameParams &P, const uint32_t *all_tiles, uint32_t *fb, hipStream_t stream) {
    const dim3 grid((P.square + 255u) / 256u, P.rendered_rows);
    if (P.square && P.rendered_rows) hipLaunchKernelGGL(ct_assemble, grid, dim3(256), 0, stream, P, all_tiles, fb);
}

}  // namespace ctd
