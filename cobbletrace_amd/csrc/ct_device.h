// ct_device.h -- device-side data layout and exact-arithmetic building blocks (gfx950).
//
// Arithmetic contract (D = binary64 op, F = rounded to binary32):
// every function below performs the reference's operations in the reference's order
// and precision.  The translation unit is compiled with -ffp-contract=off so no
// mul+add pair is fused; f64 divide / sqrt are the correctly rounded expansions.
#ifndef CT_DEVICE_H
#define CT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ctd {

// ---------------------------------------------------------------- HBM layout
// One record per INNER node of the reference tree, 128-byte aligned so a visit is a
// single L2 line: both children's boxes (the reference tests a child's box when it
// enters the child, bvh.cpp:202; testing both when the parent is opened is the same
// arithmetic on the same operands -- the right child's `tmin < ray.t` term is
// re-evaluated with the then-current ray.t when it is popped).
struct alignas(128) WideNode {
    double lmin[3], lmax[3];      // left child box   (bvh_node_t.aabbMin/aabbMax)
    double rmin[3], rmax[3];      // right child box
    uint32_t lref, rref;          // child reference: inner -> WideNode index
                                  //                  leaf  -> kLeafBit | first slot in TriRecord[]
    uint32_t pad_[6];
};
static_assert(sizeof(WideNode) == 128, "WideNode must be one 128-byte line");

// What the shared (packet) walk reads: the tree with every second level folded away.  A QuadNode
// stands for one inner node X of the reference tree and lists, in the reference's visiting order,
// the boxes the reference tests below X before it can reach anything else: for each child c of X,
// c's two children when c is an inner node whose children's boxes lie inside c's own box
// (checked at upload), otherwise c itself.  Skipping c's own test is exact: every plane parameter
// F((b - o) /D d) is monotone in b, so for nested boxes tmin(c) <= tmin(g) and tmax(c) >= tmax(g)
// in the reference's arithmetic, ray.t only shrinks, and hence g passes => c passed; g's own test
// is the one the reference makes on entering g.  Half as many dependent fetches per ray.
// Planes are rounded to binary32 -- they only FILTER: a verdict taken from them carries an error
// bound (box_verdict_f32 below) and anything inside the bound is redone from the entry's binary64 box
// (quad_boxes[]: the six doubles of the reference's node, 48 bytes per entry, one dependent fetch).
struct alignas(128) QuadNode {
    float plane[4][3][2];         // [entry][axis]{min, max}: one packed fma serves both planes of an axis
    uint32_t ref[4];              // entry: kLeafBit | first TriRecord slot, or a QuadNode index
    uint32_t count;               // entries in use (2..4)
    uint32_t axis;                // the shadow tree's nodes: their entries are stored in ascending order along this axis (3: not); else 0
    uint32_t pad_[2];
};
static_assert(sizeof(QuadNode) == 128, "QuadNode is two s_load_dwordx16");

// One record per triangle, stored in LEAF ORDER (the order of bvh indexes[]), so a leaf
// is a contiguous run and no index indirection is needed while traversing.
// e1 = p2 - p1 and e2 = p3 - p1 are the D differences the reference recomputes on every
// test (bvh.cpp:148-149); storing them is bit-identical.
struct alignas(16) TriRecord {
    double p1[3], e1[3], e2[3];
    uint32_t triangle;            // index in scene-triangle order (closestIndex value)
    uint32_t remaining;           // triangles from this slot to the end of its leaf (>=1)
};
static_assert(sizeof(TriRecord) == 80, "TriRecord is five 16-byte loads");

static constexpr uint32_t kLeafBit = 0x80000000u;
static constexpr uint32_t kNoNode = 0xffffffffu;
// The mark of an entry of a CULL NODE (below): on a leaf reference it names ONE triangle (the TriRecord at that slot,
// whatever `remaining` says), on a node reference a lower cull node.  The reference to the ROOT of a leaf's cull nodes
// sits in an ordinary QuadNode and does not carry it: that entry's box is the leaf's own, the reference's test.
// Slots and node indices stay below 2^30.
static constexpr uint32_t kOneBit = 0x40000000u;
static constexpr uint32_t kSlotMask = 0x3fffffffu;

// ---------------------------------------------------------------- cull nodes (DESIGN appendix D)
// The reference tests EVERY triangle of a leaf it enters (bvh.cpp:208-218), and its BuildBVH leaves up to dozens in
// one (a midpoint split that puts everything on one side stops: scene_import.json has 54-triangle leaves, the dragon
// stand-in 74).  A test the reference makes may be left out when its outcome is known: IntersectTriangle returns false,
// or returns true without lowering ray.t -- neither changes anything a frame's ray carries (ray.t, the closest slot).
// Both follow when the ray misses the triangle's bounding box grown by 2^-8 of its extent, by an argument about the
// reference's own rounding errors (appendix D), for rays whose magnitudes pass a gate (CullLimits).
// So a leaf of at least `big_leaf_min` triangles is replaced, in the folded tree the packet walk reads, by a small tree
// of CULL NODES: QuadNodes (same 128 bytes, same binary32 planes, appended behind the tree's own: index >= cull_first)
// whose entries are boxes of consecutive runs of the leaf's triangles -- 4^l triangles under an entry of level l, single
// triangles (kLeafBit | kOneBit | slot) at the bottom -- listed in slot order, so the walk's "first entry first, the
// others wait on the stack" reaches the triangles in the leaf's own order.  The walk treats a cull node like any node
// except for what its filter verdict means: an entry is SKIPPED by a lane only when the binary32 filter proves that the
// real slab test on the stored planes fails (`least < -margin`, the bound of appendix B) and the lane's ray is inside the
// gate; undecided lanes and lanes outside the gate go in.  There are no binary64 boxes behind cull nodes and no exact
// levels: nothing is ever decided against the reference's own arithmetic, only tests with a known outcome are left out.
struct CullLimits {
    float d_lim;        // a ray takes part in the cull only if every |d_i| <= d_lim ...
    float o_lim;        // ... and every |o_i| <= o_lim
    float tau;          // ... and, once it has a closest hit, while its ray.t >= tau (closest-hit walks)
};

// ---------------------------------------------------------------- the shadow tree (DESIGN appendix E)
// A shadow ray only asks WHETHER some triangle the reference tests lowers ray.t (raythread.cpp:303-307: no tMax, any
// update blocks the light).  Until its first update the ray carries ray.t = 1e30f, so which leaves the reference enters is
// a matter of box tests against that constant, and the answer is
//     exists triangle T:  IntersectTriangle(T) lowers ray.t  AND  the reference enters T's leaf,
// whatever the order the triangles are looked at in.  The shadow walk therefore need not walk the reference's
// midpoint-split tree: it walks a tree of its own over the same TriRecords -- four-wide QuadNodes, built by a binned
// surface-area heuristic over the triangles' boxes, single triangles (kLeafBit | kOneBit | slot) as the lowest entries --
// whose boxes only ever EXCLUDE (the cull nodes' rule, appendix D: a lane stays out of an entry only when that proves
// every triangle test below returns false or leaves ray.t alone), and a triangle that does lower ray.t counts only after
// the reference's own test of its leaf's box (binary64, the three exact levels) has passed for that ray: with nested boxes
// (checked at upload) and no zero direction component, the leaf's box passing implies every ancestor's (appendix B2).
// The boxes are the triangles' grown boxes (k_tri_boxes) grown once more by `gamma`, a scene constant that swallows every
// rounding of the binary32 slab arithmetic for rays inside the gate below -- so the walk's test is the bare
// tmax >= max(tmin, 0) with no per-entry margin.  Rays outside the gate (a zero / tiny / huge direction component, an
// origin far outside the scene) are LEFT: their shadow byte becomes kShadowLeft and a second launch of the shadow walk
// over the reference's tree (ct_walk_packet<1>) traces them.
struct ShadowTreeLimits {
    float d_min, d_max;   // a shadow ray walks the shadow tree only if every |d_i| is in [d_min, d_max] ...
    float o_max;          // ... and every |o_i| <= o_max
};
static constexpr uint8_t kShadowLeft = 3;      // shadow byte: a needed ray the shadow tree's walk did not take
static constexpr uint32_t kPrimaryLeft = 0xfffffffdu;    // hits[].slot: a primary ray ct_walk_closest_tree left to the reference-tree walk

struct Material { uint32_t color; int32_t specular; float reflection; };
struct Light { int32_t type; float intensity; double position[3]; double direction[3]; };

// ---------------------------------------------------------------- vectors
struct d3 { double x, y, z; };

__device__ __forceinline__ d3 make_d3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 load_d3(const double *p) { return make_d3(p[0], p[1], p[2]); }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return make_d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return make_d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator-(d3 a) { return make_d3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ d3 scale(double s, d3 a) { return make_d3(s * a.x, s * a.y, s * a.z); }
// mymath.h:229-232 -- D products, D sums left to right, result F
__device__ __forceinline__ float dotf(d3 a, d3 b) { return (float)(a.x * b.x + a.y * b.y + a.z * b.z); }
// mymath.h:213-216 -- F(sqrt_D(...))
__device__ __forceinline__ float lengthf(d3 a) { return (float)__builtin_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// mymath.h:234-242
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
    return make_d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// the reference's min/max macros (mymath.h:11-17): NaN in `a` selects b
__device__ __forceinline__ float lesser(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float greater(float a, float b) { return (a > b) ? a : b; }

static constexpr float kNoHit = 1e30f;
static constexpr float kFarAway = 4294967296.0f;
static constexpr uint32_t kBackground = 0x333333u;

// ---------------------------------------------------------------- slab test (bvh.cpp:165-179)
// Returns the F entry/exit parameters; the caller applies
//   tmax >= tmin && tmin < ray.t && tmax > 0.
__device__ __forceinline__ void slab(const d3 &o, const d3 &d, const double *bmin, const double *bmax,
                                     float &tmin, float &tmax) {
    const float tx1 = (float)((bmin[0] - o.x) / d.x);
    const float tx2 = (float)((bmax[0] - o.x) / d.x);
    tmin = lesser(tx1, tx2);
    tmax = greater(tx1, tx2);
    const float ty1 = (float)((bmin[1] - o.y) / d.y);
    const float ty2 = (float)((bmax[1] - o.y) / d.y);
    tmin = greater(tmin, lesser(ty1, ty2));
    tmax = lesser(tmax, greater(ty1, ty2));
    const float tz1 = (float)((bmin[2] - o.z) / d.z);
    const float tz2 = (float)((bmax[2] - o.z) / d.z);
    tmin = greater(tmin, lesser(tz1, tz2));
    tmax = lesser(tmax, greater(tz1, tz2));
}

// The same arithmetic for the rare fallback inside a register-tight kernel: the scheduler may not
// overlap the six divisions (each needs about ten temporaries), so they run one after another.
// Returns true when one of the six parameters is a NaN (0/0: a zero direction component and the
// origin exactly on that plane) -- the one situation in which a folded tree may not be used (DESIGN B2).
__device__ __forceinline__ bool slab_in_sequence(const d3 &o, const d3 &d, const double *bmin, const double *bmax,
                                                 float &tmin, float &tmax) {
    const float tx1 = (float)((bmin[0] - o.x) / d.x);
    __builtin_amdgcn_sched_barrier(0);
    const float tx2 = (float)((bmax[0] - o.x) / d.x);
    __builtin_amdgcn_sched_barrier(0);
    tmin = lesser(tx1, tx2);
    tmax = greater(tx1, tx2);
    const float ty1 = (float)((bmin[1] - o.y) / d.y);
    __builtin_amdgcn_sched_barrier(0);
    const float ty2 = (float)((bmax[1] - o.y) / d.y);
    __builtin_amdgcn_sched_barrier(0);
    tmin = greater(tmin, lesser(ty1, ty2));
    tmax = lesser(tmax, greater(ty1, ty2));
    const float tz1 = (float)((bmin[2] - o.z) / d.z);
    __builtin_amdgcn_sched_barrier(0);
    const float tz2 = (float)((bmax[2] - o.z) / d.z);
    __builtin_amdgcn_sched_barrier(0);
    tmin = greater(tmin, lesser(tz1, tz2));
    tmax = lesser(tmax, greater(tz1, tz2));
    return (tx1 != tx1) | (tx2 != tx2) | (ty1 != ty1) | (ty2 != ty2) | (tz1 != tz1) | (tz2 != tz2);
}

// ---------------------------------------------------------------- slab test, division-free
// The six quotients F((b - o) /D d) dominate the traversal (six f64 divides per box in the
// reference, 54 % of its run time).  With inv = RN(1/d) computed once per ray,
//     q = RN((b - o) * inv)       differs from       Q = RN((b - o) / d)
// by at most 3 units in the last place of a double (three roundings of relative size 2^-53
// each; zero / infinite / NaN results coincide exactly), so the approximate
// float a = F(q) is either the reference's float e = F(Q) or one of its two neighbours in
// the ordered set of floats (q and Q straddle at most one rounding boundary), with the same
// sign.  min / max in the reference's macro form are monotone, so the approximate
// tmin / tmax are likewise within ONE float step of the exact ones.
//
// Two flavours use this:
//  * slab_fast() (megakernel): proves a == e per quotient -- q must not sit within 4 ulp_D of
//    a binary32 rounding boundary; otherwise the caller recomputes with slab().
//  * slab_approx() + the ordinal tests below (wavefront pipeline): takes the approximate
//    floats as they are and proves each DECISION made from them -- `tmax >= tmin`,
//    `tmin < ray.t`, `tmax > 0` -- by requiring a margin of more than the possible error in
//    float-ordinal space; an uncertain decision is redone with slab().  Either way the
//    traversal takes exactly the reference's branches.
// Neither may be used for rays whose direction has a non-zero component outside
// [2^-999, 2^1000) (inv would be inexact beyond one rounding): direction_allows_fast_slab().
__device__ __forceinline__ bool fast_quotient(double x, double inv, float &f) {
    const double q = x * inv;
    f = (float)q;
    const uint32_t lo = (uint32_t)__double2loint(q);
    const bool near_midpoint = ((lo - 0x0FFFFFFCu) & 0x1FFFFFFFu) <= 8u;
    // v_cmp_class_f32 mask: -inf | -denormal | -0 | +0 | +denormal | +inf
    const bool odd_class = __builtin_amdgcn_classf(f, 0x2F4);
    return near_midpoint | odd_class;
}

// Returns true when the result must be recomputed with slab().
__device__ __forceinline__ bool slab_fast(const d3 &o, const d3 &inv, const double *bmin, const double *bmax,
                                          float &tmin, float &tmax) {
    float tx1, tx2, ty1, ty2, tz1, tz2;
    bool redo = fast_quotient(bmin[0] - o.x, inv.x, tx1);
    redo |= fast_quotient(bmax[0] - o.x, inv.x, tx2);
    redo |= fast_quotient(bmin[1] - o.y, inv.y, ty1);
    redo |= fast_quotient(bmax[1] - o.y, inv.y, ty2);
    redo |= fast_quotient(bmin[2] - o.z, inv.z, tz1);
    redo |= fast_quotient(bmax[2] - o.z, inv.z, tz2);
    tmin = lesser(tx1, tx2);
    tmax = greater(tx1, tx2);
    tmin = greater(tmin, lesser(ty1, ty2));
    tmax = lesser(tmax, greater(ty1, ty2));
    tmin = greater(tmin, lesser(tz1, tz2));
    tmax = lesser(tmax, greater(tz1, tz2));
    return redo;
}

// approximate entry / exit parameters, each within one float step of the reference's.
// PRECONDITION: every direction component is finite and NON-ZERO (so is inv), hence no
// quotient is NaN and the hardware min / max (one instruction each) return the same VALUE as
// the reference's `a<b?a:b` macros; they may pick the other sign of a zero, which no
// comparison downstream can see and which the verdict margins below allow for.
__device__ __forceinline__ void slab_approx(const d3 &o, const d3 &inv, const double *bmin, const double *bmax,
                                            float &tmin, float &tmax) {
    const float tx1 = (float)((bmin[0] - o.x) * inv.x), tx2 = (float)((bmax[0] - o.x) * inv.x);
    const float ty1 = (float)((bmin[1] - o.y) * inv.y), ty2 = (float)((bmax[1] - o.y) * inv.y);
    const float tz1 = (float)((bmin[2] - o.z) * inv.z), tz2 = (float)((bmax[2] - o.z) * inv.z);
    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx1, tx2), __builtin_fminf(ty1, ty2)), __builtin_fminf(tz1, tz2));
    tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx1, tx2), __builtin_fmaxf(ty1, ty2)), __builtin_fmaxf(tz1, tz2));
}

// The same with ZERO direction components allowed (inv = +-inf there, the sign of the zero): on such an
// axis (b - o) * inv is the reference's (b - o) / d exactly -- the same infinity -- unless b == o, where both
// are NaN.  Returns true when one of the six parameters is a NaN: the caller then takes the reference's own
// operation sequence (the macros' NaN behaviour depends on operand order, slab_in_sequence).
__device__ __forceinline__ bool slab_approx_checked(const d3 &o, const d3 &inv, const double *bmin, const double *bmax,
                                                    float &tmin, float &tmax) {
    const float tx1 = (float)((bmin[0] - o.x) * inv.x), tx2 = (float)((bmax[0] - o.x) * inv.x);
    const float ty1 = (float)((bmin[1] - o.y) * inv.y), ty2 = (float)((bmax[1] - o.y) * inv.y);
    const float tz1 = (float)((bmin[2] - o.z) * inv.z), tz2 = (float)((bmax[2] - o.z) * inv.z);
    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx1, tx2), __builtin_fminf(ty1, ty2)), __builtin_fminf(tz1, tz2));
    tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx1, tx2), __builtin_fmaxf(ty1, ty2)), __builtin_fmaxf(tz1, tz2));
    return __builtin_isunordered(tx1, tx2) | __builtin_isunordered(ty1, ty2) | __builtin_isunordered(tz1, tz2);
}

// position of a float in the ordered set of floats (-0 and +0 are neighbours; NaNs lie
// beyond the infinities, where no margin test is ever "close")
__device__ __forceinline__ int32_t float_ordinal(float x) {
    const int32_t b = __float_as_int(x);
    return b ^ ((b >> 31) & 0x7fffffff);
}
// |a - b| <= steps in ordinal space (wrapping arithmetic; spurious "near" only costs a redo)
__device__ __forceinline__ bool ordinals_near(int32_t a, int32_t b, uint32_t steps) {
    return (uint32_t)(a - b) + steps <= 2u * steps;
}

// Same three decisions when ray_t > 0 is known (every ray of the wavefront pipeline: t starts at
// 1e30f and only ever drops to hits > 0.0001f).  Raw bit patterns replace the ordinals:
//  * tmax within 3 steps of +-0 is flagged by its magnitude bits;
//  * for tmax >= tmin the raw-bit distance equals the ordinal distance when both have the same
//    sign, and values of opposite sign can only be neighbours next to zero, which the first
//    test has already flagged;
//  * for tmin < t, t is positive and far from zero, so only a positive tmin can be near it.
__device__ __forceinline__ bool box_verdict_positive_t(float tmin, float tmax, float ray_t, bool &pass) {
    const int32_t bmin = __float_as_int(tmin), bmax = __float_as_int(tmax), bt = __float_as_int(ray_t);
    const bool uncertain = (((uint32_t)bmax & 0x7fffffffu) <= 3u) | ordinals_near(bmax, bmin, 3u) |
                           ordinals_near(bmin, bt, 2u);
    pass = tmax >= tmin && tmin < ray_t && tmax > 0;
    return !uncertain;
}

// The three decisions of the box test from approximate tmin / tmax (each <= 1 step off):
// returns true when all three are certain, with the reference's verdict in `pass`.
// (one more step of margin than the rounding argument needs, for the -0 / +0 ordinal pair)
//   tmax >= tmin : both sides may move one step           -> needs |tmax - tmin| > 3 steps
//   tmin <  t    : t is exact                              -> needs |tmin - t|    > 2 steps
//   tmax >  0    : 0 is exact (either sign of zero)        -> needs tmax more than 3 steps from +-0
__device__ __forceinline__ bool box_verdict(float tmin, float tmax, float ray_t, int32_t ray_t_ordinal, bool &pass) {
    const int32_t kmin = float_ordinal(tmin), kmax = float_ordinal(tmax);
    const bool uncertain = ordinals_near(kmax, kmin, 3u) | ordinals_near(kmin, ray_t_ordinal, 2u) |
                           ordinals_near(kmax, 0, 3u);
    pass = tmax >= tmin && tmin < ray_t && tmax > 0;
    return !uncertain;
}

// ---------------------------------------------------------------- slab test, binary32 filter
// Level 0 of the packet walk.  Per ray, once:  invf = F(inv),  noi = -F(o * inv)  (per axis), so a
// plane parameter is ONE binary32 fma,  t^ = fma(F(b), invf, noi),  against the reference's
// e = F((b - o) /D d).  With T = (b - o)/d in real arithmetic and B >= |b| for every plane of the
// tree (plane_bound, taken over all nodes at upload):
//     |t^ - T| <= 1.001 * (B |invf| 2^-23 + |noi| 2^-24)  +  1.001 * 2^-24 |t^|
//                 [roundings of F(b), invf, noi]             [rounding of the fma]
//     |e  - T| <= (2^-24 + 2^-51) |T| + 2^-150                [the reference's own roundings]
// so  |e - t^| <= 1.01 E + 2^-22.9 |t^| + 2^-150  with  E = max over axes of the first bracket.
// ray_filter() stores  e = 1.5 E + 2^-120  and the verdict uses  M(t) = e + 2^-21 |t|,  which leaves
// more than 0.49 E + 2^-21.4 |t| for the binary32 roundings of the comparisons themselves.
// t -> t -+ M(t) is increasing, min / max are monotone, so the reference's tmin / tmax lie within
// M of the filter's; a decision is taken only when it holds over that whole interval.
// The bound needs finite, non-overflowing operands: ray_filter() sets e = +inf (every verdict
// uncertain, nothing is decided here) unless each |d| is in [2^-90, 2^90] and E < 2^80; and with
// e finite every product is below 2^81, so t^ is finite and an inf / NaN can only come from e.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct RayFilter {
    // four register pairs, laid out for the packed fma's operand selects (no splat copies):
    f32x2 inv_xy, noi_xy;         // {inv.x, inv.y}, {noi.x, noi.y}
    f32x2 inv_z_e2, noi_z_e;      // {inv.z, 2 e},   {noi.z, e}
};
// e_limit: rays whose slab parameters could exceed e_limit * 2^24 in magnitude are not taken (infinite margin: the
// exact levels decide for them).  2^80 by default; the any-hit walks pass 2^60, which keeps every parameter of the
// rays they do take below 2^84 -- far below the 1e30f such a ray carries as ray.t for as long as it is traced, so that
// "tmin < ray.t" needs no look (box_filter_f32<.., true>, and no per-lane tmin has to wait on the stack).
__device__ __forceinline__ RayFilter ray_filter(const d3 &o, const d3 &d, const d3 &inv, const float *plane_bound, float e_limit = 0x1p80f) {
    const float fi[3] = {(float)inv.x, (float)inv.y, (float)inv.z};
    const float fo[3] = {-(float)(o.x * inv.x), -(float)(o.y * inv.y), -(float)(o.z * inv.z)};
    auto ok = [](double c) {
        const uint32_t e = ((uint32_t)__double2hiint(c) >> 20) & 0x7FFu;
        return e >= 1023u - 90u && e < 1023u + 90u;
    };
    float E = 0.0f;
    for (int a = 0; a < 3; a++) {
        const float ea = __builtin_fmaf(plane_bound[a] * __builtin_fabsf(fi[a]), 0x1p-23f, __builtin_fabsf(fo[a]) * 0x1p-24f);
        E = (ea > E || !(ea == ea)) ? ea : E;          // a NaN sticks
    }
    const bool usable = ok(d.x) && ok(d.y) && ok(d.z) && E < e_limit;
    const float e = usable ? __builtin_fmaf(E, 1.5f, 0x1p-120f) : __builtin_inff();
    RayFilter f;
    f.inv_xy = f32x2{fi[0], fi[1]};  f.noi_xy = f32x2{fo[0], fo[1]};
    f.inv_z_e2 = f32x2{fi[2], e + e}; f.noi_z_e = f32x2{fo[2], e};
    return f;
}
// One box from {min, max} plane pairs (wave-uniform, in scalar registers): three packed fmas --
// written out because the operand selects are what lets {inv.x, inv.y} serve both x and y without
// a splat copy per axis (the compiler makes the copies, 6 more registers) -- then the reference's
// min / max pattern.
__device__ __forceinline__ void slab_pairs_f32(const RayFilter &f, f32x2 px, f32x2 py, f32x2 pz, float &tmin, float &tmax) {
    f32x2 tx, ty, tz;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tx) : "s"(px), "v"(f.inv_xy), "v"(f.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(ty) : "s"(py), "v"(f.inv_xy), "v"(f.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tz) : "s"(pz), "v"(f.inv_z_e2), "v"(f.noi_z_e));
    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fminf(tz.x, tz.y));
    tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fmaxf(tz.x, tz.y));
}
__device__ __forceinline__ float filter_margin(float t, float e) { return __builtin_fmaf(__builtin_fabsf(t), 0x1p-21f, e); }
// The reference's  tmax >= tmin && tmin < ray.t && tmax > 0  holds when the least of
//   tmax - tmin,  ray.t - tmin,  tmax      exceeds what the errors can take away, and fails when that
// least value is below minus the same: one margin 2 M(max |t|) >= M(tmin) + M(tmax) serves all three.
// Returns +1 (certainly passes), -1 (certainly fails) as the sign of `verdict`; |verdict| <= margin is undecided.
__device__ __forceinline__ void box_verdict_f32(float tmin, float tmax, float two_e, float ray_t, float &least, float &margin) {
    margin = __builtin_fmaf(__builtin_fmaxf(__builtin_fabsf(tmin), __builtin_fabsf(tmax)), 0x1p-20f, two_e);
    least = __builtin_fminf(__builtin_fminf(tmax - tmin, ray_t - tmin), tmax);
}
// slab_pairs_f32 + box_verdict_f32 for one box, instruction for instruction (17 VALU): written out
// because min / max on values the compiler did not produce itself get a canonicalising
// `v_max x, x` in front of each operand (6 per box), and because the scalar unit is the scarcer
// resource here -- everything per-lane stays in the vector unit until the two compares.
// NO_ROOM (the any-hit walks): ray_t is 1e30f and every parameter below 2^84 (ray_filter's e_limit), so
// ray_t - tmin is never the least of the three and is not computed: min(gap, tmax) is min3(gap, room, tmax), bit for bit.
template <bool NO_ROOM = false>
__device__ __forceinline__ void box_filter_f32(const RayFilter &f, f32x2 px, f32x2 py, f32x2 pz, float ray_t,
                                               float &tmin, float &least, float &margin) {
    f32x2 tx, ty, tz;
    float nx, ny, nz, fx, fy, fz, tmax, big, gap, room;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tx) : "s"(px), "v"(f.inv_xy), "v"(f.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(ty) : "s"(py), "v"(f.inv_xy), "v"(f.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tz) : "s"(pz), "v"(f.inv_z_e2), "v"(f.noi_z_e));
    asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(tx.x), "v"(tx.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ty.x), "v"(ty.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(tz.x), "v"(tz.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(tx.x), "v"(tx.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ty.x), "v"(ty.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(tz.x), "v"(tz.y));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(nx), "v"(ny), "v"(nz));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(fx), "v"(fy), "v"(fz));
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(big) : "v"(tmin), "v"(tmax));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(margin) : "v"(big), "s"(0x1p-20f), "v"(f.inv_z_e2.y));
    asm("v_sub_f32 %0, %1, %2" : "=v"(gap) : "v"(tmax), "v"(tmin));
    if (NO_ROOM) {
        asm("v_min_f32 %0, %1, %2" : "=v"(least) : "v"(gap), "v"(tmax));
    } else {
        asm("v_sub_f32 %0, %1, %2" : "=v"(room) : "v"(ray_t), "v"(tmin));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(least) : "v"(gap), "v"(room), "v"(tmax));
    }
}
// The same for a packet whose rays all point into one octant (OCT bit 0 / 1 / 2: the x / y / z component of every
// direction is negative): which plane of an axis is the near one is then known beforehand -- the minimum plane where
// the ray goes up, the maximum plane where it goes down -- and the six v_min / v_max that sort each axis' two
// parameters fall away (11 VALU).  Same values: t(plane) = fma(plane, 1/d, -o/d) is monotone in the plane, rising for
// 1/d > 0 and falling for 1/d < 0, and min <= max, so min(t_lo, t_hi) IS t of the plane chosen here (equal planes give
// equal parameters).  Rays with a zero component (1/d = +-inf, parameters inf or NaN) do not rely on any of it: their
// margin is infinite and every verdict of theirs goes to the exact levels, as before.
template <int OCT, bool NO_ROOM = false>
__device__ __forceinline__ void box_filter_f32_oct(const RayFilter &f, f32x2 px, f32x2 py, f32x2 pz, float ray_t,
                                                   float &tmin, float &least, float &margin) {
    if (OCT == 8) { box_filter_f32<NO_ROOM>(f, px, py, pz, ray_t, tmin, least, margin); return; }
    f32x2 tx, ty, tz;
    float tmax, big, gap, room;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tx) : "s"(px), "v"(f.inv_xy), "v"(f.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(ty) : "s"(py), "v"(f.inv_xy), "v"(f.noi_xy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(tz) : "s"(pz), "v"(f.inv_z_e2), "v"(f.noi_z_e));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"((OCT & 1) ? tx.y : tx.x), "v"((OCT & 2) ? ty.y : ty.x), "v"((OCT & 4) ? tz.y : tz.x));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"((OCT & 1) ? tx.x : tx.y), "v"((OCT & 2) ? ty.x : ty.y), "v"((OCT & 4) ? tz.x : tz.y));
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(big) : "v"(tmin), "v"(tmax));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(margin) : "v"(big), "s"(0x1p-20f), "v"(f.inv_z_e2.y));
    asm("v_sub_f32 %0, %1, %2" : "=v"(gap) : "v"(tmax), "v"(tmin));
    if (NO_ROOM) {
        asm("v_min_f32 %0, %1, %2" : "=v"(least) : "v"(gap), "v"(tmax));
    } else {
        asm("v_sub_f32 %0, %1, %2" : "=v"(room) : "v"(ray_t), "v"(tmin));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(least) : "v"(gap), "v"(room), "v"(tmax));
    }
}
// the deferred  tmin < ray.t  of a pushed child (ray.t may have dropped since): decided -> keep
__device__ __forceinline__ bool deferred_verdict_f32(float tmin, float e, float ray_t, bool &keep) {
    const float m = filter_margin(tmin, e);
    keep = tmin + m < ray_t;
    return keep | (tmin - m > ray_t);
}

// true when every component of d is non-zero and comfortably inside the binary64 normal range
// (a zero component makes inf / NaN slabs: those rays take the reference's divisions)
__device__ __forceinline__ bool direction_allows_fast_slab(const d3 &d) {
    auto ok = [](double c) {
        const uint32_t e = ((uint32_t)__double2hiint(c) >> 20) & 0x7FFu;
        return e > 23u && e < 2023u;                    // |c| in [2^-999, 2^1000): finite, normal, non-zero
    };
    return ok(d.x) && ok(d.y) && ok(d.z);
}

// the same, zero components allowed (slab_approx_checked): every component is zero or finite and normal
__device__ __forceinline__ bool direction_allows_checked_slab(const d3 &d) {
    auto ok = [](double c) {
        const uint32_t e = ((uint32_t)__double2hiint(c) >> 20) & 0x7FFu;
        return (e > 23u && e < 2023u) || c == 0.0;
    };
    return ok(d.x) && ok(d.y) && ok(d.z);
}

// ---------------------------------------------------------------- triangle test (bvh.cpp:147-163)
// Returns true exactly when the reference's IntersectTriangle returns true; ray_t is
// lowered only when t > 0.0001f.
__device__ __forceinline__ bool triangle_test(const d3 &o, const d3 &d, float &ray_t, const d3 &p1, const d3 &e1,
                                              const d3 &e2) {
    const d3 h = cross(d, e2);
    const float a = dotf(e1, h);
    if (a > -0.0001f && a < 0.0001f) return false;
    const float f = 1.0f / a;
    const d3 s = o - p1;
    const float u = f * dotf(s, h);
    if (u < 0 || u > 1) return false;
    const d3 q = cross(s, e1);
    const float v = f * dotf(d, q);
    if (v < 0 || u + v > 1) return false;
    const float t = f * dotf(e2, q);
    if (t > 0.0001f) ray_t = lesser(ray_t, t);
    return true;
}

// ---------------------------------------------------------------- colour (color.h:18-126)
__device__ __forceinline__ uint32_t pack_u8(uint32_t r, uint32_t g, uint32_t b) { return (b << 16) | (g << 8) | r; }
// float -> uint8 conversion of a non-negative value, truncating (values here are in [0,255])
__device__ __forceinline__ uint32_t to_u8(float c) { return (uint32_t)(int)c & 0xffu; }
__device__ __forceinline__ uint32_t to_u8(double c) { return (uint32_t)(int)c & 0xffu; }

// ColorToHsv . (v := value) . HsvToColor   (color.h:114-126, 49-75, 18-45)
__device__ __forceinline__ uint32_t shade_value(uint32_t material_color, float value) {
    const float r = (float)(material_color & 0xff) / 255.0f;
    const float g = (float)((material_color >> 8) & 0xff) / 255.0f;
    const float b = (float)((material_color >> 16) & 0xff) / 255.0f;
    const float hi = greater(r, greater(g, b));
    const float lo = lesser(r, lesser(g, b));
    const float delta = hi - lo;
    float h = -1.0f, s = 0.0f;
    if (hi != 0.0f) {
        s = delta / hi;
        if (r == hi) h = (g - b) / delta;
        else if (g == hi) h = 2.0f + (b - r) / delta;
        else h = 4.0f + (r - g) / delta;
        h = (float)((double)h * 60.0);
        if (h < 0) h = (float)((double)h + 360.0);
    }
    float cr, cg, cb;
    if (s == 0) {
        cr = cg = cb = value;
    } else {
        h = (float)((double)h / 60.0);
        const int sector = (int)floorf(h);
        const float frac = h - (float)sector;
        const float aa = value * (1.0f - s);
        const float bb = value * (1.0f - (s * frac));
        const float cc = value * (1.0f - (s * (1.0f - frac)));
        cr = cg = cb = 0.0f;
        switch (sector) {
            case 0: cr = value; cg = cc; cb = aa; break;
            case 1: cr = bb; cg = value; cb = aa; break;
            case 2: cr = aa; cg = value; cb = cc; break;
            case 3: cr = aa; cg = bb; cb = value; break;
            case 4: cr = cc; cg = aa; cb = value; break;
            case 5: cr = value; cg = aa; cb = bb; break;
        }
    }
    return pack_u8(to_u8(lesser(cr * 255.0f, 255.0f)), to_u8(lesser(cg * 255.0f, 255.0f)),
                   to_u8(lesser(cb * 255.0f, 255.0f)));
}

// supersample running blend (raythread.cpp:482-498)
__device__ __forceinline__ uint32_t blend_channel(uint32_t acc, uint32_t smp) {
    float c = (float)acc;
    c -= (c / 8.0f);
    c += ((float)smp / 8.0f);
    return to_u8(c);
}
__device__ __forceinline__ uint32_t blend_sample(uint32_t acc, uint32_t smp) {
    return pack_u8(blend_channel(acc & 0xff, smp & 0xff), blend_channel((acc >> 8) & 0xff, (smp >> 8) & 0xff),
                   blend_channel((acc >> 16) & 0xff, (smp >> 16) & 0xff));
}
// subsample in-between pixel (raythread.cpp:517-523)
__device__ __forceinline__ uint32_t average_channel(uint32_t a, uint32_t b) {
    return to_u8(lesser(((float)a + (float)b) / 2.0f, 255.0f));
}
__device__ __forceinline__ uint32_t average_pair(uint32_t a, uint32_t b) {
    return pack_u8(average_channel(a & 0xff, b & 0xff), average_channel((a >> 8) & 0xff, (b >> 8) & 0xff),
                   average_channel((a >> 16) & 0xff, (b >> 16) & 0xff));
}
// reflection mix (raythread.cpp:375-379)
__device__ __forceinline__ uint32_t mix_channel(uint32_t l, uint32_t r, double keep, double take) {
    return to_u8(keep * (double)l + take * (double)r);
}
__device__ __forceinline__ uint32_t mix_reflection(uint32_t local, uint32_t reflected, float reflection) {
    const double keep = (double)(1.0f - reflection), take = (double)reflection;
    return pack_u8(mix_channel(local & 0xff, reflected & 0xff, keep, take),
                   mix_channel((local >> 8) & 0xff, (reflected >> 8) & 0xff, keep, take),
                   mix_channel((local >> 16) & 0xff, (reflected >> 16) & 0xff, keep, take));
}

// jitter integers, counter based (CTGPU_RNG_HASH); same function as the oracle's
__device__ __forceinline__ int32_t hash_rand(uint64_t seed, uint64_t k) {
    uint64_t z = (seed + 1u) * 0x9E3779B97F4A7C15ull + k;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (int32_t)(z >> 33);
}

}  // namespace ctd
#endif
