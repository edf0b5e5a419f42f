// ct_shadow_tree.hip -- the SHADOW TREE: a four-wide tree of this build's own over the scene's TriRecords, walked by the
// frame's shadow rays only (ct_walk_shadow_tree, ct_pipeline.hip; what it is and why the frame stays the reference's:
// ct_device.h "the shadow tree", DESIGN appendix E).
//
// raythread.cpp:303-307 asks of a shadow ray only whether ANY triangle the reference tests lowers ray.t, so the order in
// which triangles are looked at is free and so is the shape of the tree that finds them.  The reference's BuildBVH
// (bvh.cpp:51-106) splits at the spatial midpoint of the longest axis and stops where a split leaves one side empty:
// leaves of up to 74 triangles on the dragon stand-in, 58 box tests per shadow ray.  This tree is built by the binned
// surface-area heuristic instead, down to single triangles, and collapsed to four entries a node.
//
// Made from what is already in HBM: the TriRecords (slot order = the reference's indexes[] order), the reference's node
// array.  Steps: the triangles' grown boxes (k_tri_boxes of ct_pack.hip: the boxes appendix D's lemma is about) ->
// a binary SAH tree over them (host threads at present) -> QuadNodes whose entries are single triangles or nodes, each
// node's entries ordered so that the walk's stack stays shallow -> every stored plane moved outward by `gamma`
// (appendix E.1: what the walk's binary32 arithmetic can be off by, for any ray inside the gate) and rounded outward.
#include "ctgpu_internal.h"
#include "ct_device.h"
#include "ct_shadow_tree_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

using namespace ctd;

namespace {

constexpr int kBlock = 256;

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return ctgpu_fail(CTGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// which reference node (a leaf of bvhNodes[]) holds each TriRecord slot
__global__ void k_leaf_of_slot(const ctgpu_node *nodes, uint32_t m, uint32_t n, uint32_t *leaf_of) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const uint32_t count = nodes[i].count, first = nodes[i].first;
    for (uint32_t k = 0; k < count && first + k < n; k++) leaf_of[first + k] = i;
}
// the tree's own triangle order: record s is the reference's record order[s] (a leaf of the tree is a run of records, as
// in the reference's layout), `remaining` counts to the end of ITS leaf, and leaf_of[s] is the reference leaf of that triangle
__global__ void k_gather_records(const TriRecord *recs, const uint32_t *order, const uint32_t *remaining, const uint32_t *leaf_of_slot,
                                 uint32_t n, TriRecord *out, uint32_t *leaf_of) {
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const uint32_t from = order[s];
    TriRecord t = recs[from];
    t.remaining = remaining[s];
    out[s] = t;
    leaf_of[s] = leaf_of_slot[from];
}
// do the boxes of the reference tree nest (a child's box inside its parent's, in the stored doubles)?  Appendix B2's
// "the leaf's box passes => every ancestor's does" needs it along every path; BuildBVH's output always does.
__global__ void k_boxes_nest(const ctgpu_node *nodes, uint32_t m, uint32_t *not_nested) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m || nodes[i].count) return;
    const ctgpu_node &c = nodes[i];
    bool ok = (uint64_t)c.left + 1u < m;
    for (uint32_t side = 0; ok && side < 2; side++) {
        const ctgpu_node &g = nodes[c.left + side];
        for (int a = 0; a < 3; a++) ok = ok && g.bmin[a] >= c.bmin[a] && g.bmax[a] <= c.bmax[a];
    }
    if (!ok) atomicOr(not_nested, 1u);
}

// device blocks that are freed unless the caller takes them over
struct Owned {
    std::vector<void *> blocks;
    ~Owned() { for (void *p : blocks) (void)hipFree(p); }
    template <typename T> hipError_t alloc(T *&ptr, size_t count) {
        void *raw = nullptr;
        const hipError_t e = hipMalloc(&raw, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) { ptr = (T *)raw; blocks.push_back(raw); }
        return e;
    }
    void keep(void *p) { blocks.erase(std::find(blocks.begin(), blocks.end(), p)); }
};

}  // namespace

int ctgpu_build_shadow_tree(int device, const void *d_tris, uint32_t n, const ctgpu_node *d_ref_nodes, uint32_t m, int strategy,
                            uint32_t leaf_max, float leaf_rho, ctgpu_shadow_tree *out) {
    if (!d_tris || !d_ref_nodes || !out || n == 0 || m == 0) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_build_shadow_tree: null argument");
    *out = ctgpu_shadow_tree();
    HIP_TRY(hipSetDevice(device));
    if (n >= (1u << 30)) return CTGPU_OK;                     // (slots and node indices carry two flag bits)
    const auto t0 = std::chrono::steady_clock::now();
    Owned owned;

    // ---- may this scene have one: nested reference boxes, finite magnitudes, the lemma's K K S <= 2^20
    uint32_t *d_flag = nullptr, *d_leaf_of = nullptr;
    float *d_box = nullptr;
    HIP_TRY(owned.alloc(d_flag, 1));
    HIP_TRY(owned.alloc(d_leaf_of, n));
    HIP_TRY(owned.alloc(d_box, (size_t)n * 6));
    HIP_TRY(hipMemsetAsync(d_flag, 0, 4, nullptr));
    HIP_TRY(hipMemsetAsync(d_leaf_of, 0xff, (size_t)n * 4, nullptr));
    hipLaunchKernelGGL(k_boxes_nest, dim3(blocks_for(m)), dim3(kBlock), 0, nullptr, d_ref_nodes, m, d_flag);
    hipLaunchKernelGGL(k_leaf_of_slot, dim3(blocks_for(m)), dim3(kBlock), 0, nullptr, d_ref_nodes, m, n, d_leaf_of);
    HIP_TRY(hipGetLastError());
    uint32_t stats[8];
    int rc = ctgpu_triangle_boxes_on_device(d_tris, n, d_box, stats); if (rc) return rc;
    uint32_t not_nested = 0;
    HIP_TRY(hipMemcpy(&not_nested, d_flag, 4, hipMemcpyDeviceToHost));
    float edge, vertex, plane[3];
    std::memcpy(&edge, &stats[0], 4); std::memcpy(&vertex, &stats[1], 4); std::memcpy(plane, &stats[2], 12);
    if (not_nested || stats[5] != 0u) return CTGPU_OK;
    // appendix D.3 with the shadow rays' own bound on the origin: a shadow ray starts on a surface, |o_i| <= o_max := 2 V
    // (whatever the eye), so S = |o - p1| <= 3 V, and the gate on the direction follows from K and S alone
    const double K = edge, V = vertex;
    const double o_max = 2.0 * V, s_lim = 3.0 * V * (1.0 + 0x1p-40);
    if (!(K * K * s_lim <= 0x1p20) || !(V > 0.0)) return CTGPU_OK;
    const double denom = K * std::max(K, s_lim);
    const double d_max = denom > 0.0 ? std::min(std::min(0x1p20 / denom, 0x1p12 / (K * K)), 0x1p80) : 0x1p80;
    const double d_min = 0x1p-60;
    // appendix E.1: with B >= every |stored plane| and O >= every |o_i| of a ray inside the gate, the walk's
    // t^ = fma32(plane, F(1/d), -F(o/d)) lies within 2^-22.9 (B + O) / |d| of the real (plane - o) / d: a plane moved by
    // gamma = 2^-22 (B + O) covers it with a factor to spare (and 2^-60 more for results in the denormal range)
    const double B0 = std::max(plane[0], std::max(plane[1], plane[2]));
    if (!(B0 + o_max <= 0x1p40) || !(B0 >= 0x1p-40)) return CTGPU_OK;      // (products stay finite; the 2^-60 below stays small beside B0)
    const double gamma = 0x1p-22 * (1.01 * B0 + o_max) + 0x1p-60;

    // ---- the triangles' boxes -> a binary tree -> four entries a node (ct_shadow_tree_build.h)
    std::vector<ctst::Box> boxes(n);
    static_assert(sizeof(ctst::Box) == 24, "box layout");
    static_assert(sizeof(ctst::Quad) == sizeof(QuadNode) && ctst::kLeaf == kLeafBit, "the builder's node is the walk's");
    HIP_TRY(hipMemcpy(boxes.data(), d_box, (size_t)n * sizeof(ctst::Box), hipMemcpyDeviceToHost));
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const float bound = std::nextafterf((float)(1.01 * B0), INFINITY);
    std::vector<ctst::Quad> quads;
    std::vector<uint32_t> order, remaining;
    ctst::BuildOptions opt;
    opt.strategy = strategy; opt.leaf_max = leaf_max; opt.leaf_rho = leaf_rho; opt.threads = n < (1u << 16) ? 1u : std::min(hw, 16u);      // (a one-GPU box gives a process 16 cores)
    double phase_ms[3] = {0, 0, 0};
    opt.phase_ms = phase_ms;
    const double before_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const uint32_t need = ctst::build_shadow_tree(boxes.data(), n, opt, gamma, bound, quads, order, remaining);
    if (need > 63u) return CTGPU_OK;                                         // (the walk's stack: the lanes of a register)
    if (!ctst::references_to_byte_offsets(quads, n, (uint32_t)sizeof(TriRecord))) return CTGPU_OK;

    QuadNode *d_quads = nullptr;
    TriRecord *d_records = nullptr;
    uint32_t *d_order = nullptr, *d_remaining = nullptr, *d_leaf_of_own = nullptr;
    HIP_TRY(owned.alloc(d_quads, quads.size()));
    HIP_TRY(owned.alloc(d_records, n));
    HIP_TRY(owned.alloc(d_order, n));
    HIP_TRY(owned.alloc(d_remaining, n));
    HIP_TRY(owned.alloc(d_leaf_of_own, n));
    HIP_TRY(hipMemcpy(d_quads, quads.data(), quads.size() * sizeof(QuadNode), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_remaining, remaining.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gather_records, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, (const TriRecord *)d_tris, (const uint32_t *)d_order,
                       (const uint32_t *)d_remaining, (const uint32_t *)d_leaf_of, n, d_records, d_leaf_of_own);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    owned.keep(d_quads);
    owned.keep(d_records);
    owned.keep(d_leaf_of_own);
    out->nodes = d_quads;
    out->n_nodes = (uint32_t)quads.size();
    out->records = d_records;
    out->leaf_of_slot = d_leaf_of_own;
    out->pending = need;
    out->d_min = (float)d_min;                                  // (a power of two)
    out->d_max = std::nextafterf((float)d_max, 0.0f);           // (towards zero: the gate's comparisons are the stricter side)
    out->o_max = std::nextafterf((float)o_max, 0.0f);
    out->gamma = gamma;
    out->usable = true;
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("CTGPU_DEBUG_SHADOW_TREE"))
        std::fprintf(stderr, "shadow tree: %u triangles, %zu nodes; boxes + checks + copy down %.1f ms, binary tree %.1f ms (%u threads), four-wide %.1f ms, "
                     "emit %.1f ms, copies up + records %.1f ms\n", n, quads.size(), before_ms, phase_ms[0], opt.threads, phase_ms[1], phase_ms[2],
                     out->build_ms - before_ms - phase_ms[0] - phase_ms[1] - phase_ms[2]);
    return CTGPU_OK;
}
