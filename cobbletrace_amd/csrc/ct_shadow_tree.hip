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

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
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
    t.triangle = from;            // (here: the record's slot in the REFERENCE's order -- what hits[] names, and the reference's
                                  //  visiting order, which decides between triangles hit at the same t: ct_walk_closest_tree)
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

// ------------------------------------------------------------------ the binary tree, on the device
// Agglomerative clustering in Morton order (PLOC: Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding
// Volume Hierarchy Construction", 2018): the triangles' boxes sorted along a space-filling curve are the first clusters;
// every round each cluster looks kNeighbours places to either side for the one whose union with it has the smallest box,
// pairs that chose each other merge into a new node, the list closes up.  Trees as good as a full surface-area sweep's, in
// a few dozen rounds of small kernels: 871,200 triangles in a few milliseconds, where the host's binned builder needs a
// hundred on sixteen cores.  Nodes 0 .. n-1 are the leaves (leaf i = the i-th triangle along the curve), inner nodes get
// the ids behind them in the order of their clusters' positions, round by round -- the same tree every time.
constexpr int kNeighbours = 16, kNeighboursMost = 64;
constexpr int kClusterBlock = 256;

__device__ __forceinline__ uint32_t ordered_bits(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float from_ordered_bits(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ __forceinline__ uint64_t every_third_bit(uint32_t v) {          // 21 bits
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ void k_centre_bounds(const float *box, uint32_t n, uint32_t *bounds) {       // bounds[0..2] min, [3..5] max, as ordered bits
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n)
        for (int a = 0; a < 3; a++) lo[a] = hi[a] = 0.5f * (box[(size_t)i * 6 + a] + box[(size_t)i * 6 + 3 + a]);
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(bounds + a, ordered_bits(lo[a])); atomicMax(bounds + 3 + a, ordered_bits(hi[a])); }
    }
}
__global__ void k_curve_keys(const float *box, uint32_t n, const uint32_t *bounds, uint64_t *key, uint32_t *value) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t cell[3];
    for (int a = 0; a < 3; a++) {
        const float lo = from_ordered_bits(bounds[a]), hi = from_ordered_bits(bounds[3 + a]);
        const float c = 0.5f * (box[(size_t)i * 6 + a] + box[(size_t)i * 6 + 3 + a]);
        const float unit = hi > lo ? (c - lo) / (hi - lo) : 0.0f;
        const float scaled = fminf(fmaxf(unit * 2097152.0f, 0.0f), 2097151.0f);
        cell[a] = (uint32_t)scaled;
    }
    key[i] = every_third_bit(cell[0]) | (every_third_bit(cell[1]) << 1) | (every_third_bit(cell[2]) << 2);
    value[i] = i;
}
// the leaves: node i = the i-th triangle along the curve; every cluster is its own leaf to begin with
__global__ void k_first_clusters(const float *box, const uint32_t *tri_of_leaf, uint32_t n, float *node_box, uint32_t *cluster) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = tri_of_leaf[i];
    for (int k = 0; k < 6; k++) node_box[(size_t)i * 6 + k] = box[(size_t)t * 6 + k];
    cluster[i] = i;
}
// for every cluster the neighbour (within kNeighbours places) whose union with it has the smallest box; ties: the lower place
__global__ __launch_bounds__(kClusterBlock) void k_nearest(const float *node_box, const uint32_t *cluster, uint32_t n, int radius, uint32_t *nearest) {
    __shared__ float window[(kClusterBlock + 2 * kNeighboursMost) * 6];
    const int first = (int)(blockIdx.x * kClusterBlock) - radius;
    for (int w = threadIdx.x; w < kClusterBlock + 2 * radius; w += kClusterBlock) {
        const int place = first + w;
        if (place >= 0 && place < (int)n) {
            const float *b = node_box + (size_t)cluster[place] * 6;
            for (int k = 0; k < 6; k++) window[w * 6 + k] = b[k];
        }
    }
    __syncthreads();
    const int i = (int)(blockIdx.x * kClusterBlock + threadIdx.x);
    if (i >= (int)n) return;
    const float *mine = window + (threadIdx.x + radius) * 6;
    float best = INFINITY;
    int best_place = -1;
    for (int d = -radius; d <= radius; d++) {
        const int j = i + d;
        if (d == 0 || j < 0 || j >= (int)n) continue;
        const float *other = window + (threadIdx.x + radius + d) * 6;
        const float x = fmaxf(mine[3], other[3]) - fminf(mine[0], other[0]), y = fmaxf(mine[4], other[4]) - fminf(mine[1], other[1]),
                    z = fmaxf(mine[5], other[5]) - fminf(mine[2], other[2]);
        const float area = x * y + y * z + z * x;
        if (area < best || best_place < 0) { best = area; best_place = j; }
    }
    nearest[i] = (uint32_t)best_place;
}
// pairs that chose each other merge: the lower place keeps the new cluster, the upper one leaves the list
__global__ void k_mutual(const uint32_t *nearest, uint32_t n, uint32_t *stays, uint32_t *merges) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i > n) return;
    if (i == n) { stays[i] = 0u; merges[i] = 0u; return; }        // (so that the exclusive sums end with the totals)
    const uint32_t j = nearest[i];
    const bool mutual = j < n && nearest[j] == i;
    stays[i] = (mutual && j < i) ? 0u : 1u;
    merges[i] = (mutual && i < j) ? 1u : 0u;
}
__global__ void k_merge(float *node_box, const uint32_t *cluster, const uint32_t *nearest, const uint32_t *stays_at, const uint32_t *merge_at, uint32_t n,
                        uint32_t next_node, uint32_t *cluster_out, uint32_t *left, uint32_t *right) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = nearest[i];
    const bool mutual = j < n && nearest[j] == i;
    if (mutual && j < i) return;
    uint32_t id = cluster[i];
    if (mutual) {
        const uint32_t a = id, b = cluster[j];
        id = next_node + merge_at[i];
        left[id] = a; right[id] = b;
        for (int k = 0; k < 3; k++) {
            node_box[(size_t)id * 6 + k] = fminf(node_box[(size_t)a * 6 + k], node_box[(size_t)b * 6 + k]);
            node_box[(size_t)id * 6 + 3 + k] = fmaxf(node_box[(size_t)a * 6 + 3 + k], node_box[(size_t)b * 6 + 3 + k]);
        }
    }
    cluster_out[stays_at[i]] = id;
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

namespace {
// boxes on the device -> the binary tree on the host: bin[0 .. n) the leaves in curve order, inner nodes behind them, *root
// its root; leaf_tri[i] (i < n) the triangle of leaf i (room for the pairs pair_up_leaves appends).  CTGPU_OK with
// *built = false: the clustering did not finish (the caller takes the host builder).
int cluster_on_device(const float *d_box, uint32_t n, Owned &owned, std::vector<ctst::BinNode> &bin, uint32_t *root, std::vector<uint32_t> &leaf_tri,
                      bool *built, uint32_t *rounds_out) {
    *built = false;
    if (n < 2u) return CTGPU_OK;
    const uint32_t m = 2u * n - 1u;
    uint32_t *d_bounds = nullptr, *d_value = nullptr, *d_tri_of_leaf = nullptr, *d_cluster[2] = {nullptr, nullptr}, *d_nearest = nullptr;
    uint32_t *d_stays = nullptr, *d_merges = nullptr, *d_stays_at = nullptr, *d_merge_at = nullptr, *d_left = nullptr, *d_right = nullptr;
    uint64_t *d_key = nullptr, *d_key_sorted = nullptr;
    float *d_node_box = nullptr;
    HIP_TRY(owned.alloc(d_bounds, 6));
    HIP_TRY(owned.alloc(d_key, n)); HIP_TRY(owned.alloc(d_key_sorted, n));
    HIP_TRY(owned.alloc(d_value, n)); HIP_TRY(owned.alloc(d_tri_of_leaf, n));
    HIP_TRY(owned.alloc(d_cluster[0], n)); HIP_TRY(owned.alloc(d_cluster[1], n));
    HIP_TRY(owned.alloc(d_nearest, n));
    HIP_TRY(owned.alloc(d_stays, (size_t)n + 1)); HIP_TRY(owned.alloc(d_merges, (size_t)n + 1));
    HIP_TRY(owned.alloc(d_stays_at, (size_t)n + 1)); HIP_TRY(owned.alloc(d_merge_at, (size_t)n + 1));
    HIP_TRY(owned.alloc(d_left, m)); HIP_TRY(owned.alloc(d_right, m));
    HIP_TRY(owned.alloc(d_node_box, (size_t)m * 6));
    const uint32_t start[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIP_TRY(hipMemcpy(d_bounds, start, sizeof start, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(d_left, 0xff, (size_t)m * 4, nullptr));
    HIP_TRY(hipMemsetAsync(d_right, 0xff, (size_t)m * 4, nullptr));
    hipLaunchKernelGGL(k_centre_bounds, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, d_box, n, d_bounds);
    hipLaunchKernelGGL(k_curve_keys, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, d_box, n, (const uint32_t *)d_bounds, d_key, d_value);
    HIP_TRY(hipGetLastError());
    size_t sort_bytes = 0, scan_bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_key, d_key_sorted, d_value, d_tri_of_leaf, (int)n, 0, 63, nullptr));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_stays, d_stays_at, (int)n + 1, nullptr));
    unsigned char *d_temp = nullptr;
    HIP_TRY(owned.alloc(d_temp, std::max(sort_bytes, scan_bytes)));
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, sort_bytes, d_key, d_key_sorted, d_value, d_tri_of_leaf, (int)n, 0, 63, nullptr));
    hipLaunchKernelGGL(k_first_clusters, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, d_box, (const uint32_t *)d_tri_of_leaf, n, d_node_box, d_cluster[0]);
    HIP_TRY(hipGetLastError());
    uint32_t live = n, next_node = n, rounds = 0;
    int cur = 0, radius = kNeighbours;
    if (const char *t = std::getenv("CTGPU_SHADOW_TREE_RADIUS")) radius = std::min(std::max(std::atoi(t), 1), kNeighboursMost);      // (experiments)
    while (live > 1u) {
        if (++rounds > 400u) return CTGPU_OK;                 // (every round merges at least the closest pair of all: this is a safety net)
        hipLaunchKernelGGL(k_nearest, dim3((live + kClusterBlock - 1) / kClusterBlock), dim3(kClusterBlock), 0, nullptr, (const float *)d_node_box,
                           (const uint32_t *)d_cluster[cur], live, radius, d_nearest);
        hipLaunchKernelGGL(k_mutual, dim3(blocks_for((uint64_t)live + 1)), dim3(kBlock), 0, nullptr, (const uint32_t *)d_nearest, live, d_stays, d_merges);
        size_t bytes = scan_bytes;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_temp, bytes, d_stays, d_stays_at, (int)live + 1, nullptr));
        bytes = scan_bytes;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_temp, bytes, d_merges, d_merge_at, (int)live + 1, nullptr));
        hipLaunchKernelGGL(k_merge, dim3(blocks_for(live)), dim3(kBlock), 0, nullptr, d_node_box, (const uint32_t *)d_cluster[cur], (const uint32_t *)d_nearest,
                           (const uint32_t *)d_stays_at, (const uint32_t *)d_merge_at, live, next_node, d_cluster[cur ^ 1], d_left, d_right);
        HIP_TRY(hipGetLastError());
        uint32_t stayed = 0, merged = 0;
        HIP_TRY(hipMemcpy(&stayed, d_stays_at + live, 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&merged, d_merge_at + live, 4, hipMemcpyDeviceToHost));
        if (merged == 0u || stayed + merged != live || next_node + merged > m) return CTGPU_OK;
        live = stayed; next_node += merged; cur ^= 1;
    }
    if (next_node != m) return CTGPU_OK;
    uint32_t top = 0;
    HIP_TRY(hipMemcpy(&top, d_cluster[cur], 4, hipMemcpyDeviceToHost));
    // ---- down to the host: boxes, links, the triangles along the curve
    std::vector<float> boxes((size_t)m * 6);
    std::vector<uint32_t> left(m), right(m);
    leaf_tri.assign((size_t)n + 2u * (size_t)(m - n), 0u);
    HIP_TRY(hipMemcpy(boxes.data(), d_node_box, (size_t)m * 24, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(left.data(), d_left, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(right.data(), d_right, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(leaf_tri.data(), d_tri_of_leaf, (size_t)n * 4, hipMemcpyDeviceToHost));
    bin.reserve((size_t)m + m / 8u + 16u);                    // (room for the top that resplit_top rebuilds: no second copy of 70 MB)
    bin.resize(m);
    const unsigned threads = std::min(std::max(1u, std::thread::hardware_concurrency()), 16u);
    ctst::in_chunks(m, m >= (1u << 17) ? threads : 1u, [&](unsigned, uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++) {
            ctst::BinNode &node = bin[i];
            std::memcpy(&node.box, &boxes[(size_t)i * 6], 24);
            node.left = left[i]; node.right = right[i];
            node.first = i < n ? i : 0u; node.count = i < n ? 1u : 0u;
        }
    });
    // (links that do not make a tree would send the four-wide pass round in circles: every inner node's children lie below it)
    for (uint32_t i = n; i < m; i++) if (bin[i].left >= i || bin[i].right >= i) return CTGPU_OK;
    *root = top;
    *built = top == m - 1u;
    *rounds_out = rounds;
    return CTGPU_OK;
}
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
    static_assert(sizeof(ctst::Box) == 24, "box layout");
    static_assert(sizeof(ctst::Quad) == sizeof(QuadNode) && ctst::kLeaf == kLeafBit, "the builder's node is the walk's");
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const float bound = std::nextafterf((float)(1.01 * B0), INFINITY);
    std::vector<ctst::Quad> quads;
    std::vector<uint32_t> order, remaining;
    unsigned threads = n < (1u << 16) ? 1u : std::min(hw, 16u);            // (a one-GPU box gives a process 16 cores)
    if (const char *t = std::getenv("CTGPU_SHADOW_TREE_THREADS")) threads = (unsigned)std::min(std::max(std::atoi(t), 1), 64);      // (experiments)
    double phase_ms[3] = {0, 0, 0};
    const double before_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint32_t need = 0, rounds = 0;
    bool clustered = false;
    if (strategy == 0) {
        // the binary tree by clustering on the device; four entries a node, the order of the entries, the numbering: host
        std::vector<ctst::BinNode> bin;
        std::vector<uint32_t> leaf_tri;
        uint32_t root = 0;
        rc = cluster_on_device(d_box, n, owned, bin, &root, leaf_tri, &clustered, &rounds); if (rc) return rc;
        if (clustered) {
            phase_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - before_ms;
            if (leaf_max >= 2u) ctst::pair_up_leaves(bin, n, leaf_rho, leaf_tri);
            uint32_t cut = 64u;
            if (const char *t = std::getenv("CTGPU_SHADOW_TREE_TOP")) cut = (uint32_t)std::max(std::atoi(t), 0);      // (experiments; 0: the clustering's own top)
            root = ctst::resplit_top(bin, root, n, cut);
            phase_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - before_ms;
            need = ctst::four_wide_from_binary(bin, root, leaf_tri.data(), n, gamma, bound, threads, quads, order, remaining, phase_ms);
        }
    }
    if (!clustered) {
        std::vector<ctst::Box> boxes(n);
        HIP_TRY(hipMemcpy(boxes.data(), d_box, (size_t)n * sizeof(ctst::Box), hipMemcpyDeviceToHost));
        ctst::BuildOptions opt;
        opt.strategy = strategy == 2 ? 1 : 0; opt.leaf_max = leaf_max; opt.leaf_rho = leaf_rho; opt.threads = threads;
        opt.phase_ms = phase_ms;
        need = ctst::build_shadow_tree(boxes.data(), n, opt, gamma, bound, quads, order, remaining);
    }
    if (need > 63u) return CTGPU_OK;                                         // (the walk's stack: the lanes of a register)
    if (!ctst::references_to_byte_offsets(quads, n, (uint32_t)sizeof(TriRecord), remaining)) return CTGPU_OK;

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
    out->tau = std::nextafterf((float)(2.0 * 0x1p-12 * K * K * s_lim), INFINITY);
    out->usable = true;
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("CTGPU_DEBUG_SHADOW_TREE"))
        std::fprintf(stderr, "shadow tree: %u triangles, %zu nodes; boxes + checks %.1f ms, binary tree %.1f ms (%s), four-wide %.1f ms (%u threads), "
                     "emit %.1f ms, copies up + records %.1f ms\n", n, quads.size(), before_ms, phase_ms[0],
                     clustered ? ("clustered on the device in " + std::to_string(rounds) + " rounds, copied down").c_str() : "host builder", phase_ms[1], threads, phase_ms[2],
                     out->build_ms - before_ms - phase_ms[0] - phase_ms[1] - phase_ms[2]);
    return CTGPU_OK;
}
