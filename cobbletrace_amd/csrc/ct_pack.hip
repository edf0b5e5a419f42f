// ct_pack.hip -- the scene's walk layouts, made ON THE DEVICE from the reference's arrays.
//
// Input (already in HBM, from ct_build.hip or copied up as the caller gave them): bvhNodes[] (64 B each, the
// reference's numbering), triangles.indexes[], the triangles' vertices.  Output: what ct_device.h describes --
// WideNodes (per-lane walks), the folded QuadNode tree and its unfolded twin with their binary64 boxes (packet walk),
// leaf-ordered TriRecords.  The host used to fold these serially and push ~0.5 GB through pageable copies (0.34 s
// for the dragon, 2 s for 8 M triangles); here it is a few milliseconds and the tree never crosses PCIe.
//
// Everything comes out byte for byte as the host restatement (pack_on_host in ctgpu_api.hip, CTGPU_PACK=host) lays
// it out -- tests/test_gpu_scene_pack.py compares the arrays.  Two things are order-dependent:
//  * WideNode slots: inner nodes in reference order -> an exclusive prefix count over "is inner".
//  * QuadNode slots: the host numbers QuadNodes in the order a queue discovers them from the root (each QuadNode's
//    inner entries, in entry order, append to the queue) -- breadth first.  A level of that queue is numbered from
//    the level before it: per QuadNode the count of inner entries, a prefix sum over the level, slot = end of the
//    level + sum + rank among the QuadNode's own inner entries.  One pass per level of the folded tree (~14 for the
//    dragon), each a count kernel, a scan and an emit kernel.
// The most entries ever pending on the packet walk's stack (quad_pending) rides along: a QuadNode reached with
// `before` entries pending leaves before + count - 1 - k pending while entry k is walked.
#include "ctgpu_internal.h"
#include "ct_device.h"

#include <algorithm>
#include <cmath>
#include <cstring>
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

// ---- exclusive prefix sum of uint32 (n + 1 outputs: out[n] = total), any n: block sums, their scan, the final pass
constexpr int kScanItems = 8;                         // per thread
constexpr uint32_t kScanTile = kBlock * kScanItems;
__global__ void k_scan_tiles(const uint32_t *in, uint32_t n, uint32_t *tile_sums) {
    __shared__ uint32_t partial[kBlock];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t sum = 0;
    for (int i = 0; i < kScanItems; i++) sum += (base + i < n) ? in[base + i] : 0u;
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int step = kBlock / 2; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) partial[threadIdx.x] += partial[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = partial[0];
}
__global__ void k_scan_tile_sums(uint32_t *tile_sums, uint32_t n_tiles) {     // one block, serial carry over chunks
    __shared__ uint32_t chunk[kBlock];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < n_tiles; start += kBlock) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t mine = (i < n_tiles) ? tile_sums[i] : 0u;
        chunk[threadIdx.x] = mine;
        __syncthreads();
        for (int step = 1; step < kBlock; step <<= 1) {                    // inclusive scan of the chunk
            const uint32_t add = ((int)threadIdx.x >= step) ? chunk[threadIdx.x - step] : 0u;
            __syncthreads();
            chunk[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_tiles) tile_sums[i] = carry + chunk[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == 0) carry += chunk[kBlock - 1];
        __syncthreads();
    }
}
__global__ void k_scan_finish(const uint32_t *in, uint32_t n, const uint32_t *tile_sums, uint32_t *out) {
    __shared__ uint32_t partial[kBlock];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], sum = 0;
    for (int i = 0; i < kScanItems; i++) { v[i] = (base + i < n) ? in[base + i] : 0u; sum += v[i]; }
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int step = 1; step < kBlock; step <<= 1) {
        const uint32_t add = ((int)threadIdx.x >= step) ? partial[threadIdx.x - step] : 0u;
        __syncthreads();
        partial[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = tile_sums[blockIdx.x] + partial[threadIdx.x] - sum;
    for (int i = 0; i < kScanItems; i++) {
        if (base + i <= n) out[base + i] = run;                           // (position n receives the total)
        run += v[i];
    }
}
struct Scanner {
    uint32_t *tile_sums = nullptr;
    uint32_t capacity = 0;                 // elements
    int prepare(uint32_t max_n) {
        capacity = max_n;
        HIP_TRY(hipMalloc((void **)&tile_sums, ((size_t)max_n / kScanTile + 2) * sizeof(uint32_t)));
        return CTGPU_OK;
    }
    void run(const uint32_t *in, uint32_t n, uint32_t *out, hipStream_t stream) const {     // out: n + 1 words
        const uint32_t tiles = n / kScanTile + 1;
        hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(kBlock), 0, stream, in, n, tile_sums);
        hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kBlock), 0, stream, tile_sums, tiles);
        hipLaunchKernelGGL(k_scan_finish, dim3(tiles), dim3(kBlock), 0, stream, in, n, (const uint32_t *)tile_sums, out);
    }
    ~Scanner() { if (tile_sums) (void)hipFree(tile_sums); }
};

// ---- WideNodes
__global__ void k_inner_flags(const ctgpu_node *nodes, uint32_t m, uint32_t *flag) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < m) flag[i] = nodes[i].count == 0 ? 1u : 0u;
}
__device__ __forceinline__ uint32_t child_ref(const ctgpu_node *nodes, const uint32_t *wide_of, uint32_t id) {
    return nodes[id].count ? (kLeafBit | nodes[id].first) : wide_of[id];
}
__global__ void k_wide(const ctgpu_node *nodes, uint32_t m, const uint32_t *wide_of, WideNode *wide) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m || nodes[i].count) return;
    const uint32_t left = nodes[i].left;
    const ctgpu_node l = nodes[left], r = nodes[left + 1];
    WideNode w;
    for (int c = 0; c < 3; c++) { w.lmin[c] = l.bmin[c]; w.lmax[c] = l.bmax[c]; w.rmin[c] = r.bmin[c]; w.rmax[c] = r.bmax[c]; }
    w.lref = child_ref(nodes, wide_of, left);
    w.rref = child_ref(nodes, wide_of, left + 1);
    for (int k = 0; k < 6; k++) w.pad_[k] = 0u;
    wide[wide_of[i]] = w;
}

// ---- the bound the binary32 filter needs: max |plane| per axis over every node; anything beyond 1e30 or NaN -> flag
__global__ void k_bound(const ctgpu_node *nodes, uint32_t m, unsigned long long *out) {    // out[0..2] maxima as bits, out[3] flag
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    double big[3] = {0.0, 0.0, 0.0};
    bool bad = false;
    if (i < m) {
        for (int c = 0; c < 3; c++) {
            const double lo = __builtin_fabs(nodes[i].bmin[c]), hi = __builtin_fabs(nodes[i].bmax[c]);
            if (!(lo <= 1e30) || !(hi <= 1e30)) bad = true;
            else big[c] = lo > hi ? lo : hi;
        }
    }
    // non-negative doubles order like their bit patterns
    for (int c = 0; c < 3; c++) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(big[c]);
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long other = __shfl_xor(bits, off);
            bits = other > bits ? other : bits;
        }
        if ((threadIdx.x & 63) == 0 && bits) atomicMax(out + c, bits);
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicMax(out + 3, 1ull);
}

// ---- QuadNodes, one breadth-first level at a time
// may inner node `id` be folded into its parent's QuadNode: its children's boxes lie inside its own (DESIGN B2)
__device__ __forceinline__ bool folds(const ctgpu_node *nodes, uint32_t id) {
    const ctgpu_node &c = nodes[id];
    if (c.count) return false;
    for (uint32_t side = 0; side < 2; side++) {
        const ctgpu_node &g = nodes[c.left + side];
        for (int a = 0; a < 3; a++)
            if (!(g.bmin[a] >= c.bmin[a] && g.bmax[a] <= c.bmax[a])) return false;
    }
    return true;
}
// the entries of the QuadNode headed by reference node `head`, in the reference's visiting order
__device__ __forceinline__ uint32_t quad_entries(const ctgpu_node *nodes, uint32_t head, bool fold, uint32_t entry[4]) {
    uint32_t n = 0;
    for (uint32_t side = 0; side < 2; side++) {
        const uint32_t c = nodes[head].left + side;
        if (fold && folds(nodes, c)) { entry[n++] = nodes[c].left; entry[n++] = nodes[c].left + 1; }
        else entry[n++] = c;
    }
    return n;
}
__global__ void k_quad_count(const ctgpu_node *nodes, const uint32_t *head, uint32_t begin, uint32_t end, int fold,
                             uint32_t *inner_entries) {
    const uint32_t q = begin + blockIdx.x * kBlock + threadIdx.x;
    if (q >= end) return;
    uint32_t entry[4];
    const uint32_t n = quad_entries(nodes, head[q], fold != 0, entry);
    uint32_t inner = 0;
    for (uint32_t k = 0; k < n; k++) inner += nodes[entry[k]].count == 0 ? 1u : 0u;
    inner_entries[q - begin] = inner;
}
struct QuadOut {
    QuadNode *quads;
    double *boxes;            // [quad][4][6] (the folded tree: read on the walk's undecided visits, one dependent fetch)
    uint32_t *node_of;        // [quad][4] the entries' reference nodes (the unfolded tree: its boxes are read from the reference's
                              //           own node array, by the one ray in a million that walks it; NULL for the folded tree)
    uint32_t *head, *before;  // per QuadNode: its reference node; entries pending on the stack when it is reached
    uint32_t *pending_max;
    float bound[3];
};
__global__ void k_quad_emit(const ctgpu_node *nodes, QuadOut out, uint32_t begin, uint32_t end, int fold, const uint32_t *offset) {
    const uint32_t q = begin + blockIdx.x * kBlock + threadIdx.x;
    if (q >= end) return;
    uint32_t entry[4];
    const uint32_t n = quad_entries(nodes, out.head[q], fold != 0, entry);
    QuadNode node;
    double box[4][6];
    uint32_t next = end + offset[q - begin];              // slot of this QuadNode's first inner entry
    const uint32_t before = out.before[q];
    for (uint32_t k = 0; k < 4; k++) {
        if (k < n) {
            const ctgpu_node &e = nodes[entry[k]];
            for (int a = 0; a < 3; a++) {
                node.plane[k][a][0] = (float)e.bmin[a]; node.plane[k][a][1] = (float)e.bmax[a];
                box[k][a] = e.bmin[a]; box[k][3 + a] = e.bmax[a];
            }
            if (e.count) node.ref[k] = kLeafBit | e.first;
            else {
                node.ref[k] = next;
                out.head[next] = entry[k];
                out.before[next] = before + n - 1u - k;
                next += 1;
            }
        } else {
            // absent entry: a one-point box at the corner of the plane bound (never entered, see ct_walk_packet)
            for (int a = 0; a < 3; a++) { node.plane[k][a][0] = node.plane[k][a][1] = out.bound[a]; box[k][a] = 0.0; box[k][3 + a] = 0.0; }
            node.ref[k] = 0u;
        }
    }
    node.count = n;
    node.axis = node.pad_[0] = node.pad_[1] = 0u;
    out.quads[q] = node;
    if (out.boxes) {
        double *dst = out.boxes + (size_t)q * 24;
        for (int i = 0; i < 24; i++) dst[i] = (&box[0][0])[i];
    }
    if (out.node_of)
        for (uint32_t k = 0; k < 4; k++) out.node_of[(size_t)q * 4 + k] = k < n ? entry[k] : 0u;
    atomicMax(out.pending_max, before + n - 1u);
}

// ---- TriRecords (leaf order) and where scene triangle 0 went
__global__ void k_tri_records(const double *verts, const uint32_t *indexes, uint32_t n, TriRecord *recs, uint32_t *slot0) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= n) return;
    const uint32_t tri = indexes[slot];
    const double *v = verts + (size_t)tri * 9;
    TriRecord t;
    for (int c = 0; c < 3; c++) {
        t.p1[c] = v[c];
        t.e1[c] = v[3 + c] - v[c];      // p2 - p1, as bvh.cpp:148
        t.e2[c] = v[6 + c] - v[c];      // p3 - p1, as bvh.cpp:149
    }
    t.triangle = tri;
    t.remaining = 1u;                   // (k_leaf_remaining overwrites it; every slot lies in a leaf)
    recs[slot] = t;
    if (tri == 0u) *slot0 = slot;
}
__global__ void k_leaf_remaining(const ctgpu_node *nodes, uint32_t m, TriRecord *recs) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const uint32_t count = nodes[i].count, first = nodes[i].first;
    for (uint32_t k = 0; k < count; k++) recs[first + k].remaining = count - k;
}

struct Owned {
    std::vector<void *> blocks;
    ~Owned() { for (void *p : blocks) (void)hipFree(p); }
    template <typename T> hipError_t alloc(T *&ptr, size_t count) {
        void *raw = nullptr;
        const hipError_t e = hipMalloc(&raw, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) { ptr = (T *)raw; blocks.push_back(raw); }
        return e;
    }
    void keep(void *p) { blocks.erase(std::find(blocks.begin(), blocks.end(), p)); }     // the caller's from now on
};

// one of the two QuadNode trees; arrays sized for the worst case (every inner node heads a QuadNode) and left so
int build_quads(const ctgpu_node *d_nodes, bool root_is_inner, uint32_t n_inner, bool fold, const float bound[3], Owned &owned,
                const Scanner &scan, uint32_t *d_counts, uint32_t *d_offsets, uint32_t *d_pending, QuadNode *&quads,
                double *&boxes, uint32_t *&node_of, uint32_t &n_quads, uint32_t &root, uint32_t &pending) {
    const uint32_t room = std::max<uint32_t>(n_inner, 1);
    HIP_TRY(owned.alloc(quads, room));
    boxes = nullptr; node_of = nullptr;
    if (fold) HIP_TRY(owned.alloc(boxes, (size_t)room * 24));
    else HIP_TRY(owned.alloc(node_of, (size_t)room * 4));
    root = kLeafBit | 0u;
    pending = 1;
    if (!root_is_inner) {                             // a single leaf: one all-zero QuadNode nobody visits
        HIP_TRY(hipMemsetAsync(quads, 0, sizeof(QuadNode), nullptr));
        if (boxes) HIP_TRY(hipMemsetAsync(boxes, 0, 24 * sizeof(double), nullptr));
        if (node_of) HIP_TRY(hipMemsetAsync(node_of, 0, 4 * sizeof(uint32_t), nullptr));
        n_quads = 1;
        return CTGPU_OK;
    }
    QuadOut out;
    out.quads = quads; out.boxes = boxes; out.node_of = node_of; out.pending_max = d_pending;
    std::memcpy(out.bound, bound, sizeof out.bound);
    HIP_TRY(owned.alloc(out.head, room));
    HIP_TRY(owned.alloc(out.before, room));
    const uint32_t first[2] = {0u, 1u};               // QuadNode 0 is headed by reference node 0, nothing pending; pending starts at 1
    HIP_TRY(hipMemcpy(out.head, &first[0], 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(out.before, &first[0], 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_pending, &first[1], 4, hipMemcpyHostToDevice));
    root = 0;
    uint32_t begin = 0, end = 1;
    for (uint32_t level = 0; begin < end; level++) {
        if (level > 2u * CTGPU_MAX_TREE_DEPTH + 2u) return ctgpu_fail(CTGPU_ERR_ARG, "node links form a cycle");
        const uint32_t width = end - begin;
        hipLaunchKernelGGL(k_quad_count, dim3(blocks_for(width)), dim3(kBlock), 0, nullptr, d_nodes, (const uint32_t *)out.head,
                           begin, end, fold ? 1 : 0, d_counts);
        scan.run(d_counts, width, d_offsets, nullptr);
        HIP_TRY(hipGetLastError());
        // (before anything is emitted: the level's QuadNodes write head / before / quads of the NEXT level's slots, and a
        //  caller's node array that shares children between parents would reach more inner nodes than it has)
        uint32_t made = 0;
        HIP_TRY(hipMemcpy(&made, d_offsets + width, 4, hipMemcpyDeviceToHost));
        if ((uint64_t)end + made > room) return ctgpu_fail(CTGPU_ERR_ARG, "node links form a cycle");
        hipLaunchKernelGGL(k_quad_emit, dim3(blocks_for(width)), dim3(kBlock), 0, nullptr, d_nodes, out, begin, end, fold ? 1 : 0,
                           (const uint32_t *)d_offsets);
        HIP_TRY(hipGetLastError());
        begin = end;
        end += made;
    }
    n_quads = end;
    HIP_TRY(hipMemcpy(&pending, d_pending, 4, hipMemcpyDeviceToHost));
    owned.keep(quads);
    if (boxes) owned.keep(boxes);
    if (node_of) owned.keep(node_of);
    return CTGPU_OK;
}

// ---- cull nodes: the big leaves' own small trees of triangle boxes (ct_device.h "cull nodes", DESIGN appendix D)
// next binary32 value below / above a finite x (zero steps to the smallest denormal of the other / same sign)
__device__ __forceinline__ float float_down(float x) {
    const uint32_t b = __float_as_uint(x);
    if (x > 0.0f) return __uint_as_float(b - 1u);
    if (x < 0.0f) return __uint_as_float(b + 1u);
    return __uint_as_float(0x80000001u);
}
__device__ __forceinline__ float float_up(float x) { return -float_down(-x); }

// One thread per TriRecord slot: the box of the triangle {p1, p1 + e1, p1 + e2} (the vertices the reference's test
// works with, bvh.cpp:148-149), grown on every side by (1.01 / 256) of its extent on that axis -- the 2 * eta * extent of
// appendix D with eta = 2^-9 -- plus 2^-45 of the largest coordinate (which swallows the binary64 roundings made here),
// then rounded OUTWARD to binary32: box[slot] = {lo xyz, hi xyz}.  Also the scene's magnitudes for the gate and the
// filter: largest |e1|, |e2| component, largest |vertex coordinate|, largest |box plane| per axis; a flag for anything
// that is not finite.
__global__ void k_tri_boxes(const TriRecord *recs, uint32_t n, float *box, uint32_t *stats) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    float edge = 0.0f, coord = 0.0f, plane[3] = {0.0f, 0.0f, 0.0f};
    bool bad = false;
    if (slot < n) {
        const TriRecord t = recs[slot];
        for (int a = 0; a < 3; a++) {
            const double v0 = t.p1[a], v1 = t.p1[a] + t.e1[a], v2 = t.p1[a] + t.e2[a];
            const double lo = fmin(v0, fmin(v1, v2)), hi = fmax(v0, fmax(v1, v2));
            const double big = fmax(fabs(lo), fabs(hi));
            const double pad = (hi - lo) * (1.01 / 256.0) + big * 0x1p-45;
            if (!(big <= 1e30) || !(fabs(t.e1[a]) <= 1e30) || !(fabs(t.e2[a]) <= 1e30)) bad = true;      // also NaN
            const float lo_f = float_down((float)(lo - pad)), hi_f = float_up((float)(hi + pad));
            box[(size_t)slot * 6 + a] = lo_f;
            box[(size_t)slot * 6 + 3 + a] = hi_f;
            edge = fmaxf(edge, float_up((float)fmax(fabs(t.e1[a]), fabs(t.e2[a]))));
            coord = fmaxf(coord, float_up((float)big));
            plane[a] = fmaxf(fabsf(lo_f), fabsf(hi_f));
        }
    }
    // (non-negative floats order like their bit patterns)
    uint32_t v[5] = {__float_as_uint(edge), __float_as_uint(coord), __float_as_uint(plane[0]), __float_as_uint(plane[1]), __float_as_uint(plane[2])};
    for (int k = 0; k < 5; k++) {
        for (int off = 32; off > 0; off >>= 1) { const uint32_t other = __shfl_xor(v[k], off); v[k] = other > v[k] ? other : v[k]; }
        if ((threadIdx.x & 63) == 0 && v[k]) atomicMax(stats + k, v[k]);
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicMax(stats + 5, 1u);
}

// A leaf of c triangles becomes a tree of cull nodes: level 1 holds the triangles four to a node (entry = ONE triangle,
// kLeafBit | kOneBit | slot), level l + 1 the level-l nodes four to a node, up to a single root; entries in slot order, so
// the walk's "first entry first, the others wait on the stack" visits the triangles in the leaf's own order.  The nodes of a
// leaf lie together, root first, then each lower level.
constexpr uint32_t kCullMaxLevels = 6;               // leaves of up to 4^6 triangles
constexpr uint32_t kCullMaxLeaf = 4096;
__host__ __device__ inline uint32_t cull_levels(uint32_t c, uint32_t per_level[kCullMaxLevels]) {
    uint32_t levels = 0, width = c;
    do { width = (width + 3u) / 4u; per_level[levels++] = width; } while (width > 1u && levels < kCullMaxLevels);
    return levels;
}
// the leaf an entry of the folded tree refers to, if it is to get cull nodes: entry e = 4 * quad + k; e = 4 * n_quads is
// the tree's root when the whole tree is one leaf
__device__ __forceinline__ bool cull_candidate(const QuadNode *quads, uint32_t n_quads, uint32_t root, const TriRecord *recs, uint32_t n,
                                               uint32_t big_min, uint32_t e, uint32_t &first, uint32_t &count) {
    uint32_t ref;
    if (e == 4u * n_quads) ref = root;
    else {
        const QuadNode &q = quads[e >> 2];
        if ((e & 3u) >= q.count) return false;
        ref = q.ref[e & 3u];
    }
    if (!(ref & kLeafBit)) return false;
    first = ref & kSlotMask;
    if (first >= n) return false;
    count = recs[first].remaining;
    return count >= big_min && count <= kCullMaxLeaf && first + count <= n;
}
__global__ void k_cull_sizes(const QuadNode *quads, uint32_t n_quads, uint32_t root, const TriRecord *recs, uint32_t n, uint32_t big_min,
                             uint32_t *size, uint32_t *stats) {
    const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
    if (e > 4u * n_quads) return;
    uint32_t first, count, nodes = 0;
    if (cull_candidate(quads, n_quads, root, recs, n, big_min, e, first, count)) {
        uint32_t per_level[kCullMaxLevels];
        const uint32_t levels = cull_levels(count, per_level);
        for (uint32_t l = 0; l < levels; l++) nodes += per_level[l];
        atomicMax(stats + 6, levels);
        atomicAdd(stats + 7, 1u);
        atomicAdd(stats + 8, count);
    }
    size[e] = nodes;
}
__global__ void k_cull_emit(QuadNode *quads, uint32_t n_quads, uint32_t root, uint32_t *root_out, const TriRecord *recs, uint32_t n,
                            uint32_t big_min, const uint32_t *offset, const float *box, const float *bound) {
    const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
    if (e > 4u * n_quads) return;
    uint32_t first, count;
    if (!cull_candidate(quads, n_quads, root, recs, n, big_min, e, first, count)) return;
    uint32_t per_level[kCullMaxLevels];
    const uint32_t levels = cull_levels(count, per_level);
    const uint32_t base = n_quads + offset[e];
    // start[l]: where level l + 1's nodes begin (root level first)
    uint32_t start[kCullMaxLevels];
    uint32_t at = base;
    for (int l = (int)levels - 1; l >= 0; l--) { start[l] = at; at += per_level[l]; }
    uint32_t span = 1;                                       // triangles under one entry of the level
    for (uint32_t l = 0; l < levels; l++) {
        const uint32_t children = l == 0 ? count : per_level[l - 1];
        for (uint32_t i = 0; i < per_level[l]; i++) {
            QuadNode node;
            uint32_t used = 0;
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t child = 4u * i + k;
                if (child < children) {
                    const uint32_t t0 = child * span, t1 = (t0 + span < count) ? t0 + span : count;
                    float lo[3], hi[3];
                    for (int a = 0; a < 3; a++) { lo[a] = box[(size_t)(first + t0) * 6 + a]; hi[a] = box[(size_t)(first + t0) * 6 + 3 + a]; }
                    for (uint32_t t = t0 + 1; t < t1; t++)
                        for (int a = 0; a < 3; a++) {
                            lo[a] = fminf(lo[a], box[(size_t)(first + t) * 6 + a]);
                            hi[a] = fmaxf(hi[a], box[(size_t)(first + t) * 6 + 3 + a]);
                        }
                    for (int a = 0; a < 3; a++) { node.plane[k][a][0] = lo[a]; node.plane[k][a][1] = hi[a]; }
                    node.ref[k] = l == 0 ? (kLeafBit | kOneBit | (first + child)) : (kOneBit | (start[l - 1] + child));
                    used += 1;
                } else {
                    // absent entry: a one-point box (never entered), as in every QuadNode
                    for (int a = 0; a < 3; a++) node.plane[k][a][0] = node.plane[k][a][1] = bound[a];
                    node.ref[k] = 0u;
                }
            }
            node.count = used;
            node.axis = node.pad_[0] = node.pad_[1] = 0u;
            quads[start[l] + i] = node;
        }
        span *= 4u;
    }
    // the leaf's reference becomes the reference of its root cull node
    if (e == 4u * n_quads) *root_out = base;
    else quads[e >> 2].ref[e & 3u] = base;
}

}  // namespace

int ctgpu_triangle_boxes_on_device(const void *d_tris, uint32_t n, float *d_box, uint32_t stats[8]) {
    uint32_t *d_stats = nullptr;
    HIP_TRY(hipMalloc((void **)&d_stats, 12 * sizeof(uint32_t)));
    struct Free { uint32_t *p; ~Free() { (void)hipFree(p); } } free_stats{d_stats};
    HIP_TRY(hipMemsetAsync(d_stats, 0, 12 * sizeof(uint32_t), nullptr));
    hipLaunchKernelGGL(k_tri_boxes, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, (const TriRecord *)d_tris, n, d_box, d_stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(stats, d_stats, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CTGPU_OK;
}

// The cull nodes of the folded tree's big leaves (whoever packed the tree: ct_pack's kernels or the host restatement).
// d_quads (n_quads QuadNodes, root reference quad_root) is left as it is; when any leaf gets cull nodes, out->quads is a
// second, longer array -- the tree's own QuadNodes first, unchanged but for the references to those leaves, then the cull
// nodes -- with out->quad_root its root reference.
int ctgpu_pack_cull_on_device(int device, const void *d_tris, uint32_t n, const void *d_quads, uint32_t n_quads, uint32_t quad_root,
                              uint32_t big_min, uint32_t share_percent, const float plane_bound[3], ctgpu_cull_scene *out) {
    if (!d_tris || !d_quads || !out || n == 0) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_pack_cull_on_device: null argument");
    HIP_TRY(hipSetDevice(device));
    Owned owned;
    float *box = nullptr, *d_bound = nullptr;
    uint32_t *d_stats = nullptr, *d_size = nullptr, *d_offset = nullptr, *d_root = nullptr;
    const uint32_t entries = 4u * n_quads + 1u;
    HIP_TRY(owned.alloc(box, (size_t)n * 6));
    HIP_TRY(owned.alloc(d_stats, 12));
    HIP_TRY(owned.alloc(d_size, entries));
    HIP_TRY(owned.alloc(d_offset, (size_t)entries + 1));
    HIP_TRY(owned.alloc(d_root, 1));
    HIP_TRY(owned.alloc(d_bound, 3));
    HIP_TRY(hipMemsetAsync(d_stats, 0, 12 * sizeof(uint32_t), nullptr));
    const TriRecord *recs = (const TriRecord *)d_tris;
    hipLaunchKernelGGL(k_tri_boxes, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, recs, n, box, d_stats);
    HIP_TRY(hipGetLastError());
    uint32_t stats[12];
    HIP_TRY(hipMemcpy(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost));
    std::memcpy(&out->edge_bound, &stats[0], 4);
    std::memcpy(&out->vertex_bound, &stats[1], 4);
    std::memcpy(out->plane_bound, &stats[2], 12);
    out->usable = stats[5] == 0u;
    out->cull_first = kNoNode; out->n_cull = 0; out->levels = 0; out->n_big_leaves = 0;
    if (!out->usable || big_min == 0u) return CTGPU_OK;
    // (the gate of appendix D can never open when K K S > 2^20 with S >= 3 V even for an eye inside the scene: a scene
    //  of such magnitudes -- scene_file_cube.json's 200-unit floor triangles -- would only pay for cull nodes)
    if (!((double)out->edge_bound * out->edge_bound * 3.0 * out->vertex_bound <= 0x1p20)) return CTGPU_OK;

    // the bound the filter will use (and the absent entries sit at): the tree's planes and the triangle boxes'
    float bound[3];
    for (int c = 0; c < 3; c++) bound[c] = std::max(plane_bound[c], std::nextafterf(out->plane_bound[c], INFINITY));
    HIP_TRY(hipMemcpy(d_bound, bound, sizeof bound, hipMemcpyHostToDevice));
    Scanner scan;
    int rc = scan.prepare(entries); if (rc) return rc;
    const QuadNode *quads = (const QuadNode *)d_quads;
    hipLaunchKernelGGL(k_cull_sizes, dim3(blocks_for(entries)), dim3(kBlock), 0, nullptr, quads, n_quads, quad_root, recs, n, big_min, d_size, d_stats);
    scan.run(d_size, entries, d_offset, nullptr);
    HIP_TRY(hipGetLastError());
    uint32_t n_cull = 0;
    HIP_TRY(hipMemcpy(&n_cull, d_offset + entries, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost));
    out->triangles_in_big_leaves = stats[8];
    if (n_cull == 0u) return CTGPU_OK;
    // cull nodes pay where the leaves they replace are where the walk's time goes; a scene with few triangles in big
    // leaves keeps its plain tree and the walk kernels that do not carry the cull code (share_percent of all triangles)
    if ((uint64_t)stats[8] * 100u < (uint64_t)share_percent * n) return CTGPU_OK;
    if ((uint64_t)n_quads + n_cull >= (uint64_t)kOneBit) return CTGPU_OK;                     // (references stay below the flag bits)
    QuadNode *longer = nullptr;
    HIP_TRY(owned.alloc(longer, (size_t)n_quads + n_cull));
    HIP_TRY(hipMemcpy(longer, quads, (size_t)n_quads * sizeof(QuadNode), hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(d_root, &quad_root, 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cull_emit, dim3(blocks_for(entries)), dim3(kBlock), 0, nullptr, longer, n_quads, quad_root, d_root, recs, n, big_min,
                       (const uint32_t *)d_offset, (const float *)box, (const float *)d_bound);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(&out->quad_root, d_root, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipDeviceSynchronize());
    owned.keep(longer);
    out->quads = longer;
    out->cull_first = n_quads;
    out->n_cull = n_cull;
    out->levels = stats[6];
    out->n_big_leaves = stats[7];
    return CTGPU_OK;
}

int ctgpu_pack_on_device(int device, const ctgpu_node *d_nodes, uint32_t m, const uint32_t *d_indexes, const double *d_verts,
                         uint32_t n, ctgpu_packed_scene *out) {
    if (!d_nodes || !d_indexes || !d_verts || !out || m == 0 || n == 0) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_pack_on_device: null argument");
    HIP_TRY(hipSetDevice(device));
    Owned owned;
    struct Events {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, nullptr));

    // ---- the root and the plane bound
    ctgpu_node root_node;
    HIP_TRY(hipMemcpy(&root_node, d_nodes, sizeof root_node, hipMemcpyDeviceToHost));
    std::memcpy(out->root_min, root_node.bmin, 24);
    std::memcpy(out->root_max, root_node.bmax, 24);
    unsigned long long *d_bound = nullptr;
    HIP_TRY(owned.alloc(d_bound, 4));
    HIP_TRY(hipMemsetAsync(d_bound, 0, 4 * sizeof(unsigned long long), nullptr));
    hipLaunchKernelGGL(k_bound, dim3(blocks_for(m)), dim3(kBlock), 0, nullptr, d_nodes, m, d_bound);

    // ---- WideNodes: inner nodes in reference order
    Scanner scan;
    int rc = scan.prepare(m); if (rc) return rc;
    uint32_t *d_flag = nullptr, *d_wide_of = nullptr;
    HIP_TRY(owned.alloc(d_flag, m));
    HIP_TRY(owned.alloc(d_wide_of, (size_t)m + 1));
    hipLaunchKernelGGL(k_inner_flags, dim3(blocks_for(m)), dim3(kBlock), 0, nullptr, d_nodes, m, d_flag);
    scan.run(d_flag, m, d_wide_of, nullptr);
    HIP_TRY(hipGetLastError());
    uint32_t n_inner = 0;
    HIP_TRY(hipMemcpy(&n_inner, d_wide_of + m, 4, hipMemcpyDeviceToHost));
    unsigned long long bound_bits[4];
    HIP_TRY(hipMemcpy(bound_bits, d_bound, sizeof bound_bits, hipMemcpyDeviceToHost));
    for (int c = 0; c < 3; c++) {
        double big;
        std::memcpy(&big, &bound_bits[c], 8);
        out->plane_bound[c] = bound_bits[3] ? INFINITY : std::nextafterf((float)big, INFINITY);
    }
    WideNode *wide = nullptr;
    out->n_wide = std::max<uint32_t>(n_inner, 1);
    HIP_TRY(owned.alloc(wide, out->n_wide));
    if (n_inner == 0) HIP_TRY(hipMemsetAsync(wide, 0, sizeof(WideNode), nullptr));
    hipLaunchKernelGGL(k_wide, dim3(blocks_for(m)), dim3(kBlock), 0, nullptr, d_nodes, m, (const uint32_t *)d_wide_of, wide);
    const bool root_is_inner = root_node.count == 0;
    out->root_ref = root_is_inner ? 0u : (kLeafBit | root_node.first);       // (inner root: WideNode 0)

    // ---- the two QuadNode trees
    uint32_t *d_counts = nullptr, *d_offsets = nullptr, *d_pending = nullptr;
    HIP_TRY(owned.alloc(d_counts, std::max<uint32_t>(n_inner, 1)));
    HIP_TRY(owned.alloc(d_offsets, (size_t)std::max<uint32_t>(n_inner, 1) + 1));
    HIP_TRY(owned.alloc(d_pending, 1));
    QuadNode *quads = nullptr, *quads_flat = nullptr;
    uint32_t pending = 1, pending_flat = 1;
    uint32_t *no_ids = nullptr;
    double *no_boxes = nullptr;
    rc = build_quads(d_nodes, root_is_inner, n_inner, true, out->plane_bound, owned, scan, d_counts, d_offsets, d_pending, quads,
                     out->quad_boxes, no_ids, out->n_quads, out->quad_root, pending);
    if (rc) return rc;
    rc = build_quads(d_nodes, root_is_inner, n_inner, false, out->plane_bound, owned, scan, d_counts, d_offsets, d_pending, quads_flat,
                     no_boxes, out->quad_node_of_flat, out->n_quads_flat, out->quad_root_flat, pending_flat);
    if (rc) return rc;
    if (!root_is_inner) { owned.keep(quads); owned.keep(out->quad_boxes); owned.keep(quads_flat); owned.keep(out->quad_node_of_flat); }
    out->quads = quads;
    out->quads_flat = quads_flat;
    out->quad_pending = std::max(pending, pending_flat);

    // ---- TriRecords
    TriRecord *recs = nullptr;
    uint32_t *d_slot0 = nullptr;
    HIP_TRY(owned.alloc(recs, n));
    HIP_TRY(owned.alloc(d_slot0, 1));
    HIP_TRY(hipMemsetAsync(d_slot0, 0, 4, nullptr));
    hipLaunchKernelGGL(k_tri_records, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, d_verts, d_indexes, n, recs, d_slot0);
    hipLaunchKernelGGL(k_leaf_remaining, dim3(blocks_for(m)), dim3(kBlock), 0, nullptr, d_nodes, m, recs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(&out->slot_of_triangle0, d_slot0, 4, hipMemcpyDeviceToHost));

    HIP_TRY(hipEventRecord(ev.b, nullptr));
    HIP_TRY(hipEventSynchronize(ev.b));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    out->device_ms = ms;
    owned.keep(wide);
    owned.keep(recs);
    out->wide = wide;
    out->tris = recs;
    return CTGPU_OK;
}
