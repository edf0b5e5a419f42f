// ct_build.hip -- BuildBVH (bvh.cpp:16-121) on the device, node for node and index for index the tree the
// reference's recursion builds.
//
// The reference subdivides depth first: a node with more than two triangles picks the longest axis of its box,
// the binary32 midpoint of that extent, and partitions its stretch of the index array IN PLACE with a two-pointer
// swap loop (bvh.cpp:82-93); if both sides are non-empty the children get the next two node slots and their boxes
// are folded from their triangles' corners with the a<b?a:b macros (bvh.cpp:30-49).  Three things of that are
// order-dependent and have to come out identical here, level by level in parallel:
//
//  * the permutation the swap loop leaves.  It has a closed form.  Call an element "left" when its centroid
//    component is below the plane.  Let r_1 < r_2 < ... be the positions of the right elements from the front and
//    l_1 > l_2 > ... those of the left elements from the back, M = #{m : r_m < l_m}, l_0 = end.  Then the loop moves
//        a[l_m] -> r_m                 for m <= M            (a left element from the back fills the hole)
//        a[r_m] -> l_{m-1} - 1         for m <= M, and for m = M+1 if r_{M+1} < l_M
//        every other right element     one position down (q -> q - 1)
//    and nothing else (verified against the loop itself on 300k random cases, tools/partition_closed_form.py; the
//    tests compare whole trees).  So: one prefix count, two compactions, one binary search per node, one scatter.
//    The loop also runs -- and permutes -- when it then finds one side empty and the node stays a leaf.
//  * the node numbering: children get consecutive slots when their parent is subdivided, parents are subdivided
//    in depth-first pre-order, so left child = 1 + 2 x (inner nodes before the parent in pre-order).  Subtree
//    counts bottom-up, pre-order ranks top-down, then the nodes are written to their slots.
//  * the boxes: lesser(a, b) = a < b ? a : b returns b on a tie, so of -0.0 and +0.0 the LATER corner wins.  That
//    fold ("minimum, last one among equals") is associative, so it reduces in parallel as long as the operands
//    stay in sequence: every thread folds a contiguous run, partial results are combined in run order.
//
// Work per level is O(n) over flat arrays (a few dozen kernel launches per level, ~27 levels for the dragon
// stand-in); everything stays in HBM until the finished node and index arrays are copied out.
#include "ctgpu_internal.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kSmallSegment = 48;        // boxes of nodes up to this many triangles: one thread each
constexpr uint32_t kHugeSegment = 4096;       // above this: 256 blocks fold a node's triangles, one more combines them

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return ctgpu_fail(CTGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct BuildArrays {
    const double *verts;       // [n][9]
    double *cent;              // [n][3]
    uint32_t *idx_in, *idx_out;
    int32_t *seg, *seg_next;   // active slot of a position, -1: its node is finished
    uint32_t *right_flag;      // [n] 1: the element belongs right (active positions only)
    uint32_t *right_before;    // [n + 1] exclusive prefix count of right_flag
    uint32_t *block_sums;      // scan scratch
    uint32_t *rpos, *lpos;     // [n] positions of right elements from the front / left elements from the back, per segment
    // nodes, temporary numbering (creation order)
    uint32_t *t_first, *t_count, *t_left;      // t_left: temporary id of the left child, 0 = leaf
    int32_t *t_slot;           // active slot of a node in the NEXT level (-1: finished)
    double *t_min, *t_max;     // [nodes][3]
    uint32_t *t_inner_below;   // inner nodes in the subtree (the node included)
    uint32_t *t_rank;          // pre-order rank among inner nodes
    uint32_t *t_final;         // final slot
    // active nodes of the current level
    uint32_t *a_node, *a_node_next;            // temporary ids
    int32_t *a_axis;
    float *a_plane;
    uint32_t *a_left_count, *a_pairs;          // nL, M
    // work lists of the boxes to fit
    uint32_t *box_small, *box_big, *box_huge;
    double *huge_runs;         // [huge nodes][256][6] partial boxes
    uint32_t *counters;        // [0] nodes used, [1] next level's active nodes, [2] small, [3] big, [4] huge boxes; [7] a NaN vertex was seen
    uint32_t n;
};

// ---- the reference's comparisons
__device__ __forceinline__ double lesser_d(double a, double b) { return (a < b) ? a : b; }      // mymath.h:15-17
__device__ __forceinline__ double greater_d(double a, double b) { return (a > b) ? a : b; }     // mymath.h:11-13

__global__ void k_centroids(BuildArrays A) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= A.n) return;
    const double third = (double)0.3333f;              // bvh.cpp:111: (p1 + p2 + p3) * 0.3333f
    const double *v = A.verts + (size_t)i * 9;
    for (int c = 0; c < 3; c++) A.cent[(size_t)i * 3 + c] = ((v[c] + v[3 + c]) + v[6 + c]) * third;
    bool nan = false;
    for (int c = 0; c < 9; c++) nan = nan || (v[c] != v[c]);
    if (nan) A.counters[7] = 1u;                       // the ordered box folds below assume NaN-free corners (see ctgpu_build_bvh)
    A.idx_in[i] = i;
    A.seg[i] = (A.n > 2) ? 0 : -1;
}

// ---- per active node: axis and plane (bvh.cpp:60-66; V3ByIndex returns float)
__global__ void k_plan(BuildArrays A, uint32_t n_active) {
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_active) return;
    const uint32_t a = A.a_node[s];
    const double *lo = A.t_min + (size_t)a * 3, *hi = A.t_max + (size_t)a * 3;
    const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    int axis = 0;
    if (ey > ex) axis = 1;
    const double chosen = (axis == 0) ? ex : ey;
    if (ez > (double)(float)chosen) axis = 2;
    const double extent = (axis == 0) ? ex : (axis == 1 ? ey : ez);
    A.a_axis[s] = axis;
    A.a_plane[s] = (float)lo[axis] + (float)extent * 0.5f;
}

__global__ void k_classify(BuildArrays A) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= A.n) return;
    const int32_t s = A.seg[p];
    uint32_t right = 0;
    if (s >= 0) {
        const uint32_t tri = A.idx_in[p];
        right = ((float)A.cent[(size_t)tri * 3 + A.a_axis[s]] < A.a_plane[s]) ? 0u : 1u;
    }
    A.right_flag[p] = right;
}

// ---- exclusive prefix sum of right_flag (n + 1 entries): per-block sums, their scan, the final pass
constexpr int kScanItems = 4;                          // per thread
__global__ void k_scan_blocks(BuildArrays A) {
    __shared__ uint32_t wave_sums[kBlock / 64];
    const uint32_t base = blockIdx.x * kBlock * kScanItems + threadIdx.x * kScanItems;
    uint32_t sum = 0;
    for (int k = 0; k < kScanItems; k++)
        if (base + k < A.n) sum += A.right_flag[base + k];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) A.block_sums[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
}
__global__ void k_scan_sums(BuildArrays A, uint32_t n_blocks) {          // one block: a serial carry over chunks of kBlock
    __shared__ uint32_t tile[kBlock];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < n_blocks; start += kBlock) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = (i < n_blocks) ? A.block_sums[i] : 0u;
        tile[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {                     // inclusive scan of the tile
            const uint32_t add = (threadIdx.x >= (uint32_t)off) ? tile[threadIdx.x - off] : 0u;
            __syncthreads();
            tile[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_blocks) A.block_sums[i] = carry + tile[threadIdx.x] - v;   // exclusive
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry += tile[kBlock - 1];
        __syncthreads();
    }
}
__global__ void k_scan_final(BuildArrays A) {
    __shared__ uint32_t thread_sums[kBlock];
    const uint32_t base = blockIdx.x * kBlock * kScanItems + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], sum = 0;
    for (int k = 0; k < kScanItems; k++) {
        v[k] = (base + k < A.n) ? A.right_flag[base + k] : 0u;
        sum += v[k];
    }
    thread_sums[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        const uint32_t add = (threadIdx.x >= (uint32_t)off) ? thread_sums[threadIdx.x - off] : 0u;
        __syncthreads();
        thread_sums[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t running = A.block_sums[blockIdx.x] + thread_sums[threadIdx.x] - sum;
    for (int k = 0; k < kScanItems; k++) {
        if (base + k <= A.n) A.right_before[base + k] = running;           // (entry n holds the total)
        running += v[k];
    }
}

// ---- positions of the right elements from the front and of the left elements from the back, per segment
__global__ void k_lists(BuildArrays A) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= A.n) return;
    const int32_t s = A.seg[p];
    if (s < 0) return;
    const uint32_t a = A.a_node[s], first = A.t_first[a], count = A.t_count[a];
    const uint32_t rights_before = A.right_before[p] - A.right_before[first];
    const uint32_t n_right = A.right_before[first + count] - A.right_before[first], n_left = count - n_right;
    if (A.right_flag[p]) {
        A.rpos[first + rights_before] = p;
    } else {
        const uint32_t lefts_before = (p - first) - rights_before;
        A.lpos[first + (n_left - 1u - lefts_before)] = p;
    }
}

// ---- per active node: how many (front right, back left) pairs the swap loop exchanges
__global__ void k_pairs(BuildArrays A, uint32_t n_active) {
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_active) return;
    const uint32_t a = A.a_node[s], first = A.t_first[a], count = A.t_count[a];
    const uint32_t n_right = A.right_before[first + count] - A.right_before[first], n_left = count - n_right;
    uint32_t lo = 0, hi = n_right < n_left ? n_right : n_left;          // r_m < l_m holds for m < M, fails from M on
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (A.rpos[first + mid] < A.lpos[first + mid]) lo = mid + 1; else hi = mid;
    }
    A.a_left_count[s] = n_left;
    A.a_pairs[s] = lo;
}

__global__ void k_scatter(BuildArrays A) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= A.n) return;
    const int32_t s = A.seg[p];
    uint32_t dst = p;
    if (s >= 0) {
        const uint32_t a = A.a_node[s], first = A.t_first[a], count = A.t_count[a], end = first + count;
        const uint32_t n_left = A.a_left_count[s], pairs = A.a_pairs[s];
        const uint32_t rights_before = A.right_before[p] - A.right_before[first];
        if (A.right_flag[p]) {
            const uint32_t m = rights_before;                                // 0-based rank from the front
            const uint32_t last_paired_left = pairs ? A.lpos[first + pairs - 1u] : end;
            if (m < pairs || (m == pairs && p < last_paired_left)) dst = (m ? A.lpos[first + m - 1u] : end) - 1u;
            else dst = p - 1u;
        } else {
            const uint32_t from_back = n_left - 1u - ((p - first) - rights_before);
            if (from_back < pairs) dst = A.rpos[first + from_back];
        }
    }
    A.idx_out[dst] = A.idx_in[p];
}

// ---- per active node: children (bvh.cpp:95-104), the next level's active list, the boxes to fit
// One atomic per wave and counter instead of one per lane: an atomic on a single word takes ~11 ns, and the deep
// levels have hundreds of thousands of nodes asking.  Returns this lane's first slot (meaningful where `amount` > 0).
__device__ __forceinline__ uint32_t wave_reserve(uint32_t *counter, uint32_t amount) {
    uint32_t before = amount;                                   // inclusive prefix over the wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(before, off);
        if ((int)(threadIdx.x & 63) >= off) before += v;
    }
    const uint32_t total = __shfl(before, 63);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 63 && total) base = atomicAdd(counter, total);
    base = __shfl(base, 63);
    return base + before - amount;
}

__global__ void k_children(BuildArrays A, uint32_t n_active) {
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    const bool live = s < n_active;
    uint32_t a = 0, first = 0, count = 0, n_left = 0;
    if (live) {
        a = A.a_node[s]; first = A.t_first[a]; count = A.t_count[a];
        n_left = A.a_left_count[s];
    }
    // a side stayed empty: the node remains a leaf (its indexes were permuted all the same)
    const bool splits = live && !(n_left == 0 || n_left == count);
    if (live && !splits) A.t_left[a] = 0;
    const uint32_t counts[2] = {n_left, count - n_left}, firsts[2] = {first, first + n_left};
    const uint32_t left = wave_reserve(&A.counters[0], splits ? 2u : 0u);
    uint32_t n_next = 0, n_small = 0, n_big = 0, n_huge = 0;
    for (int c = 0; c < 2; c++) {
        if (!splits) break;
        n_next += counts[c] > 2u ? 1u : 0u;                                    // bvh.cpp:53: two or fewer stay a leaf
        if (counts[c] <= kSmallSegment) n_small += 1; else if (counts[c] <= kHugeSegment) n_big += 1; else n_huge += 1;
    }
    uint32_t slot_next = wave_reserve(&A.counters[1], n_next), slot_small = wave_reserve(&A.counters[2], n_small),
             slot_big = wave_reserve(&A.counters[3], n_big), slot_huge = wave_reserve(&A.counters[4], n_huge);
    if (!splits) return;
    A.t_left[a] = left;
    for (int c = 0; c < 2; c++) {
        const uint32_t child = left + c;
        A.t_first[child] = firsts[c];
        A.t_count[child] = counts[c];
        A.t_left[child] = 0;
        int32_t slot = -1;
        if (counts[c] > 2u) {
            slot = (int32_t)slot_next++;
            A.a_node_next[slot] = child;
        }
        A.t_slot[child] = slot;
        if (counts[c] <= kSmallSegment) A.box_small[slot_small++] = child;
        else if (counts[c] <= kHugeSegment) A.box_big[slot_big++] = child;
        else A.box_huge[slot_huge++] = child;
    }
}

__global__ void k_next_segments(BuildArrays A) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= A.n) return;
    const int32_t s = A.seg[p];
    int32_t next = -1;
    if (s >= 0) {
        const uint32_t a = A.a_node[s], left = A.t_left[a];
        if (left) next = A.t_slot[(p < A.t_first[a] + A.a_left_count[s]) ? left : left + 1u];
    }
    A.seg_next[p] = next;
}

// ---- boxes (bvh.cpp:30-49): per coordinate the fold  lo = lesser(lo, corner), hi = greater(hi, corner)  over the
// node's triangles in index order, corners p1 p2 p3, from +-1e30f.  "The later one among equals" is kept by folding
// contiguous runs and combining them in order.
struct Box { double lo[3], hi[3]; };
__device__ __forceinline__ void box_start(Box &b) {
    const double big = (double)1e30f;
    for (int c = 0; c < 3; c++) { b.lo[c] = big; b.hi[c] = -big; }
}
__device__ __forceinline__ void box_add_triangle(Box &b, const double *v) {
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) b.lo[c] = lesser_d(b.lo[c], v[k * 3 + c]);
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) b.hi[c] = greater_d(b.hi[c], v[k * 3 + c]);
}
__device__ __forceinline__ void box_append(Box &earlier, const Box &later) {      // fold(earlier run ++ later run)
    for (int c = 0; c < 3; c++) {
        earlier.lo[c] = lesser_d(earlier.lo[c], later.lo[c]);
        earlier.hi[c] = greater_d(earlier.hi[c], later.hi[c]);
    }
}
__device__ __forceinline__ void box_store(const BuildArrays &A, uint32_t node, const Box &b) {
    for (int c = 0; c < 3; c++) { A.t_min[(size_t)node * 3 + c] = b.lo[c]; A.t_max[(size_t)node * 3 + c] = b.hi[c]; }
}

__global__ void k_boxes_small(BuildArrays A, uint32_t n_nodes, const uint32_t *idx) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t node = A.box_small[i], first = A.t_first[node], count = A.t_count[node];
    Box b;
    box_start(b);
    for (uint32_t k = 0; k < count; k++) box_add_triangle(b, A.verts + (size_t)idx[first + k] * 9);
    box_store(A, node, b);
}

__global__ void k_boxes_big(BuildArrays A, const uint32_t *idx) {                // one block per node
    __shared__ Box runs[kBlock];
    const uint32_t node = A.box_big[blockIdx.x], first = A.t_first[node], count = A.t_count[node];
    const uint32_t per_thread = (count + kBlock - 1) / kBlock;
    const uint32_t lo = threadIdx.x * per_thread, hi = (lo + per_thread < count) ? lo + per_thread : count;
    Box b;
    box_start(b);                        // (a run that starts from +-1e30f again: harmless, 1e30f never beats a tie it should lose)
    for (uint32_t k = lo; k < hi; k++) box_add_triangle(b, A.verts + (size_t)idx[first + k] * 9);
    runs[threadIdx.x] = b;
    __syncthreads();
    for (int width = 1; width < kBlock; width <<= 1) {                           // ordered tree: run i takes in run i + width
        if ((threadIdx.x % (2 * width)) == 0 && threadIdx.x + width < kBlock) box_append(runs[threadIdx.x], runs[threadIdx.x + width]);
        __syncthreads();
    }
    if (threadIdx.x == 0) box_store(A, node, runs[0]);
}

// nodes of many triangles: 256 blocks fold consecutive stretches, one block puts their results together in order
__global__ void k_boxes_huge_runs(BuildArrays A, const uint32_t *idx) {           // grid (256, huge nodes)
    __shared__ Box runs[kBlock];
    const uint32_t node = A.box_huge[blockIdx.y], first = A.t_first[node], count = A.t_count[node];
    const uint32_t per_block = (count + 255u) / 256u;
    const uint32_t b_lo = blockIdx.x * per_block, b_hi = (b_lo + per_block < count) ? b_lo + per_block : count;
    const uint32_t span = b_hi > b_lo ? b_hi - b_lo : 0u;
    const uint32_t per_thread = (span + kBlock - 1) / kBlock;
    const uint32_t lo = b_lo + threadIdx.x * per_thread, hi = (lo + per_thread < b_hi) ? lo + per_thread : b_hi;
    Box b;
    box_start(b);
    for (uint32_t k = lo; k < hi; k++) box_add_triangle(b, A.verts + (size_t)idx[first + k] * 9);
    runs[threadIdx.x] = b;
    __syncthreads();
    for (int width = 1; width < kBlock; width <<= 1) {
        if ((threadIdx.x % (2 * width)) == 0 && threadIdx.x + width < kBlock) box_append(runs[threadIdx.x], runs[threadIdx.x + width]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double *out = A.huge_runs + ((size_t)blockIdx.y * 256u + blockIdx.x) * 6u;
        for (int c = 0; c < 3; c++) { out[c] = runs[0].lo[c]; out[3 + c] = runs[0].hi[c]; }
    }
}
__global__ void k_boxes_huge_join(BuildArrays A) {                                // one block of 256 per huge node
    __shared__ Box runs[kBlock];
    const double *in = A.huge_runs + ((size_t)blockIdx.x * 256u + threadIdx.x) * 6u;
    for (int c = 0; c < 3; c++) { runs[threadIdx.x].lo[c] = in[c]; runs[threadIdx.x].hi[c] = in[3 + c]; }
    __syncthreads();
    for (int width = 1; width < kBlock; width <<= 1) {
        if ((threadIdx.x % (2 * width)) == 0 && threadIdx.x + width < kBlock) box_append(runs[threadIdx.x], runs[threadIdx.x + width]);
        __syncthreads();
    }
    if (threadIdx.x == 0) box_store(A, A.box_huge[blockIdx.x], runs[0]);
}

// ---- numbering: inner nodes below (bottom-up), pre-order rank and final slot (top-down), then the nodes themselves
__global__ void k_count_inner(BuildArrays A, uint32_t begin, uint32_t end) {
    const uint32_t a = begin + blockIdx.x * kBlock + threadIdx.x;
    if (a >= end) return;
    const uint32_t left = A.t_left[a];
    A.t_inner_below[a] = left ? 1u + A.t_inner_below[left] + A.t_inner_below[left + 1u] : 0u;
}
__global__ void k_number(BuildArrays A, uint32_t begin, uint32_t end) {
    const uint32_t a = begin + blockIdx.x * kBlock + threadIdx.x;
    if (a >= end) return;
    const uint32_t left = A.t_left[a];
    if (!left) return;
    const uint32_t rank = A.t_rank[a];
    A.t_final[left] = 1u + 2u * rank;            // children take the next two slots when their parent is subdivided
    A.t_final[left + 1u] = 2u + 2u * rank;
    A.t_rank[left] = rank + 1u;                  // pre-order: the left subtree's inner nodes come first
    A.t_rank[left + 1u] = rank + 1u + A.t_inner_below[left];
}
__global__ void k_emit(BuildArrays A, uint32_t n_nodes, ctgpu_node *out) {
    const uint32_t a = blockIdx.x * kBlock + threadIdx.x;
    if (a >= n_nodes) return;
    ctgpu_node node;
    for (int c = 0; c < 3; c++) { node.bmin[c] = A.t_min[(size_t)a * 3 + c]; node.bmax[c] = A.t_max[(size_t)a * 3 + c]; }
    const uint32_t left = A.t_left[a];
    node.left = left ? A.t_final[left] : 0u;
    node.first = A.t_first[a];                   // (an inner node keeps the first index it had as a leaf, bvh.cpp:99-104)
    node.count = left ? 0u : A.t_count[a];
    node.pad_ = 0u;
    out[A.t_final[a]] = node;
}

template <typename T>
hipError_t device_alloc(T *&ptr, size_t count, std::vector<void *> &owned) {
    void *raw = nullptr;
    const hipError_t e = hipMalloc(&raw, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) { ptr = (T *)raw; owned.push_back(raw); }
    return e;
}

inline unsigned blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

// BuildBVH on vertices that are already in HBM; the finished arrays stay there (see ctgpu_internal.h).
//
// Precondition: no NaN among the vertex coordinates.  The reference folds a node's box with a < b ? a : b in
// sequence, where a NaN corner RESTARTS the fold (lesser(x, NaN) = NaN, lesser(NaN, c) = c); the ordered parallel
// fold here does not reproduce that restart, so a mesh with a NaN vertex is refused (CTGPU_ERR_ARG) and the caller
// builds that tree on the host (host/bvh_build.cpp follows the sequence literally).  Infinities are fine.
int ctgpu_build_bvh_resident(int device, const double *d_verts, uint32_t n, ctgpu_node **d_nodes_out, uint32_t **d_indexes_out,
                             uint32_t *n_nodes_out, uint32_t *depth_out, double *device_ms) {
    if (!d_verts || !d_nodes_out || !d_indexes_out || !n_nodes_out || n == 0)
        return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_build_bvh: null argument or no triangles");
    if (n > (1u << 30)) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_build_bvh: %u triangles are too many", n);
    HIP_TRY(hipSetDevice(device));

    std::vector<void *> owned;
    struct Release {
        std::vector<void *> &v;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        ~Release() {
            for (void *p : v) (void)hipFree(p);
            if (ev0) (void)hipEventDestroy(ev0);
            if (ev1) (void)hipEventDestroy(ev1);
        }
    } release{owned};
    BuildArrays A;
    std::memset(&A, 0, sizeof A);
    A.n = n;
    const size_t max_nodes = 2 * (size_t)n;
    ctgpu_node *d_out = nullptr;
    HIP_TRY(device_alloc(A.cent, (size_t)n * 3, owned));
    HIP_TRY(device_alloc(A.idx_in, n, owned));      HIP_TRY(device_alloc(A.idx_out, n, owned));
    HIP_TRY(device_alloc(A.seg, n, owned));         HIP_TRY(device_alloc(A.seg_next, n, owned));
    HIP_TRY(device_alloc(A.right_flag, n, owned));  HIP_TRY(device_alloc(A.right_before, (size_t)n + 1, owned));
    const uint32_t scan_blocks = (n + kBlock * kScanItems) / (kBlock * kScanItems) + 1;
    HIP_TRY(device_alloc(A.block_sums, scan_blocks, owned));
    HIP_TRY(device_alloc(A.rpos, n, owned));        HIP_TRY(device_alloc(A.lpos, n, owned));
    HIP_TRY(device_alloc(A.t_first, max_nodes, owned)); HIP_TRY(device_alloc(A.t_count, max_nodes, owned));
    HIP_TRY(device_alloc(A.t_left, max_nodes, owned));  HIP_TRY(device_alloc(A.t_slot, max_nodes, owned));
    HIP_TRY(device_alloc(A.t_min, max_nodes * 3, owned)); HIP_TRY(device_alloc(A.t_max, max_nodes * 3, owned));
    HIP_TRY(device_alloc(A.t_inner_below, max_nodes, owned)); HIP_TRY(device_alloc(A.t_rank, max_nodes, owned));
    HIP_TRY(device_alloc(A.t_final, max_nodes, owned));
    HIP_TRY(device_alloc(A.a_node, n, owned));      HIP_TRY(device_alloc(A.a_node_next, n, owned));
    HIP_TRY(device_alloc(A.a_axis, n, owned));      HIP_TRY(device_alloc(A.a_plane, n, owned));
    HIP_TRY(device_alloc(A.a_left_count, n, owned)); HIP_TRY(device_alloc(A.a_pairs, n, owned));
    HIP_TRY(device_alloc(A.box_small, max_nodes, owned)); HIP_TRY(device_alloc(A.box_big, max_nodes, owned));
    const size_t max_huge = (size_t)n / kHugeSegment + 2;
    HIP_TRY(device_alloc(A.box_huge, max_huge, owned));
    HIP_TRY(device_alloc(A.huge_runs, max_huge * 256 * 6, owned));
    HIP_TRY(device_alloc(A.counters, 8, owned));
    HIP_TRY(hipMemset(A.counters, 0, 8 * sizeof(uint32_t)));
    A.verts = d_verts;

    HIP_TRY(hipEventCreate(&release.ev0));
    HIP_TRY(hipEventCreate(&release.ev1));
    HIP_TRY(hipEventRecord(release.ev0, nullptr));

    // root: node 0 over all triangles (bvh.cpp:113-118)
    const uint32_t root_setup[3] = {0u, n, 0u};
    HIP_TRY(hipMemcpy(A.t_first, &root_setup[0], 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(A.t_count, &root_setup[1], 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(A.t_left, &root_setup[2], 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(A.a_node, &root_setup[0], 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(A.box_huge, &root_setup[0], 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_centroids, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, A);
    hipLaunchKernelGGL(k_boxes_huge_runs, dim3(256, 1), dim3(kBlock), 0, nullptr, A, (const uint32_t *)A.idx_in);
    hipLaunchKernelGGL(k_boxes_huge_join, dim3(1), dim3(kBlock), 0, nullptr, A);

    std::vector<uint32_t> level_begin{0u};          // temporary ids of the nodes created at each depth
    uint32_t nodes_used = 1, n_active = (n > 2) ? 1u : 0u;
    while (n_active) {
        level_begin.push_back(nodes_used);
        // (a tree deeper than the walks' stacks is reported here, by the build, with the depth it reached)
        if (level_begin.size() > CTGPU_MAX_TREE_DEPTH + 1)
            return ctgpu_fail(CTGPU_ERR_STACK, "ctgpu_build_bvh: the tree is deeper than %d", CTGPU_MAX_TREE_DEPTH);
        const uint32_t counters[5] = {nodes_used, 0u, 0u, 0u, 0u};
        HIP_TRY(hipMemcpy(A.counters, counters, sizeof counters, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_plan, dim3(blocks_for(n_active)), dim3(kBlock), 0, nullptr, A, n_active);
        hipLaunchKernelGGL(k_classify, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, A);
        hipLaunchKernelGGL(k_scan_blocks, dim3(scan_blocks), dim3(kBlock), 0, nullptr, A);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, nullptr, A, scan_blocks);
        hipLaunchKernelGGL(k_scan_final, dim3(scan_blocks), dim3(kBlock), 0, nullptr, A);
        hipLaunchKernelGGL(k_lists, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, A);
        hipLaunchKernelGGL(k_pairs, dim3(blocks_for(n_active)), dim3(kBlock), 0, nullptr, A, n_active);
        hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, A);
        hipLaunchKernelGGL(k_children, dim3(blocks_for(n_active)), dim3(kBlock), 0, nullptr, A, n_active);
        hipLaunchKernelGGL(k_next_segments, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, A);
        uint32_t after[8];
        HIP_TRY(hipMemcpy(after, A.counters, sizeof after, hipMemcpyDeviceToHost));
        if (after[7]) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_build_bvh: a vertex coordinate is NaN (build this tree on the host)");
        if (after[2]) hipLaunchKernelGGL(k_boxes_small, dim3(blocks_for(after[2])), dim3(kBlock), 0, nullptr, A, after[2], (const uint32_t *)A.idx_out);
        if (after[3]) hipLaunchKernelGGL(k_boxes_big, dim3(after[3]), dim3(kBlock), 0, nullptr, A, (const uint32_t *)A.idx_out);
        if (after[4]) {
            hipLaunchKernelGGL(k_boxes_huge_runs, dim3(256, after[4]), dim3(kBlock), 0, nullptr, A, (const uint32_t *)A.idx_out);
            hipLaunchKernelGGL(k_boxes_huge_join, dim3(after[4]), dim3(kBlock), 0, nullptr, A);
        }
        HIP_TRY(hipGetLastError());
        nodes_used = after[0];
        n_active = after[1];
        std::swap(A.idx_in, A.idx_out);
        std::swap(A.seg, A.seg_next);
        std::swap(A.a_node, A.a_node_next);
    }
    level_begin.push_back(nodes_used);               // level_begin[d] .. level_begin[d + 1]: nodes of depth d
    {
        uint32_t nan_seen = 0;                       // (trees of one or two triangles never entered the loop)
        HIP_TRY(hipMemcpy(&nan_seen, A.counters + 7, 4, hipMemcpyDeviceToHost));
        if (nan_seen) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_build_bvh: a vertex coordinate is NaN (build this tree on the host)");
    }

    // numbering in the reference's allocation order
    HIP_TRY(device_alloc(d_out, nodes_used, owned));
    const size_t depths = level_begin.size() - 1;
    for (size_t d = depths; d-- > 0;) {
        const uint32_t b = level_begin[d], e = level_begin[d + 1];
        if (e > b) hipLaunchKernelGGL(k_count_inner, dim3(blocks_for(e - b)), dim3(kBlock), 0, nullptr, A, b, e);
    }
    const uint32_t zero = 0;
    HIP_TRY(hipMemcpy(A.t_rank, &zero, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(A.t_final, &zero, 4, hipMemcpyHostToDevice));
    for (size_t d = 0; d < depths; d++) {
        const uint32_t b = level_begin[d], e = level_begin[d + 1];
        if (e > b) hipLaunchKernelGGL(k_number, dim3(blocks_for(e - b)), dim3(kBlock), 0, nullptr, A, b, e);
    }
    hipLaunchKernelGGL(k_emit, dim3(blocks_for(nodes_used)), dim3(kBlock), 0, nullptr, A, nodes_used, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(release.ev1, nullptr));
    HIP_TRY(hipEventSynchronize(release.ev1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, release.ev0, release.ev1));

    // the two results leave the scratch list: they are the caller's now
    for (void *keep : {(void *)d_out, (void *)A.idx_in})
        owned.erase(std::find(owned.begin(), owned.end(), keep));
    *d_nodes_out = d_out;
    *d_indexes_out = A.idx_in;
    *n_nodes_out = nodes_used;
    if (depth_out) {                                 // levels that hold a node (the last pass may have split nothing)
        uint32_t levels = 0;
        for (size_t d = 0; d < depths; d++) levels += level_begin[d + 1] > level_begin[d] ? 1u : 0u;
        *depth_out = levels;
    }
    if (device_ms) *device_ms = ms;
    return CTGPU_OK;
}

// nodes_out: room for 2n - 1 nodes; indexes_out: n.  *device_ms (optional): time of the build on the device, copies
// excluded.  Returns the reference's arrays: node i is bvhNodes[i], indexes_out is triangles.indexes.
extern "C" int ctgpu_build_bvh(int device, const double *vertices, uint32_t n, ctgpu_node *nodes_out, uint32_t *indexes_out,
                               uint32_t *n_nodes_out, double *device_ms) {
    if (!vertices || !nodes_out || !indexes_out || !n_nodes_out || n == 0)
        return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_build_bvh: null argument or no triangles");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return ctgpu_fail(CTGPU_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= n_dev) return ctgpu_fail(CTGPU_ERR_ARG, "device %d out of range (0..%d)", device, n_dev - 1);
    HIP_TRY(hipSetDevice(device));
    std::vector<void *> owned;
    struct Release { std::vector<void *> &v; ~Release() { for (void *p : v) if (p) (void)hipFree(p); } } release{owned};
    double *d_verts = nullptr;
    HIP_TRY(device_alloc(d_verts, (size_t)n * 9, owned));
    HIP_TRY(hipMemcpy(d_verts, vertices, (size_t)n * 9 * sizeof(double), hipMemcpyHostToDevice));
    ctgpu_node *d_nodes = nullptr;
    uint32_t *d_idx = nullptr;
    uint32_t used = 0;
    const int rc = ctgpu_build_bvh_resident(device, d_verts, n, &d_nodes, &d_idx, &used, nullptr, device_ms);
    if (rc) return rc;
    owned.push_back(d_nodes);
    owned.push_back(d_idx);
    HIP_TRY(hipMemcpy(nodes_out, d_nodes, (size_t)used * sizeof(ctgpu_node), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(indexes_out, d_idx, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_nodes_out = used;
    return CTGPU_OK;
}
