// bvh_build.cpp -- midpoint-split BVH builder behind InitializeBVHState / BuildBVH.
//
// The device traversal reproduces the reference's hit selection only if it walks
// the very same tree, so the builder must return the reference's node numbering,
// bounds and index permutation bit for bit (bvh.cpp:16-121):
//   * centroid = ((p1 + p2) + p3) * (double)0.3333f
//   * bounds start at +-(double)1e30f and fold p1, p2, p3 of each triangle with
//     "a < b ? a : b" / "a > b ? a : b"
//   * split axis: y if extent.y > extent.x; z if extent.z > (double)(float)extent[axis]
//   * split plane in binary32: (float)min[axis] + (float)extent[axis] * 0.5f
//   * Hoare-style sweep comparing (float)centroid[axis] < plane, swapping with the tail
//   * a node stays a leaf when it has <= 2 triangles or one side would be empty
//   * children take the next two node slots (left, then right); the left subtree
//     is completed before the right one is descended
// Inner nodes keep their old firstTriangleIndex, as in the reference.
//
// The build runs on several threads for big meshes and still returns the reference's numbering:
// a node's descendants occupy one contiguous stretch of node slots right after the pair its
// split allocated (the recursion finishes a subtree before anything else allocates), and a subtree
// only permutes its own stretch of the index array.  So the top of the tree is split on one
// thread down to subtrees of a few thousand triangles, those are built independently with slot
// numbers relative to their own root, and one pass in the reference's order hands out the
// absolute slots and copies them in (CT_BVH_THREADS=1 forces the plain recursion;
// tests/test_bvh_build.py compares the two byte for byte).
#include "cobbletrace_host.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline double component(const v3_t &v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
inline float component_f32(const v3_t &v, int axis) { return (float)component(v, axis); }

inline double lesser(double a, double b) { return (a < b) ? a : b; }     // mymath.h:15-17
inline double greater(double a, double b) { return (a > b) ? a : b; }    // mymath.h:11-13

// a node's box from its triangles' corners, MinV3 / MaxV3 from +-1e30f (bvh.cpp:30-49)
void fit_bounds_of(const bvh_state_t *s, bvh_node_t &node) {
    const double big = (double)1e30f;
    v3_t lo{big, big, big}, hi{-big, -big, -big};
    for (uint32_t k = 0; k < node.triangleCount; k++) {
        const triangle_t &t = s->triangles.data[s->triangles.indexes[node.firstTriangleIndex + k]];
        const v3_t *corners[3] = {&t.p1, &t.p2, &t.p3};
        for (const v3_t *p : corners) {
            lo.x = lesser(lo.x, p->x); lo.y = lesser(lo.y, p->y); lo.z = lesser(lo.z, p->z);
        }
        for (const v3_t *p : corners) {
            hi.x = greater(hi.x, p->x); hi.y = greater(hi.y, p->y); hi.z = greater(hi.z, p->z);
        }
    }
    node.aabbMin = lo;
    node.aabbMax = hi;
}

void fit_bounds(bvh_state_t *s, uint32_t node_id) { fit_bounds_of(s, s->bvhNodes[node_id]); }

// Partitions the node's triangles about its split plane; returns false when the node stays a leaf,
// else the sizes / first indices of the two children.
bool partition(bvh_state_t *s, const bvh_node_t &node, uint32_t &left_count, uint32_t &right_first) {
    if (node.triangleCount <= 2) return false;
    const v3_t extent{node.aabbMax.x - node.aabbMin.x, node.aabbMax.y - node.aabbMin.y,
                      node.aabbMax.z - node.aabbMin.z};
    int axis = 0;
    if (extent.y > extent.x) axis = 1;
    if (extent.z > (double)component_f32(extent, axis)) axis = 2;
    const float plane = component_f32(node.aabbMin, axis) + component_f32(extent, axis) * 0.5f;

    uint32_t *order = s->triangles.indexes;
    int i = (int)node.firstTriangleIndex;
    int j = i + (int)node.triangleCount - 1;
    while (i <= j) {
        if (component_f32(s->triangles.data[order[i]].centroid, axis) < plane) {
            i++;
        } else {
            const uint32_t tmp = order[i];
            order[i] = order[j];
            order[j--] = tmp;
        }
    }
    left_count = (uint32_t)(i - (int)node.firstTriangleIndex);
    right_first = (uint32_t)i;
    return !(left_count == 0 || left_count == node.triangleCount);
}

// the reference's recursion, in place (bvh.cpp:51-106)
void split(bvh_state_t *s, uint32_t node_id) {
    bvh_node_t *node = &s->bvhNodes[node_id];
    uint32_t left_count, right_first;
    if (!partition(s, *node, left_count, right_first)) return;
    const uint32_t left = s->nodesUsed++;
    const uint32_t right = s->nodesUsed++;
    node->leftNode = left;
    s->bvhNodes[left].firstTriangleIndex = node->firstTriangleIndex;
    s->bvhNodes[left].triangleCount = left_count;
    s->bvhNodes[right].firstTriangleIndex = right_first;
    s->bvhNodes[right].triangleCount = node->triangleCount - left_count;
    node->triangleCount = 0;
    fit_bounds(s, left);
    fit_bounds(s, right);
    split(s, left);
    split(s, right);
}

// The same recursion into a private node list: nodes[0] is the subtree's root (bounds already fitted),
// descendants follow in allocation order, leftNode counts from the start of this list.
void split_local(const bvh_state_t *s, std::vector<bvh_node_t> &nodes, uint32_t at) {
    uint32_t left_count, right_first;
    if (!partition(const_cast<bvh_state_t *>(s), nodes[at], left_count, right_first)) return;
    const uint32_t left = (uint32_t)nodes.size();
    nodes.resize(nodes.size() + 2);
    bvh_node_t &node = nodes[at];
    node.leftNode = left;
    nodes[left].firstTriangleIndex = node.firstTriangleIndex;
    nodes[left].triangleCount = left_count;
    nodes[left + 1].firstTriangleIndex = right_first;
    nodes[left + 1].triangleCount = node.triangleCount - left_count;
    node.triangleCount = 0;
    fit_bounds_of(s, nodes[left]);
    fit_bounds_of(s, nodes[left + 1]);
    split_local(s, nodes, left);
    split_local(s, nodes, left + 1);
}

struct TopNode {
    bvh_node_t node;          // leftNode is filled in when absolute slots are handed out
    int left = -1;            // top-tree index of the left child (right = left + 1); -1: leaf or a job
    int job = -1;             // index of the subtree built on a worker thread
};

void build_parallel(bvh_state_t *s, unsigned threads) {
    const auto t_begin = std::chrono::steady_clock::now();
    const uint32_t grain = s->triangles.size / (threads * 8u) + 1024u;
    std::vector<TopNode> top(1);
    top[0].node = s->bvhNodes[s->rootNodeIdx];
    std::vector<std::vector<bvh_node_t>> jobs;
    // top of the tree, one thread, the reference's order (left subtree first)
    std::vector<int> todo{0};
    while (!todo.empty()) {
        const int at = todo.back();
        todo.pop_back();
        if (top[at].node.triangleCount <= grain) {
            top[at].job = (int)jobs.size();
            jobs.emplace_back(1, top[at].node);
            continue;
        }
        uint32_t left_count, right_first;
        if (!partition(s, top[at].node, left_count, right_first)) continue;
        const int left = (int)top.size();
        top.resize(top.size() + 2);
        TopNode &parent = top[at];
        parent.left = left;
        top[left].node = bvh_node_t{};
        top[left + 1].node = bvh_node_t{};
        top[left].node.firstTriangleIndex = parent.node.firstTriangleIndex;
        top[left].node.triangleCount = left_count;
        top[left + 1].node.firstTriangleIndex = right_first;
        top[left + 1].node.triangleCount = parent.node.triangleCount - left_count;
        parent.node.triangleCount = 0;
        fit_bounds_of(s, top[left].node);
        fit_bounds_of(s, top[left + 1].node);
        todo.push_back(left + 1);
        todo.push_back(left);
    }
    const auto t_top = std::chrono::steady_clock::now();
    // the subtrees, each on whichever thread is free (disjoint stretches of the index array)
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t k = next.fetch_add(1); k < jobs.size(); k = next.fetch_add(1)) {
            jobs[k].reserve((size_t)jobs[k][0].triangleCount * 2u);
            split_local(s, jobs[k], 0);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    const auto t_jobs = std::chrono::steady_clock::now();
    // absolute slots in the reference's allocation order
    struct Placed { int top; uint32_t slot; };
    std::vector<Placed> walk{{0, s->rootNodeIdx}};
    while (!walk.empty()) {
        const Placed p = walk.back();
        walk.pop_back();
        const TopNode &t = top[p.top];
        bvh_node_t out = t.node;
        if (t.job >= 0) {
            const std::vector<bvh_node_t> &local = jobs[t.job];
            const uint32_t base = s->nodesUsed;                     // local slot j >= 1 lands on base + j - 1
            out = local[0];
            if (local.size() > 1) out.leftNode = base + local[0].leftNode - 1u;
            for (size_t j = 1; j < local.size(); j++) {
                bvh_node_t n = local[j];
                if (n.triangleCount == 0) n.leftNode = base + n.leftNode - 1u;
                s->bvhNodes[base + j - 1] = n;
            }
            s->nodesUsed += (uint32_t)local.size() - 1u;
        } else if (t.left >= 0) {
            const uint32_t left = s->nodesUsed;
            s->nodesUsed += 2;
            out.leftNode = left;
            walk.push_back({t.left + 1, left + 1});                 // the right subtree's descendants come later
            // the left subtree's descendants come next: place it first
            s->bvhNodes[p.slot] = out;
            // (stack order: left is handled before right)
            walk.push_back({t.left, left});
            continue;
        }
        s->bvhNodes[p.slot] = out;
    }
    if (std::getenv("CT_BVH_TIMING")) {
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "bvh build on %u threads: top %.1f ms (%zu nodes, %zu subtrees), subtrees %.1f ms, numbering %.1f ms\n",
                     threads, ms(t_begin, t_top), top.size(), jobs.size(), ms(t_top, t_jobs), ms(t_jobs, t_end));
    }
}

}  // namespace

bvh_state_t InitializeBVHState(triangle_t *triangles, int size) {
    bvh_state_t s;
    s.bvhNodes = (bvh_node_t *)std::calloc((size_t)(size > 0 ? size * 2 - 1 : 1), sizeof(bvh_node_t));
    s.nodesUsed = 1;
    s.rootNodeIdx = 0;
    s.triangles.size = (uint32_t)size;
    s.triangles.data = triangles;
    s.triangles.indexes = (uint32_t *)std::calloc((size_t)(size > 0 ? size : 1), sizeof(uint32_t));
    for (int i = 0; i < size; i++) s.triangles.indexes[i] = (uint32_t)i;
    return s;
}

void BuildBVH(bvh_state_t *s) {
    const double third = (double)0.3333f;
    for (uint32_t i = 0; i < s->triangles.size; i++) {
        triangle_t &t = s->triangles.data[i];
        t.centroid = v3_t{((t.p1.x + t.p2.x) + t.p3.x) * third, ((t.p1.y + t.p2.y) + t.p3.y) * third,
                          ((t.p1.z + t.p2.z) + t.p3.z) * third};
    }
    bvh_node_t &root = s->bvhNodes[s->rootNodeIdx];
    root.leftNode = 0;
    root.firstTriangleIndex = 0;
    root.triangleCount = s->triangles.size;
    fit_bounds(s, s->rootNodeIdx);
    unsigned threads = std::thread::hardware_concurrency();
    if (threads > 16) threads = 16;
    if (const char *t = std::getenv("CT_BVH_THREADS")) threads = (unsigned)std::atoi(t);
    if (threads > 1 && s->triangles.size >= 50000u) build_parallel(s, threads);
    else split(s, s->rootNodeIdx);
}
