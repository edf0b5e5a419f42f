// ct_shadow_tree_build.h -- the shadow tree's builder: plain C++ (no HIP), so that the CPU suite can hold it to its
// contract (tests/test_shadow_tree_build.py compiles this header with g++): every triangle under exactly one entry, every
// entry's box holding what lies below it, the walk's stack bound, no node reachable twice.
// What the tree is for and why any shape of it gives the reference's answer: ct_device.h "the shadow tree", DESIGN appendix E.
#ifndef CT_SHADOW_TREE_BUILD_H
#define CT_SHADOW_TREE_BUILD_H

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace ctst {

// the walk's node (ct_device.h QuadNode, same 128 bytes) and its reference bits, restated here so that the header stands alone
struct Quad {
    float plane[4][3][2];         // [entry][axis]{min, max}
    uint32_t ref[4];              // kLeaf | first slot of a run of triangles (the tree's own slot order), or a node index
    uint32_t count;
    uint32_t pad_[3];
};
static_assert(sizeof(Quad) == 128, "QuadNode");
constexpr uint32_t kLeaf = 0x80000000u;

struct Box {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const Box &b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};

// binary tree: node i covers order[first .. first + count); an inner node's children are `left` and left + 1; a LEAF
// (left == 0) holds its count triangles as one run
struct BinNode { Box box; uint32_t left, first, count; };

constexpr int kBins = 16;
constexpr uint32_t kSequentialBelow = 1u << 15;      // ranges a worker thread builds alone
constexpr uint32_t kMedianBelowDepth = 56;           // deeper than this the splits are medians (a bound on the depth)

struct Builder {
    const Box *tri;                   // [n] the triangles' boxes
    std::vector<uint32_t> order;      // slots, permuted as the ranges are split
    std::vector<BinNode> nodes;       // 2n - 1
    std::atomic<uint32_t> used{1};
    bool median_only = false;
    uint32_t leaf_max = 1;            // most triangles a leaf may hold ...
    float leaf_rho = 0.2f;            // ... and it holds c > 1 only if  c - sum(area_i) / area(leaf) <= rho c:  the triangles' boxes
                                      // fill the leaf's (the two halves of a grid cell have the cell's box each), so that a box test
                                      // per triangle (rho of a triangle test's cost) would exclude too little to pay for itself

    std::mutex lock;
    std::condition_variable wake;
    std::vector<std::pair<uint32_t, uint32_t>> todo;       // {node, depth}
    uint32_t busy = 0;
    bool done = false;

    static float centre(const Box &b, int a) { return 0.5f * (b.lo[a] + b.hi[a]); }

    // split node `id` in place (its range of `order`); false: it stays a leaf (one triangle)
    bool split(uint32_t id, uint32_t depth) {
        BinNode &node = nodes[id];
        if (node.count <= 1u) return false;
        uint32_t *idx = order.data() + node.first;
        const uint32_t count = node.count;
        if (count <= leaf_max) {
            const float whole = node.box.half_area();
            float parts = 0.0f;
            for (uint32_t i = 0; i < count; i++) parts += tri[idx[i]].half_area();
            if (!(whole > 0.0f) || (float)count - parts / whole <= leaf_rho * (float)count) return false;
        }
        uint32_t mid = 0;
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < count; i++)
            for (int a = 0; a < 3; a++) { const float c = centre(tri[idx[i]], a); clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c); }
        int best_axis = -1, best_bin = 0;
        float best_cost = INFINITY;
        if (!median_only && depth < kMedianBelowDepth) {
            for (int a = 0; a < 3; a++) {
                const float extent = chi[a] - clo[a];
                if (!(extent > 0.0f) || !std::isfinite(extent)) continue;
                Box bin_box[kBins];
                uint32_t bin_n[kBins];
                for (int b = 0; b < kBins; b++) { bin_box[b].clear(); bin_n[b] = 0; }
                const float scale = (float)kBins / extent;
                for (uint32_t i = 0; i < count; i++) {
                    int b = (int)((centre(tri[idx[i]], a) - clo[a]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    bin_n[b] += 1; bin_box[b].grow(tri[idx[i]]);
                }
                float right_area[kBins];
                uint32_t right_n[kBins];
                Box acc; acc.clear();
                uint32_t n_acc = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    if (bin_n[b]) acc.grow(bin_box[b]);
                    n_acc += bin_n[b];
                    right_area[b] = n_acc ? acc.half_area() : 0.0f; right_n[b] = n_acc;
                }
                acc.clear(); n_acc = 0;
                for (int b = 0; b + 1 < kBins; b++) {
                    if (bin_n[b]) acc.grow(bin_box[b]);
                    n_acc += bin_n[b];
                    if (n_acc == 0 || right_n[b + 1] == 0) continue;
                    const float cost = acc.half_area() * (float)n_acc + right_area[b + 1] * (float)right_n[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
        }
        if (best_axis >= 0) {
            const int a = best_axis;
            const float scale = (float)kBins / (chi[a] - clo[a]);
            mid = (uint32_t)(std::partition(idx, idx + count, [&](uint32_t s) {
                int b = (int)((centre(tri[s], a) - clo[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            }) - idx);
        }
        if (mid == 0 || mid == count) {
            // no plane separates the centres (or the depth bound is reached): the median along the widest axis
            int a = 0;
            for (int c = 1; c < 3; c++) if (chi[c] - clo[c] > chi[a] - clo[a]) a = c;
            mid = count / 2;
            std::nth_element(idx, idx + mid, idx + count, [&](uint32_t x, uint32_t y) {
                const float cx = centre(tri[x], a), cy = centre(tri[y], a);
                return cx < cy || (cx == cy && x < y);
            });
        }
        const uint32_t left = used.fetch_add(2u);
        BinNode &l = nodes[left], &r = nodes[left + 1];
        l.first = node.first; l.count = mid; l.left = 0;
        r.first = node.first + mid; r.count = count - mid; r.left = 0;
        l.box.clear(); r.box.clear();
        for (uint32_t i = 0; i < mid; i++) l.box.grow(tri[idx[i]]);
        for (uint32_t i = mid; i < count; i++) r.box.grow(tri[idx[i]]);
        node.left = left;
        return true;
    }
    void build_alone(uint32_t id, uint32_t depth) {
        std::vector<std::pair<uint32_t, uint32_t>> stack{{id, depth}};
        while (!stack.empty()) {
            const auto [at, d] = stack.back();
            stack.pop_back();
            if (!split(at, d)) continue;
            stack.push_back({nodes[at].left, d + 1});
            stack.push_back({nodes[at].left + 1, d + 1});
        }
    }
    void worker() {
        for (;;) {
            std::pair<uint32_t, uint32_t> job;
            {
                std::unique_lock<std::mutex> hold(lock);
                wake.wait(hold, [&] { return done || !todo.empty(); });
                if (todo.empty()) return;
                job = todo.back();
                todo.pop_back();
                busy += 1;
            }
            if (nodes[job.first].count < kSequentialBelow) build_alone(job.first, job.second);
            else if (split(job.first, job.second)) {
                std::lock_guard<std::mutex> hold(lock);
                todo.push_back({nodes[job.first].left, job.second + 1});
                todo.push_back({nodes[job.first].left + 1, job.second + 1});
                wake.notify_all();
            }
            {
                std::lock_guard<std::mutex> hold(lock);
                busy -= 1;
                if (busy == 0 && todo.empty()) { done = true; wake.notify_all(); }
            }
        }
    }
    void run(uint32_t n, unsigned threads) {
        order.resize(n);
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        nodes.resize(2u * (size_t)n);
        BinNode &root = nodes[0];
        root.first = 0; root.count = n; root.left = 0;
        root.box.clear();
        for (uint32_t i = 0; i < n; i++) root.box.grow(tri[i]);
        todo.push_back({0u, 0u});
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back([this] { worker(); });
        worker();
        for (auto &t : pool) t.join();
    }
};

// ------------------------------------------------------------------ four entries a node
struct QuadEntry { Box box; uint32_t bin; bool triangle; };        // bin: the binary node the entry stands for; triangle: a leaf (a run of triangles)
struct QuadDraft {
    QuadEntry entry[4];
    uint32_t count;
    uint32_t need;          // most entries waiting on the walk's stack anywhere below, this node's own included
    uint32_t child[4];      // draft index of an inner entry
};

struct Collapser {
    const std::vector<BinNode> &bin;
    const std::vector<uint32_t> &order;
    std::vector<QuadDraft> drafts;
    explicit Collapser(const Builder &b) : bin(b.nodes), order(b.order) {}

    // the node that stands for binary node `id` (an inner one): its two children, the one with the largest box opened in
    // turn until four entries are there or only triangles are left
    uint32_t make(uint32_t root_bin) {
        // (explicit stack: post-order, so that a node's `need` sees its children's)
        struct Frame { uint32_t bin, draft, next_child; };
        std::vector<Frame> stack;
        const auto open = [&](uint32_t id) -> uint32_t {
            QuadDraft q;
            q.count = 0; q.need = 0;
            uint32_t ids[4];
            ids[q.count++] = bin[id].left; ids[q.count++] = bin[id].left + 1;
            while (q.count < 4u) {
                int widest = -1;
                float area = -1.0f;
                for (uint32_t k = 0; k < q.count; k++)
                    if (bin[ids[k]].left != 0u && bin[ids[k]].box.half_area() > area) { area = bin[ids[k]].box.half_area(); widest = (int)k; }
                if (widest < 0) break;
                const uint32_t opened = ids[widest];
                ids[widest] = bin[opened].left;
                ids[q.count++] = bin[opened].left + 1;
            }
            for (uint32_t k = 0; k < q.count; k++) {
                q.entry[k].box = bin[ids[k]].box; q.entry[k].bin = ids[k]; q.entry[k].triangle = bin[ids[k]].left == 0u;
                q.child[k] = 0;
            }
            drafts.push_back(q);
            return (uint32_t)drafts.size() - 1u;
        };
        const uint32_t root = open(root_bin);
        stack.push_back({root_bin, root, 0u});
        while (!stack.empty()) {
            Frame &f = stack.back();
            const uint32_t me = f.draft;
            if (f.next_child < drafts[me].count) {
                const uint32_t k = f.next_child++;
                if (!drafts[me].entry[k].triangle) {
                    const uint32_t child_bin = drafts[me].entry[k].bin;
                    const uint32_t made = open(child_bin);         // (may move `drafts`: no reference is held across it)
                    drafts[me].child[k] = made;
                    stack.push_back({child_bin, made, 0u});
                }
                continue;
            }
            // all children are final: order the entries -- triangles first, then subtrees by what they need of the stack,
            // least first -- so that the deepest subtree is entered with nothing of this node waiting
            QuadDraft &q = drafts[me];
            uint32_t perm[4] = {0, 1, 2, 3};
            const auto key = [&](uint32_t k) { return q.entry[k].triangle ? 0u : 1u + drafts[q.child[k]].need; };
            std::stable_sort(perm, perm + q.count, [&](uint32_t x, uint32_t y) { return key(x) < key(y); });
            QuadDraft sorted = q;
            for (uint32_t k = 0; k < q.count; k++) { sorted.entry[k] = q.entry[perm[k]]; sorted.child[k] = q.child[perm[k]]; }
            sorted.need = 0;
            for (uint32_t k = 0; k < q.count; k++) {
                const uint32_t below = sorted.entry[k].triangle ? 0u : drafts[sorted.child[k]].need;
                sorted.need = std::max(sorted.need, q.count - 1u - k + below);
            }
            q = sorted;
            stack.pop_back();
        }
        return root;
    }
};


// a plane moved outward by gamma and rounded outward to binary32
inline float plane_down(float p, double gamma) {
    const double x = (double)p - gamma;
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float plane_up(float p, double gamma) { return -plane_down(-p, gamma); }

// The whole build: boxes[n] (the triangles' grown boxes, slot order) -> nodes (root = node 0), every stored plane moved
// outward by gamma; `bound` >= every |stored plane| (the absent entries' inside-out boxes sit there).  Returns the most
// entries the walk ever has waiting (its stack need).  strategy 0: binned surface-area heuristic, 1: object medians.
// order[s]: the input triangle that sits at slot s of the tree's own triangle order (a leaf is a run of slots: an entry's
// reference is kLeaf | first slot); remaining[s]: triangles from slot s to the end of its leaf.
struct BuildOptions { int strategy = 0; uint32_t leaf_max = 2; float leaf_rho = 0.2f; unsigned threads = 1; };
inline uint32_t build_shadow_tree(const Box *boxes, uint32_t n, const BuildOptions &opt, double gamma, float bound, std::vector<Quad> &quads,
                                  std::vector<uint32_t> &order, std::vector<uint32_t> &remaining) {
    quads.clear(); order.clear(); remaining.clear();
    // An entry a node does not have: a one-point box at a corner beyond every plane -- a ray's line enters it only by
    // passing through that very point -- and, for the line that does, a reference that is always safe to follow: some
    // triangle below this node.  (Testing a triangle more than the boxes ask for never changes the answer: the test is the
    // reference's own, and it counts only after the reference's own test of its leaf's box.)  The walk looks at all four
    // entries of every node and never asks how many there are.
    const auto absent = [&](Quad &node, int k, uint32_t some_slot) {
        for (int a = 0; a < 3; a++) node.plane[k][a][0] = node.plane[k][a][1] = bound;
        node.ref[k] = kLeaf | some_slot;
    };
    const auto store = [&](Quad &node, int k, const Box &b) {
        for (int a = 0; a < 3; a++) { node.plane[k][a][0] = plane_down(b.lo[a], gamma); node.plane[k][a][1] = plane_up(b.hi[a], gamma); }
    };
    if (n == 0u) return 0u;
    if (n == 1u) {
        Quad node;
        std::memset(&node, 0, sizeof node);
        store(node, 0, boxes[0]);
        node.ref[0] = kLeaf | 0u;
        for (int k = 1; k < 4; k++) absent(node, k, 0u);
        node.count = 1;
        quads.push_back(node);
        order.assign(1, 0u); remaining.assign(1, 1u);
        return 0u;
    }
    Builder builder;
    builder.tri = boxes;
    builder.median_only = opt.strategy == 1;
    builder.leaf_max = std::max(1u, opt.leaf_max);
    builder.leaf_rho = opt.leaf_rho;
    builder.run(n, std::max(1u, opt.threads));
    remaining.assign(n, 1u);
    if (builder.nodes[0].left == 0u) {
        // the whole scene is one leaf: a root node with that one entry
        Quad node;
        std::memset(&node, 0, sizeof node);
        store(node, 0, builder.nodes[0].box);
        node.ref[0] = kLeaf | 0u;
        for (int k = 1; k < 4; k++) absent(node, k, 0u);
        node.count = 1;
        quads.push_back(node);
        for (uint32_t i = 0; i < n; i++) remaining[i] = n - i;
        order = builder.order;
        return 0u;
    }
    Collapser collapse(builder);
    const uint32_t root = collapse.make(0u);
    const uint32_t need = collapse.drafts[root].need;
    // numbered in the walk's own order (depth first, entries as stored): a subtree's nodes lie together
    std::vector<uint32_t> number(collapse.drafts.size(), 0u);
    std::vector<uint32_t> stack{root};
    std::vector<uint32_t> sequence;
    sequence.reserve(collapse.drafts.size());
    while (!stack.empty()) {
        const uint32_t at = stack.back();
        stack.pop_back();
        number[at] = (uint32_t)sequence.size();
        sequence.push_back(at);
        const QuadDraft &q = collapse.drafts[at];
        for (int k = (int)q.count - 1; k >= 0; k--) if (!q.entry[k].triangle) stack.push_back(q.child[k]);
    }
    quads.resize(sequence.size());
    for (size_t i = 0; i < sequence.size(); i++) {
        const QuadDraft &q = collapse.drafts[sequence[i]];
        Quad &node = quads[i];
        std::memset(&node, 0, sizeof node);
        for (uint32_t k = 0; k < 4; k++) {
            if (k >= q.count) { absent(node, (int)k, builder.nodes[q.entry[0].bin].first); continue; }
            store(node, (int)k, q.entry[k].box);
            const BinNode &b = builder.nodes[q.entry[k].bin];
            node.ref[k] = q.entry[k].triangle ? (kLeaf | b.first) : number[q.child[k]];
            if (q.entry[k].triangle) for (uint32_t i = 0; i < b.count; i++) remaining[b.first + i] = b.count - i;
        }
        node.count = q.count;
    }
    order = builder.order;
    return need;
}

// What the walk reads: the same nodes with every reference turned into the BYTE OFFSET of what it names (a node: index *
// 128; a leaf: kLeaf | first record * record_bytes), so that a visit's scalar loads take the reference as it is.
// false: an offset would not fit below kLeaf.
inline bool references_to_byte_offsets(std::vector<Quad> &quads, uint32_t n, uint32_t record_bytes) {
    if ((uint64_t)quads.size() * sizeof(Quad) >= (uint64_t)kLeaf || (uint64_t)n * record_bytes >= (uint64_t)kLeaf) return false;
    for (Quad &q : quads)
        for (int k = 0; k < 4; k++)
            q.ref[k] = (q.ref[k] & kLeaf) ? (kLeaf | ((q.ref[k] & ~kLeaf) * record_bytes)) : q.ref[k] * (uint32_t)sizeof(Quad);
    return true;
}

}  // namespace ctst
#endif
