// ct_shadow_tree_build.h -- the shadow tree's builder: plain C++ (no HIP), so that the CPU suite can hold it to its
// contract (tests/test_shadow_tree_build.py compiles this header with g++): every triangle under exactly one entry, every
// entry's box holding what lies below it, the walk's stack bound, no node reachable twice.
// What the tree is for and why any shape of it gives the reference's answer: ct_device.h "the shadow tree", DESIGN appendix E.
#ifndef CT_SHADOW_TREE_BUILD_H
#define CT_SHADOW_TREE_BUILD_H

#include <algorithm>
#include <atomic>
#include <chrono>
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
constexpr uint32_t kParallelAbove = 1u << 17;        // ranges whose passes over the triangles are cut into chunks for several threads
constexpr uint32_t kMedianBelowDepth = 56;           // deeper than this the splits are medians (a bound on the depth)

// fn(chunk, begin, end) over [0, count) cut into `chunks` pieces, one thread each (the caller's included)
template <typename F>
inline void in_chunks(uint32_t count, unsigned chunks, F &&fn) {
    if (chunks <= 1u) { fn(0u, 0u, count); return; }
    std::vector<std::thread> pool;
    const uint32_t step = (count + chunks - 1u) / chunks;
    for (unsigned c = 1; c < chunks; c++) {
        const uint32_t b = std::min(count, c * step), e = std::min(count, (c + 1u) * step);
        pool.emplace_back([&fn, c, b, e] { fn(c, b, e); });
    }
    fn(0u, 0u, std::min(count, step));
    for (auto &t : pool) t.join();
}

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
    unsigned threads = 1;

    std::mutex lock;
    std::condition_variable wake;
    std::vector<std::pair<uint32_t, uint32_t>> todo;       // {node, depth}
    uint32_t busy = 0;
    bool done = false;

    static float centre(const Box &b, int a) { return 0.5f * (b.lo[a] + b.hi[a]); }
    struct Bins { Box box[3][kBins]; uint32_t n[3][kBins]; };

    // split node `id` in place (its range of `order`); false: it stays a leaf
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
        const unsigned chunks = (count >= kParallelAbove && threads > 1u) ? std::min(threads, 16u) : 1u;
        // pass 1: the centres' bounds
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        {
            std::vector<float> part((size_t)chunks * 6);
            in_chunks(count, chunks, [&](unsigned c, uint32_t b, uint32_t e) {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (uint32_t i = b; i < e; i++)
                    for (int a = 0; a < 3; a++) { const float v = centre(tri[idx[i]], a); lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
                for (int a = 0; a < 3; a++) { part[(size_t)c * 6 + a] = lo[a]; part[(size_t)c * 6 + 3 + a] = hi[a]; }
            });
            for (unsigned c = 0; c < chunks; c++)
                for (int a = 0; a < 3; a++) { clo[a] = std::min(clo[a], part[(size_t)c * 6 + a]); chi[a] = std::max(chi[a], part[(size_t)c * 6 + 3 + a]); }
        }
        uint32_t mid = 0;
        Box left_box, right_box;
        left_box.clear(); right_box.clear();
        bool boxes_known = false;
        float scale[3];
        bool usable[3];
        bool any_axis = false;
        for (int a = 0; a < 3; a++) {
            const float extent = chi[a] - clo[a];
            usable[a] = extent > 0.0f && std::isfinite(extent);
            scale[a] = usable[a] ? (float)kBins / extent : 0.0f;
            any_axis = any_axis || usable[a];
        }
        const auto bin_of = [&](uint32_t s, int a) {
            int b = (int)((centre(tri[s], a) - clo[a]) * scale[a]);
            return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
        };
        if (!median_only && depth < kMedianBelowDepth && any_axis) {
            // pass 2: the three axes' bins at once
            std::vector<Bins> part(chunks);
            in_chunks(count, chunks, [&](unsigned c, uint32_t b, uint32_t e) {
                Bins &B = part[c];
                for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) { B.box[a][k].clear(); B.n[a][k] = 0; }
                for (uint32_t i = b; i < e; i++) {
                    const Box &t = tri[idx[i]];
                    for (int a = 0; a < 3; a++) {
                        if (!usable[a]) continue;
                        const int k = bin_of(idx[i], a);
                        B.n[a][k] += 1; B.box[a][k].grow(t);
                    }
                }
            });
            Bins &B = part[0];
            for (unsigned c = 1; c < chunks; c++)
                for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) if (part[c].n[a][k]) { B.n[a][k] += part[c].n[a][k]; B.box[a][k].grow(part[c].box[a][k]); }
            int best_axis = -1, best_bin = 0;
            float best_cost = INFINITY;
            for (int a = 0; a < 3; a++) {
                if (!usable[a]) continue;
                float right_area[kBins];
                uint32_t right_n[kBins];
                Box acc; acc.clear();
                uint32_t n_acc = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    if (B.n[a][b]) acc.grow(B.box[a][b]);
                    n_acc += B.n[a][b];
                    right_area[b] = n_acc ? acc.half_area() : 0.0f; right_n[b] = n_acc;
                }
                acc.clear(); n_acc = 0;
                for (int b = 0; b + 1 < kBins; b++) {
                    if (B.n[a][b]) acc.grow(B.box[a][b]);
                    n_acc += B.n[a][b];
                    if (n_acc == 0 || right_n[b + 1] == 0) continue;
                    const float cost = acc.half_area() * (float)n_acc + right_area[b + 1] * (float)right_n[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const int a = best_axis;
                mid = (uint32_t)(std::partition(idx, idx + count, [&](uint32_t s) { return bin_of(s, a) <= best_bin; }) - idx);
                for (int b = 0; b < kBins; b++) if (B.n[a][b]) (b <= best_bin ? left_box : right_box).grow(B.box[a][b]);
                boxes_known = true;
            }
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
            boxes_known = false;
        }
        const uint32_t left = used.fetch_add(2u);
        BinNode &l = nodes[left], &r = nodes[left + 1];
        l.first = node.first; l.count = mid; l.left = 0;
        r.first = node.first + mid; r.count = count - mid; r.left = 0;
        if (boxes_known) { l.box = left_box; r.box = right_box; }
        else {
            l.box.clear(); r.box.clear();
            for (uint32_t i = 0; i < mid; i++) l.box.grow(tri[idx[i]]);
            for (uint32_t i = mid; i < count; i++) r.box.grow(tri[idx[i]]);
        }
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
    void run(uint32_t n, unsigned n_threads) {
        threads = std::max(1u, n_threads);
        order.resize(n);
        nodes.resize(2u * (size_t)n);
        BinNode &root = nodes[0];
        root.first = 0; root.count = n; root.left = 0;
        const unsigned chunks = (n >= kParallelAbove && threads > 1u) ? std::min(threads, 16u) : 1u;
        std::vector<Box> part(chunks);
        in_chunks(n, chunks, [&](unsigned c, uint32_t b, uint32_t e) {
            part[c].clear();
            for (uint32_t i = b; i < e; i++) { order[i] = i; part[c].grow(tri[i]); }
        });
        root.box.clear();
        for (unsigned c = 0; c < chunks; c++) root.box.grow(part[c]);
        todo.push_back({0u, 0u});
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back([this] { worker(); });
        worker();
        for (auto &t : pool) t.join();
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

// ------------------------------------------------------------------ four entries a node
// The node that stands for an inner binary node: its two children, the one with the largest box opened in turn until four
// entries are there or only leaves are left.  Per binary node that heads a four-wide node: its entries (binary node ids, in
// the order the walk will meet them), how many, what its subtree needs of the walk's stack, how many four-wide nodes it
// holds.  Entries are ordered leaves first, then subtrees by their need, least first: the deepest subtree is entered with
// nothing of this node waiting, need(node) = max over k of (count - 1 - k + need(entry k)).
struct FourWide {
    const std::vector<BinNode> &bin;
    std::vector<uint32_t> entry;      // [bin][4]
    std::vector<uint8_t> count;       // [bin] 0: not (yet) a four-wide node's head
    std::vector<uint32_t> need, size; // [bin]
    explicit FourWide(const Builder &b) : bin(b.nodes), entry(b.nodes.size() * 4u), count(b.nodes.size(), 0), need(b.nodes.size(), 0), size(b.nodes.size(), 0) {}

    uint32_t open(uint32_t id, uint32_t ids[4]) const {
        uint32_t n = 0;
        ids[n++] = bin[id].left; ids[n++] = bin[id].left + 1;
        while (n < 4u) {
            int widest = -1;
            float area = -1.0f;
            for (uint32_t k = 0; k < n; k++)
                if (bin[ids[k]].left != 0u && bin[ids[k]].box.half_area() > area) { area = bin[ids[k]].box.half_area(); widest = (int)k; }
            if (widest < 0) break;
            const uint32_t opened = ids[widest];
            ids[widest] = bin[opened].left;
            ids[n++] = bin[opened].left + 1;
        }
        return n;
    }
    // post-order over the subtree of `root` (explicit stack): entries, order, need, size of every four-wide node in it
    void settle(uint32_t root) {
        struct Frame { uint32_t id; uint32_t ids[4]; uint32_t n, next; };
        std::vector<Frame> stack;
        const auto push = [&](uint32_t id) { Frame f; f.id = id; f.n = open(id, f.ids); f.next = 0; stack.push_back(f); };
        if (count[root]) return;
        push(root);
        while (!stack.empty()) {
            Frame &f = stack.back();
            if (f.next < f.n) {
                const uint32_t child = f.ids[f.next++];
                if (bin[child].left != 0u && !count[child]) push(child);       // (may move `stack`: f is not used after it)
                continue;
            }
            uint32_t key[4], perm[4] = {0, 1, 2, 3};
            for (uint32_t k = 0; k < f.n; k++) key[k] = bin[f.ids[k]].left == 0u ? 0u : 1u + need[f.ids[k]];
            for (uint32_t i = 1; i < f.n; i++)                                  // (insertion sort, stable)
                for (uint32_t j = i; j > 0 && key[perm[j]] < key[perm[j - 1]]; j--) std::swap(perm[j], perm[j - 1]);
            uint32_t worst = 0, nodes = 1;
            for (uint32_t k = 0; k < f.n; k++) {
                const uint32_t id = f.ids[perm[k]];
                entry[(size_t)f.id * 4 + k] = id;
                const bool leaf = bin[id].left == 0u;
                worst = std::max(worst, f.n - 1u - k + (leaf ? 0u : need[id]));
                nodes += leaf ? 0u : size[id];
            }
            need[f.id] = worst; size[f.id] = nodes;
            count[f.id] = (uint8_t)f.n;
            stack.pop_back();
        }
    }
};

// The whole build: boxes[n] (the triangles' grown boxes, slot order) -> nodes (root = node 0), every stored plane moved
// outward by gamma; `bound` >= every |stored plane| (the absent entries' corner point sits there).  Returns the most
// entries the walk ever has waiting (its stack need).  strategy 0: binned surface-area heuristic, 1: object medians.
// order[s]: the input triangle that sits at slot s of the tree's own triangle order (a leaf is a run of slots: an entry's
// reference is kLeaf | first slot); remaining[s]: triangles from slot s to the end of its leaf.
struct BuildOptions { int strategy = 0; uint32_t leaf_max = 2; float leaf_rho = 0.2f; unsigned threads = 1; double *phase_ms = nullptr; /* [3] binary tree, four-wide, emit */ };
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
    const auto clock0 = std::chrono::steady_clock::now();
    Builder builder;
    builder.tri = boxes;
    builder.median_only = opt.strategy == 1;
    builder.leaf_max = std::max(1u, opt.leaf_max);
    builder.leaf_rho = opt.leaf_rho;
    builder.run(n, std::max(1u, opt.threads));
    const auto clock1 = std::chrono::steady_clock::now();
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
        order = std::move(builder.order);
        return 0u;
    }
    // ---- four entries a node: the subtrees a few levels down settle side by side, then the top
    const unsigned threads = std::max(1u, opt.threads);
    FourWide four(builder);
    std::vector<uint32_t> frontier;
    std::vector<uint8_t> is_frontier(builder.nodes.size(), 0);
    if (threads > 1u && n >= kParallelAbove) {
        std::vector<std::pair<uint32_t, uint32_t>> walk{{0u, 0u}};
        while (!walk.empty()) {
            const auto [id, depth] = walk.back();
            walk.pop_back();
            if (depth == 5u) { frontier.push_back(id); is_frontier[id] = 1; continue; }
            uint32_t ids[4];
            const uint32_t c = four.open(id, ids);
            for (uint32_t k = 0; k < c; k++) if (builder.nodes[ids[k]].left != 0u) walk.push_back({ids[k], depth + 1u});
        }
    }
    const auto over_frontier = [&](auto &&fn) {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        const auto run = [&] { for (size_t i; (i = next.fetch_add(1)) < frontier.size();) fn(i); };
        for (unsigned t = 1; t < std::min<size_t>(threads, frontier.size()); t++) pool.emplace_back(run);
        run();
        for (auto &t : pool) t.join();
    };
    if (!frontier.empty()) over_frontier([&](size_t i) { four.settle(frontier[i]); });
    four.settle(0u);
    const uint32_t need = four.need[0];
    const auto clock2 = std::chrono::steady_clock::now();
    // ---- numbered in the walk's own order (depth first, entries as stored: a subtree's nodes lie together), written out
    quads.resize(four.size[0]);
    const std::vector<BinNode> &bin = builder.nodes;
    std::vector<std::pair<uint32_t, uint32_t>> jobs;          // {head, index} of the subtrees written side by side
    const auto write = [&](uint32_t root, uint32_t root_index, bool top) {
        std::vector<std::pair<uint32_t, uint32_t>> stack{{root, root_index}};
        while (!stack.empty()) {
            const auto [id, index] = stack.back();
            stack.pop_back();
            if (top && is_frontier[id]) { jobs.push_back({id, index}); continue; }
            Quad &node = quads[index];
            std::memset(&node, 0, sizeof node);
            const uint32_t c = four.count[id];
            uint32_t running = index + 1u;
            for (uint32_t k = 0; k < 4u; k++) {
                if (k >= c) { absent(node, (int)k, bin[four.entry[(size_t)id * 4]].first); continue; }
                const uint32_t e = four.entry[(size_t)id * 4 + k];
                const BinNode &b = bin[e];
                store(node, (int)k, b.box);
                if (b.left == 0u) {
                    node.ref[k] = kLeaf | b.first;
                    for (uint32_t i = 0; i < b.count; i++) remaining[b.first + i] = b.count - i;
                } else {
                    node.ref[k] = running;
                    stack.push_back({e, running});
                    running += four.size[e];
                }
            }
            node.count = c;
        }
    };
    write(0u, 0u, !frontier.empty());
    if (!jobs.empty()) {
        frontier.clear();          // (over_frontier runs over `jobs` now)
        frontier.resize(jobs.size());
        over_frontier([&](size_t i) { write(jobs[i].first, jobs[i].second, false); });
    }
    order = std::move(builder.order);
    if (opt.phase_ms) {
        const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        opt.phase_ms[0] = ms(clock0, clock1); opt.phase_ms[1] = ms(clock1, clock2); opt.phase_ms[2] = ms(clock2, std::chrono::steady_clock::now());
    }
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
