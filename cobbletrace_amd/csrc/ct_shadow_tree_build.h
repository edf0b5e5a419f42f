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
    uint32_t axis;                // the entries are listed in ascending order of their boxes' centres along this axis (0 x, 1 y, 2 z)
    uint32_t pad_[2];
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

// binary tree: an inner node's children are `left` and `right`; a LEAF (left == kNone) holds `count` triangles, the entries
// leaf_tri[first .. first + count) of the tree's triangle list
constexpr uint32_t kNone = 0xffffffffu;
struct BinNode { Box box; uint32_t left, right, first, count; };

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
    std::vector<uint32_t> scratch;    // (the big splits move their triangles through it)
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

    // split node `id` in place (its range of `order`); false: it stays a leaf.  wide: every thread works on this one split
    // (the big ranges at the top, one after another), else the calling thread alone (the many ranges below, side by side)
    bool split(uint32_t id, uint32_t depth, bool wide = false) {
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
        const unsigned chunks = (wide && count >= kParallelAbove && threads > 1u) ? threads : 1u;
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
                if (chunks > 1u) {
                    // every chunk counts its two sides, the counts become places, every chunk moves its triangles there
                    std::vector<uint32_t> lefts(chunks + 1u, 0u);
                    const uint32_t step = (count + chunks - 1u) / chunks;
                    in_chunks(count, chunks, [&](unsigned c, uint32_t b, uint32_t e) {
                        uint32_t l = 0;
                        for (uint32_t i = b; i < e; i++) l += bin_of(idx[i], a) <= best_bin ? 1u : 0u;
                        lefts[c + 1u] = l;
                    });
                    for (unsigned c = 0; c < chunks; c++) lefts[c + 1u] += lefts[c];
                    mid = lefts[chunks];
                    scratch.resize(order.size());
                    uint32_t *out = scratch.data() + node.first;
                    in_chunks(count, chunks, [&](unsigned c, uint32_t b, uint32_t e) {
                        uint32_t l = lefts[c], r = mid + (std::min(count, c * step) - lefts[c]);
                        for (uint32_t i = b; i < e; i++) { if (bin_of(idx[i], a) <= best_bin) out[l++] = idx[i]; else out[r++] = idx[i]; }
                    });
                    in_chunks(count, chunks, [&](unsigned, uint32_t b, uint32_t e) { std::memcpy(idx + b, out + b, (size_t)(e - b) * 4u); });
                } else
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
        l.first = node.first; l.count = mid; l.left = l.right = kNone;
        r.first = node.first + mid; r.count = count - mid; r.left = r.right = kNone;
        if (boxes_known) { l.box = left_box; r.box = right_box; }
        else {
            l.box.clear(); r.box.clear();
            for (uint32_t i = 0; i < mid; i++) l.box.grow(tri[idx[i]]);
            for (uint32_t i = mid; i < count; i++) r.box.grow(tri[idx[i]]);
        }
        node.left = left; node.right = left + 1u;
        return true;
    }
    void build_alone(uint32_t id, uint32_t depth) {
        std::vector<std::pair<uint32_t, uint32_t>> stack{{id, depth}};
        while (!stack.empty()) {
            const auto [at, d] = stack.back();
            stack.pop_back();
            if (!split(at, d)) continue;
            stack.push_back({nodes[at].left, d + 1});
            stack.push_back({nodes[at].right, d + 1});
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
                todo.push_back({nodes[job.first].right, job.second + 1});
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
        root.first = 0; root.count = n; root.left = root.right = kNone;
        const unsigned chunks = (n >= kParallelAbove && threads > 1u) ? threads : 1u;
        std::vector<Box> part(chunks);
        in_chunks(n, chunks, [&](unsigned c, uint32_t b, uint32_t e) {
            part[c].clear();
            for (uint32_t i = b; i < e; i++) { order[i] = i; part[c].grow(tri[i]); }
        });
        root.box.clear();
        for (unsigned c = 0; c < chunks; c++) root.box.grow(part[c]);
        // the big ranges one after another, every thread on each; what they leave, side by side
        std::vector<std::pair<uint32_t, uint32_t>> big{{0u, 0u}};
        while (!big.empty()) {
            const auto [id, depth] = big.back();
            big.pop_back();
            if (nodes[id].count < kParallelAbove || threads == 1u) { todo.push_back({id, depth}); continue; }
            if (!split(id, depth, true)) continue;
            big.push_back({nodes[id].left, depth + 1u});
            big.push_back({nodes[id].right, depth + 1u});
        }
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
// the order they are stored), how many, what its subtree needs of the walk's stack, how many four-wide nodes and triangles
// it holds.  Entries are stored in ascending order of their boxes' centres along the axis on which those centres spread
// most: a closest-hit walk meets them front to back by going through them forwards or backwards, as its rays point up or
// down that axis (the any-hit walk does not care).  Either way may happen, so a node can leave count - 1 entries waiting
// whichever entry the walk is in: need(node) = count - 1 + the largest need among its entries.
struct FourWide {
    const std::vector<BinNode> &bin;
    std::vector<uint32_t> entry;      // [bin][4]
    std::vector<uint8_t> count;       // [bin] 0: not (yet) a four-wide node's head
    std::vector<uint8_t> axis;        // [bin] the axis its entries are ordered along
    std::vector<uint32_t> need, size, tris; // [bin]
    bool by_need = false;             // the fallback order: leaves first, then subtrees by their need, least first, walked forwards
                                      // only -- need(node) = max over k of (count - 1 - k + need(entry k)), the least any order gives;
                                      // such a node says axis = 3 (the closest-hit walk, which wants front to back, stays off the tree)
    explicit FourWide(const std::vector<BinNode> &nodes) : bin(nodes), entry(nodes.size() * 4u), count(nodes.size(), 0), axis(nodes.size(), 0), need(nodes.size(), 0),
                                                            size(nodes.size(), 0), tris(nodes.size(), 0) {}

    uint32_t open(uint32_t id, uint32_t ids[4]) const {
        uint32_t n = 0;
        ids[n++] = bin[id].left; ids[n++] = bin[id].right;
        while (n < 4u) {
            int widest = -1;
            float area = -1.0f;
            for (uint32_t k = 0; k < n; k++)
                if (bin[ids[k]].left != kNone && bin[ids[k]].box.half_area() > area) { area = bin[ids[k]].box.half_area(); widest = (int)k; }
            if (widest < 0) break;
            const uint32_t opened = ids[widest];
            ids[widest] = bin[opened].left;
            ids[n++] = bin[opened].right;
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
                if (bin[child].left != kNone && !count[child]) push(child);     // (may move `stack`: f is not used after it)
                continue;
            }
            float centre[4][3], lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t k = 0; k < f.n; k++)
                for (int a = 0; a < 3; a++) {
                    centre[k][a] = 0.5f * (bin[f.ids[k]].box.lo[a] + bin[f.ids[k]].box.hi[a]);
                    lo[a] = std::min(lo[a], centre[k][a]); hi[a] = std::max(hi[a], centre[k][a]);
                }
            int along = 0;
            for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[along] - lo[along]) along = a;
            uint32_t perm[4] = {0, 1, 2, 3}, key[4];
            for (uint32_t k = 0; k < f.n; k++) key[k] = bin[f.ids[k]].left == kNone ? 0u : 1u + need[f.ids[k]];
            for (uint32_t i = 1; i < f.n; i++)                                  // (insertion sort, stable)
                for (uint32_t j = i; j > 0 && (by_need ? key[perm[j]] < key[perm[j - 1]] : centre[perm[j]][along] < centre[perm[j - 1]][along]); j--)
                    std::swap(perm[j], perm[j - 1]);
            uint32_t deepest = 0, worst = 0, nodes = 1, below = 0;
            for (uint32_t k = 0; k < f.n; k++) {
                const uint32_t id = f.ids[perm[k]];
                entry[(size_t)f.id * 4 + k] = id;
                const bool leaf = bin[id].left == kNone;
                deepest = std::max(deepest, leaf ? 0u : need[id]);
                worst = std::max(worst, f.n - 1u - k + (leaf ? 0u : need[id]));
                nodes += leaf ? 0u : size[id];
                below += leaf ? bin[id].count : tris[id];
            }
            need[f.id] = by_need ? worst : f.n - 1u + deepest; size[f.id] = nodes; tris[f.id] = below;
            axis[f.id] = by_need ? 3u : (uint8_t)along;
            count[f.id] = (uint8_t)f.n;
            stack.pop_back();
        }
    }
};

// From a binary tree over the triangles (any builder's: this header's, or the device's -- ct_shadow_tree.hip) to what the
// walk reads.  bin[root] is the root; a leaf's triangles are leaf_tri[first .. first + count).  Every stored plane moved
// outward by gamma; `bound` >= every |stored plane| (the absent entries' corner point sits there).  Returns the most
// entries the walk ever has waiting (its stack need).
// order[s]: the input triangle that sits at slot s of the tree's own triangle order (a leaf is a run of slots: an entry's
// reference is kLeaf | first slot); remaining[s]: triangles from slot s to the end of its leaf.
inline uint32_t four_wide_from_binary(const std::vector<BinNode> &bin, uint32_t root, const uint32_t *leaf_tri, uint32_t n, double gamma, float bound,
                                      unsigned threads, std::vector<Quad> &quads, std::vector<uint32_t> &order, std::vector<uint32_t> &remaining,
                                      double *phase_ms = nullptr) {
    // An entry a node does not have: a one-point box at a corner beyond every plane -- a ray's line enters it only by
    // passing through that very point -- and, for the line that does, a reference that is always safe to follow: some
    // leaf below this node.  (Testing a triangle more than the boxes ask for never changes the answer: the test is the
    // reference's own, and it counts only after the reference's own test of its leaf's box.)  The walk looks at all four
    // entries of every node and never asks how many there are.
    const auto absent = [&](Quad &node, int k, uint32_t some_slot) {
        for (int a = 0; a < 3; a++) node.plane[k][a][0] = node.plane[k][a][1] = bound;
        node.ref[k] = kLeaf | some_slot;
    };
    const auto store = [&](Quad &node, int k, const Box &b) {
        for (int a = 0; a < 3; a++) { node.plane[k][a][0] = plane_down(b.lo[a], gamma); node.plane[k][a][1] = plane_up(b.hi[a], gamma); }
    };
    const auto clock1 = std::chrono::steady_clock::now();
    quads.clear();
    order.assign(n, 0u); remaining.assign(n, 1u);
    if (bin[root].left == kNone) {
        // the whole scene is one leaf: a root node with that one entry
        Quad node;
        std::memset(&node, 0, sizeof node);
        store(node, 0, bin[root].box);
        node.ref[0] = kLeaf | 0u;
        for (int k = 1; k < 4; k++) absent(node, k, 0u);
        node.count = 1;
        quads.push_back(node);
        for (uint32_t i = 0; i < n; i++) { remaining[i] = n - i; order[i] = leaf_tri[bin[root].first + i]; }
        return 0u;
    }
    // ---- four entries a node: the subtrees a few levels down settle side by side, then the top
    threads = std::max(1u, threads);
    FourWide four_by_place(bin);
    std::vector<uint32_t> frontier;
    std::vector<uint8_t> is_frontier(bin.size(), 0);
    if (threads > 1u && n >= kParallelAbove) {
        std::vector<std::pair<uint32_t, uint32_t>> walk{{root, 0u}};
        while (!walk.empty()) {
            const auto [id, depth] = walk.back();
            walk.pop_back();
            if (depth == 5u) { frontier.push_back(id); is_frontier[id] = 1; continue; }
            uint32_t ids[4];
            const uint32_t c = four_by_place.open(id, ids);
            for (uint32_t k = 0; k < c; k++) if (bin[ids[k]].left != kNone) walk.push_back({ids[k], depth + 1u});
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
    if (!frontier.empty()) over_frontier([&](size_t i) { four_by_place.settle(frontier[i]); });
    four_by_place.settle(root);
    // (the walk's stack is the 64 lanes of a register: a tree whose positional order could leave more waiting falls back to
    //  the order that needs least -- and tells the closest-hit walk, by axis = 3, to stay off it)
    FourWide four_by_need(bin);
    four_by_need.by_need = true;
    const bool fallback = four_by_place.need[root] > 63u;
    if (fallback) {
        if (!frontier.empty()) over_frontier([&](size_t i) { four_by_need.settle(frontier[i]); });
        four_by_need.settle(root);
    }
    FourWide &four = fallback ? four_by_need : four_by_place;
    const uint32_t need = four.need[root];
    const auto clock2 = std::chrono::steady_clock::now();
    // ---- numbered in the walk's own order (depth first, entries as stored: a subtree's nodes and triangles lie together)
    quads.resize(four.size[root]);
    struct Job { uint32_t head, index, slot; };
    std::vector<Job> jobs;                                    // the subtrees written side by side
    const auto write = [&](Job from, bool top) {
        std::vector<Job> stack{from};
        while (!stack.empty()) {
            const Job at = stack.back();
            stack.pop_back();
            if (top && is_frontier[at.head]) { jobs.push_back(at); continue; }
            Quad &node = quads[at.index];
            std::memset(&node, 0, sizeof node);
            const uint32_t c = four.count[at.head];
            uint32_t running = at.index + 1u, slot = at.slot;
            for (uint32_t k = 0; k < 4u; k++) {
                if (k >= c) { absent(node, (int)k, at.slot); continue; }        // (at.slot: where this node's first leaf begins)
                const uint32_t e = four.entry[(size_t)at.head * 4 + k];
                const BinNode &b = bin[e];
                store(node, (int)k, b.box);
                if (b.left == kNone) {
                    node.ref[k] = kLeaf | slot;
                    for (uint32_t i = 0; i < b.count; i++) { order[slot + i] = leaf_tri[b.first + i]; remaining[slot + i] = b.count - i; }
                    slot += b.count;
                } else {
                    node.ref[k] = running;
                    stack.push_back(Job{e, running, slot});
                    running += four.size[e];
                    slot += four.tris[e];
                }
            }
            node.count = c;
            node.axis = four.axis[at.head];
        }
    };
    write(Job{root, 0u, 0u}, !frontier.empty());
    if (!jobs.empty()) {
        frontier.assign(jobs.size(), 0u);                     // (over_frontier runs over `jobs` now)
        over_frontier([&](size_t i) { write(jobs[i], false); });
    }
    if (phase_ms) {
        const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        phase_ms[1] = ms(clock1, clock2); phase_ms[2] = ms(clock2, std::chrono::steady_clock::now());
    }
    return need;
}

// The whole build on the host: boxes[n] (the triangles' grown boxes, slot order) -> a binary tree by the binned surface-area
// heuristic (strategy 0) or object medians (1), leaves of up to leaf_max triangles -> four_wide_from_binary.
struct BuildOptions { int strategy = 0; uint32_t leaf_max = 2; float leaf_rho = 0.2f; unsigned threads = 1; double *phase_ms = nullptr; /* [3] binary tree, four-wide, emit */ };
inline uint32_t build_shadow_tree(const Box *boxes, uint32_t n, const BuildOptions &opt, double gamma, float bound, std::vector<Quad> &quads,
                                  std::vector<uint32_t> &order, std::vector<uint32_t> &remaining) {
    quads.clear(); order.clear(); remaining.clear();
    if (n == 0u) return 0u;
    const auto clock0 = std::chrono::steady_clock::now();
    Builder builder;
    builder.tri = boxes;
    builder.median_only = opt.strategy == 1;
    builder.leaf_max = std::max(1u, opt.leaf_max);
    builder.leaf_rho = opt.leaf_rho;
    builder.run(n, std::max(1u, opt.threads));
    if (opt.phase_ms) opt.phase_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - clock0).count();
    return four_wide_from_binary(builder.nodes, 0u, builder.order.data(), n, gamma, bound, opt.threads, quads, order, remaining, opt.phase_ms);
}

// The top of a tree rebuilt by the surface-area heuristic.  Bottom-up clustering makes good small subtrees and mediocre tops
// (the last merges join whatever is left along the curve); the top is also where every packet of rays passes.  So the tree
// is cut where subtrees hold at most `cut` triangles, those subtrees' boxes become the primitives of a binned top-down
// build (a few thousand of them: a millisecond), and its leaves are the subtrees.  bin grows by the new inner nodes; the
// nodes above the cut stay in the array, unreferenced.  count_below: scratch.  Returns the new root.
inline uint32_t resplit_top(std::vector<BinNode> &bin, uint32_t root, uint32_t n, uint32_t cut) {
    if (cut == 0u || bin[root].left == kNone) return root;
    std::vector<uint32_t> below(bin.size(), 0u);
    for (size_t i = 0; i < bin.size(); i++)                       // (children lie below their parents in the array)
        below[i] = bin[i].left == kNone ? bin[i].count : below[bin[i].left] + below[bin[i].right];
    std::vector<uint32_t> heads;
    std::vector<uint32_t> stack{root};
    while (!stack.empty()) {
        const uint32_t id = stack.back();
        stack.pop_back();
        if (bin[id].left == kNone || below[id] <= cut) { heads.push_back(id); continue; }
        stack.push_back(bin[id].right); stack.push_back(bin[id].left);
    }
    if (heads.size() < 3u) return root;
    std::vector<Box> boxes(heads.size());
    for (size_t k = 0; k < heads.size(); k++) boxes[k] = bin[heads[k]].box;
    Builder top;
    top.tri = boxes.data();
    top.leaf_max = 1;
    top.run((uint32_t)heads.size(), 1u);
    const uint32_t made = top.used.load();
    const uint32_t base = (uint32_t)bin.size();
    std::vector<uint32_t> id_of(made, kNone);
    uint32_t fresh = 0;
    for (uint32_t j = 0; j < made; j++)
        id_of[j] = top.nodes[j].left == kNone ? heads[top.order[top.nodes[j].first]] : base + fresh++;
    bin.resize((size_t)base + fresh);
    for (uint32_t j = 0; j < made; j++) {
        if (top.nodes[j].left == kNone) continue;
        BinNode &node = bin[id_of[j]];
        node.box = top.nodes[j].box;
        node.left = id_of[top.nodes[j].left]; node.right = id_of[top.nodes[j].right];
        node.first = 0; node.count = 0;
    }
    (void)n;
    return id_of[0];
}

// A binary tree that came without leaf decisions (the device's agglomerative build: every leaf one triangle, leaf i =
// triangle tri_of_leaf[i], inner nodes behind them): an inner node whose two children are single triangles becomes a leaf of
// two where their boxes fill its own -- Builder's rule, c - sum(area_i) / area <= rho c -- its pair appended to leaf_tri.
// bin[0 .. n) are the leaves on entry (first = i, count = 1); leaf_tri must have room for n + 2 * (bin.size() - n) entries.
inline void pair_up_leaves(std::vector<BinNode> &bin, uint32_t n, float leaf_rho, std::vector<uint32_t> &leaf_tri) {
    uint32_t at = n;
    for (size_t id = n; id < bin.size(); id++) {
        BinNode &b = bin[id];
        if (b.left == kNone || b.left >= n || b.right >= n) continue;
        const float whole = b.box.half_area(), parts = bin[b.left].box.half_area() + bin[b.right].box.half_area();
        if (whole > 0.0f && 2.0f - parts / whole > leaf_rho * 2.0f) continue;
        leaf_tri[at] = leaf_tri[b.left]; leaf_tri[at + 1u] = leaf_tri[b.right];
        b.first = at; b.count = 2u; b.left = b.right = kNone;
        at += 2u;
    }
}

// What the walk reads: the same nodes with every reference turned into the BYTE OFFSET of what it names (a node: index *
// 128; a leaf: kLeaf | first record * record_bytes | its number of triangles - 1 in the four low bits, which a multiple of
// 16 leaves free), so that a visit's scalar loads take the reference as it is and a leaf's size is known without its
// records.  false: an offset would not fit below kLeaf, or a leaf holds more than 16 triangles.
inline bool references_to_byte_offsets(std::vector<Quad> &quads, uint32_t n, uint32_t record_bytes, const std::vector<uint32_t> &remaining) {
    if ((uint64_t)quads.size() * sizeof(Quad) >= (uint64_t)kLeaf || (uint64_t)n * record_bytes >= (uint64_t)kLeaf || record_bytes % 16u != 0u) return false;
    for (Quad &q : quads)
        for (int k = 0; k < 4; k++) {
            if (!(q.ref[k] & kLeaf)) { q.ref[k] *= (uint32_t)sizeof(Quad); continue; }
            const uint32_t slot = q.ref[k] & ~kLeaf;
            if (slot >= n || remaining[slot] < 1u || remaining[slot] > 16u) return false;
            q.ref[k] = kLeaf | (slot * record_bytes) | (remaining[slot] - 1u);
        }
    return true;
}

}  // namespace ctst
#endif
