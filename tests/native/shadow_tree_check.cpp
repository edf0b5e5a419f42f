// shadow_tree_check.cpp -- CPU harness of the shadow tree's builder (cobbletrace_amd/csrc/ct_shadow_tree_build.h), driven
// by tests/test_shadow_tree_build.py: structure of the tree, and the walk's control flow (the stack discipline of
// ct_walk_shadow_tree, one ray a packet) against a loop over all triangles' boxes.
//   usage: shadow_tree_check <n> <seed> <strategy> <layout> <threads> [leaf_max] [leaf_fill_percent]
#include "ct_shadow_tree_build.h"
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace ctst;

// The device's clustering (ct_shadow_tree.hip: k_nearest, k_mutual, k_merge) restated for the CPU, so that what the host does
// with such a tree -- pair_up_leaves, four_wide_from_binary over leaves that are NOT runs of one array -- is held to the same
// contract as the host builder's trees.  Clusters in input order (the device sorts along a curve first; any order is a tree).
static void cluster_on_cpu(const std::vector<Box> &boxes, std::vector<BinNode> &bin, uint32_t &root, std::vector<uint32_t> &leaf_tri) {
    const uint32_t n = (uint32_t)boxes.size(), m = 2u * n - 1u;
    bin.assign(m, BinNode{});
    leaf_tri.assign((size_t)n + 2u * (m - n), 0u);
    std::vector<uint32_t> cluster(n), next;
    for (uint32_t i = 0; i < n; i++) { bin[i].box = boxes[i]; bin[i].left = bin[i].right = kNone; bin[i].first = i; bin[i].count = 1; leaf_tri[i] = i; cluster[i] = i; }
    uint32_t next_node = n;
    const int r = 16;
    while (cluster.size() > 1) {
        const int live = (int)cluster.size();
        std::vector<int> nearest(live);
        for (int i = 0; i < live; i++) {
            float best = INFINITY; int place = -1;
            for (int d = -r; d <= r; d++) {
                const int j = i + d;
                if (d == 0 || j < 0 || j >= live) continue;
                Box u = bin[cluster[i]].box; u.grow(bin[cluster[j]].box);
                const float area = u.half_area();
                if (area < best || place < 0) { best = area; place = j; }
            }
            nearest[i] = place;
        }
        next.clear();
        for (int i = 0; i < live; i++) {
            const int j = nearest[i];
            const bool mutual = nearest[j] == i;
            if (mutual && j < i) continue;
            uint32_t id = cluster[i];
            if (mutual) {
                id = next_node++;
                bin[id].left = cluster[i]; bin[id].right = cluster[j]; bin[id].first = 0; bin[id].count = 0;
                bin[id].box = bin[cluster[i]].box; bin[id].box.grow(bin[cluster[j]].box);
            }
            next.push_back(id);
        }
        cluster.swap(next);
    }
    root = cluster[0];
}

static bool enters(const float plane[3][2], const float inv[3], const float noi[3]) {
    float tmin = -INFINITY, tmax = INFINITY;
    for (int a = 0; a < 3; a++) {
        const float t0 = std::fma(plane[a][0], inv[a], noi[a]), t1 = std::fma(plane[a][1], inv[a], noi[a]);
        tmin = std::max(tmin, std::min(t0, t1));
        tmax = std::min(tmax, std::max(t0, t1));
    }
    return tmax >= std::max(tmin, 0.0f);
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const uint32_t n = (uint32_t)std::atoll(argv[1]);
    const uint32_t seed = (uint32_t)std::atoll(argv[2]);
    const int strategy = std::atoi(argv[3]), layout = std::atoi(argv[4]);
    const unsigned threads = (unsigned)std::atoi(argv[5]);
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<Box> boxes(n);
    for (uint32_t i = 0; i < n; i++) {
        float c[3], h[3];
        for (int a = 0; a < 3; a++) {
            switch (layout) {
                case 0: c[a] = U(rng) * 20.0f - 10.0f; h[a] = U(rng) * 0.05f; break;                    // a soup of small boxes
                case 1: c[a] = a == 1 ? 0.0f : U(rng) * 20.0f - 10.0f; h[a] = a == 1 ? 0.0f : 0.02f; break;   // flat: a sheet
                case 2: c[a] = 1.0f; h[a] = 0.25f; break;                                                 // all the same box
                case 3: c[a] = std::ldexp(1.0f, -(int)(i % 40u)) * (a + 1); h[a] = c[a] * 0.01f; break;   // exponentially spaced
                default: c[a] = (float)((i >> (a * 3)) & 7u); h[a] = U(rng) * 3.0f; break;                // big overlapping boxes on a lattice
            }
        }
        for (int a = 0; a < 3; a++) { boxes[i].lo[a] = c[a] - h[a]; boxes[i].hi[a] = c[a] + h[a]; }
    }
    const double gamma = 1e-5;
    const float bound = 64.0f;
    std::vector<Quad> quads;
    std::vector<uint32_t> order, remaining;
    BuildOptions opt;
    opt.strategy = strategy; opt.threads = threads;
    opt.leaf_max = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 2u;
    opt.leaf_rho = argc > 7 ? (float)std::atoi(argv[7]) * 0.01f : 0.2f;
    double phase_ms[3] = {0, 0, 0};
    opt.phase_ms = phase_ms;
    uint32_t need = 0;
    if (strategy == 2 && n >= 2) {
        std::vector<BinNode> bin;
        std::vector<uint32_t> leaf_tri;
        uint32_t root = 0;
        cluster_on_cpu(boxes, bin, root, leaf_tri);
        if (opt.leaf_max >= 2u) pair_up_leaves(bin, n, opt.leaf_rho, leaf_tri);
        root = resplit_top(bin, root, n, n >= 64 ? 16u : 2u);          // (the top rebuilt over the subtrees of at most that many triangles)
        need = four_wide_from_binary(bin, root, leaf_tri.data(), n, gamma, bound, threads, quads, order, remaining, phase_ms);
        if (opt.leaf_max > 2u) opt.leaf_max = 2u;          // (pairs are all this path makes)
    } else {
        need = build_shadow_tree(boxes.data(), n, opt, gamma, bound, quads, order, remaining);
    }
    if (quads.empty() || order.size() != n || remaining.size() != n) { std::printf("FAIL no nodes / orders\n"); return 1; }
    {   // order is a permutation; remaining counts down to 1 within each leaf
        std::vector<uint8_t> once(n, 0);
        for (uint32_t s = 0; s < n; s++) { if (order[s] >= n || once[order[s]]) { std::printf("FAIL order is no permutation\n"); return 1; } once[order[s]] = 1; }
        for (uint32_t s = 0; s < n; s++) {
            if (remaining[s] < 1 || remaining[s] > opt.leaf_max || s + remaining[s] > n || (remaining[s] > 1 && remaining[s + 1] != remaining[s] - 1)) {
                std::printf("FAIL remaining[%u] = %u\n", s, remaining[s]); return 1;
            }
        }
    }

    // ---- structure: a tree (every node reached once), every triangle under exactly one entry, boxes hold what is below
    std::vector<uint8_t> node_seen(quads.size(), 0), tri_seen(n, 0);
    struct Item { uint32_t node; uint32_t waiting; };
    std::vector<Item> stack{{0u, 0u}};
    uint32_t worst = 0;
    std::vector<Box> below(quads.size());
    std::vector<uint32_t> order_visit;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (it.node >= quads.size() || node_seen[it.node]) { std::printf("FAIL node %u reached twice or out of range\n", it.node); return 1; }
        node_seen[it.node] = 1;
        order_visit.push_back(it.node);
        const Quad &q = quads[it.node];
        if (q.count < 1 || q.count > 4) { std::printf("FAIL count\n"); return 1; }
        worst = std::max(worst, it.waiting + q.count - 1u);
        if (q.axis > 3u) { std::printf("FAIL axis\n"); return 1; }
        for (uint32_t k = 1; k < q.count && q.axis < 3u; k++) {          // stored in ascending order of the boxes' centres along the node's axis
            const float c0 = 0.5f * (q.plane[k - 1][q.axis][0] + q.plane[k - 1][q.axis][1]), c1 = 0.5f * (q.plane[k][q.axis][0] + q.plane[k][q.axis][1]);
            if (c1 < c0 - 1e-3f * (std::fabs(c0) + std::fabs(c1) + 1e-6f)) { std::printf("FAIL entries not in order along the axis\n"); return 1; }
        }
        for (uint32_t k = 0; k < q.count; k++) {
            if (q.ref[k] & kLeaf) {
                const uint32_t first = q.ref[k] & 0x3fffffffu;
                if (first >= n || (first > 0 && remaining[first - 1] > 1)) { std::printf("FAIL leaf entry does not start a leaf\n"); return 1; }
                for (uint32_t i = 0; i < remaining[first]; i++) {
                    const uint32_t t = order[first + i];
                    if (tri_seen[t]) { std::printf("FAIL triangle under two entries\n"); return 1; }
                    tri_seen[t] = 1;
                    for (int a = 0; a < 3; a++)
                        if (!(q.plane[k][a][0] <= boxes[t].lo[a] - (float)gamma * 0.999f && q.plane[k][a][1] >= boxes[t].hi[a] + (float)gamma * 0.999f)) {
                            std::printf("FAIL triangle %u not inside its entry's box by gamma\n", t); return 1;
                        }
                }
            } else {
                stack.push_back({q.ref[k], it.waiting + q.count - 1u - (q.axis == 3u ? k : 0u)});        // (axis < 3: the walk may go through the entries either way)
            }
        }
        for (uint32_t k = q.count; k < 4; k++) {
            // an entry that is not there: a point beyond every plane, and a reference that is safe to follow (one triangle)
            for (int a = 0; a < 3; a++) if (!(q.plane[k][a][0] == bound && q.plane[k][a][1] == bound)) { std::printf("FAIL absent entry is not the corner point\n"); return 1; }
            if (!(q.ref[k] & kLeaf) || (q.ref[k] & 0x3fffffffu) >= n || ((q.ref[k] & 0x3fffffffu) > 0 && remaining[(q.ref[k] & 0x3fffffffu) - 1] > 1)) { std::printf("FAIL absent entry's reference\n"); return 1; }
        }
    }
    for (uint32_t t = 0; t < n; t++) if (!tri_seen[t]) { std::printf("FAIL triangle %u under no entry\n", t); return 1; }
    for (size_t i = 0; i < quads.size(); i++) if (!node_seen[i]) { std::printf("FAIL node %zu unreachable\n", i); return 1; }
    if (worst != need && n > 1) { std::printf("FAIL stack need %u, builder says %u\n", worst, need); return 1; }
    // an inner entry's box holds its node's entries' boxes (bottom-up: children have larger indices in depth-first order)
    for (size_t i = quads.size(); i-- > 0;) {
        const Quad &q = quads[i];
        Box b; b.clear();
        for (uint32_t k = 0; k < q.count; k++) {
            Box e;
            for (int a = 0; a < 3; a++) { e.lo[a] = q.plane[k][a][0]; e.hi[a] = q.plane[k][a][1]; }
            b.grow(e);
            if (!(q.ref[k] & kLeaf)) {
                if (q.ref[k] <= i) { std::printf("FAIL child index not behind its parent\n"); return 1; }
                for (int a = 0; a < 3; a++)
                    if (!(e.lo[a] <= below[q.ref[k]].lo[a] && e.hi[a] >= below[q.ref[k]].hi[a])) { std::printf("FAIL entry box does not hold its node\n"); return 1; }
            }
        }
        below[i] = b;
    }

    // ---- the walk: ct_walk_shadow_tree's loop for one ray; the triangles it reaches must include every triangle whose
    //      OWN stored box the ray's line enters (computed the same way), and the stack never exceeds `need`
    uint32_t rays = n > 500000u ? 3u : 200u, mismatches = 0;
    for (uint32_t r = 0; r < rays; r++) {
        float o[3], d[3], inv[3], noi[3];
        for (int a = 0; a < 3; a++) { o[a] = U(rng) * 24.0f - 12.0f; d[a] = (U(rng) - 0.5f); if (std::fabs(d[a]) < 1e-3f) d[a] = 0.01f; inv[a] = 1.0f / d[a]; noi[a] = -o[a] * inv[a]; }
        std::vector<uint8_t> reached(n, 0);
        uint32_t cur = 0, sp = 0, visits = 0;
        uint32_t stack_ref[64];
        bool active = true;
        while (active) {
            if (++visits > 5u * (quads.size() + n) + 1) { std::printf("FAIL walk does not end\n"); return 1; }
            bool popped = false;
            if (cur & kLeaf) {
                const uint32_t first = cur & 0x3fffffffu;
                for (uint32_t i = 0; i < remaining[first]; i++) reached[order[first + i]] = 1;
            } else {
                const Quad &q = quads[cur];
                uint32_t next_ref = 0; bool have = false;
                const bool backwards = q.axis < 3u && d[q.axis] < 0.0f;          // front to back: up or down the node's axis
                for (int j = 3; j >= 0; j--) {
                    const int k = backwards ? 3 - j : j;
                    if (enters(q.plane[k], inv, noi)) {
                        if (have) { if (sp >= 64 || sp + 1 > need) { std::printf("FAIL stack %u beyond need %u\n", sp + 1, need); return 1; } stack_ref[sp++] = next_ref; }
                        next_ref = q.ref[k]; have = true;
                    }
                }
                if (have) { cur = next_ref; popped = true; }
            }
            if (!popped) {
                if (sp == 0) active = false;
                else cur = stack_ref[--sp];
            }
        }
        // (every triangle whose box the line enters lies under entries whose boxes the line enters too: boxes nest)
        for (uint32_t t = 0; t < n; t++) {
            float plane[3][2];
            for (int a = 0; a < 3; a++) { plane[a][0] = plane_down(boxes[t].lo[a], gamma); plane[a][1] = plane_up(boxes[t].hi[a], gamma); }
            if (enters(plane, inv, noi) && !reached[t]) mismatches++;
        }
    }
    // (binary32 monotonicity makes "child entered => parent entered" exact here: same fma, planes further out)
    if (mismatches) { std::printf("FAIL %u triangles a ray's line enters were not reached\n", mismatches); return 1; }
    std::printf("OK n %u nodes %zu need %u (binary tree %.1f ms, four-wide %.1f ms, written %.1f ms)\n", n, quads.size(), need, phase_ms[0], phase_ms[1], phase_ms[2]);
    return 0;
}
