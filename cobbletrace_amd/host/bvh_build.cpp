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
#include "cobbletrace_host.h"

#include <cstdlib>
#include <vector>

namespace {

inline double component(const v3_t &v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
inline float component_f32(const v3_t &v, int axis) { return (float)component(v, axis); }

inline double lesser(double a, double b) { return (a < b) ? a : b; }     // mymath.h:15-17
inline double greater(double a, double b) { return (a > b) ? a : b; }    // mymath.h:11-13

void fit_bounds(bvh_state_t *s, uint32_t node_id) {
    bvh_node_t &node = s->bvhNodes[node_id];
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

void split(bvh_state_t *s, uint32_t node_id) {
    bvh_node_t *node = &s->bvhNodes[node_id];
    if (node->triangleCount <= 2) return;

    const v3_t extent{node->aabbMax.x - node->aabbMin.x, node->aabbMax.y - node->aabbMin.y,
                      node->aabbMax.z - node->aabbMin.z};
    int axis = 0;
    if (extent.y > extent.x) axis = 1;
    if (extent.z > (double)component_f32(extent, axis)) axis = 2;
    const float plane = component_f32(node->aabbMin, axis) + component_f32(extent, axis) * 0.5f;

    uint32_t *order = s->triangles.indexes;
    int i = (int)node->firstTriangleIndex;
    int j = i + (int)node->triangleCount - 1;
    while (i <= j) {
        if (component_f32(s->triangles.data[order[i]].centroid, axis) < plane) {
            i++;
        } else {
            const uint32_t tmp = order[i];
            order[i] = order[j];
            order[j--] = tmp;
        }
    }

    const uint32_t left_count = (uint32_t)(i - (int)node->firstTriangleIndex);
    if (left_count == 0 || left_count == node->triangleCount) return;

    const uint32_t left = s->nodesUsed++;
    const uint32_t right = s->nodesUsed++;
    node->leftNode = left;
    s->bvhNodes[left].firstTriangleIndex = node->firstTriangleIndex;
    s->bvhNodes[left].triangleCount = left_count;
    s->bvhNodes[right].firstTriangleIndex = (uint32_t)i;
    s->bvhNodes[right].triangleCount = node->triangleCount - left_count;
    node->triangleCount = 0;

    fit_bounds(s, left);
    fit_bounds(s, right);
    split(s, left);
    split(s, right);
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
    split(s, s->rootNodeIdx);
}
