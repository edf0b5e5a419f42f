// ct_spheres.h -- the scene's sphere objects (CTGPU_F_SPHERES), shared by the general path (ct_kernels.hip) and the
// wavefront pipeline (ct_pipeline.hip).
#ifndef CT_SPHERES_H
#define CT_SPHERES_H

#include "ct_params.h"

namespace ctd {

// ---- spheres (CTGPU_F_SPHERES; the reference parses them but has not rendered them since it got its BVH,
// raythread.cpp:207 "TODO add sphere's to the bvh").  IntersectRaySphere as written (raythread.cpp:158-174): float
// coefficients of double dot products; a negative discriminant falls through to the square root, whose NaN no
// comparison accepts.
__device__ __forceinline__ static void sphere_roots(const d3 &o, const d3 &d, const ctgpu_sphere &s, float &t1, float &t2) {
    const float radius = s.radius;
    const d3 co = o - load_d3(s.center);
    const float a = dotf(d, d);
    const float b = 2 * dotf(co, d);
    const float c = dotf(co, co) - (radius * radius);
    const float discriminant = b * b - 4 * a * c;
    t1 = (-b + __builtin_sqrtf(discriminant)) / (2 * a);
    t2 = (-b - __builtin_sqrtf(discriminant)) / (2 * a);
}

// The spheres' tree, depth first, left before right, leaf entries in index order; a box is entered when the slab
// test of bvh.cpp:165-179 passes against the closest distance so far (tmin <= it).  A root is accepted by the rule
// of the reference's own pre-BVH loop (the commented block raythread.cpp:230-246): tmin <= t <= tmax and
// t <= tclosest, t1 before t2.  Small scenes only: one lane, one private stack, the reference's divisions.
__device__ __noinline__ static void walk_spheres(const FrameParams &P, const d3 &o, const d3 &d, float t_min, float t_max,
                                          float &tclosest, int &sphere) {
    uint32_t pending[CTGPU_MAX_TREE_DEPTH + 1];
    int sp = 0;
    uint32_t node_id = 0;
    for (;;) {
        const ctgpu_node *n = P.sphere_nodes + node_id;
        float lo, hi;
        slab(o, d, n->bmin, n->bmax, lo, hi);
        bool descend = hi >= lo && lo <= tclosest && hi > 0;
        if (descend && n->count) {
            for (uint32_t k = 0; k < n->count; k++) {
                const uint32_t id = P.sphere_indexes[n->first + k];
                float t1, t2;
                sphere_roots(o, d, P.spheres[id], t1, t2);
                if (t_min <= t1 && t1 <= t_max && t1 <= tclosest) { tclosest = t1; sphere = (int)id; }
                if (t_min <= t2 && t2 <= t_max && t2 <= tclosest) { tclosest = t2; sphere = (int)id; }
            }
            descend = false;
        }
        if (descend) {
            pending[sp++] = n->left + 1;
            node_id = n->left;
            continue;
        }
        if (sp == 0) return;
        node_id = pending[--sp];
    }
}


}  // namespace ctd
#endif
