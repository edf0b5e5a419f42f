/*
 * ct_oracle.c -- CPU ORACLE (test infrastructure only; see ct_oracle.h).
 *
 * Arithmetic notation used in the comments: D = IEEE binary64 operation,
 * F = result rounded to binary32.  Build with -ffp-contract=off.
 */
#include "ct_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vectors */
typedef struct { double x, y, z; } v3;

static inline v3 v3_make(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_load(const double *p) { return v3_make(p[0], p[1], p[2]); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_scale(double s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }   /* mymath.h:47-56 */
/* mymath.h:229-232: the products and sums are D, the RESULT is F */
static inline float v3_dot(v3 a, v3 b) { return (float)(a.x * b.x + a.y * b.y + a.z * b.z); }
/* mymath.h:213-216 */
static inline float v3_length(v3 a) { return (float)sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
/* mymath.h:234-242 */
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* the reference's min/max are macros: (a<b)?a:b and (a>b)?a:b  (mymath.h:11-17) */
#define LESSER(a, b) (((a) < (b)) ? (a) : (b))
#define GREATER(a, b) (((a) > (b)) ? (a) : (b))

static const float kNoHit = 1e30f;               /* ray.t sentinel              */
static const float kFarAway = 4294967296.0f;     /* FINF, raythread.cpp:58      */
static const uint32_t kBackground = 0x333333u;   /* raythread.cpp:59            */

/* ------------------------------------------------------------------ colour (color.h) */
static inline uint32_t pack_u8(uint8_t r, uint8_t g, uint8_t b) {          /* color.h:77-80 */
    return ((uint32_t)b << 16) | ((uint32_t)g << 8) | (uint32_t)r;
}
static inline void unpack_u8(uint32_t c, uint8_t *r, uint8_t *g, uint8_t *b) {   /* color.h:87-94 */
    *r = c & 0xff; *g = (c >> 8) & 0xff; *b = (c >> 16) & 0xff;
}

void cto_color_to_hsv(uint32_t color, float hsv[3]) {      /* color.h:114-120 + 49-75 */
    uint8_t r8, g8, b8;
    unpack_u8(color, &r8, &g8, &b8);
    const float r = (float)r8 / 255.0f, g = (float)g8 / 255.0f, b = (float)b8 / 255.0f;
    const float gb_max = GREATER(g, b), gb_min = LESSER(g, b);
    const float hi = GREATER(r, gb_max), lo = LESSER(r, gb_min);
    const float delta = hi - lo;
    float h, s;
    if (hi != 0.0) {
        s = delta / hi;
    } else {
        hsv[0] = -1.0f; hsv[1] = 0.0f; hsv[2] = hi;
        return;
    }
    if (r == hi) h = (g - b) / delta;
    else if (g == hi) h = 2.0f + (b - r) / delta;
    else h = 4.0f + (r - g) / delta;
    h = (float)((double)h * 60.0);
    if (h < 0) h = (float)((double)h + 360.0);
    hsv[0] = h; hsv[1] = s; hsv[2] = hi;
}

uint32_t cto_hsv_to_color(const float hsv[3]) {            /* color.h:122-126 + 18-45 */
    float h = hsv[0];
    const float s = hsv[1], v = hsv[2];
    float r, g, b;
    if (s == 0) {
        r = g = b = v;
    } else {
        h = (float)((double)h / 60.0);
        const int sector = (int)floorf(h);
        const float frac = h - (float)sector;
        const float aa = v * (1.0f - s);
        const float bb = v * (1.0f - (s * frac));
        const float cc = v * (1.0f - (s * (1.0f - frac)));
        r = g = b = 0.0f;   /* sector outside 0..5 leaves the reference's rgb indeterminate */
        switch (sector) {
            case 0: r = v;  g = cc; b = aa; break;
            case 1: r = bb; g = v;  b = aa; break;
            case 2: r = aa; g = v;  b = cc; break;
            case 3: r = aa; g = bb; b = v;  break;
            case 4: r = cc; g = aa; b = v;  break;
            case 5: r = v;  g = aa; b = bb; break;
        }
    }
    /* min(c*0xff, 0xff) as the macro, then float -> uint8 truncation */
    const float r255 = r * 255.0f, g255 = g * 255.0f, b255 = b * 255.0f;
    const float rc = LESSER(r255, 255.0f), gc = LESSER(g255, 255.0f), bc = LESSER(b255, 255.0f);
    return pack_u8((uint8_t)rc, (uint8_t)gc, (uint8_t)bc);
}

uint32_t cto_shade_value(uint32_t material_color, float intensity) {
    float hsv[3];
    cto_color_to_hsv(material_color, hsv);
    hsv[2] = intensity;
    return cto_hsv_to_color(hsv);
}

uint32_t cto_blend_sample(uint32_t accumulated, uint32_t sample) {    /* raythread.cpp:482-498 */
    uint8_t ar, ag, ab, sr, sg, sb;
    unpack_u8(accumulated, &ar, &ag, &ab);
    unpack_u8(sample, &sr, &sg, &sb);
    const float denom = 8.0f;
    float r = (float)ar, g = (float)ag, b = (float)ab;
    r -= (r / denom); r += ((float)sr / denom);
    g -= (g / denom); g += ((float)sg / denom);
    b -= (b / denom); b += ((float)sb / denom);
    return pack_u8((uint8_t)r, (uint8_t)g, (uint8_t)b);
}

uint32_t cto_average_pair(uint32_t a, uint32_t b) {                   /* raythread.cpp:517-523 */
    uint8_t ar, ag, ab, br, bg, bb;
    unpack_u8(a, &ar, &ag, &ab);
    unpack_u8(b, &br, &bg, &bb);
    const float r = ((float)ar + (float)br) / 2.0f, g = ((float)ag + (float)bg) / 2.0f,
                bl = ((float)ab + (float)bb) / 2.0f;
    const float rc = LESSER(r, 255.0f), gc = LESSER(g, 255.0f), bc = LESSER(bl, 255.0f);
    return pack_u8((uint8_t)rc, (uint8_t)gc, (uint8_t)bc);
}

uint32_t cto_mix_reflection(uint32_t local, uint32_t reflected, float reflection) {  /* raythread.cpp:375-379 */
    uint8_t lr, lg, lb, rr, rg, rb;
    unpack_u8(local, &lr, &lg, &lb);
    unpack_u8(reflected, &rr, &rg, &rb);
    const double keep = (double)(1.0f - reflection), take = (double)reflection;
    const double r = keep * (double)lr + take * (double)rr;
    const double g = keep * (double)lg + take * (double)rg;
    const double b = keep * (double)lb + take * (double)rb;
    return pack_u8((uint8_t)r, (uint8_t)g, (uint8_t)b);
}

/* ------------------------------------------------------------------ camera */
void cto_camera_rotation(float yaw, float pitch, float roll, double rot[9]) {   /* raythread.cpp:564-572 */
    const float cy = cosf(yaw), sy = sinf(yaw), cp = cosf(pitch), sp = sinf(pitch), cr = cosf(roll),
                sr = sinf(roll);
    rot[0] = cy * cp;
    rot[1] = cy * sp * sr - sy * cr;
    rot[2] = cy * sp * cr + sy * sr;
    rot[3] = sy * cp;
    rot[4] = sy * sp * sr + cy * cr;
    rot[5] = sy * sp * cr - cy * sr;
    rot[6] = -sy;
    rot[7] = cp * sr;
    rot[8] = cp * cr;
}

void cto_primary_direction(int32_t height, float px, float py, const double rot[9], double dir[3]) {
    /* CanvasToViewport (raythread.cpp:186-194): viewport {1,1,1}; the scale is the F quotient
     * 1.f/(float)height promoted to D; then row-vector x 3x3 (mymath.h:68-75). */
    const float scale = 1.0f / (float)height;
    const double vx = (double)px * (double)scale, vy = (double)py * (double)scale, vz = (double)1.0f;
    dir[0] = vx * rot[0] + vy * rot[3] + vz * rot[6];
    dir[1] = vx * rot[1] + vy * rot[4] + vz * rot[7];
    dir[2] = vx * rot[2] + vy * rot[5] + vz * rot[8];
}

/* ------------------------------------------------------------------ jitter source */
int32_t cto_hash_rand(uint64_t seed, uint64_t k) {
    uint64_t z = (seed + 1u) * 0x9E3779B97F4A7C15ull + k;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (int32_t)(z >> 33);     /* 31 bits, like rand() */
}

uint64_t cto_fnv1a64(const void *data, uint64_t n) {
    const unsigned char *b = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

/* ------------------------------------------------------------------ intersection */
typedef struct { v3 origin, direction; float t; } ray;

typedef struct {
    const ctgpu_scene_desc *scene;
    uint64_t node_visits, triangle_tests;
    uint64_t rays_primary, rays_shadow, rays_reflect;
    int count_work;
    int shadow_early_exit;
    int spheres;        /* CTGPU_F_SPHERES: the scene's sphere objects are rendered too */
    int reflect_fix;    /* CTGPU_F_REFLECT_FIX: reflected rays start at ray.t = 1e30f instead of 0 */
} tracer;

/* bvh.cpp:165-179: six D quotients stored to F, F min/max macros */
static int slab_test(const ray *r, const double bmin[3], const double bmax[3]) {
    const float tx1 = (float)((bmin[0] - r->origin.x) / r->direction.x);
    const float tx2 = (float)((bmax[0] - r->origin.x) / r->direction.x);
    float tmin = LESSER(tx1, tx2);
    float tmax = GREATER(tx1, tx2);
    const float ty1 = (float)((bmin[1] - r->origin.y) / r->direction.y);
    const float ty2 = (float)((bmax[1] - r->origin.y) / r->direction.y);
    tmin = GREATER(tmin, LESSER(ty1, ty2));
    tmax = LESSER(tmax, GREATER(ty1, ty2));
    const float tz1 = (float)((bmin[2] - r->origin.z) / r->direction.z);
    const float tz2 = (float)((bmax[2] - r->origin.z) / r->direction.z);
    tmin = GREATER(tmin, LESSER(tz1, tz2));
    tmax = LESSER(tmax, GREATER(tz1, tz2));
    return tmax >= tmin && tmin < r->t && tmax > 0;
}

/* bvh.cpp:147-163 */
static int triangle_test(ray *r, const double *tri) {
    const v3 p1 = v3_load(tri), p2 = v3_load(tri + 3), p3 = v3_load(tri + 6);
    const v3 e1 = v3_sub(p2, p1), e2 = v3_sub(p3, p1);
    const v3 h = v3_cross(r->direction, e2);
    const float a = v3_dot(e1, h);
    if (a > -0.0001f && a < 0.0001f) return 0;
    const float f = 1.0f / a;
    const v3 s = v3_sub(r->origin, p1);
    const float u = f * v3_dot(s, h);
    if (u < 0 || u > 1) return 0;
    const v3 q = v3_cross(s, e1);
    const float v = f * v3_dot(r->direction, q);
    if (v < 0 || u + v > 1) return 0;
    const float t = f * v3_dot(e2, q);
    if (t > 0.0001f) r->t = LESSER(r->t, t);
    return 1;
}

int cto_intersect_aabb(const double o[3], const double d[3], float ray_t, const double bmin[3],
                       const double bmax[3]) {
    ray r = {v3_load(o), v3_load(d), ray_t};
    return slab_test(&r, bmin, bmax);
}

int cto_intersect_triangle(const double o[3], const double d[3], float *ray_t, const double tri[9]) {
    ray r = {v3_load(o), v3_load(d), *ray_t};
    const int hit = triangle_test(&r, tri);
    *ray_t = r.t;
    return hit;
}

/* bvh.cpp:198-222.  any_hit_exit: leave as soon as ray.t has moved (shadow rays only; see
 * ct_oracle.h cto_options.shadow_early_exit).  Returns 1 when the early exit fired. */
static int walk_closest(tracer *tr, ray *r, uint32_t node_id, float *tclosest, uint32_t *closest,
                        int any_hit_exit) {
    const ctgpu_node *node = &tr->scene->nodes[node_id];
    if (tr->count_work) tr->node_visits++;
    if (!slab_test(r, node->bmin, node->bmax)) return 0;

    if (node->count > 0) {
        for (uint32_t k = 0; k < node->count; k++) {
            const uint32_t tri = tr->scene->indexes[node->first + k];
            if (tr->count_work) tr->triangle_tests++;
            const int hit = triangle_test(r, tr->scene->vertices + (size_t)tri * 9);
            if (hit && r->t != kNoHit && r->t < *tclosest) {
                *closest = tri;
                *tclosest = r->t;
            }
            if (any_hit_exit && r->t != kNoHit) return 1;
        }
        return 0;
    }
    if (walk_closest(tr, r, node->left, tclosest, closest, any_hit_exit)) return 1;
    return walk_closest(tr, r, node->left + 1, tclosest, closest, any_hit_exit);
}

/* ---- spheres (CTGPU_F_SPHERES; off = the reference's pictures, which have none: README.md:20) ----
 * IntersectRaySphere, raythread.cpp:158-174, as written: the quadratic's coefficients are floats of double dot
 * products; a negative discriminant sets both roots to FINF and then FALLS THROUGH to the square root, whose NaN
 * replaces them (and no comparison below accepts a NaN). */
static void sphere_roots(const ray *r, const ctgpu_sphere *s, float *t1, float *t2) {
    const float radius = s->radius;
    const v3 co = v3_sub(r->origin, v3_load(s->center));
    const float a = v3_dot(r->direction, r->direction);
    const float b = 2 * v3_dot(co, r->direction);
    const float c = v3_dot(co, co) - (radius * radius);
    const float discriminant = b * b - 4 * a * c;
    if (discriminant < 0) { *t1 = kFarAway; *t2 = kFarAway; }
    /* sqrt of a float picks C++'s float overload; the sum and the quotient are float operations */
    *t1 = (-b + sqrtf(discriminant)) / (2 * a);
    *t2 = (-b - sqrtf(discriminant)) / (2 * a);
}

/* The spheres' tree, walked like the triangles' (left subtree first, leaf entries in index order); a box is
 * entered when the slab test of bvh.cpp:165-179 passes against the closest distance so far (tmin <= it: a sphere
 * at the very same distance still replaces the holder, as the rule below lets it).  Acceptance is the reference's
 * own, from its pre-BVH loop (the commented block raythread.cpp:230-246). */
static void walk_spheres(tracer *tr, const ray *r, uint32_t node_id, float tmin, float tmax, float *tclosest, int *sphere) {
    const ctgpu_scene_desc *sc = tr->scene;
    const ctgpu_node *node = &sc->sphere_nodes[node_id];
    {
        /* slab parameters as slab_test computes them, compared against *tclosest with <= */
        const float tx1 = (float)((node->bmin[0] - r->origin.x) / r->direction.x);
        const float tx2 = (float)((node->bmax[0] - r->origin.x) / r->direction.x);
        float lo = LESSER(tx1, tx2), hi = GREATER(tx1, tx2);
        const float ty1 = (float)((node->bmin[1] - r->origin.y) / r->direction.y);
        const float ty2 = (float)((node->bmax[1] - r->origin.y) / r->direction.y);
        lo = GREATER(lo, LESSER(ty1, ty2));
        hi = LESSER(hi, GREATER(ty1, ty2));
        const float tz1 = (float)((node->bmin[2] - r->origin.z) / r->direction.z);
        const float tz2 = (float)((node->bmax[2] - r->origin.z) / r->direction.z);
        lo = GREATER(lo, LESSER(tz1, tz2));
        hi = LESSER(hi, GREATER(tz1, tz2));
        if (!(hi >= lo && lo <= *tclosest && hi > 0)) return;
    }
    if (node->count > 0) {
        for (uint32_t k = 0; k < node->count; k++) {
            const uint32_t id = sc->sphere_indexes[node->first + k];
            float t1, t2;
            sphere_roots(r, &sc->spheres[id], &t1, &t2);
            if (tmin <= t1 && t1 <= tmax && t1 <= *tclosest) { *tclosest = t1; *sphere = (int)id; }
            if (tmin <= t2 && t2 <= tmax && t2 <= *tclosest) { *tclosest = t2; *sphere = (int)id; }
        }
        return;
    }
    walk_spheres(tr, r, node->left, tmin, tmax, tclosest, sphere);
    walk_spheres(tr, r, node->left + 1, tmin, tmax, tclosest, sphere);
}

/* raythread.cpp:197-228; *sphere >= 0 when a sphere holds the closest distance (spheres enabled only).
 * tmin / tmax: what the caller passes to ClosestIntersection -- the reference ignores them for triangles
 *  and so does this; they bound the sphere roots. */
static int closest_intersection(tracer *tr, ray r, float tmin, float tmax, float *tclosest, uint32_t *closest, int *sphere,
                                int any_hit_exit) {
    *tclosest = kFarAway;
    *closest = 0;
    *sphere = -1;
    walk_closest(tr, &r, 0, tclosest, closest, any_hit_exit);
    int found = r.t != kNoHit;
    if (tr->spheres && tr->scene->n_spheres && !(any_hit_exit && found)) {
        walk_spheres(tr, &r, 0, tmin, tmax, tclosest, sphere);
        found = found || *sphere >= 0;
    }
    return found;
}

/* ------------------------------------------------------------------ shading */
/* raythread.cpp:275-327 */
static float compute_lighting(tracer *tr, v3 position, v3 normal, v3 view, int specular) {
    const ctgpu_scene_desc *sc = tr->scene;
    float intensity = 0.0f;
    for (uint32_t i = 0; i < sc->n_lights; i++) {
        const ctgpu_light *light = &sc->lights[i];
        v3 to_light;
        if (light->type == CTGPU_LIGHT_AMBIENT) {
            intensity += light->intensity;
            continue;
        } else if (light->type == CTGPU_LIGHT_POINT) {
            to_light = v3_sub(v3_load(light->position), position);
        } else if (light->type == CTGPU_LIGHT_DIRECTIONAL) {
            to_light = v3_load(light->direction);
        } else {
            continue;
        }

        /* shadow ray: full traversal from the surface point, no distance limit for triangles (tMax is
         * computed and ignored, raythread.cpp:289-307); 0.001 .. tMax bound a sphere's roots */
        float tclosest; uint32_t closest; int sphere;
        ray shadow = {position, to_light, kNoHit};
        const float t_max = (light->type == CTGPU_LIGHT_POINT) ? 1.0f : kFarAway;
        tr->rays_shadow++;
        if (closest_intersection(tr, shadow, 0.001f, t_max, &tclosest, &closest, &sphere, tr->shadow_early_exit)) continue;

        const float n_dot_l = v3_dot(normal, to_light);
        if (n_dot_l > 0)
            intensity += light->intensity * n_dot_l / (v3_length(normal) * v3_length(to_light));

        if (specular != -1) {
            /* ReflectRay(L, N) = (2.0 * N) * (D)F(N.L) - L   (raythread.cpp:270-273) */
            const double nl = (double)v3_dot(normal, to_light);
            const v3 reflected = v3_sub(v3_scale(nl, v3_scale(2.0, normal)), to_light);
            const float r_dot_v = v3_dot(reflected, view);
            if (r_dot_v > 0) {
                const float base = r_dot_v / (v3_length(reflected) * v3_length(view));
                intensity = (float)((double)intensity +
                                    (double)light->intensity * pow((double)base, (double)specular));
            }
        }
    }
    return intensity;
}

/* raythread.cpp:353-386 (+ NormalOfSceneObject :329-350) */
static uint32_t trace_ray(tracer *tr, ray r, int depth, int is_reflection) {
    float tclosest; uint32_t closest; int sphere;
    if (is_reflection) tr->rays_reflect++; else tr->rays_primary++;
    /* TraceRay(.., 1, FINF, ..) for camera rays, (.., 0.001, FINF, ..) for reflected ones (raythread.cpp:480, 373) */
    if (!closest_intersection(tr, r, is_reflection ? 0.001f : 1.0f, kFarAway, &tclosest, &closest, &sphere, 0)) return kBackground;

    const ctgpu_scene_desc *sc = tr->scene;
    const double *tri = sc->vertices + (size_t)closest * 9;
    const ctgpu_material *mat = sphere >= 0 ? &sc->spheres[sphere].material : &sc->materials[closest];

    const v3 position = v3_add(r.origin, v3_scale((double)tclosest, r.direction));
    v3 normal;
    if (sphere >= 0) {
        /* NormalOfSceneObject, OT_SPHERE (raythread.cpp:331-334): n = P - C; n / Magnitude(n) = (1.0f / (D)F|n|) * n */
        const v3 n = v3_sub(position, v3_load(sc->spheres[sphere].center));
        normal = v3_scale((double)1.0f / (double)v3_length(n), n);
    } else {
        const v3 p1 = v3_load(tri), p2 = v3_load(tri + 3), p3 = v3_load(tri + 6);
        normal = v3_cross(v3_sub(p2, p1), v3_sub(p3, p1));
        if (!(v3_dot(normal, r.direction) < 0)) normal = v3_neg(normal);
    }

    const float value = compute_lighting(tr, position, normal, v3_neg(r.direction), mat->specular);
    const uint32_t local = cto_shade_value(mat->color, value);

    if (depth <= 0 || mat->reflection <= 0) return local;

    /* ReflectRay(-dir, N); the reflected ray starts with t = 0 (sic, raythread.cpp:373) */
    const v3 incoming = v3_neg(r.direction);
    const double ni = (double)v3_dot(normal, incoming);
    ray bounce = {position, v3_sub(v3_scale(ni, v3_scale(2.0, normal)), incoming), tr->reflect_fix ? kNoHit : 0.0f};
    const uint32_t reflected = trace_ray(tr, bounce, depth - 1, 1);
    return cto_mix_reflection(local, reflected, mat->reflection);
}

/* ------------------------------------------------------------------ frame geometry */
typedef struct {
    int32_t width, height, half, bands, rows_per_band;
} geometry;

static geometry frame_geometry(const ctgpu_frame_desc *f) {
    geometry g;
    g.width = f->width; g.height = f->height;
    g.half = f->height / 2;                                /* raythread.cpp:454, 574 */
    g.bands = f->num_threads;
    g.rows_per_band = f->height / f->num_threads;          /* raythread.cpp:576 */
    return g;
}

/* CanvasPutPixel + PutPixel (raythread.cpp:176-184, draw2d.h:8-20) */
static void put_pixel(uint32_t *fb, const geometry *g, int x, int y, uint32_t color, uint64_t *written) {
    const int col = x + g->width / 2, row = g->height / 2 - y;
    if (row < 0 || row >= g->height) return;
    if (col < 0 || col >= g->width) return;
    fb[(size_t)row * g->width + col] = color;
    if (written) (*written)++;
}

/* traced rows of one band under the subsampling step rule (raythread.cpp:526-535) */
static int next_row(int y, int y_end, int subsampling) {
    if (!subsampling) return y + 1;
    return (y + 2 >= y_end) ? y + 1 : y + 2;
}

typedef struct {
    const ctgpu_scene_desc *scene;
    const ctgpu_frame_desc *frame;
    const cto_options *opt;
    uint32_t *fb;
    geometry g;
    const int32_t *rand_table;     /* LIBC mode: rand() values in one-worker call order */
    const int32_t *row_ordinal;    /* LIBC mode: traced-row ordinal per (y + half), -1 if not traced */
    int32_t traced_rows;
    /* work queue */
    pthread_mutex_t lock;
    int next_band, end_band, band_stride;
    /* totals */
    ctgpu_stats totals;
} frame_job;

static inline int32_t jitter_int(const frame_job *job, int x, int y, int sample, int axis) {
    const ctgpu_frame_desc *f = job->frame;
    const uint64_t xi = (uint64_t)(x + job->g.half), yi = (uint64_t)(y + job->g.half);
    if (f->rng_mode == CTGPU_RNG_LIBC) {
        const uint64_t row = (uint64_t)job->row_ordinal[yi];
        const uint64_t k = ((xi * (uint64_t)job->traced_rows + row) * 16u + (uint64_t)sample) * 2u + (uint64_t)axis;
        return job->rand_table[k];
    }
    const uint64_t k = ((xi * (uint64_t)job->g.height + yi) * 16u + (uint64_t)sample) * 2u + (uint64_t)axis;
    return cto_hash_rand(f->rng_seed, k);
}

/* one pixel (raythread.cpp:459-509) */
static uint32_t render_pixel(const frame_job *job, tracer *tr, int x, int y) {
    const ctgpu_frame_desc *f = job->frame;
    const v3 eye = v3_load(f->cam_pos);
    double dir[3];
    if (!(f->flags & CTGPU_F_SUPERSAMPLE)) {
        cto_primary_direction(f->height, (float)x, (float)y, f->cam_rot, dir);
        ray r = {eye, v3_load(dir), kNoHit};
        return trace_ray(tr, r, f->max_depth, 0);
    }
    const float step = 1.0f / 4.0f, jitter = step / 8.0f;
    float sx = (float)x;
    uint32_t color = 0;
    int first = 1;
    for (int xs = 0; xs < 4; xs++) {
        float sy = (float)y;
        for (int ys = 0; ys < 4; ys++) {
            const int sample = xs * 4 + ys;
            float rx = (float)jitter_int(job, x, y, sample, 0) / 2147483648.0f;   /* (float)RAND_MAX */
            float ry = (float)jitter_int(job, x, y, sample, 1) / 2147483648.0f;
            rx = (rx * (2.0f * jitter)) - jitter;
            ry = (ry * (2.0f * jitter)) - jitter;
            sx += rx;
            sy += ry;
            cto_primary_direction(f->height, sx, sy, f->cam_rot, dir);
            ray r = {eye, v3_load(dir), kNoHit};
            const uint32_t sample_color = trace_ray(tr, r, f->max_depth, 0);
            if (first) { first = 0; color = sample_color; }
            else color = cto_blend_sample(color, sample_color);
            sy -= ry;
            sy += step;
            sx -= rx;
        }
        sx += step;
    }
    return color;
}

/* one worker's band (raythread.cpp:454-537).  Deterministic rule for the one racy write
 * of the reference: the in-between pixel a band's FIRST row would put into the row below
 * it (which belongs to the neighbouring band, raythread.cpp:513-524) is not written --
 * the neighbour's own traced pixel stands. */
static void render_band(frame_job *job, tracer *tr, int band, uint64_t *written) {
    const geometry *g = &job->g;
    const int subsampling = (job->frame->flags & CTGPU_F_SUBSAMPLE) != 0;
    const int y_start = -g->half + band * g->rows_per_band;
    const int y_end = y_start + g->rows_per_band;
    for (int x = -g->half; x < g->half; x++) {
        uint32_t last = 0;
        for (int y = y_start; y < y_end; y = next_row(y, y_end, subsampling)) {
            const uint32_t color = render_pixel(job, tr, x, y);
            put_pixel(job->fb, g, x, y, color, written);
            if (subsampling && y != y_start)
                put_pixel(job->fb, g, x, y - 1, cto_average_pair(last, color), written);
            last = color;
        }
    }
}

static void *worker_main(void *arg) {
    frame_job *job = (frame_job *)arg;
    tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = job->scene;
    tr.count_work = job->opt->count_work;
    tr.shadow_early_exit = job->opt->shadow_early_exit;
    tr.spheres = (job->frame->flags & CTGPU_F_SPHERES) != 0;
    tr.reflect_fix = (job->frame->flags & CTGPU_F_REFLECT_FIX) != 0;
    uint64_t written = 0;
    for (;;) {
        pthread_mutex_lock(&job->lock);
        int band = -1;
        if (job->next_band < job->end_band) { band = job->next_band; job->next_band += job->band_stride; }
        pthread_mutex_unlock(&job->lock);
        if (band < 0) break;
        render_band(job, &tr, band, &written);
    }
    pthread_mutex_lock(&job->lock);
    job->totals.rays_primary += tr.rays_primary;
    job->totals.rays_shadow += tr.rays_shadow;
    job->totals.rays_reflect += tr.rays_reflect;
    job->totals.node_visits += tr.node_visits;
    job->totals.triangle_tests += tr.triangle_tests;
    job->totals.pixels_written += written;
    pthread_mutex_unlock(&job->lock);
    return NULL;
}

int cto_render(const ctgpu_scene_desc *scene, const ctgpu_frame_desc *frame, const cto_options *opt_in,
               uint32_t *fb, ctgpu_stats *stats) {
    cto_options opt;
    if (opt_in) opt = *opt_in; else memset(&opt, 0, sizeof opt);
    if (!scene || !frame || !fb || frame->num_threads <= 0 || frame->width <= 0 || frame->height <= 0 ||
        scene->n_triangles == 0 || scene->n_nodes == 0)
        return CTGPU_ERR_ARG;

    frame_job job;
    memset(&job, 0, sizeof job);
    job.scene = scene; job.frame = frame; job.opt = &opt; job.fb = fb;
    job.g = frame_geometry(frame);
    pthread_mutex_init(&job.lock, NULL);
    job.band_stride = opt.band_stride > 1 ? opt.band_stride : 1;
    job.next_band = opt.band_count > 0 ? opt.band_first : 0;
    job.end_band = opt.band_count > 0 ? opt.band_first + (opt.band_count - 1) * job.band_stride + 1 : job.g.bands;
    if (job.next_band < 0 || job.end_band > job.g.bands) return CTGPU_ERR_ARG;

    int32_t *table = NULL, *ordinal = NULL;
    if ((frame->flags & CTGPU_F_SUPERSAMPLE) && frame->rng_mode == CTGPU_RNG_LIBC) {
        /* rand() in the call order of a single worker: x outer, traced rows inner,
         * 16 samples, rx before ry (raythread.cpp:455-471). */
        const int subsampling = (frame->flags & CTGPU_F_SUBSAMPLE) != 0;
        const int rows = job.g.bands * job.g.rows_per_band;
        ordinal = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1));
        for (int i = 0; i < rows; i++) ordinal[i] = -1;
        int n = 0;
        for (int b = 0; b < job.g.bands; b++) {
            const int y0 = b * job.g.rows_per_band, y1 = y0 + job.g.rows_per_band;
            for (int y = y0; y < y1; y = next_row(y, y1, subsampling)) ordinal[y] = n++;
        }
        job.traced_rows = n;
        const uint64_t count = (uint64_t)(2 * job.g.half) * (uint64_t)n * 32u;
        table = (int32_t *)malloc(sizeof(int32_t) * (size_t)(count ? count : 1));
        srand((unsigned)frame->rng_seed);
        for (uint64_t i = 0; i < frame->rng_offset; i++) (void)rand();
        for (uint64_t i = 0; i < count; i++) table[i] = rand();
        job.rand_table = table;
        job.row_ordinal = ordinal;
    }

    int n_threads = opt.os_threads > 0 ? opt.os_threads
                                       : (job.end_band - job.next_band + job.band_stride - 1) / job.band_stride;
    if (n_threads < 1) n_threads = 1;
    pthread_t *threads = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int i = 0; i < n_threads; i++) pthread_create(&threads[i], NULL, worker_main, &job);
    for (int i = 0; i < n_threads; i++) pthread_join(threads[i], NULL);
    free(threads);
    free(table);
    free(ordinal);
    pthread_mutex_destroy(&job.lock);
    if (stats) *stats = job.totals;
    return CTGPU_OK;
}

int cto_trace_rays(const ctgpu_scene_desc *scene, const ctgpu_ray *rays, uint64_t n, ctgpu_hit *hits) {
    if (!scene || !rays || !hits) return CTGPU_ERR_ARG;
    tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = scene;
    for (uint64_t i = 0; i < n; i++) {
        ray r = {v3_load(rays[i].origin), v3_load(rays[i].direction), rays[i].t};
        float tclosest = kFarAway; uint32_t closest = 0;
        walk_closest(&tr, &r, 0, &tclosest, &closest, 0);
        hits[i].t = r.t;
        hits[i].tclosest = tclosest;
        hits[i].triangle = closest;
        hits[i].found = (r.t != kNoHit);
    }
    return CTGPU_OK;
}

/* ------------------------------------------------------------------ BvhTest (bvh.cpp:137-145, 223-310) */
void cto_bvhtest_triangles(double verts[256 * 9]) {
    srand(1);
    for (int i = 0; i < 256; i++) {
        double r[9];
        for (int k = 0; k < 9; k++) r[k] = (double)((float)rand() / (float)RAND_MAX);   /* RandVertex */
        double *t = verts + i * 9;
        for (int c = 0; c < 3; c++) {
            t[c] = 9.0 * r[c] - (double)5.0f;       /* p1 = r0 * 9 - v5 */
            t[3 + c] = t[c] + r[3 + c];             /* p2 = p1 + r1    */
            t[6 + c] = t[c] + r[6 + c];             /* p3 = p1 + r2    */
        }
    }
}

/* bvh.cpp:181-196: any-order traversal without closest bookkeeping */
static void walk_any(const ctgpu_scene_desc *sc, ray *r, uint32_t node_id) {
    const ctgpu_node *node = &sc->nodes[node_id];
    if (!slab_test(r, node->bmin, node->bmax)) return;
    if (node->count > 0) {
        for (uint32_t k = 0; k < node->count; k++)
            triangle_test(r, sc->vertices + (size_t)sc->indexes[node->first + k] * 9);
    } else {
        walk_any(sc, r, node->left);
        walk_any(sc, r, node->left + 1);
    }
}

int cto_bvhtest_render(const ctgpu_scene_desc *scene, uint32_t *fb, uint64_t *hit_pixels) {
    if (!scene || !fb) return CTGPU_ERR_ARG;
    const v3 cam = {0, 0, -18}, p0 = {-1, 1, -15}, p1 = {1, 1, -15}, p2 = {-1, -1, -15};
    const v3 du = v3_sub(p1, p0), dv = v3_sub(p2, p0);
    uint64_t hits = 0;
    for (int y = 0; y < 640; y++) {
        for (int x = 0; x < 640; x++) {
            /* p0 + (p1-p0)*(x/640.0f) + (p2-p0)*(y/640.0f): the fractions are F */
            const double fx = (double)((float)x / 640.0f), fy = (double)((float)y / 640.0f);
            const v3 pixel = v3_add(v3_add(p0, v3_scale(fx, du)), v3_scale(fy, dv));
            v3 d = v3_sub(pixel, cam);
            const double len = (double)v3_length(d);      /* Normalize: Magnitude is F (mymath.h:218-227) */
            if (len != 0) { d.x /= len; d.y /= len; d.z /= len; }
            ray r = {cam, d, kNoHit};
            walk_any(scene, &r, 0);
            if (r.t != kNoHit) {
                fb[(size_t)y * 640 + x] = pack_u8(135, 105, 134);
                hits++;
            }
        }
    }
    if (hit_pixels) *hit_pixels = hits;
    return CTGPU_OK;
}

/* ------------------------------------------------------------------ packet-coherence experiment
 * How many lanes of a 64-ray group want each node the group visits, if the group walks the tree
 * TOGETHER in the reference's DFS order (every ray keeps exactly its own visits; the group visits
 * the union)?  Inputs for a design decision, not part of the parity checks. */
typedef struct { ray r; float tclosest; uint32_t closest; int active; int any_hit; } packet_lane;

static void packet_walk(const ctgpu_scene_desc *sc, packet_lane *lanes, const unsigned char *in_mask, int n,
                        uint32_t node_id, uint64_t *stats) {
    const ctgpu_node *node = &sc->nodes[node_id];
    unsigned char mask[64];
    int any = 0, wanted = 0;
    for (int i = 0; i < n; i++) {
        mask[i] = 0;
        if (!in_mask[i] || !lanes[i].active) continue;
        wanted++;
        if (slab_test(&lanes[i].r, node->bmin, node->bmax)) { mask[i] = 1; any = 1; }
    }
    if (wanted == 0) return;
    stats[0] += 1;                 /* group visits  */
    stats[1] += (uint64_t)wanted;  /* lane visits   */
    if (!any) return;
    if (node->count > 0) {
        for (uint32_t k = 0; k < node->count; k++) {
            const uint32_t tri = sc->indexes[node->first + k];
            int lanes_here = 0;
            for (int i = 0; i < n; i++) {
                if (!mask[i] || !lanes[i].active) continue;
                lanes_here++;
                const int hit = triangle_test(&lanes[i].r, sc->vertices + (size_t)tri * 9);
                if (hit && lanes[i].r.t != kNoHit && lanes[i].r.t < lanes[i].tclosest) {
                    lanes[i].closest = tri;
                    lanes[i].tclosest = lanes[i].r.t;
                }
                if (lanes[i].any_hit && lanes[i].r.t != kNoHit) lanes[i].active = 0;
            }
            if (lanes_here) { stats[2] += 1; stats[3] += (uint64_t)lanes_here; }
        }
        return;
    }
    packet_walk(sc, lanes, mask, n, node->left, stats);
    packet_walk(sc, lanes, mask, n, node->left + 1, stats);
}

/* stats[0..3]: primary group visits, lane visits, group triangle tests, lane triangle tests;
 * stats[4..7]: the same for shadow rays (any-hit), grouped like their primary samples. */
int cto_packet_stats(const ctgpu_scene_desc *sc, const ctgpu_frame_desc *f, int group_w, int group_h,
                     uint64_t stats[8]) {
    memset(stats, 0, 8 * sizeof(uint64_t));
    const geometry g = frame_geometry(f);
    const int rows = g.bands * g.rows_per_band;
    const int ss = (f->flags & CTGPU_F_SUPERSAMPLE) != 0;
    const int per_pixel = ss ? 16 : 1;
    frame_job job;
    memset(&job, 0, sizeof job);
    job.scene = sc; job.frame = f; job.g = g;
    const v3 eye = v3_load(f->cam_pos);
    unsigned char all[64];
    memset(all, 1, sizeof all);
    for (int y0 = -g.half + 1; y0 < -g.half + rows; y0 += group_h) {
        for (int x0 = -g.half; x0 < g.half; x0 += group_w) {
            packet_lane lanes[64];
            int n = 0;
            for (int yy = 0; yy < group_h; yy++) for (int xx = 0; xx < group_w; xx++) {
                const int x = x0 + xx, y = y0 + yy;
                if (x >= g.half || y >= -g.half + rows) continue;
                float sx = (float)x;
                for (int xs = 0; xs < (ss ? 4 : 1); xs++) {
                    float sy = (float)y;
                    for (int ys = 0; ys < (ss ? 4 : 1); ys++) {
                        float px = (float)x, py = (float)y;
                        float rx = 0, ry = 0;
                        if (ss) {
                            const float jitter = 0.25f / 8.0f;
                            rx = (float)jitter_int(&job, x, y, xs * 4 + ys, 0) / 2147483648.0f;
                            ry = (float)jitter_int(&job, x, y, xs * 4 + ys, 1) / 2147483648.0f;
                            rx = (rx * (2.0f * jitter)) - jitter; ry = (ry * (2.0f * jitter)) - jitter;
                            sx += rx; sy += ry; px = sx; py = sy;
                        }
                        double dir[3];
                        cto_primary_direction(f->height, px, py, f->cam_rot, dir);
                        if (n < 64) {
                            lanes[n].r.origin = eye; lanes[n].r.direction = v3_load(dir); lanes[n].r.t = kNoHit;
                            lanes[n].tclosest = kFarAway; lanes[n].closest = 0; lanes[n].active = 1; lanes[n].any_hit = 0;
                            n++;
                        }
                        if (ss) { sy -= ry; sy += 0.25f; sx -= rx; }
                    }
                    if (ss) sx += 0.25f;
                }
            }
            (void)per_pixel;
            packet_walk(sc, lanes, all, n, 0, stats);
            /* shadow rays of the hits, one group per non-ambient light */
            for (uint32_t li = 0; li < sc->n_lights; li++) {
                const ctgpu_light *light = &sc->lights[li];
                if (light->type == CTGPU_LIGHT_AMBIENT) continue;
                packet_lane sh[64];
                int m = 0;
                for (int i = 0; i < n; i++) {
                    if (lanes[i].r.t == kNoHit) continue;
                    const v3 p = v3_add(lanes[i].r.origin, v3_scale((double)lanes[i].tclosest, lanes[i].r.direction));
                    sh[m].r.origin = p;
                    sh[m].r.direction = light->type == CTGPU_LIGHT_POINT ? v3_sub(v3_load(light->position), p) : v3_load(light->direction);
                    sh[m].r.t = kNoHit; sh[m].tclosest = kFarAway; sh[m].closest = 0; sh[m].active = 1; sh[m].any_hit = 1;
                    m++;
                }
                if (m) packet_walk(sc, sh, all, m, 0, stats + 4);
            }
        }
    }
    return CTGPU_OK;
}
