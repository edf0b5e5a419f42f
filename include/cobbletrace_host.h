/*
 * cobbletrace_host.h -- host-side interface of the MI355X build.
 *
 * These declarations mirror, field for field, the structures and entry points the
 * reference application passes across its ray-tracer boundary, so that the
 * reference's own main loop can link against this library instead of its
 * Boss/Worker thread pool.  Only the interface is mirrored; every definition
 * behind it is this repository's own.
 *
 *   boundary            reference declaration
 *   ------------------  -----------------------------------------------
 *   RayThread           raythread.h:5      (impl raythread.cpp:641-667)
 *   InitSceneData       scenefile.h:111    (impl scenefile.cpp:11-29)
 *   ParseSceneFile      scenefile.h:110    (impl scenefile.cpp:256-299)
 *   ImportObject        objectLoader.h:4   (impl objectLoader.cpp:282-288)
 *   InitializeBVHState  bvh.h:29           (impl bvh.cpp:16-28)
 *   BuildBVH            bvh.h:30           (impl bvh.cpp:108-121)
 *   IntersectBVH / IntersectBVHClosest   bvh.h:31-32   (impl bvh.cpp:181-222)
 *   scene_t & friends   scenefile.h:13-108, mymath.h:19-43
 *   environment_t       environment.h:7-20, eventQueue.h:9-42
 *   bvh_node_t ...      bvh.h:5-25
 *
 * Layouts are asserted in cobbletrace_amd/host/layout_check.cpp.
 */
#ifndef COBBLETRACE_HOST_H
#define COBBLETRACE_HOST_H

#include <stdint.h>
#include <stddef.h>

/* ---- linear algebra carriers (mymath.h:19-43): all components are double ---- */
struct v2_t { double x, y; };
struct v3_t { double x, y, z; };
struct v4_t { double x, y, z, w; };
struct m3x3_t { double data[3][3]; };
struct m4x4_t { double data[4][4]; };

/* ---- string view used by import records (ctstring.h:3-6) ---- */
struct string_t { int size; char *data; };

/* ---- scene model (scenefile.h:9-108) ---- */
#define MAX_OBJECTS 1000000
#define MAX_LIGHTS  100

enum lightType_t     { LT_POINT, LT_DIRECTIONAL, LT_AMBIENT };
enum object_t        { OT_SPHERE, OT_TRIANGLE, OT_IMPORT };
enum import_format_t { IT_BLENDER, IT_PLY };
enum stack_type_t    { ST_LIGHT, ST_OBJECT };

struct viewport_t { float width, height, d; };
struct material_t { uint32_t color; int specular; float reflection; };
struct sphere_t   { v3_t center; float radius; };
struct triangle_t { v3_t p1, p2, p3, centroid; };
struct import_t {
    string_t filename;
    v3_t position, rotation, scale;
    import_format_t format;
};
struct scene_object_t {
    object_t type;
    union {
        sphere_t   sphere;
        triangle_t triangle;
        import_t   import;
    };
    material_t material;
};
struct light_t {
    lightType_t type;
    float intensity;
    v3_t position, direction;
};
struct camera_t { v3_t position; m3x3_t rotation; };
struct scene_stack_t {
    int size, index;
    stack_type_t type;
    union {
        light_t        *lights;
        scene_object_t *objects;
    };
};
struct settings_t {
    int  numberOfThreads;
    bool subsampling, wireframe, supersampling;
};
struct scene_triangle_lookup_t { uint32_t triangleCount; uint32_t *indexes; };
struct scene_t {
    scene_stack_t lightStack;
    scene_stack_t objectStack;
    scene_triangle_lookup_t triangleLookup;
    camera_t   camera;
    viewport_t viewport;
    settings_t settings;
};
struct ray_t { v3_t origin, direction; float t; };

/* ---- platform hand-off (environment.h:7-20, eventQueue.h:9-42) ---- */
enum eventType_t     { ET_MOUSE_MOVE, ET_MOUSE_CLICK, ET_KEY_UP, ET_KEY_DOWN };
enum eventModifier_t { EM_NONE, EM_MOUSE_LBUTTON, EM_MOUSE_RBUTTON, EM_CONTROL,
                       EM_DOWN, EM_UP, EM_LEFT, EM_RIGHT };
struct event_t {
    eventType_t type;
    eventModifier_t modifier;
    v2_t cursor;
    uint32_t value;
};
struct eventManager_t {
    event_t *queue;
    uint32_t writeIndex, readIndex, capacity, active;
};
struct SDL_Renderer;                      /* opaque; never dereferenced here */
struct bitmapSettings_t {
    void  *memory;                        /* uint32 0x00BBGGRR, row-major, pitch 4*width */
    int    width, height;
    float *depthBuffer;                   /* unused by the ray tracer */
};
struct environment_t {
    SDL_Renderer     *renderer;
    bitmapSettings_t *bitmap;
    eventManager_t    events;
    int clientWidth, clientHeight;
};

/* ---- acceleration structure (bvh.h:5-25) ---- */
struct bvh_node_t {
    v3_t aabbMin, aabbMax;
    uint32_t leftNode, firstTriangleIndex, triangleCount;
};
struct bvh_triangles_t { uint32_t size; triangle_t *data; uint32_t *indexes; };
struct bvh_state_t {
    bvh_node_t *bvhNodes;
    uint32_t rootNodeIdx = 0;
    uint32_t nodesUsed   = 1;
    bvh_triangles_t triangles;
};

/* ---- entry points (C++ linkage, as in the reference) ---- */
void InitSceneData(scene_t *scene);
void ParseSceneFile(char *filename, scene_t *scene);
void ImportObject(scene_object_t importObject, scene_t *scene);
bvh_state_t InitializeBVHState(triangle_t *triangles, int size);
void BuildBVH(bvh_state_t *bvhState);
/* The operator-level boundary (bvh.h:31-32, impl bvh.cpp:181-222): one ray against a tree made by BuildBVH.  Here a
 * thin host wrapper that sends the ray through the HIP library (the tree is uploaded once per bvh_state_t and kept;
 * the walk is the frame kernel's, ctgpu_trace_rays_packet) -- there is no CPU walk in this build, so without the
 * library or a GPU these exit(2) like RayThread.  Supported as the reference calls them: nodeIdx = the tree's root,
 * *tclosest = 4294967296.f on entry (raythread.cpp:204-213); anything else is a fatal error, not a wrong answer.
 * A call costs a kernel launch: they are for known-answer tests and tools, frames go through RayThread. */
void IntersectBVH(ray_t *ray, uint32_t nodeIdx, bvh_state_t *bvhState);
void IntersectBVHClosest(ray_t *ray, uint32_t nodeIdx, bvh_state_t *bvhState, float *tclosest, uint32_t *closestIndex);
event_t *ReadEvent(eventManager_t *em);                 /* eventQueue.cpp:14-24 */
void AddEvent(eventManager_t *em, event_t event);       /* eventQueue.cpp:5-12  */

/* The drop-in: renders env->bitmap through the HIP library (include/ctgpu.h).
 * Returns true always, like the reference.  Fatal conditions (no GPU library,
 * no device, upload failure) print the ctgpu error and exit(2), matching the
 * reference's exit-on-fatal convention (raythread.cpp:590-592, 608). */
bool RayThread(environment_t *env, scene_t *scene);

/* Build-specific controls that have no counterpart in the reference (which
 * hard-codes them): all optional, process-wide. */
struct ct_raythread_options {
    int      rng_mode;     /* CTGPU_RNG_* from ctgpu.h; default CTGPU_RNG_HASH     */
    uint64_t rng_seed;     /* hash-mode seed, default 0                             */
    int      device;       /* first HIP device ordinal, default 0                   */
    int      n_devices;    /* GPUs to split every frame over (row tiles, RCCL gather);
                              0 or 1 = one GPU.  Ordinals wrap around the visible devices, so a
                              larger count than GPUs present still works (contexts share a GPU). */
    int      async;        /* 0: a call that renders returns with the pixels in the bitmap.
                              1: it only queues the frame; later calls find out that it has finished, as
                              the reference's boss polls its workers (one GPU only; the bitmap must stay
                              where it is while a frame is in flight) */
    unsigned frame_flags;  /* extra CTGPU_F_* bits for every frame: CTGPU_F_SPHERES, CTGPU_F_REFLECT_FIX (what the
                              reference's source has but its pictures do not; default 0 = the reference's frame) */
    int      keep_bitmap_pinned;  /* 1: env->bitmap->memory is ONE buffer that lives until RayThreadReset (the reference's
                              application: cobbletrace.cpp:57) -- its page-lock then stays between frames
                              (ctgpu_keep_framebuffer_pinned).  0 (default): RayThread is handed env anew on every call and
                              assumes nothing about the bitmap's life: it is page-locked for one frame at a time. */
};
void RayThreadSetOptions(const ct_raythread_options *opt);
bool RayThreadBusy(void);                 /* async: a queued frame has not been seen finished yet */
uint64_t RayThreadFramesFinished(void);   /* frames whose pixels have reached the bitmap */

/* The frame's consumer, draw2d.h:22-64: hands env->bitmap to the display.  Built with SDL2 (make SDL=1, or when
 * sdl2-config is found) it is the reference's sequence -- SDL_CreateRGBSurfaceFrom(memory, W, H, 32, 4*W,
 * R=0xff, G=0xff00, B=0xff0000, A=0) -> texture -> SDL_RenderCopy to {0,0,W,H} -> SDL_RenderPresent, surface and
 * texture destroyed each frame.  Without SDL2, or with env->renderer == NULL, it is a frame sink: it counts the
 * frames it was shown and, when CT_BLIT_PPM names a file, writes the bitmap there as a binary PPM. */
void Blit(environment_t *environment);
uint64_t BlitFrameCount(void);
/* Releases device buffers and the function-local state so that a new scene can
 * be rendered in the same process (the reference never frees, raythread.cpp:643). */
void RayThreadReset(void);

#endif
