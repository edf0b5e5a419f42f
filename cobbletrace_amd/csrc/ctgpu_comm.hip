// ctgpu_comm.hip -- the frame's one collective, native: an RCCL gather of the ranks' row-tile buffers to
// the root over xGMI, issued from this library on the frame's own stream.
//
// The reference shards a frame over worker threads by row bands that all write one framebuffer
// (raythread.cpp:574-618).  Across GPUs the shards are interleaved row tiles (ctgpu_render_partition),
// each rank traces into a compact tile buffer in its own HBM, and ONE collective per frame brings the
// buffers to the root, where ct_assemble writes the framebuffer.  xGMI is point to point: a gather is
// seven peers each sending 1/8 of the image straight to the root, no ring.
//
// librccl is opened at run time (preferring a copy the process has already loaded, e.g. PyTorch's, so
// that there is one RCCL and one HIP runtime per process); without it the comm entry points fail with
// CTGPU_ERR_COMM -- nothing here has a host-memory fallback (ctgpu_render_multi, one process owning several
// devices, then hands tiles over with peer-to-peer device copies instead: ctgpu_api.hip).
#include "ctgpu_internal.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommAbort) comm_abort = nullptr;
    decltype(&ncclGather) gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;              // (optional: what the communicator itself says)
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    std::string why;            // set when loading failed
};

Rccl &rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        // an explicit choice is final: a CTGPU_RCCL_LIBRARY that cannot be loaded means "no RCCL", not "some other RCCL"
        if (const char *override_path = std::getenv("CTGPU_RCCL_LIBRARY")) {
            api.lib = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
            if (!api.lib) { api.why = std::string("CTGPU_RCCL_LIBRARY: ") + dlerror(); return; }
        }
        // a copy the process already holds (PyTorch maps its own librccl.so without a soname) ...
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        // ... else ROCm's
        for (const char *n : {"librccl.so.1", "librccl.so"})
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) { api.why = std::string("librccl not found: ") + dlerror(); return; }
        bool ok = true;
#define RESOLVE(field, symbol)                                                              \
        api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, symbol));          \
        if (!api.field) { ok = false; api.why = std::string("librccl lacks ") + symbol; }
        RESOLVE(get_unique_id, "ncclGetUniqueId");
        RESOLVE(comm_init_rank, "ncclCommInitRank");
        RESOLVE(comm_init_all, "ncclCommInitAll");
        RESOLVE(comm_destroy, "ncclCommDestroy");
        RESOLVE(comm_abort, "ncclCommAbort");
        RESOLVE(gather, "ncclGather");
        RESOLVE(group_start, "ncclGroupStart");
        RESOLVE(group_end, "ncclGroupEnd");
        RESOLVE(error_string, "ncclGetErrorString");
#undef RESOLVE
        if (ok) {
            api.comm_count = reinterpret_cast<decltype(api.comm_count)>(dlsym(api.lib, "ncclCommCount"));
            api.comm_user_rank = reinterpret_cast<decltype(api.comm_user_rank)>(dlsym(api.lib, "ncclCommUserRank"));
        }
        if (!ok) api.lib = nullptr;
    });
    return api;
}

int rccl_ready() {
    Rccl &r = rccl();
    if (!r.lib) return ctgpu_fail(CTGPU_ERR_COMM, "%s", r.why.c_str());
    return CTGPU_OK;
}

#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return ctgpu_fail(CTGPU_ERR_COMM, "%s: %s", #expr, rccl().error_string(r_));    \
    } while (0)

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return ctgpu_fail(CTGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct ctgpu_comm_state {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int device = 0;
    int users = 1;              // contexts holding this communicator (ctgpu_comm_share)
    uint32_t joined = 1;        // ... ever (the next sharer's place among them)
};

// ---- the order of the gathers
// RCCL runs the collectives of one communicator in the order each rank issued them.  With several frames in flight
// (contexts sharing the communicator) that is the right frames meeting only if every rank issues its contexts' frames in
// the same order -- which the caller promises (bench.py: round robin) and nobody could see being broken: tiles of two
// different frames would assemble into a plausible picture.  So every sharded frame's gather travels with a second, tiny
// one: {the issuing context's place among the sharers, the number of frames it has issued}.  The root compares every rank's
// pair with its own expectation and raises bit 2 of the context's error word, which the next counter read reports.
__global__ void k_stamp(unsigned long long *mine, unsigned long long place, unsigned long long frames) { mine[0] = place; mine[1] = frames; }
__global__ void k_compare_stamps(const unsigned long long *all, int world, unsigned long long place, unsigned long long frames, uint32_t *flag) {
    const int r = (int)threadIdx.x;
    if (r < world && (all[2 * r] != place || all[2 * r + 1] != frames)) atomicOr(flag, 4u);
}
namespace {
int gather_with_stamp(ctgpu_ctx *ctx, ctgpu_comm_state *s, const void *d_tiles, uint64_t n, void *d_all, hipStream_t stream) {
    ctgpu_comm_stamp &stamp = ctgpu_comm_stamp_of(ctx);
    unsigned long long *mine = nullptr, *all = nullptr;
    uint32_t *flag = nullptr;
    int rc = ctgpu_comm_stamp_buffers(ctx, s->world, s->rank == 0, &mine, &all, &flag); if (rc) return rc;
    const unsigned long long expected = stamp.frames;
    stamp.frames += 1;
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, stream, mine, (unsigned long long)stamp.index, (unsigned long long)((long long)expected + stamp.skew));
    HIP_TRY(hipGetLastError());
    if (!s->comm) return ctgpu_fail(CTGPU_ERR_COMM, "this context's communicator was aborted after an earlier failure");
    RCCL_TRY(rccl().group_start());
    const ncclResult_t a = rccl().gather(d_tiles, d_all, (size_t)n, ncclUint32, 0, s->comm, stream);
    const ncclResult_t b = rccl().gather(mine, all, 2, ncclUint64, 0, s->comm, stream);
    const ncclResult_t e = rccl().group_end();
    if (a != ncclSuccess) return ctgpu_fail(CTGPU_ERR_COMM, "ncclGather (tiles): %s", rccl().error_string(a));
    if (b != ncclSuccess) return ctgpu_fail(CTGPU_ERR_COMM, "ncclGather (stamps): %s", rccl().error_string(b));
    if (e != ncclSuccess) return ctgpu_fail(CTGPU_ERR_COMM, "ncclGroupEnd: %s", rccl().error_string(e));
    if (s->rank == 0) {
        if (s->world > 1024) return CTGPU_OK;
        hipLaunchKernelGGL(k_compare_stamps, dim3(1), dim3(1024), 0, stream, (const unsigned long long *)all, s->world,
                           (unsigned long long)stamp.index, expected, flag);
        HIP_TRY(hipGetLastError());
    }
    return CTGPU_OK;
}
}  // namespace

// test hook: from now on this context stamps its sharded frames `delta` frames off (what a rank that issued its contexts'
// frames in another order than the root would send); 0 ends it
extern "C" int ctgpu_comm_debug_skew(ctgpu_ctx *ctx, int delta) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_debug_skew: null argument");
    ctgpu_comm_stamp_of(ctx).skew = delta;
    return CTGPU_OK;
}

static_assert(CTGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ctgpu.h promises room for an ncclUniqueId");

extern "C" int ctgpu_comm_unique_id(void *id) {
    if (!id) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_unique_id: null argument");
    int rc = rccl_ready(); if (rc) return rc;
    ncclUniqueId uid;
    RCCL_TRY(rccl().get_unique_id(&uid));
    std::memcpy(id, &uid, sizeof uid);
    return CTGPU_OK;
}

extern "C" int ctgpu_comm_init(ctgpu_ctx *ctx, const void *id, int rank, int world) {
    if (!ctx || !id) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_init: rank %d of %d", rank, world);
    int rc = rccl_ready(); if (rc) return rc;
    ctgpu_comm_state *&slot = ctgpu_comm_slot(ctx);
    if (slot) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_init: this context already has a communicator");
    HIP_TRY(hipSetDevice(ctgpu_device_of(ctx)));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    RCCL_TRY(rccl().comm_init_rank(&comm, world, uid, rank));
    slot = new ctgpu_comm_state();
    slot->comm = comm; slot->rank = rank; slot->world = world; slot->device = ctgpu_device_of(ctx);
    ctgpu_comm_stamp_of(ctx) = ctgpu_comm_stamp();
    return CTGPU_OK;
}

// Several contexts of one process on one device (frames in flight) gather through ONE communicator: RCCL orders
// the collectives of a communicator among themselves, whichever streams they were queued on, so every rank runs
// them in the same order -- with a communicator each, two gathers could wait for each other across ranks.
extern "C" int ctgpu_comm_share(ctgpu_ctx *ctx, ctgpu_ctx *owner) {
    if (!ctx || !owner) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_share: null argument");
    ctgpu_comm_state *&slot = ctgpu_comm_slot(ctx);
    ctgpu_comm_state *theirs = ctgpu_comm_slot(owner);
    if (slot) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_share: this context already has a communicator");
    if (!theirs) return ctgpu_fail(CTGPU_ERR_COMM, "ctgpu_comm_share: the other context has no communicator");
    if (theirs->device != ctgpu_device_of(ctx)) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_share: the contexts sit on different devices");
    theirs->users += 1;
    ctgpu_comm_stamp_of(ctx) = ctgpu_comm_stamp();
    ctgpu_comm_stamp_of(ctx).index = theirs->joined++;
    slot = theirs;
    return CTGPU_OK;
}

extern "C" int ctgpu_comm_destroy(ctgpu_ctx *ctx) {
    if (!ctx) return CTGPU_OK;
    ctgpu_comm_state *&slot = ctgpu_comm_slot(ctx);
    if (!slot) return CTGPU_OK;
    if (--slot->users == 0) {
        if (slot->comm && rccl().lib) (void)rccl().comm_destroy(slot->comm);      // (an aborted communicator is already gone)
        delete slot;
    }
    slot = nullptr;
    return CTGPU_OK;
}

extern "C" int ctgpu_comm_rank(ctgpu_ctx *ctx, int *rank, int *world) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_comm_rank: null argument");
    ctgpu_comm_state *s = ctgpu_comm_slot(ctx);
    if (!s) return ctgpu_fail(CTGPU_ERR_COMM, "this context has no communicator (ctgpu_comm_init)");
    if (rank) *rank = s->rank;
    if (world) *world = s->world;
    // (what the communicator itself says, where this RCCL can be asked: a bench line then shows the ranks RCCL sees)
    if (s->comm && rccl().comm_count && rccl().comm_user_rank) {
        int n = 0, r = 0;
        if (rccl().comm_count(s->comm, &n) == ncclSuccess && world) *world = n;
        if (rccl().comm_user_rank(s->comm, &r) == ncclSuccess && rank) *rank = r;
    }
    return CTGPU_OK;
}

// every rank's n uint32 -> back to back in d_all on rank `root` (ncclGather, asynchronous on `stream`)
extern "C" int ctgpu_gather_tiles(ctgpu_ctx *ctx, const void *d_tiles, uint64_t n, void *d_all, int root, void *stream) {
    if (!ctx || !d_tiles) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_gather_tiles: null argument");
    ctgpu_comm_state *s = ctgpu_comm_slot(ctx);
    if (!s) return ctgpu_fail(CTGPU_ERR_COMM, "this context has no communicator (ctgpu_comm_init)");
    if (root < 0 || root >= s->world) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_gather_tiles: root %d of %d", root, s->world);
    if (s->rank == root && !d_all) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_gather_tiles: the root needs a receive buffer");
    if (!s->comm) return ctgpu_fail(CTGPU_ERR_COMM, "this context's communicator was aborted after an earlier failure");
    HIP_TRY(hipSetDevice(ctgpu_device_of(ctx)));
    RCCL_TRY(rccl().gather(d_tiles, d_all, (size_t)n, ncclUint32, root, s->comm, (hipStream_t)stream));
    return CTGPU_OK;
}

// One frame, this rank's part of it: trace my row tiles, gather to rank 0, assemble there -- all queued on
// `stream`, nothing synchronises.  d_fb: the root's device framebuffer (width * height uint32); ignored elsewhere.
//
// A collective that one rank never queues blocks every other rank for good, so whatever fails locally here must not
// keep this rank out of the gather.  Argument errors are the same on every rank (the frame and the partition rule
// are shared) and return before anything is queued anywhere.  After that, a local failure (no memory for the
// buffers, a kernel launch error) still takes part: the rank sends its tile buffer zeroed -- "nothing rendered",
// the value the reference's bitmap starts with -- and reports its error; only when there is no buffer to send
// does it abort the communicator, which makes the peers' gather return with an error instead of waiting.
extern "C" int ctgpu_render_sharded(ctgpu_ctx *ctx, const ctgpu_frame_desc *frame, int tile_rows, void *d_fb, void *stream) {
    if (!ctx) return ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_render_sharded: null argument");
    ctgpu_comm_state *s = ctgpu_comm_slot(ctx);
    if (!s) return ctgpu_fail(CTGPU_ERR_COMM, "this context has no communicator (ctgpu_comm_init)");
    if (!s->comm) return ctgpu_fail(CTGPU_ERR_COMM, "this context's communicator was aborted after an earlier failure");
    const ctgpu_partition part = {tile_rows, s->rank, s->world};
    uint64_t n = 0;
    int rc = ctgpu_partition_pixels(frame, &part, &n); if (rc) return rc;                 // same verdict on every rank
    if (s->rank == 0 && !d_fb) rc = ctgpu_fail(CTGPU_ERR_ARG, "ctgpu_render_sharded: rank 0 needs a framebuffer");
    void *tiles = nullptr, *all = nullptr;
    const int have = ctgpu_shard_buffers(ctx, n, s->rank == 0 ? n * (uint64_t)s->world : 0, &tiles, &all);
    if (have != CTGPU_OK) {                                                               // nothing to send at all
        if (s->comm) { (void)rccl().comm_abort(s->comm); s->comm = nullptr; }
        return have;
    }
    (void)ctgpu_phase_mark(ctx, 0, (hipStream_t)stream);
    if (rc == CTGPU_OK) rc = ctgpu_render_partition(ctx, frame, &part, tiles, stream);
    if (rc != CTGPU_OK) (void)hipMemsetAsync(tiles, 0, n * sizeof(uint32_t), (hipStream_t)stream);
    const std::string first_error = rc != CTGPU_OK ? ctgpu_last_error() : "";
    (void)ctgpu_phase_mark(ctx, 1, (hipStream_t)stream);
    HIP_TRY(hipSetDevice(ctgpu_device_of(ctx)));
    const int gathered = gather_with_stamp(ctx, s, tiles, n, all, (hipStream_t)stream);
    (void)ctgpu_phase_mark(ctx, 2, (hipStream_t)stream);
    if (rc != CTGPU_OK) return ctgpu_fail(rc, "%s", first_error.c_str());
    if (gathered != CTGPU_OK) return gathered;
    if (s->rank == 0) { rc = ctgpu_assemble(ctx, frame, &part, all, d_fb, stream); if (rc) return rc; }
    return ctgpu_phase_mark(ctx, 3, (hipStream_t)stream);
}

// ------------------------------------------------------------------ one process, several devices
// ctgpu_render_multi's gather: one communicator per device (ncclCommInitAll), the gathers of all devices in
// one group.  comms are cached by the caller (the root context) for as long as the device list is the same.
int ctgpu_multi_comms(const int *devices, int n, std::vector<void *> &comms) {
    int rc = rccl_ready(); if (rc) return rc;
    std::vector<ncclComm_t> c((size_t)n, nullptr);
    RCCL_TRY(rccl().comm_init_all(c.data(), n, devices));
    comms.assign(c.begin(), c.end());
    return CTGPU_OK;
}

// abort: the communicators may have a collective queued that not every device joined (a failure in mid-frame):
// ncclCommDestroy would wait for it, ncclCommAbort ends it
void ctgpu_multi_comms_release(std::vector<void *> &comms, bool abort) {
    if (rccl().lib)
        for (void *c : comms)
            if (c) (void)(abort ? rccl().comm_abort((ncclComm_t)c) : rccl().comm_destroy((ncclComm_t)c));
    comms.clear();
}

int ctgpu_multi_gather(const std::vector<void *> &comms, const int *devices, void *const *d_tiles, uint64_t n, void *d_all,
                       hipStream_t const *streams) {
    RCCL_TRY(rccl().group_start());
    for (size_t i = 0; i < comms.size(); i++) {
        HIP_TRY(hipSetDevice(devices[i]));
        RCCL_TRY(rccl().gather(d_tiles[i], i == 0 ? d_all : nullptr, (size_t)n, ncclUint32, 0, (ncclComm_t)comms[i], streams[i]));
    }
    RCCL_TRY(rccl().group_end());
    return CTGPU_OK;
}
