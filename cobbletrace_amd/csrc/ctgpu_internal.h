// ctgpu_internal.h -- what ctgpu_api.hip and ctgpu_comm.hip share (not part of the C ABI).
#ifndef CTGPU_INTERNAL_H
#define CTGPU_INTERNAL_H

#include "ct_params.h"

#include <string>
#include <vector>

#define CTGPU_HIDDEN __attribute__((visibility("hidden")))

struct ctgpu_comm_state;                                        // ctgpu_comm.hip

// records the message ctgpu_last_error() returns, gives back `code`
CTGPU_HIDDEN int ctgpu_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
CTGPU_HIDDEN ctgpu_comm_state *&ctgpu_comm_slot(ctgpu_ctx *ctx);
CTGPU_HIDDEN int ctgpu_device_of(const ctgpu_ctx *ctx);
// the context's own tile buffer (n_tiles uint32) and, where n_all > 0, its gather buffer
CTGPU_HIDDEN int ctgpu_shard_buffers(ctgpu_ctx *ctx, uint64_t n_tiles, uint64_t n_all, void **tiles, void **all);

// ctgpu_render_sharded's phase marks: which = 0 start, 1 traced, 2 gathered, 3 assembled (no-op unless timing is on)
CTGPU_HIDDEN int ctgpu_phase_mark(ctgpu_ctx *ctx, int which, hipStream_t stream);

// one process, several devices (ctgpu_render_multi)
CTGPU_HIDDEN int ctgpu_multi_comms(const int *devices, int n, std::vector<void *> &comms);
CTGPU_HIDDEN void ctgpu_multi_comms_release(std::vector<void *> &comms);
CTGPU_HIDDEN int ctgpu_multi_gather(const std::vector<void *> &comms, const int *devices, void *const *d_tiles, uint64_t n,
                                    void *d_all, hipStream_t const *streams);

#endif
