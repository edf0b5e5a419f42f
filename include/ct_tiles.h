/*
 * ct_tiles.h -- how the rendered rows of a frame are dealt to N partitions (GPUs / ranks) and
 * merged again.  ONE definition, compiled into both sides of the product: the ct_assemble kernel
 * of libctgpu.so (tile buffers gathered in HBM) and cth_assemble_tiles of libcthost.so (tile
 * buffers that arrive in host memory, e.g. gathered between nodes over MPI / gloo).
 *
 * The reference shards by contiguous row bands, yStep = H / T rows per worker
 * (raythread.cpp:574-588), all workers writing one framebuffer.  Here the rendered rows
 * yi = 0 .. T*(H/T)-1 (yi = y + H/2) are cut into tiles of tile_rows rows, tile t goes to
 * partition t % part_count (interleaved: the geometry sits mid-image), and every partition
 * stores its rows compactly: local row (t / part_count) * tile_rows + yi % tile_rows.
 * Plain C, no dependencies.
 */
#ifndef CT_TILES_H
#define CT_TILES_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CT_TILES_FN __host__ __device__ static inline
#else
#define CT_TILES_FN static inline
#endif

typedef struct ct_tile_layout {
    int32_t  width, height, half;      /* bitmap; half = height / 2                              */
    uint32_t square;                   /* 2 * half: columns of the rendered square               */
    uint32_t rendered_rows;            /* T * (height / T)                                       */
    uint32_t rows_per_band;            /* height / T                                             */
    uint32_t subsample;                /* settings.subsampling                                   */
    uint32_t tile_rows, part_count;
    uint32_t local_rows;               /* rows of one partition's buffer (same for all)          */
} ct_tile_layout;

/* rows of one partition's compact buffer */
CT_TILES_FN uint32_t ct_tiles_local_rows(uint32_t rendered_rows, uint32_t tile_rows, uint32_t part_count) {
    const uint32_t tiles = (rendered_rows + tile_rows - 1u) / tile_rows;
    const uint32_t per_part = (tiles + part_count - 1u) / part_count;
    return per_part * tile_rows;
}

/* which partition holds rendered row yi, and at which of its local rows */
CT_TILES_FN void ct_tiles_locate(const ct_tile_layout *L, uint32_t yi, uint32_t *owner, uint32_t *local_row) {
    const uint32_t tile = yi / L->tile_rows;
    *owner = tile % L->part_count;
    *local_row = (tile / L->part_count) * L->tile_rows + yi % L->tile_rows;
}

/* index of pixel (xi, yi) in the back-to-back buffers of all partitions */
CT_TILES_FN uint64_t ct_tiles_index(const ct_tile_layout *L, uint32_t xi, uint32_t yi) {
    uint32_t owner, local_row;
    ct_tiles_locate(L, yi, &owner, &local_row);
    return ((uint64_t)owner * L->local_rows + local_row) * L->square + xi;
}

/* subsample in-between pixel (raythread.cpp:517-523): per channel min((a + b) / 2, 255) in float, truncated */
CT_TILES_FN uint32_t ct_tiles_average_channel(uint32_t a, uint32_t b) {
    float c = ((float)a + (float)b) / 2.0f;
    if (!(c < 255.0f)) c = 255.0f;
    return (uint32_t)(int)c & 0xffu;
}
CT_TILES_FN uint32_t ct_tiles_average(uint32_t a, uint32_t b) {
    return (ct_tiles_average_channel((a >> 16) & 0xffu, (b >> 16) & 0xffu) << 16) |
           (ct_tiles_average_channel((a >> 8) & 0xffu, (b >> 8) & 0xffu) << 8) |
           ct_tiles_average_channel(a & 0xffu, b & 0xffu);
}

/* The colour of rendered pixel (xi, yi) from the partitions' buffers.  Without subsampling: the
 * traced pixel.  With it (raythread.cpp:512-535): a band traces its even row offsets and its
 * last row; every traced row except the band's first also writes the row below it with the
 * average of itself and the previously traced row, in row order, so the later write wins:
 *   odd offset, not the last row -> never traced: the average of its two neighbours
 *   offset q-2 when q is even     -> traced, then overwritten by row q-1's average
 * The in-between write of a band's FIRST row (into the neighbouring band) is the reference's
 * one racy store; it is dropped. */
CT_TILES_FN uint32_t ct_tiles_pixel(const ct_tile_layout *L, const uint32_t *all_tiles, uint32_t xi, uint32_t yi) {
    if (!L->subsample) return all_tiles[ct_tiles_index(L, xi, yi)];
    const uint32_t q = L->rows_per_band, off = yi % q;
    if ((off & 1u) && off != q - 1u)
        return ct_tiles_average(all_tiles[ct_tiles_index(L, xi, yi - 1u)], all_tiles[ct_tiles_index(L, xi, yi + 1u)]);
    if (!(q & 1u) && q >= 2u && off == q - 2u)
        return ct_tiles_average(all_tiles[ct_tiles_index(L, xi, yi)], all_tiles[ct_tiles_index(L, xi, yi + 1u)]);
    return all_tiles[ct_tiles_index(L, xi, yi)];
}

/* where rendered pixel (xi, yi) lands in the bitmap (raythread.cpp:176-184, draw2d.h:8-20); 0 = dropped */
CT_TILES_FN int ct_tiles_bitmap_slot(const ct_tile_layout *L, uint32_t xi, uint32_t yi, uint64_t *slot) {
    const int row = L->height / 2 - ((int)yi - L->half), col = (int)xi - L->half + L->width / 2;
    if (row < 0 || row >= L->height || col < 0 || col >= L->width) return 0;
    *slot = (uint64_t)row * (uint64_t)L->width + (uint64_t)col;
    return 1;
}

#endif
