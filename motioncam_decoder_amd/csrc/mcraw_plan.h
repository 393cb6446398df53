// mcraw_plan.h -- structures shared by the host side of the C ABI (mcraw_abi.hip)
// and the gfx950 kernels (mcraw_type7.hip, mcraw_type6.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcraw {

// Internal status bit on top of include/mcraw_hip.h: the coded geometry in the
// frame header differs from the one the host planned with (ceil64(width) x
// ceil4(height)); the host re-plans that frame from the real header.
constexpr int32_t E_GEOMETRY = 0x1000;
// ... or: the bits side stream does not end before the refs side stream begins; the host
// re-plans the frame without the extent hint (Plan7::full_extent).
constexpr int32_t E_LAYOUT = 0x2000;

constexpr int GROUP_BLOCKS = 64;   // blocks per side-stream record = per decode group
constexpr int GROUP_TILES = 16;    // 64x4 tiles per group
// A decode ITEM is the unit one wave of k7_tiles handles: 1/ITEM_SPLIT of a group.  Halving the
// item halves the LDS a wave needs for its payload span (4 KiB instead of 8) and so doubles the
// waves a CU can keep in flight; the payload offset table has one entry per item.
constexpr int ITEM_SPLIT = 2;
constexpr int ITEM_BLOCKS = GROUP_BLOCKS / ITEM_SPLIT;
constexpr int ITEM_TILES = GROUP_TILES / ITEM_SPLIT;
constexpr int ITEM_SPAN = ITEM_BLOCKS * 128; // largest payload span of one item (all raw-16)
constexpr int SPAN_MAX = 64 * 128; // largest payload span of one group (all raw-16)
constexpr int CH7 = 1024;          // bytes of side stream per transition-map chunk
constexpr int PH7 = 65;            // entry offsets 0,2,..,128 (record stride 2 + LEN <= 130, all even)

// Optional stage fused behind the decode (what a DNG writer does next with the mosaic,
// example.cpp:80-92): black-level subtraction and 12-bit strip packing.  Batch-wide, passed by value.
struct Post {
    uint32_t mode;             // POST_* bits; 0 = plain uint16 mosaic (the reference's output)
    uint32_t black01, black23; // black levels of CFA positions (row & 1, col & 1): (0,0) | (0,1) << 16 and (1,0) | (1,1) << 16
};
constexpr uint32_t POST_BLACK = 1;  // sample = max(sample - black[row & 1][col & 1], 0)
constexpr uint32_t POST_PACK12 = 2; // rows of min(sample, 4095) packed MSB-first, 3 bytes per 2 samples (TIFF/DNG BitsPerSample 12)

// Bytes of one output row of `width` samples.
__host__ __device__ inline uint32_t post_row_bytes(uint32_t width, uint32_t mode)
{
    return (mode & POST_PACK12) ? (width * 12u + 7u) >> 3 : width * 2u;
}

// Per-frame plan of the current ("type 7") encoding; lives in HBM for the
// duration of one batch.
struct Plan7 {
    const uint8_t *in;   // frame buffer (lib/RawData.cpp:528 `input`)
    uint16_t *out;       // width x rows mosaic
    uint32_t len;
    int32_t width;       // output columns kept (crop of encW, :598-608)
    int32_t rows;        // output rows kept = min(height, encH)
    uint32_t encW, encH; // coded geometry the plan assumes (:500-511)
    uint32_t tilesX;     // encW / 64
    uint32_t nblk;       // N = 4 * tilesX * encH/4  (payload blocks = side-stream entries used)
    uint32_t ngroups;    // R = ceil(N / 64)
    uint32_t fast_store; // 1: out 16-B aligned and width % 8 == 0
    uint32_t full_extent; // 1: do not assume the bits stream ends where the refs stream starts
};

// Batch-wide view of the type-7 work, passed to the kernels BY VALUE (kernarg):
// every workspace array has the same per-frame stride (sized for the largest
// frame of the batch), so a workgroup finds its slice from (frame, group)
// without any dependent pointer load.
struct Work7 {
    const Plan7 *plans;  // [n7]
    int32_t *status;     // [n7] status word of every type-7 frame
    uint32_t *cmap;      // [n7][2][nch][PH7] transition map of every side-stream chunk (exit phase | count << 8)
    uint32_t *centry;    // [n7][2][nch]      resolved entry of every chunk (phase | first record << 8)
    uint4 *sinfo;        // [n7][2]           per stream: first record offset, chunks to map, extent hinted, -
    uint4 *list_maps;    // work list of k7_maps:    (stream, first of 3 chunks, its byte offset, chunks of the stream)
    uint4 *list_recs;    // work lists of k7_records: sparse items (stream, first chunk | chunks << 24, byte offset, end of
                         // the record range) from the front, dense items (stream, records, byte offset, entry) from the back
    uint32_t *counters;  // [0] k7_maps items, [1] sparse / [2] dense k7_records items (zeroed by the table upload)
    uint32_t list_cap;   // capacity of list_recs (dense items are filled in from the back)
    uint8_t *bits;       // [n7][Rmax*64]  decoded `bits` stream  (:557)
    uint16_t *refs;      // [n7][Rmax*64]  decoded `refs` stream  (:560)
    uint32_t *grp_off;   // [n7][Rmax*ITEM_SPLIT+1] payload byte offset of every decode item (:562 + prefix of LEN)
    uint32_t Rmax;       // largest ngroups in the batch
    uint32_t nch;        // side-stream chunks planned per stream (covers Rmax records of 130 bytes)
    int32_t n7;
    Post post;           // fused post-decode stage (mode 0: none)
    // k7_tiles is launched once per SIZE CLASS of the batch (plans are sorted by ngroups, descending):
    // frames [class_first[k], class_first[k+1]) get class_groups[k] decode groups each, so a batch that
    // mixes small and large frames does not pay the largest frame's grid for every frame
    static constexpr int MAX_CLASSES = 8;
    uint32_t nclasses;
    uint32_t class_first[MAX_CLASSES + 1];
    uint32_t class_groups[MAX_CLASSES];
};

// Per-frame plan of the legacy ("type 6") encoding.
struct Plan6 {
    const uint8_t *in;
    uint16_t *out;
    uint32_t len;
    int32_t width, height;
    uint32_t padded;       // ceil32(width)                        (lib/RawData_Legacy.cpp:34-36)
    uint32_t recs_per_row; // 2 * padded / 32
    uint32_t nrec;         // height * recs_per_row
    uint32_t nchunks;      // ceil(len / CHUNK6)
    uint32_t nsuper;       // ceil(nchunks / SUPER6)
    uint32_t fast_store;
    uint32_t *cmap;        // [nchunks][17] per-chunk transition map  (exit phase | count)
    uint32_t *smap;        // [nsuper][17]  per-super-chunk map
    uint32_t *centry;      // [nchunks]     resolved entry of every chunk (phase | first record)
    uint32_t *sentry;      // [nsuper]      resolved entry of every super-chunk
    int32_t *status;
};

constexpr int CHUNK6 = 1024; // bytes of legacy stream per transition-map chunk
constexpr int SUPER6 = 64;   // chunks per super-chunk
constexpr int ROWS_CH = 4;   // chunks per wave of k6_rows
constexpr int PHASES6 = 17;  // entry offsets 0,2,..,32 (record stride <= 34, all even)

} // namespace mcraw
