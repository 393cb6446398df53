// mcraw_plan.h -- structures shared by the host side of the C ABI (mcraw_abi.hip)
// and the gfx950 kernels (mcraw_type7.hip, mcraw_type6.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcraw {

// Internal status bit on top of include/mcraw_hip.h: the frame's header describes a coded frame
// with more blocks than the host planned for (it plans from the caller's width x height, or from
// the header where it can read it); the host plans that frame again from the real header.
constexpr int32_t E_GEOMETRY = 0x1000;

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

// Optional stage fused behind the decode (what a DNG writer does next with the mosaic,
// example.cpp:80-92): black-level subtraction and 12-bit strip packing.  Batch-wide, passed by value.
struct Post {
    uint32_t mode;             // POST_* bits; 0 = plain uint16 mosaic (the reference's output)
    uint32_t black01, black23; // black levels of CFA positions (row & 1, col & 1): (0,0) | (0,1) << 16 and (1,0) | (1,1) << 16
};
constexpr uint32_t POST_BLACK = 1;  // sample = max(sample - black[row & 1][col & 1], 0)
constexpr uint32_t POST_PACK12 = 2; // rows of min(sample, 4095) packed MSB-first, 3 bytes per 2 samples (TIFF/DNG BitsPerSample 12)
constexpr uint32_t POST_PACK10 = 4; // ... min(sample, 1023), 5 bytes per 4 samples (BitsPerSample 10)
constexpr uint32_t POST_PACK14 = 8; // ... min(sample, 16383), 7 bytes per 4 samples (BitsPerSample 14)
constexpr uint32_t POST_PACKED = POST_PACK12 | POST_PACK10 | POST_PACK14;

// Bits per sample of an output row.
__host__ __device__ inline uint32_t post_bits(uint32_t mode)
{
    return (mode & POST_PACK12) ? 12u : (mode & POST_PACK10) ? 10u : (mode & POST_PACK14) ? 14u : 16u;
}

// Bytes of one output row of `width` samples.
__host__ __device__ inline uint32_t post_row_bytes(uint32_t width, uint32_t mode)
{
    return (width * post_bits(mode) + 7u) >> 3;
}

// Per-frame plan of the current ("type 7") encoding, written by the host into PINNED HOST memory
// that k7_side reads directly (one 48-byte read per stream over the link: no upload copy in front
// of the kernels).  Geometry is NOT part of it: the kernels take it from the frame header
// (lib/RawData.cpp:500-524); the plan only says how much workspace and grid the frame was given.
struct Plan7 {
    const uint8_t *in;   // frame buffer (lib/RawData.cpp:528 `input`)
    uint16_t *out;       // width x rows mosaic
    uint32_t len;
    int32_t width;       // output columns kept (crop of encW, :598-608)
    int32_t height;      // output rows the caller has room for; rows kept = min(height, encH)
    uint32_t ngroups;    // decode groups (side-stream records) the workspace and the k7_tiles grid provide for
    uint32_t fast_store; // 1: out 16-B aligned and width % 8 == 0
    uint32_t pad;
};

// What k7_tiles needs to know about a frame, in HBM: written by k7_side (the workgroup of the bits
// stream) from the plan and the frame header.
struct Frame7 {
    const uint8_t *in;
    uint16_t *out;
    uint32_t len;
    int32_t width;       // output columns kept
    uint32_t rows;       // output rows kept = min(height, encH) (RawData.cpp:571, :598-608)
    uint32_t tilesX;     // encW / 64
    uint32_t nblk;       // N = 4 * tilesX * encH/4 payload blocks; 0: nothing to decode (the header was rejected)
    uint32_t fast_store;
    uint32_t encH;
    // A side stream can be resolved by several workgroups ("parts", Work7::nsplit[]), each of which owns a range of the stream's
    // 32 KiB pieces.  A part of the bits stream writes the payload offsets of ITS items relative to its own first item; what
    // k7_tiles adds to an offset: part_len[q] for every q with part_item[q] <= item.
    uint32_t part_item[3]; // first decode item of parts 1..3 of the bits stream (0xFFFFFFFF: no such part, or it had nothing to do)
    uint32_t part_len[3];  // payload bytes of parts 0..2 (part 0: up to the end of its last item, the 16-byte header included)
    uint32_t pad;
};
constexpr int MAX_SPLIT7 = 4; // parts per side stream, at most

// Batch-wide view of the type-7 work, passed to the kernels BY VALUE (kernarg):
// every workspace array has the same per-frame stride (sized for the largest
// frame of the batch), so a workgroup finds its slice from (frame, group)
// without any dependent pointer load.
struct Work7 {
    const Plan7 *plans;  // [n7] in pinned host memory (read by k7_side only)
    Frame7 *frames;      // [n7] in HBM (written by k7_side, read by k7_tiles)
    int32_t *status;     // [nstatus] status words: nsplit[0] + nsplit[1] per type-7 frame (one per part of its bits / refs stream,
                         // each written once, by a plain store: nothing to initialise), then one per legacy frame, then
                         // [n7] the coded height of every type-7 frame (read back with the statuses)
    uint8_t *bits;       // [n7][Rmax*64]  decoded `bits` stream  (:557)
    uint16_t *refs;      // [n7][Rmax*64]  decoded `refs` stream  (:560)
    uint32_t *grp_off;   // [n7][Rmax*ITEM_SPLIT+1] payload byte offset of every decode item (:562 + prefix of LEN)
    uint32_t Rmax;       // largest ngroups in the batch
    uint32_t nstatus;    // status words in front of the coded heights ((nsplit[0] + nsplit[1]) * n7 + legacy frames + 1)
    uint32_t nsplit[2];  // workgroups ("parts") per bits / refs stream, 1 .. MAX_SPLIT7: long streams of small batches are cut up
    uint64_t *sync;      // [n7][2][MAX_SPLIT7][2] what a part tells the next one (epoch-tagged words, never cleared): records up to
                         // the end of its pieces; where the chain enters the next part's first piece
    uint32_t epoch;      // ... of this launch
    uint32_t side_lastc; // k7_side, streams in parts: the last part of a stream counts its pieces too while it waits for the part in front
                         // (small batches: the chain of a stream gets a third shorter; a full chip only gets more to do)
    uint16_t *rpos;      // [n7][2][MAX_SPLIT7][Rmax] (only when a stream has parts) where the records of a part's pieces start, in chain
                         // order: candidate index inside its piece | 0x8000 on the first record of a piece -- written by the part's
                         // count, read back by its decode, which then need not follow the chain a second time
    int32_t n7;
    Post post;           // fused post-decode stage (mode 0: none)
    uint32_t xcd_chunk;  // k7_tiles: logical items per XCD run (0: one run per XCD = the whole grid in eight parts)
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
    uint32_t fast_store;
    int32_t *status;
};

constexpr int CHUNK6 = 1024; // bytes of legacy stream per transition-map chunk
constexpr int TICKET_STRIDE6 = 64; // uint32 words between the segment ticket counters of two legacy frames (256 bytes)
constexpr int ROWS_CH = 4;   // chunks per unpacking wave of k6_decode
constexpr int SEG_WAVES6 = 4;                                        // waves per workgroup of k6_decode
constexpr int SEG_CHUNKS6 = SEG_WAVES6 * ROWS_CH;                    // chunks per workgroup = per SEGMENT
constexpr int PHASES6 = 17;  // entry offsets 0,2,..,32 (record stride <= 34, all even)

// Look-back state of the legacy frames' segments (k6_decode), in a buffer that lives as long as its slot and is never
// cleared: every 64-bit word carries the epoch of the launch that wrote it in its high half (a poll of 64
// predecessors reads four cache lines of an array).  Per segment (index frame * smax + segment):
//   res  state << 30 | records << 5 | exit phase: state 1 = records of this segment alone, 2 = records of the frame up to
//        the end of this segment; exit phase = the phase at which the NEXT segment's first chunk is entered
//   ex   the same exit phase said earlier, before the segment knows its own entry: kind << 30 | phase, kind 1 = a phase
//        (one of the segment's chunk maps is unanimous), 2 = it depends on the segment's entry: the map is in hm
//   hm   [3] that map, entry phase -> exit phase, six 5-bit phases per word
struct Look6 {
    uint64_t *res, *ex, *hm;
};

} // namespace mcraw
