/*
 * mcraw_hip.h -- C ABI of the MI355X (gfx950) MCRAW frame-decode path.
 *
 * This is the drop-in boundary: the entry points a maintainer of
 * mirsadm/motioncam-decoder binds instead of the CPU codec.  Plain pointers
 * and sizes only; no C++ or torch types.  INTEGRATION.md shows the
 * reference-side change (lib/Decoder.cpp:224-231).
 *
 * There is NO CPU fallback behind these symbols: without a HIP device every
 * entry point fails (returns 0 / a negative status).
 */
#ifndef MCRAW_HIP_H
#define MCRAW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCRAW_TYPE_LEGACY 6 /* lib/Decoder.cpp:20 MOTIONCAM_COMPRESSION_TYPE_LEGACY */
#define MCRAW_TYPE_BLOCK  7 /* lib/Decoder.cpp:21 MOTIONCAM_COMPRESSION_TYPE        */

/* Per-frame status bits reported by mcraw_decode_batch (0 = decoded). */
#define MCRAW_OK            0
#define MCRAW_E_ARGS        0x0001 /* bad width/height/type/pointer                      */
#define MCRAW_E_HEADER      0x0002 /* offsets > len, encodedWidth%64, encodedWidth<width */
                                   /*   (reference returns 0: lib/RawData.cpp:547-554)   */
#define MCRAW_E_TRUNCATED   0x0004 /* a block or record crosses `len` (the reference     */
                                   /*   skips it and leaves stale samples, :419-420)     */
#define MCRAW_E_SIDESTREAM  0x0008 /* side-stream entry count < blocks, or bits > 16     */
#define MCRAW_E_CAPACITY    0x0010 /* out_capacity < width * rows                        */
#define MCRAW_E_DEVICE      0x0100 /* HIP runtime error                                  */

/* Where the `in` / `out` pointers of a batch live. */
#define MCRAW_MEM_DEVICE 0 /* HBM of the context's device: no copies, decode only      */
#define MCRAW_MEM_HOST   1 /* host memory: staged H2D / D2H on the context's streams,   */
                           /*   copies of one sub-batch overlap decode of the previous  */

typedef struct mcraw_ctx mcraw_ctx;

/* One frame of a batch.  Mirrors the arguments of motioncam::raw::Decode /
 * DecodeLegacy (lib/include/motioncam/RawData.hpp:25-37) plus the explicit
 * output capacity the reference lacks. */
typedef struct mcraw_frame {
    const uint8_t *in;   /* compressed frame buffer (BUFFER item payload), any      */
                         /*   byte alignment -- e.g. a payload inside a .mcraw file */
                         /*   image resident in HBM; with MCRAW_MEM_DEVICE up to 3  */
                         /*   bytes behind in + len may be read (never used)        */
    size_t len;          /* its length in bytes                                    */
    int32_t width;       /* frame JSON "width"   (lib/Decoder.cpp:216)             */
    int32_t height;      /* frame JSON "height"  (lib/Decoder.cpp:217)             */
    int32_t type;        /* frame JSON "compressionType": 6 or 7 (:218)            */
    int32_t reserved;
    uint16_t *out;       /* row-major uint16 LE mosaic, width * height             */
    size_t out_capacity; /* in uint16 elements                                     */
} mcraw_frame;

/* ---- context ------------------------------------------------------------ */

/* Create a decode context on HIP device `device` (-1: env MCRAW_DEVICE, else
 * the current device).  Returns 0 or a negative hipError. */
int mcraw_ctx_create(int device, mcraw_ctx **ctx);
void mcraw_ctx_destroy(mcraw_ctx *ctx);
const char *mcraw_last_error(void);

/* ---- drop-in single-frame entry points ---------------------------------- */

/* Replace motioncam::raw::Decode (lib/RawData.cpp:528-612) and
 * motioncam::raw::DecodeLegacy (lib/RawData_Legacy.cpp:445-495): same five
 * arguments, same return convention (uint16 elements written, 0 = failure).
 * Host pointers; `output` must hold width*height elements.  They run on a
 * process-wide default context (device: MCRAW_DEVICE or 0). */
size_t mcraw_decode7(uint16_t *output, int width, int height, const uint8_t *input, size_t len);
size_t mcraw_decode6(uint16_t *output, int width, int height, const uint8_t *input, size_t len);

/* ---- batched entry point (replaces the per-frame loop, lib/Decoder.cpp:184-235,
 *      example.cpp:187-195) ------------------------------------------------ */

/* Decode `nframes` independent frames.  `mem` says where in/out live.
 * `stream` is a hipStream_t (NULL = the context's own stream); with
 * MCRAW_MEM_DEVICE all work is enqueued on it and the call returns without
 * synchronising unless `written`/`status` are requested:
 *   written[i] : uint16 elements produced for frame i (0 on failure), or NULL
 *   status[i]  : MCRAW_* bits for frame i, or NULL
 * Passing either forces a stream synchronisation before returning.
 * Returns 0, or a negative value when the batch could not be submitted. */
int mcraw_decode_batch(mcraw_ctx *ctx, const mcraw_frame *frames, int nframes, int mem,
                       void *stream, size_t *written, int32_t *status);

/* Asynchronous host-memory batches (MCRAW_MEM_HOST semantics; buffers should be pinned): the call
 * returns when the batch is queued, so the next one can be submitted while this one is still moving
 * over PCIe -- the upload / kernel / download lanes then run back to back ACROSS batches, which a
 * sequence of synchronous mcraw_decode_batch calls cannot do.  `frames` is copied; the in / out buffers
 * must stay valid until mcraw_ticket_wait has returned for the ticket.  mcraw_ticket_wait blocks for that
 * batch only, fills `written` / `status` (either may be NULL) and releases the ticket.  Wait for every
 * ticket before mcraw_ctx_destroy.  Returns 0 or a negative value.
 * Scheduling: the copy lanes run best with two batches of a few hundred MB under way (about 3 000 UHD frames/s
 * host to host).  The library keeps to that by itself -- a third ticket's submission waits for the oldest one's
 * downloads (not for its mcraw_ticket_wait), and a batch of more than 384 MB is dealt out in such pieces inside the
 * call, which then returns when the last piece is queued -- so neither the size of a batch nor the number of tickets
 * a caller keeps in flight (two is enough) has to be tuned.  How the per-frame statuses travel (written home behind the
 * kernels, or fetched at the wait) is measured by every context on its first batches, because the better way depends on
 * what the process did with the GPU before the context existed; MCRAW_SHORT_WAY=0|1 in the environment decides it
 * beforehand (1: written home), MCRAW_TRACE=1 prints what was measured. */
typedef struct mcraw_ticket mcraw_ticket;
int mcraw_decode_batch_async(mcraw_ctx *ctx, const mcraw_frame *frames, int nframes, mcraw_ticket **ticket);
int mcraw_ticket_wait(mcraw_ticket *ticket, size_t *written, int32_t *status);

/* Wait for everything submitted on the context; fetch the statuses of the
 * last batch (status may be NULL).  Returns 0 or negative. */
int mcraw_ctx_synchronize(mcraw_ctx *ctx, int32_t *status, int nframes);
/* Device-memory batches submitted WITHOUT a status request, several in a row (the reference's loop, example.cpp:187-195, run as
 * batches that follow each other on the GPU): every device-memory batch of a context has a serial number (the last one
 * submitted: mcraw_ctx_last_serial); mcraw_ctx_batch_status waits for that batch and returns its statuses (0; 1 when the
 * serial is not one of the last 64 such batches); mcraw_ctx_errors returns the OR of the statuses of all frames of such
 * batches whose statuses became known since the last call with `reset` (mcraw_ctx_synchronize makes all of them known). */
uint64_t mcraw_ctx_last_serial(mcraw_ctx *ctx);
int mcraw_ctx_batch_status(mcraw_ctx *ctx, uint64_t serial, int32_t *status, int nframes);
int32_t mcraw_ctx_errors(mcraw_ctx *ctx, int reset);

/* ---- several GPUs of one node (device pool) --------------------------------------------------
 *
 * The reference walks a clip frame by frame on one thread (example.cpp:187-195 over
 * lib/Decoder.cpp:184-235).  Frames are independent, so a batch shards by frame index: frame i is
 * decoded by pool member i mod G -- no exchange between devices.  Every member is a context of its
 * own, driven by one host thread of its own that is bound to the CPUs of its GPU's NUMA node (those of them the
 * process may run on).  Results do not depend on the pool size.  Buffers are host memory (MCRAW_MEM_HOST semantics)
 * except for mcraw_pool_decode_batch_device. */
typedef struct mcraw_pool mcraw_pool;
typedef struct mcraw_pool_ticket mcraw_pool_ticket;
struct mcraw_post;

/* The partition rule, usable without a GPU: which member decodes frame `index` (index mod ndevices;
 * -1 on bad arguments), and how many of `nframes` frames member `member` gets. */
int mcraw_shard_of(long index, int ndevices);
int mcraw_shard_count(long nframes, int member, int ndevices);

/* devices[0..ndevices): HIP device indices.  ndevices == 0: env MCRAW_DEVICES ("all" or "0,1,5"),
 * else one member on MCRAW_DEVICE / the current device.  Returns 0 or a negative value. */
int mcraw_pool_create(const int *devices, int ndevices, mcraw_pool **pool);
void mcraw_pool_destroy(mcraw_pool *pool);
const char *mcraw_pool_last_error(void);
int mcraw_pool_size(const mcraw_pool *pool);
int mcraw_pool_device(const mcraw_pool *pool, int member);    /* its HIP device index */
int mcraw_pool_numa_cpus(const mcraw_pool *pool, int member); /* CPUs its host thread is bound to (0: not bound) */
mcraw_ctx *mcraw_pool_ctx(mcraw_pool *pool, int member);      /* for the measurement calls below */
int mcraw_pool_set_post(mcraw_pool *pool, const struct mcraw_post *post);
/* Pinned host memory allocated by the member's own (NUMA-bound) thread: local to its GPU.  Free with mcraw_host_free. */
void *mcraw_pool_host_alloc(mcraw_pool *pool, int member, size_t bytes);
/* One batch over all members; the asynchronous form returns when every member has queued its share.
 * The pool may be used from several host threads at once (batches are dealt one at a time, every member runs its tasks
 * in the order they were handed to it); a ticket is waited for by one thread. */
int mcraw_pool_decode_batch(mcraw_pool *pool, const mcraw_frame *frames, int nframes, size_t *written, int32_t *status);
/* The same for buffers that are already resident: frames[i].in / .out are device pointers in the HBM of the GPU that
 * decodes frame i, mcraw_pool_device(pool, i % mcraw_pool_size(pool)) -- BASELINE config 5's form (a clip sharded over
 * the node's GPUs by frame index; lib/Decoder.cpp:184-235 run as one batch).  With `written` or `status` it returns when
 * every member's share is decoded.  Where a frame's buffers live is CHECKED (hipPointerGetAttributes): a frame whose `in`
 * or `out` is not device memory of the GPU that decodes it gets MCRAW_E_ARGS and is not decoded -- it would be decoded
 * over xGMI at a fraction of the rate, or fault. */
int mcraw_pool_decode_batch_device(mcraw_pool *pool, const mcraw_frame *frames, int nframes, size_t *written, int32_t *status);
/* With `written` and `status` both NULL the call above only queues every member's share (each on its context's own stream)
 * and returns; several batches in a row then run back to back on every GPU.  mcraw_pool_synchronize waits for everything
 * the members have queued and fetches the statuses of the CALLING THREAD's last such batch (status may be NULL) -- the pool
 * may be used from several host threads at once, each sees its own.  It returns a negative value for a runtime failure, else
 * the OR of the statuses of all frames of all queued batches (of any thread) whose outcome became known since the last
 * call: 0 = every frame of every queued batch decoded. */
int mcraw_pool_synchronize(mcraw_pool *pool, int32_t *status, int nframes);
int mcraw_pool_decode_batch_async(mcraw_pool *pool, const mcraw_frame *frames, int nframes, mcraw_pool_ticket **ticket);
int mcraw_pool_ticket_wait(mcraw_pool_ticket *ticket, size_t *written, int32_t *status);

/* The order in which the legacy kernel's workgroups take the segments (16 KiB of stream each) of a batch's legacy frames,
 * as the library builds it for every batch; exported so that the rule can be checked without a GPU.  `nseg[n]`: segments per
 * frame.  The launch goes round by round -- round r = segment r of every frame that has one --, the frames by falling
 * number of segments; stage t = the rounds in which all but the t smallest frames are in play.  `tab` receives 3 n + 1
 * words: [0, n]: first workgroup of every stage (tab[n] = workgroups in all = segments of all frames); [n + 1, 2n]: first
 * round of every stage; [2n + 1, 3n]: the frames in that order.  Workgroup b of stage t works on frame
 * tab[2n + 1 + (b - tab[t]) % (n - t)] and is given -- if workgroups start in order -- segment
 * tab[n + 1 + t] + (b - tab[t]) / (n - t). */
void mcraw_legacy_launch_order(const uint32_t *nseg, int n, uint32_t *tab);

/* The order the tile kernel's workgroups take their work in: block `b` of a grid of `n` blocks works on logical workgroup
 * mcraw_tile_order(b, n, runs) -- runs of `runs` consecutive workgroups per XCD (0: the grid in eight parts; see
 * mcraw_ctx_xcd_runs).  A permutation of 0..n-1 for every n and runs; exported so that this can be checked without a GPU. */
uint32_t mcraw_tile_order(uint32_t b, uint32_t n, uint32_t runs);

/* ---- measurement -------------------------------------------------------- */

/* Kernel ids for mcraw_ctx_kernel_ms. */
#define MCRAW_K7_SIDE    0 /* side streams: chain, records, payload offsets (one launch) */
                           /* ids 1 and 2 are retired (former separate chain kernels)  */
#define MCRAW_K7_TILES   3 /* tile decode (the roofline kernel)   */
                           /* ids 4 and 5 are retired (former legacy map / resolve kernels) */
#define MCRAW_K6_DECODE  6 /* legacy: the whole decode (one launch) */
#define MCRAW_K_COUNT    7

/* hipEvent bracketing of kernel launches on the launch stream: 0 = off, 1 = every
 * kernel, MCRAW_PROFILE_ONLY(id) [| MCRAW_PROFILE_ONLY(id2) ...] = those kernels only
 * (each bracket costs two event records in the stream).  mcraw_ctx_kernel_ms returns,
 * for kernel `id`, the summed duration (ms) and launch count since the last reset
 * (synchronises). */
#define MCRAW_PROFILE_ONLY(id) (2 << (id))
int mcraw_ctx_profile(mcraw_ctx *ctx, int enable);
/* Bracket only every n-th launch of a profiled kernel (n >= 1; default 1): an event pair costs the stream
 * several microseconds, a sample of the launches gives the same average duration. */
int mcraw_ctx_profile_every(mcraw_ctx *ctx, int n);
int mcraw_ctx_kernel_ms(mcraw_ctx *ctx, int id, double *ms, int *launches, int reset);
/* How the tile kernel's workgroups are dealt to the GPU's eight XCDs for large resident batches: the library measures two
 * mappings on the first launches of a geometry (frames per batch, groups per frame) -- which one is faster depends on where
 * the caller's buffers lie in physical memory --, keeps the faster, and times one launch in 64 afterwards (the chosen mapping
 * and the other one in turn) so that the choice follows the caller's buffers; a caller that never reuses a buffer is not kept
 * measuring.  Returns the choice for the geometry of the last such batch: the length of the runs in workgroups (0: the grid
 * in eight parts), -1 while the first measurements are under way or nothing was measured (small or host-memory batches use
 * runs of 128).  Environment MCRAW_XCD_CHUNK pins the mapping (then always -1 here). */
int mcraw_ctx_xcd_runs(mcraw_ctx *ctx);
/* How many workgroups resolve each side stream of the frames of large-frame resident batches (lib/RawData.cpp:463-498 is one chain
 * per stream; which of a frame's two streams is the slow one depends on its content): chosen by measurement on the first launches
 * of a geometry, re-checked by one timed launch in 64.  Returns 16 * (parts of the bits stream) + (parts of the refs stream) for
 * the geometry of the last such batch, -1 while measuring or when nothing was measured. */
int mcraw_ctx_side_parts(mcraw_ctx *ctx);
/* Host-memory batches (MCRAW_MEM_HOST): how the status words of a large batch come home -- 0: fetched when the batch is waited
 * for, 1: written into pinned memory behind the kernels, -1: the context is still comparing the two on its own batches (which is
 * faster depends on what else the process has done with the GPU; MCRAW_SHORT_WAY decides beforehand). */
int mcraw_ctx_host_way(mcraw_ctx *ctx);

/* Optional stage fused behind the decode, for consumers that take the mosaic further on the
 * device or ship it as a DNG strip (what example.cpp:80-92 hands to the DNG writer: the raw strip,
 * BlackLevel, BitsPerSample).  It applies to every batch submitted on `ctx` after the call;
 * NULL (or flags 0) restores the reference's output, the plain uint16 mosaic.
 *   MCRAW_POST_BLACK   sample = max(sample - black[(row & 1) * 2 + (col & 1)], 0)   (after the decode)
 *   MCRAW_POST_PACK12  rows are written as 12-bit strips: ceil(width * 12 / 8) bytes per row, rows
 *                      back to back, samples MSB-first, 3 bytes per 2 samples (TIFF / DNG
 *                      BitsPerSample = 12, FillOrder 1); samples above 4095 saturate.  `out` must be
 *                      2-byte aligned (4-byte aligned and width % 8 == 0 for the vector-store path);
 *                      `out_capacity` still counts uint16 units (2 bytes) of the buffer and `written`
 *                      still counts samples. */
#define MCRAW_POST_BLACK  1u
#define MCRAW_POST_PACK12 2u
/* The same strip form at 10 or 14 bits per sample (5 bytes per 4 samples / 7 bytes per 4 samples; samples above
 * 1023 / 16383 saturate): pick the width from the container's whiteLevel, so that 10-bit footage crosses the
 * host link at 1.25 bytes per sample.  At most one of the three PACK flags.  `out` 2-byte aligned. */
#define MCRAW_POST_PACK10 4u
#define MCRAW_POST_PACK14 8u
typedef struct mcraw_post {
    uint32_t flags;
    uint16_t black[4];
} mcraw_post;
int mcraw_ctx_set_post(mcraw_ctx *ctx, const mcraw_post *post);

/* ---- environment ------------------------------------------------------------------------------
 * Read when a context (or pool) is created, never afterwards:
 *   MCRAW_DEVICE=n, MCRAW_DEVICES=all|0,1,5   default device of the five-argument entry points / members of a default pool
 *   MCRAW_XCD_CHUNK=n                          pins the tile kernel's workgroup-to-XCD mapping (mcraw_ctx_xcd_runs)
 *   MCRAW_SIDE_SPLIT=b[,r]                     pins the workgroups per bits / refs side stream (mcraw_ctx_side_parts)
 *   MCRAW_SIDE_LASTC=0|1                       the last part of a split side stream counts its pieces too (default: by batch size)
 *   MCRAW_SHORT_WAY=0|1                        host-memory pipeline: status words fetched at the wait / written home behind the
 *                                              kernels (default: every context measures which is faster in its process)
 *   MCRAW_TRACE=1                              the verdicts of the contexts' own measurements on stderr
 * Of the HIP runtime (docs/lab_notes.md): GPU_MAX_HW_QUEUES (hardware queues per process and priority). */

/* Pinned host memory for MCRAW_MEM_HOST batches (hipHostMalloc / hipHostFree). */
void *mcraw_host_alloc(size_t bytes);
void mcraw_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* MCRAW_HIP_H */
