/*
 * mcraw_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain scalar C) of the MCRAW frame codecs of
 * mirsadm/motioncam-decoder.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped decode
 * path (motioncam_decoder_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py compares every
 * function here with the real reference compiled into oracle/_ref/ (recipe:
 * oracle/Makefile) and tests/golden/ holds vectors produced by that build.
 *
 * Reference followed (file:line under /root/reference):
 *   type 7  lib/RawData.cpp:27-45 (lengths) :106-110 (header) :112-408 (unpack)
 *           :410-461 (block) :463-498 (side stream) :500-524 (frame header)
 *           :528-612 (frame)
 *   type 6  lib/RawData_Legacy.cpp:13-36 :38-370 :372-442 :445-495
 */
#ifndef MCRAW_ORACLE_H
#define MCRAW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same five arguments and return convention as motioncam::raw::Decode
 * (lib/include/motioncam/RawData.hpp:25-30): number of uint16 written, 0 on
 * failure.  Deviations, all outside the reference's well-formed domain
 * (SURVEY Appendix A.4): rows are clipped to min(height, encodedHeight)
 * (the reference writes encodedHeight rows, RawData.cpp:571), and truncated
 * blocks / bits > 16 / short side streams return 0 where the reference skips
 * the block or reads out of bounds (RawData.cpp:419-420). */
size_t mcraw_oracle_decode7(uint16_t *out, int width, int height,
                            const uint8_t *in, size_t len);

/* motioncam::raw::DecodeLegacy (RawData.hpp:32-37, RawData_Legacy.cpp:445-495).
 * Returns width*height or 0.  A record that fails the reference's '>=' bound
 * (RawData_Legacy.cpp:387-388,398-399) makes the frame fail instead of
 * leaving stale samples. */
size_t mcraw_oracle_decode6(uint16_t *out, int width, int height,
                            const uint8_t *in, size_t len);

/* One 64-sample type-7 block (RawData.cpp:410-461): bytes consumed. */
int mcraw_oracle_block7(uint16_t out[64], int bits, const uint8_t *in);

/* One 16-sample legacy block payload (RawData_Legacy.cpp:401-439). */
int mcraw_oracle_block6(uint16_t out[16], int bits, const uint8_t *in);

/* Algorithmic input bytes of a type-7 frame: header + payload + both side
 * streams (SURVEY 8d "len_used"); 0 if malformed. */
size_t mcraw_oracle_len_used7(const uint8_t *in, size_t len);

/* The optional stage the product can fuse behind the decode (include/mcraw_hip.h,
 * mcraw_ctx_set_post) -- there is no reference function for it; this is its definition, applied to
 * a decoded mosaic: what a DNG writer is handed next (example.cpp:80-92: strip, BlackLevel,
 * BitsPerSample).  flags bit 0: sample = max(sample - black[(row & 1) * 2 + (col & 1)], 0);
 * bit 1: rows packed as 12-bit MSB-first strips of ceil(width * 12 / 8) bytes, samples saturating
 * at 4095.  Without bit 1 the output is the uint16 little-endian mosaic.  Returns bytes written. */
size_t mcraw_oracle_post(uint8_t *out, const uint16_t *img, int width, int height,
                         unsigned flags, const uint16_t black[4]);

/* Frame-parallel timing helper for bench.py's cpu_baseline: decodes
 * `nframes` buffers with `nthreads` pthreads (one frame per task), `reps`
 * passes, returns seconds of wall time for all passes, <0 on decode error. */
double mcraw_oracle_time_batch(int type, int width, int height,
                               const uint8_t *const *ins, const size_t *lens,
                               int nframes, int nthreads, int reps);

#ifdef __cplusplus
}
#endif
#endif
