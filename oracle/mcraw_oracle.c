/*
 * mcraw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see mcraw_oracle.h).
 *
 * Scalar restatement of the two MCRAW codecs.  Written from the format
 * (SURVEY.md Appendix A); every routine cites the reference lines it follows.
 * Parity: pinned against the compiled reference (oracle/_ref, oracle/Makefile)
 * by tests/test_oracle_vs_reference.py and against tests/golden/.
 */
#define _POSIX_C_SOURCE 200809L
#include "mcraw_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---------------------------------------------------------------- type 7 */

/* RawData.cpp:27-45 */
static const int LEN7[17] = {0, 8, 16, 24, 32, 40, 48, 64, 64,
                             80, 80, 128, 128, 128, 128, 128, 128};

static inline uint32_t rd_u32le(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
           ((uint32_t)p[3] << 24);
}

/* P(i, j): byte j of 8-byte plane i of the block. */
#define P(i, j) ((uint16_t)in[8 * (i) + (j)])

int mcraw_oracle_block7(uint16_t out[64], int bits, const uint8_t *in)
{
    int j, k, i, h, g;
    switch (bits) {
    case 0: /* RawData.cpp:425-427 */
        memset(out, 0, 64 * sizeof(uint16_t));
        return 0;
    case 1: /* :112-136 */
        for (k = 0; k < 8; k++)
            for (j = 0; j < 8; j++)
                out[8 * k + j] = (P(0, j) >> k) & 1;
        return 8;
    case 2: /* :138-162 */
        for (h = 0; h < 2; h++)
            for (k = 0; k < 4; k++)
                for (j = 0; j < 8; j++)
                    out[32 * h + 8 * k + j] = (P(h, j) >> (2 * k)) & 3;
        return 16;
    case 3: /* :164-199 */
        for (j = 0; j < 8; j++) {
            out[j]      = P(0, j) & 7;
            out[8 + j]  = (P(0, j) >> 3) & 7;
            out[16 + j] = ((P(0, j) >> 6) & 3) | (((P(2, j) >> 6) & 1) << 2);
            out[24 + j] = P(1, j) & 7;
            out[32 + j] = (P(1, j) >> 3) & 7;
            out[40 + j] = ((P(1, j) >> 6) & 3) | (((P(2, j) >> 7) & 1) << 2);
            out[48 + j] = P(2, j) & 7;
            out[56 + j] = (P(2, j) >> 3) & 7;
        }
        return 24;
    case 4: /* :201-223 */
        for (g = 0; g < 4; g++)
            for (j = 0; j < 8; j++) {
                out[16 * g + j]     = P(g, j) & 15;
                out[16 * g + 8 + j] = (P(g, j) >> 4) & 15;
            }
        return 32;
    case 5: /* :225-262 */
        for (j = 0; j < 8; j++) {
            for (i = 0; i < 5; i++)
                out[8 * i + j] = P(i, j) & 31;
            out[40 + j] = ((P(0, j) >> 5) & 7) | (((P(3, j) >> 5) & 3) << 3);
            out[48 + j] = ((P(1, j) >> 5) & 7) | (((P(4, j) >> 5) & 3) << 3);
            out[56 + j] = ((P(2, j) >> 5) & 7) | (((P(3, j) >> 7) & 1) << 3) |
                          (((P(4, j) >> 7) & 1) << 4);
        }
        return 40;
    case 6: /* :264-304 */
        for (j = 0; j < 8; j++) {
            for (i = 0; i < 6; i++)
                out[8 * i + j] = P(i, j) & 63;
            out[48 + j] = ((P(0, j) >> 6) & 3) | (((P(1, j) >> 6) & 3) << 2) |
                          (((P(2, j) >> 6) & 3) << 4);
            out[56 + j] = ((P(3, j) >> 6) & 3) | (((P(4, j) >> 6) & 3) << 2) |
                          (((P(5, j) >> 6) & 3) << 4);
        }
        return 48;
    case 7:
    case 8: /* :306-326, dispatch :446-449 */
        for (i = 0; i < 64; i++)
            out[i] = in[i];
        return 64;
    case 9:
    case 10: /* :328-374, dispatch :450-453 */
        for (h = 0; h < 2; h++)
            for (i = 0; i < 4; i++)
                for (j = 0; j < 8; j++)
                    out[32 * h + 8 * i + j] =
                        P(5 * h + i, j) |
                        (((P(5 * h + 4, j) >> (2 * i)) & 3) << 8);
        return 80;
    default: /* 11.. : :376-408, dispatch :454-457 (host-endian = LE) */
        for (i = 0; i < 64; i++)
            out[i] = (uint16_t)(in[2 * i] | (in[2 * i + 1] << 8));
        return 128;
    }
}
#undef P

/* Side stream (RawData.cpp:463-498).  Decodes ceil(count/64) records into
 * `vals` (capacity cap, multiple of 64); returns entry count or -1. */
static long side_stream7(const uint8_t *in, size_t len, size_t off,
                         uint16_t *vals, size_t cap, size_t need,
                         size_t *end_off)
{
    if (off + 4 > len)
        return -1;
    uint32_t count = rd_u32le(in + off); /* :470-474 */
    off += 4;
    if ((size_t)count < need)
        return -1;
    size_t nrec = (need + 63) / 64; /* only the entries the frame uses */
    if (nrec * 64 > cap)
        return -1;
    for (size_t r = 0; r < nrec; r++) {
        if (off + 2 > len)
            return -1;
        int hbits = in[off] >> 4;                          /* :106-110 */
        uint16_t ref = (uint16_t)(((in[off] & 15) << 8) | in[off + 1]);
        off += 2;                                          /* :488 */
        if (off + (size_t)LEN7[hbits] > len)               /* :419-420 */
            return -1;
        uint16_t *d = vals + 64 * r;
        off += (size_t)mcraw_oracle_block7(d, hbits, in + off);
        for (int x = 0; x < 64; x++)                       /* :491-492 */
            d[x] = (uint16_t)(d[x] + ref);
    }
    if (end_off)
        *end_off = off;
    return (long)count;
}

static size_t decode7_impl(uint16_t *out, int width, int height,
                           const uint8_t *in, size_t len, size_t *len_used)
{
    if (len < 16 || width <= 0 || height <= 0)
        return 0;
    uint32_t encW = rd_u32le(in), encH = rd_u32le(in + 4); /* :500-524 */
    uint32_t bitsOff = rd_u32le(in + 8), refsOff = rd_u32le(in + 12);
    if (bitsOff > len || refsOff > len)                    /* :547-548 */
        return 0;
    if (encW % 64 != 0 || encW < (uint32_t)width || encW == 0) /* :550-554 */
        return 0;
    if (encH % 4 != 0 || encH == 0)                        /* loop step :571 */
        return 0;

    size_t tilesX = encW / 64, tilesY = encH / 4;
    size_t nblk = 4 * tilesX * tilesY;
    size_t cap = (nblk + 63) / 64 * 64;
    uint16_t *bits = (uint16_t *)malloc(cap * sizeof(uint16_t));
    uint16_t *refs = (uint16_t *)malloc(cap * sizeof(uint16_t));
    size_t ret = 0, bitsEnd = 0, refsEnd = 0;
    if (!bits || !refs)
        goto done;
    if (side_stream7(in, len, bitsOff, bits, cap, nblk, &bitsEnd) < 0) /* :557 */
        goto done;
    if (side_stream7(in, len, refsOff, refs, cap, nblk, &refsEnd) < 0) /* :560 */
        goto done;

    size_t rows = (size_t)height < encH ? (size_t)height : encH;
    size_t off = 16;                                       /* :562 */
    size_t m = 0;
    uint16_t p[4][64];
    for (size_t ty = 0; ty < tilesY; ty++) {               /* :571 */
        for (size_t tx = 0; tx < tilesX; tx++, m += 4) {   /* :572 */
            for (int b = 0; b < 4; b++) {                  /* :576-579 */
                int bb = bits[m + b];
                if (bb > 16)
                    goto done;
                if (off + (size_t)LEN7[bb] > len)          /* :419-420 */
                    goto done;
                off += (size_t)mcraw_oracle_block7(p[b], bb, in + off);
            }
            /* :581-593 -- block b is the colour plane (row parity b>>1,
             * column parity b&1); samples 0..31 first row pair, 32..63 second */
            if (4 * ty + 4 <= rows && 64 * tx + 64 <= (size_t)width) {
                /* interior tile: no crop needed */
                for (int b = 0; b < 4; b++) {
                    uint16_t ref = refs[m + b];
                    uint16_t *o0 = out + (4 * ty + (size_t)(b >> 1)) * (size_t)width + 64 * tx + (size_t)(b & 1);
                    uint16_t *o1 = o0 + 2 * (size_t)width;
                    for (int s = 0; s < 32; s++) {
                        o0[2 * s] = (uint16_t)(p[b][s] + ref);
                        o1[2 * s] = (uint16_t)(p[b][32 + s] + ref);
                    }
                }
                continue;
            }
            for (int b = 0; b < 4; b++) {
                uint16_t ref = refs[m + b];
                for (int s = 0; s < 64; s++) {
                    size_t y = 4 * ty + (size_t)(b >> 1) + 2 * (size_t)(s >> 5);
                    size_t x = 64 * tx + 2 * (size_t)(s & 31) + (size_t)(b & 1);
                    if (y < rows && x < (size_t)width)     /* crop :598-608 */
                        out[y * (size_t)width + x] = (uint16_t)(p[b][s] + ref);
                }
            }
        }
    }
    ret = (size_t)width * rows;                            /* :611 */
    if (len_used)
        *len_used = off + (bitsEnd - bitsOff) + (refsEnd - refsOff);
done:
    free(bits);
    free(refs);
    return ret;
}

size_t mcraw_oracle_decode7(uint16_t *out, int width, int height,
                            const uint8_t *in, size_t len)
{
    return decode7_impl(out, width, height, in, len, NULL);
}

size_t mcraw_oracle_len_used7(const uint8_t *in, size_t len)
{
    if (len < 16)
        return 0;
    uint32_t encW = rd_u32le(in), encH = rd_u32le(in + 4);
    if (encW == 0 || encH == 0 || encW > (1u << 20) || encH > (1u << 20))
        return 0;
    uint16_t *tmp = (uint16_t *)malloc((size_t)encW * encH * 2);
    size_t used = 0;
    if (tmp && !decode7_impl(tmp, (int)encW, (int)encH, in, len, &used))
        used = 0;
    free(tmp);
    return used;
}

/* ---------------------------------------------------------------- post stage */

size_t mcraw_oracle_post(uint8_t *out, const uint16_t *img, int width, int height,
                         unsigned flags, const uint16_t black[4])
{
    /* strip width: flag 2 = 12 bits, 4 = 10 bits, 8 = 14 bits per sample, none = uint16 LE */
    const unsigned bits = (flags & 2u) ? 12u : (flags & 4u) ? 10u : (flags & 8u) ? 14u : 16u;
    const size_t row_bytes = ((size_t)width * bits + 7) / 8;
    for (int y = 0; y < height; y++) {
        uint8_t *row = out + (size_t)y * row_bytes;
        if (bits != 16u)
            memset(row, 0, row_bytes);
        for (int x = 0; x < width; x++) {
            unsigned v = img[(size_t)y * width + x];
            if (flags & 1u) {
                unsigned b = black[(y & 1) * 2 + (x & 1)];
                v = v > b ? v - b : 0;
            }
            if (bits != 16u) {
                if (v > (1u << bits) - 1u)
                    v = (1u << bits) - 1u;
                /* MSB-first: sample x occupies bits [bits*x, bits*x + bits) of the row */
                size_t bit = (size_t)x * bits;
                uint32_t t = (uint32_t)v << (24u - bits - (unsigned)(bit & 7)); /* at most 14 + 7 bits: three bytes */
                row[bit >> 3] |= (uint8_t)(t >> 16);
                if ((bit >> 3) + 1 < row_bytes)
                    row[(bit >> 3) + 1] |= (uint8_t)(t >> 8);
                if ((bit >> 3) + 2 < row_bytes)
                    row[(bit >> 3) + 2] |= (uint8_t)t;
            } else {
                row[2 * x] = (uint8_t)(v & 255);
                row[2 * x + 1] = (uint8_t)(v >> 8);
            }
        }
    }
    return row_bytes * (size_t)height;
}

/* ---------------------------------------------------------------- type 6 */

/* RawData_Legacy.cpp:13-32 */
static const int LEN6[17] = {0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20,
                             32, 32, 32, 32, 32, 32};

int mcraw_oracle_block6(uint16_t out[16], int bits, const uint8_t *in)
{
    if (bits == 0) { /* :402-404 */
        memset(out, 0, 16 * sizeof(uint16_t));
        return 0;
    }
    if (bits > 10) { /* :360-370 big-endian u16 */
        for (int i = 0; i < 16; i++)
            out[i] = (uint16_t)((in[2 * i] << 8) | in[2 * i + 1]);
        return 32;
    }
    /* :38-358 -- all ten routines are one MSB-first bitstream: sample k is
     * bits [k*b, (k+1)*b) counted from the MSB of byte 0. */
    for (int k = 0; k < 16; k++) {
        unsigned v = 0;
        for (int t = 0; t < bits; t++) {
            int pos = k * bits + t;
            v = (v << 1) | ((in[pos >> 3] >> (7 - (pos & 7))) & 1u);
        }
        out[k] = (uint16_t)v;
    }
    return 2 * bits;
}

size_t mcraw_oracle_decode6(uint16_t *out, int width, int height,
                            const uint8_t *in, size_t len)
{
    if (width <= 0 || height <= 0 || len == 0)
        return 0;
    int padded = 32 * ((width + 31) / 32); /* :34-36 */
    size_t off = 0;
    uint16_t p[2][16], ref[2];
    for (int y = 0; y < height; y++) {     /* :478 */
        for (int x = 0; x < padded; x += 32) { /* :479 */
            for (int b = 0; b < 2; b++) {  /* :480-481 */
                if (off + 2 >= len)        /* :387-388 */
                    return 0;
                int bits = in[off] >> 4;   /* :372-375 */
                ref[b] = (uint16_t)(((in[off] & 15) << 8) | in[off + 1]);
                if (off + 2 + (size_t)LEN6[bits] >= len) /* :398-399 */
                    return 0;
                mcraw_oracle_block6(p[b], bits, in + off + 2);
                off += 2 + (size_t)LEN6[bits]; /* :441 */
            }
            for (int i = 0; i < 16; i++) { /* :483-486 */
                int xe = x + 2 * i, xo = x + 2 * i + 1;
                if (xe < width)            /* crop :490 */
                    out[(size_t)y * width + xe] = (uint16_t)(p[0][i] + ref[0]);
                if (xo < width)
                    out[(size_t)y * width + xo] = (uint16_t)(p[1][i] + ref[1]);
            }
        }
    }
    return (size_t)width * (size_t)height; /* :494 */
}

/* ------------------------------------------------- cpu_baseline timing leg */

typedef struct {
    int type, width, height, nframes, reps;
    const uint8_t *const *ins;
    const size_t *lens;
    int tid, nthreads;
    int err;
} work_t;

static void *worker(void *arg)
{
    work_t *w = (work_t *)arg;
    uint16_t *out = (uint16_t *)malloc((size_t)w->width * ((size_t)w->height + 4) * 2);
    if (!out) {
        w->err = 1;
        return NULL;
    }
    for (int r = 0; r < w->reps; r++)
        for (int f = w->tid; f < w->nframes; f += w->nthreads) {
            size_t n = w->type == 7
                           ? mcraw_oracle_decode7(out, w->width, w->height, w->ins[f], w->lens[f])
                           : mcraw_oracle_decode6(out, w->width, w->height, w->ins[f], w->lens[f]);
            if (n == 0)
                w->err = 1;
        }
    free(out);
    return NULL;
}

double mcraw_oracle_time_batch(int type, int width, int height,
                               const uint8_t *const *ins, const size_t *lens,
                               int nframes, int nthreads, int reps)
{
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > 1024)
        nthreads = 1024;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    work_t *ws = (work_t *)malloc(sizeof(work_t) * (size_t)nthreads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; t++) {
        ws[t] = (work_t){type, width, height, nframes, reps, ins, lens, t, nthreads, 0};
        pthread_create(&th[t], NULL, worker, &ws[t]);
    }
    int err = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        err |= ws[t].err;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(ws);
    if (err)
        return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
