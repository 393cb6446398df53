/*
 * mcraw_synth.c -- input synthesis for the MCRAW decode path (host, CPU).
 *
 * The reference ships a decoder only (no encoder, no sample file: SURVEY 4),
 * so every test and benchmark input is produced here: an encoder for the
 * current ("type 7") and legacy ("type 6") frame encodings, the inverse of
 * the format the decoders read, and a deterministic Bayer image generator.
 * This is NOT a decode path and nothing here is called by the HIP library.
 *
 * Format followed (file:line under /root/reference, decode direction):
 *   type 7: lib/RawData.cpp:27-45,106-110,112-408,463-498,500-524,562-593
 *   type 6: lib/RawData_Legacy.cpp:13-36,38-370,372-375,478-491
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const int LEN7[17] = {0, 8, 16, 24, 32, 40, 48, 64, 64, 80, 80, 128, 128, 128, 128, 128, 128};
static const int LEN6[17] = {0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 32, 32, 32, 32, 32, 32};

static int bit_length(unsigned v)
{
    int n = 0;
    while (v) {
        n++;
        v >>= 1;
    }
    return n;
}

static void wr_u32le(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

/* Pack 64 residuals (each < 2^bits, or any u16 for bits >= 11) at storage
 * class `bits`; returns LEN7[bits].  P(i)[j] = out[8*i+j]. */
int mcraw_synth_pack_block7(uint8_t *out, int bits, const uint16_t v[64])
{
    int n = LEN7[bits > 16 ? 16 : bits];
    memset(out, 0, (size_t)n);
#define P(i, j) out[8 * (i) + (j)]
    for (int j = 0; j < 8; j++) {
        switch (bits) {
        case 0:
            break;
        case 1:
            for (int k = 0; k < 8; k++)
                P(0, j) |= (uint8_t)((v[8 * k + j] & 1) << k);
            break;
        case 2:
            for (int h = 0; h < 2; h++)
                for (int k = 0; k < 4; k++)
                    P(h, j) |= (uint8_t)((v[32 * h + 8 * k + j] & 3) << (2 * k));
            break;
        case 3:
            P(0, j) = (uint8_t)((v[j] & 7) | ((v[8 + j] & 7) << 3) | ((v[16 + j] & 3) << 6));
            P(1, j) = (uint8_t)((v[24 + j] & 7) | ((v[32 + j] & 7) << 3) | ((v[40 + j] & 3) << 6));
            P(2, j) = (uint8_t)((v[48 + j] & 7) | ((v[56 + j] & 7) << 3) |
                                (((v[16 + j] >> 2) & 1) << 6) | (((v[40 + j] >> 2) & 1) << 7));
            break;
        case 4:
            for (int g = 0; g < 4; g++)
                P(g, j) = (uint8_t)((v[16 * g + j] & 15) | ((v[16 * g + 8 + j] & 15) << 4));
            break;
        case 5:
            for (int i = 0; i < 5; i++)
                P(i, j) = (uint8_t)(v[8 * i + j] & 31);
            P(0, j) |= (uint8_t)((v[40 + j] & 7) << 5);
            P(3, j) |= (uint8_t)(((v[40 + j] >> 3) & 3) << 5);
            P(1, j) |= (uint8_t)((v[48 + j] & 7) << 5);
            P(4, j) |= (uint8_t)(((v[48 + j] >> 3) & 3) << 5);
            P(2, j) |= (uint8_t)((v[56 + j] & 7) << 5);
            P(3, j) |= (uint8_t)(((v[56 + j] >> 3) & 1) << 7);
            P(4, j) |= (uint8_t)(((v[56 + j] >> 4) & 1) << 7);
            break;
        case 6:
            for (int i = 0; i < 6; i++)
                P(i, j) = (uint8_t)(v[8 * i + j] & 63);
            for (int t = 0; t < 3; t++) {
                P(t, j) |= (uint8_t)(((v[48 + j] >> (2 * t)) & 3) << 6);
                P(3 + t, j) |= (uint8_t)(((v[56 + j] >> (2 * t)) & 3) << 6);
            }
            break;
        case 7:
        case 8:
            for (int k = 0; k < 8; k++)
                P(k, j) = (uint8_t)v[8 * k + j];
            break;
        case 9:
        case 10:
            for (int h = 0; h < 2; h++)
                for (int i = 0; i < 4; i++) {
                    uint16_t s = v[32 * h + 8 * i + j];
                    P(5 * h + i, j) = (uint8_t)s;
                    P(5 * h + 4, j) |= (uint8_t)(((s >> 8) & 3) << (2 * i));
                }
            break;
        default:
            for (int k = 0; k < 8; k++) {
                uint16_t s = v[8 * k + j];
                out[2 * (8 * k + j)] = (uint8_t)s;
                out[2 * (8 * k + j) + 1] = (uint8_t)(s >> 8);
            }
            break;
        }
    }
#undef P
    return n;
}

/* Side stream: u32 count, then per 64 entries {hbits<<4|ref>>8, ref&255, block}.
 * `vals` has `padded` entries (multiple of 64); `count` is the value written to
 * the count field. */
static size_t put_side_stream7(uint8_t *out, const uint16_t *vals, size_t padded, uint32_t count, int extra_records)
{
    size_t o = 0;
    wr_u32le(out, count);
    o += 4;
    for (size_t r = 0; r < padded + 64 * (size_t)extra_records; r += 64) {
        if (r >= padded) { /* records behind the ones the frame uses: entries 7, 8, 9, ... (never read by a decoder) */
            uint16_t res[64];
            for (int i = 0; i < 64; i++)
                res[i] = (uint16_t)(i & 7);
            out[o] = (uint8_t)((3 << 4) | 0);
            out[o + 1] = 7;
            o += 2;
            o += (size_t)mcraw_synth_pack_block7(out + o, 3, res);
            continue;
        }
        unsigned mn = 65535, mx = 0;
        for (int i = 0; i < 64; i++) {
            if (vals[r + i] < mn)
                mn = vals[r + i];
            if (vals[r + i] > mx)
                mx = vals[r + i];
        }
        unsigned ref = mn > 4095 ? 4095 : mn; /* header reference is 12 bits */
        int hb = bit_length(mx - ref);
        if (hb > 15)
            hb = 15; /* header nibble; 11..15 all mean raw-16 storage */
        uint16_t res[64];
        for (int i = 0; i < 64; i++)
            res[i] = (uint16_t)(vals[r + i] - ref);
        out[o] = (uint8_t)((hb << 4) | (ref >> 8));
        out[o + 1] = (uint8_t)(ref & 255);
        o += 2;
        o += (size_t)mcraw_synth_pack_block7(out + o, hb, res);
    }
    return o;
}

/* Worst-case encoded size of a type-7 frame. */
size_t mcraw_synth_bound7(int width, int height)
{
    size_t encW = ((size_t)width + 63) / 64 * 64, encH = ((size_t)height + 3) / 4 * 4;
    size_t nblk = encW * encH / 64, padded = (nblk + 63) / 64 * 64;
    return 16 + nblk * 128 + 2 * (4 + (padded / 64 + 3) * 130) + 64 + 64;
}

/* Encode a width x height uint16 Bayer mosaic as a type-7 frame buffer.
 *   min_bits : optional per-block lower bound on `bits` (NULL = natural), to
 *              force storage classes; indexed like the bits side stream.
 *   flags    : bit0 = write the UNROUNDED entry count in the side streams
 *              (real files may; the reference then overflows);
 *              layouts other writers may produce (the decoder takes every position from the header,
 *              lib/RawData.cpp:500-524):
 *              bit1 = the refs stream in front of the bits stream;
 *              bit2 = unused bytes (0xEE) between payload and streams, between the streams and behind
 *                     them (13, 5 and 3 bytes: the streams then start on odd addresses);
 *              bit3 = three more records than the frame uses in each stream, counted in its entry count.
 * Returns bytes written (0 on bad arguments). */
size_t mcraw_synth_encode7(uint8_t *out, size_t cap, const uint16_t *img, int width, int height,
                           const uint8_t *min_bits, int flags)
{
    if (width <= 0 || height <= 0 || cap < mcraw_synth_bound7(width, height))
        return 0;
    size_t encW = ((size_t)width + 63) / 64 * 64, encH = ((size_t)height + 3) / 4 * 4;
    size_t tilesX = encW / 64, tilesY = encH / 4, nblk = 4 * tilesX * tilesY;
    size_t padded = (nblk + 63) / 64 * 64;
    uint16_t *bits = (uint16_t *)calloc(padded, 2), *refs = (uint16_t *)calloc(padded, 2);
    if (!bits || !refs) {
        free(bits);
        free(refs);
        return 0;
    }
    size_t o = 16, m = 0;
    for (size_t ty = 0; ty < tilesY; ty++)
        for (size_t tx = 0; tx < tilesX; tx++)
            for (int b = 0; b < 4; b++, m++) {
                uint16_t s[64];
                unsigned mn = 65535, mx = 0;
                for (int i = 0; i < 64; i++) {
                    size_t y = 4 * ty + (size_t)(b >> 1) + 2 * (size_t)(i >> 5);
                    size_t x = 64 * tx + 2 * (size_t)(i & 31) + (size_t)(b & 1);
                    /* padding: clamp to the last pixel of the same Bayer parity */
                    while (y >= (size_t)height && y >= 2)
                        y -= 2;
                    if (y >= (size_t)height)
                        y = (size_t)height - 1;
                    while (x >= (size_t)width && x >= 2)
                        x -= 2;
                    if (x >= (size_t)width)
                        x = (size_t)width - 1;
                    s[i] = img[y * (size_t)width + x];
                    if (s[i] < mn)
                        mn = s[i];
                    if (s[i] > mx)
                        mx = s[i];
                }
                int nb = bit_length(mx - mn);
                if (min_bits && min_bits[m] > nb)
                    nb = min_bits[m] > 16 ? 16 : min_bits[m];
                for (int i = 0; i < 64; i++)
                    s[i] = (uint16_t)(s[i] - mn);
                bits[m] = (uint16_t)nb;
                refs[m] = (uint16_t)mn;
                o += (size_t)mcraw_synth_pack_block7(out + o, nb, s);
            }
    const int extra = (flags & 8) ? 3 : 0;
    uint32_t count = ((flags & 1) ? (uint32_t)nblk : (uint32_t)padded) + 64u * (uint32_t)extra;
    wr_u32le(out + 0, (uint32_t)encW);
    wr_u32le(out + 4, (uint32_t)encH);
    if (flags & 4) {
        memset(out + o, 0xEE, 13);
        o += 13;
    }
    for (int k = 0; k < 2; k++) {
        const int which = (flags & 2) ? 1 - k : k; /* 0 = bits, 1 = refs */
        wr_u32le(out + (which ? 12 : 8), (uint32_t)o);
        o += put_side_stream7(out + o, which ? refs : bits, padded, count, extra);
        if (flags & 4) {
            memset(out + o, 0xEE, k ? 3 : 5);
            o += k ? 3 : 5;
        }
    }
    free(bits);
    free(refs);
    return o;
}

/* ---------------------------------------------------------------- type 6 */

size_t mcraw_synth_bound6(int width, int height)
{
    size_t padded = ((size_t)width + 31) / 32 * 32;
    return padded / 16 * 34 * (size_t)height + 16 + 5 * ((size_t)height / 8 + 2);
}

/* MSB-first bit writer for one 16-sample legacy block. */
static int pack_block6(uint8_t *out, int bits, const uint16_t v[16])
{
    int n = LEN6[bits > 16 ? 16 : bits];
    memset(out, 0, (size_t)n);
    if (bits == 0)
        return 0;
    if (bits > 10) {
        for (int i = 0; i < 16; i++) {
            out[2 * i] = (uint8_t)(v[i] >> 8); /* big-endian */
            out[2 * i + 1] = (uint8_t)v[i];
        }
        return 32;
    }
    for (int k = 0; k < 16; k++)
        for (int t = 0; t < bits; t++) {
            int pos = k * bits + t;
            if ((v[k] >> (bits - 1 - t)) & 1)
                out[pos >> 3] |= (uint8_t)(0x80 >> (pos & 7));
        }
    return n;
}

/* Encode as a legacy (type 6) stream.  min_bits: optional per-record lower
 * bound (record index = (y * paddedWidth/32 + group) * 2 + parity).
 * flags bit0: append a restart-offset trailer ([non-0xFF][u32 BE][0xFF])
 * like real files (RawData_Legacy.cpp:455-469) instead of a single 0x00. */
size_t mcraw_synth_encode6(uint8_t *out, size_t cap, const uint16_t *img, int width, int height,
                           const uint8_t *min_bits, int flags)
{
    if (width <= 0 || height <= 0 || cap < mcraw_synth_bound6(width, height))
        return 0;
    int padded = (width + 31) / 32 * 32;
    size_t o = 0, rec = 0;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < padded; x += 32)
            for (int par = 0; par < 2; par++, rec++) {
                uint16_t s[16];
                unsigned mn = 65535, mx = 0;
                for (int i = 0; i < 16; i++) {
                    int xx = x + 2 * i + par;
                    while (xx >= width && xx >= 2)
                        xx -= 2;
                    if (xx >= width)
                        xx = width - 1;
                    s[i] = img[(size_t)y * (size_t)width + (size_t)xx];
                    if (s[i] < mn)
                        mn = s[i];
                    if (s[i] > mx)
                        mx = s[i];
                }
                unsigned ref = mn > 4095 ? 4095 : mn; /* 12-bit reference */
                int nb = bit_length(mx - ref);
                if (min_bits && min_bits[rec] > nb)
                    nb = min_bits[rec];
                if (nb > 15)
                    nb = 15; /* header nibble; 11..15 = raw-16 big-endian */
                for (int i = 0; i < 16; i++)
                    s[i] = (uint16_t)(s[i] - ref);
                out[o] = (uint8_t)((nb << 4) | (ref >> 8));
                out[o + 1] = (uint8_t)(ref & 255);
                o += 2;
                o += (size_t)pack_block6(out + o, nb, s);
            }
    if (flags & 2) { /* a trailer of many restart records, one per 8 rows ([u32 BE position][0xFF], RawData_Legacy.cpp:455-469) */
        out[o++] = 0x00;
        for (int y = 0; y < height; y += 8) {
            uint32_t pos = (uint32_t)((size_t)y * 977u);
            out[o++] = (uint8_t)(pos >> 24);
            out[o++] = (uint8_t)(pos >> 16);
            out[o++] = (uint8_t)(pos >> 8);
            out[o++] = (uint8_t)pos;
            out[o++] = 0xFF;
        }
    } else if (flags & 1) {
        uint32_t pos = (uint32_t)(o / 2);
        out[o++] = 0x00;
        out[o++] = (uint8_t)(pos >> 24);
        out[o++] = (uint8_t)(pos >> 16);
        out[o++] = (uint8_t)(pos >> 8);
        out[o++] = (uint8_t)pos;
        out[o++] = 0xFF;
    } else {
        out[o++] = 0x00;
    }
    return o;
}

/* --------------------------------------------------------- image generator */

static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Deterministic synthetic Bayer frame, values in [0, 2^nbits - 1].
 *   dist 0 ("U")  : i.i.d. uniform samples -> one storage class per frame.
 *   dist 1 ("Nat"): smooth field 0.8*max*(0.5 + 0.45*sin(x/211)*cos(y/173)) +
 *                   black level max/16 + noise of std `sigma` (sum of four
 *                   uniforms, near-Gaussian), clipped -> mixed classes.
 * (SURVEY 8d distributions; the bulk generator trades numpy's exact Gaussian
 * for speed, tests/golden uses numpy.) */
void mcraw_synth_image(uint16_t *img, int width, int height, int nbits, int dist, double sigma,
                       uint64_t seed)
{
    const double maxv = (double)((1u << nbits) - 1);
    uint64_t st = seed * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull;
    /* per-row/column separable smooth field: sin(x/211) and cos(y/173) tables */
    double *sx = (double *)malloc(sizeof(double) * (size_t)width);
    for (int x = 0; x < width; x++)
        sx[x] = sin((double)x / 211.0);
    /* sum of 4 uniform[0,1) has std sqrt(4/12); scale to sigma */
    const double nscale = sigma / 0.5773502691896258 / 65536.0;
    for (int y = 0; y < height; y++) {
        double cy = cos((double)y / 173.0);
        uint16_t *row = img + (size_t)y * (size_t)width;
        for (int x = 0; x < width; x++) {
            uint64_t r = splitmix64(&st);
            if (dist == 0) {
                row[x] = (uint16_t)(r >> (64 - nbits));
            } else {
                double n = (double)((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) +
                                    (r >> 48)) - 2.0 * 65535.0;
                double v = 0.8 * maxv * (0.5 + 0.45 * sx[x] * cy) + maxv / 16.0 + n * nscale;
                if (v < 0.0)
                    v = 0.0;
                if (v > maxv)
                    v = maxv;
                row[x] = (uint16_t)(v + 0.5);
            }
        }
    }
    free(sx);
}
