/* abi_probe.c -- the C ABI header must be plain C (C99), and a C program must be able to link the
 * library and call it.  Without a GPU every entry point fails cleanly (no CPU fallback); with one,
 * a tiny all-zero type-7 frame decodes.  Prints one line per observation for tests/test_abi_library.py. */
#include "mcraw_hip.h"

#include <stdio.h>
#include <string.h>

int main(void)
{
    mcraw_ctx *ctx = NULL;
    int rc = mcraw_ctx_create(-1, &ctx);
    printf("ctx_create rc=%d ctx=%s\n", rc, ctx ? "yes" : "no");
    /* 64x4 frame of zeros: header (encW 64, encH 4, bits stream at 16, refs stream at 22), no payload,
     * two side streams of one all-zero record each: count 64 (one record), header bytes 0 0 -- byte for
     * byte what the encoder emits for a black 64x4 image */
    unsigned char buf[32];
    memset(buf, 0, sizeof buf);
    buf[0] = 64; buf[4] = 4; buf[8] = 16; buf[12] = 22;
    buf[16] = 64; /* bits stream: u32 count, then record header (class 0, reference 0) */
    buf[22] = 64; /* refs stream */
    unsigned short out[64 * 4];
    memset(out, 0xA5, sizeof out);
    size_t n = mcraw_decode7(out, 64, 4, buf, 28);
    printf("decode7 returned %zu first=%u\n", n, (unsigned)out[0]);
    if (rc != 0) {
        printf("last_error: %s\n", mcraw_last_error());
        return n == 0 ? 0 : 1; /* no device: must have failed */
    }
    mcraw_post post;
    memset(&post, 0, sizeof post);
    post.flags = MCRAW_POST_BLACK;
    printf("set_post rc=%d\n", mcraw_ctx_set_post(ctx, &post));
    printf("set_post(NULL) rc=%d\n", mcraw_ctx_set_post(ctx, NULL));
    mcraw_ctx_destroy(ctx);
    return n == 64 * 4 && out[0] == 0 ? 0 : 1;
}
