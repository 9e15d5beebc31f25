/*
 * bitcursor.c -- the byte/bit cursor of the reference API (bitstream.h:4-22).
 *
 * The segment parsers in segments.c consume markers and header fields through
 * next_byte()/get_byte().  The bit-level calls (next_bit, next_bit_size,
 * check_marker, next_byte_restart_marker) are kept with the reference's semantics
 * for callers that link against them, but the decode path does not use them: on
 * the GPU, FF00 unstuffing and restart-marker discovery are done by the scan
 * kernels over the whole entropy-coded segment at once.
 */
#include "mi355jpeg_compat.h"

uint8_t get_byte(Bitstream *bs) {
    return bs->encoded_data[0];
}

uint8_t next_byte(Bitstream *bs) {
    uint8_t v = bs->encoded_data[0];
    bs->encoded_data += 1;
    bs->size += 1;
    bs->offset = 0;
    return v;
}

int progress(Bitstream *bs, int len) {
    bs->encoded_data += len;
    return 0;
}

/* bitstream.c:22-59: leave an exhausted byte; a data FF takes its stuffed 00 with it. */
static void leave_byte(Bitstream *bs) {
    const uint8_t *p = bs->encoded_data;
    bs->encoded_data += (p[0] == 0xFF && p[1] == 0x00) ? 2 : 1;
    bs->offset = 0;
}

uint8_t next_bit(Bitstream *bs) {
    uint8_t bit = (uint8_t) ((bs->encoded_data[0] >> (7 - bs->offset)) & 1u);
    bs->offset++;
    if (bs->offset >= 8) {
        leave_byte(bs);
    }
    return bit;
}

int16_t next_bit_size(Bitstream *bs, uint8_t size) {
    int16_t v = 0;
    for (uint8_t i = 0; i < size; i++) {
        v = (int16_t) ((v << 1) + next_bit(bs));
    }
    return v;
}

static int is_rst(uint8_t b) {
    return b >= 0xD0 && b <= 0xD7;
}

/* bitstream.c:84-134.  Looks at the current (possibly partly consumed) byte and
 * up to three bytes after it. */
uint8_t check_marker(Bitstream *bs) {
    const uint8_t *p = bs->encoded_data;
    uint8_t b0 = p[0], b1 = p[1];
    if (b0 == 0xFF) {
        if (b1 == 0xD9) return 2;
        if (is_rst(b1)) return 1;
        if (b1 != 0x00) return 2;
        /* FF 00 followed by a marker: the stuffed byte is the interval's last */
        if (p[2] == 0xFF) {
            if (is_rst(p[3])) return 1;
            return 2;
        }
        return 0;
    }
    if (b1 == 0xFF) {
        uint8_t b2 = p[2];
        if (is_rst(b2)) return 1;
        if (b2 != 0x00) return 2;
    }
    return 0;
}

/* bitstream.c:142-176: move to the byte after an FF Dn that starts at the cursor
 * or one byte later; otherwise advance one byte (two over a stuffed FF 00). */
int next_byte_restart_marker(Bitstream *bs) {
    uint8_t *p = bs->encoded_data;
    bs->offset = 0;
    if (p[0] == 0xFF) {
        if (p[1] == 0x00) {
            bs->encoded_data = p + 2;
            return 2;
        }
        if (is_rst(p[1])) {
            bs->encoded_data = p + 2;
            return 1;
        }
    } else if (p[1] == 0xFF && is_rst(p[2])) {
        bs->encoded_data = p + 3;
        return 1;
    }
    bs->encoded_data = p + 1;
    return 0;
}
