/*
 * jpeg_decoder_main.c -- headless stand-in for the reference's viewer.
 *
 * Same call sequence as the reference's main() (jpeg_decoder.c:44-104): map the
 * file, decode_jpeg_buffer(), pick y_rgb / ycbcr_rgb by component count, convert
 * into an RGB24 surface, free.  Instead of opening an SDL window it prints the
 * frame size and an FNV-1a checksum of the pixels, and optionally writes a PPM.
 *
 *   jpeg_decoder <file.jpg> [-o out.ppm] [-n repeats]
 */
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* the include list of the reference's viewer, name for name (jpeg_decoder.c:14-21) */
#include "colour_conversion.h"
#include "debug.h"
#include "decode.h"
#include "frame_header.h"
#include "huff_table.h"
#include "restart_interval.h"
#include "scan_header.h"

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

int main(int argc, char **argv) {
    const char *path = NULL, *out = NULL;
    int repeats = 1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-n") && i + 1 < argc) repeats = atoi(argv[++i]);
        else path = argv[i];
    }
    if (!path) {
        fprintf(stderr, "Need file path\n");
        return 1;
    }
    int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) {
        perror(path);
        return 1;
    }
    uint8_t *buf = (uint8_t *) mmap(NULL, (size_t) st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (buf == MAP_FAILED) {
        perror("mmap");
        return 1;
    }
    debug_print("length: %lu\n", (unsigned long) st.st_size);   /* jpeg_decoder.c:48 */
    uint8_t *pixels = NULL;
    uint16_t width = 0, height = 0, pitch = 0;
    double t0 = now_s();
    for (int r = 0; r < repeats; r++) {
        Image *img = NULL;
        FrameHeader fh;
        fh.process = -1;
        if (decode_jpeg_buffer(buf, (size_t) st.st_size, &fh, &img) != 0) {
            fprintf(stderr, "Decoding jpeg buffer failed\n");
            return 1;
        }
        width = fh.X;
        height = fh.Y;
        /* rows as an SDL RGB24 surface lays them out: padded to four bytes (jpeg_decoder.c:62-77) */
        pitch = (uint16_t) ((3u * width + 3u) & ~3u);
        if (!pixels) pixels = (uint8_t *) calloc((size_t) pitch * height, 1);
        int rc = -1;
        if (fh.ncs == 1) rc = y_rgb(&fh, img, pixels, width, height, pitch);
        else if (fh.ncs == 3) rc = ycbcr_rgb(&fh, img, pixels, width, height, pitch);
        else if (fh.ncs == 4) rc = yccb_rgb(&fh, img, pixels, width, height, pitch);
        free_img(img);
        free_frame_header(&fh);
        if (rc != 0) {
            fprintf(stderr, "Colour conversion failed\n");
            return 1;
        }
    }
    double dt = now_s() - t0;
    uint64_t h = 1469598103934665603ull;
    for (size_t y = 0; y < height; y++) {      /* the pixels, not the row padding */
        for (size_t i = 0; i < (size_t) 3 * width; i++) h = (h ^ pixels[y * pitch + i]) * 1099511628211ull;
    }
    printf("%s: %ux%u rgb24 fnv1a=%016llx  %.3f ms/decode\n", path, width, height, (unsigned long long) h,
           1e3 * dt / repeats);
    if (out) {
        FILE *f = fopen(out, "wb");
        if (!f) {
            perror(out);
            return 1;
        }
        fprintf(f, "P6\n%u %u\n255\n", width, height);
        for (size_t y = 0; y < height; y++) fwrite(pixels + y * pitch, 1, (size_t) 3 * width, f);
        fclose(f);
    }
    free(pixels);
    munmap(buf, (size_t) st.st_size);
    return 0;
}
