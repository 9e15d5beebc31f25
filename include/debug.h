/* Drop-in name for the reference's src/debug.h (debug.h:1-8): a compile-time switch
 * and a printf-style macro that writes to stderr only when the switch is on.  The
 * reference's jpeg_decoder.c includes it (jpeg_decoder.c:15) and calls debug_print
 * (jpeg_decoder.c:48).  The library's own diagnostics do not use this macro: its
 * print_* helpers are switched at run time by the MJ_DEBUG environment variable. */
#ifndef MI355JPEG_FWD_DEBUG_H
#define MI355JPEG_FWD_DEBUG_H
#include <stdio.h>

#ifndef DEBUG
#define DEBUG 0
#endif

#define debug_print(...)                  \
    do {                                  \
        if (DEBUG) {                      \
            fprintf(stderr, __VA_ARGS__); \
        }                                 \
    } while (0)

#endif
