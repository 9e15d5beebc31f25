/* Drop-in name for the reference's src/decode.h: the declarations live in
 * mi355jpeg_compat.h so that a translation unit written against the reference's
 * headers (e.g. its jpeg_decoder.c) compiles unchanged against this library. */
#ifndef MI355JPEG_FWD_DECODE_H
#define MI355JPEG_FWD_DECODE_H
#include "mi355jpeg_compat.h"
#endif
