/* bmp.h -- drop-in for the reference's lib/bmp.h (host only; nothing here touches the device).
 *
 * model/cifar_unet.c dumps its 32 x 32 predictions through this interface (run(), :1936): three 8-bit colour planes
 * of width x height pixels each, row 0 at the top, written as one uncompressed 24-bit BMP.  The struct layout is the
 * interface -- callers fill its fields directly -- so field names, types and order are the reference's; everything
 * else in this file is this build's. */
#ifndef BLA_DROPIN_BMP_H
#define BLA_DROPIN_BMP_H

#include <stdint.h>

typedef struct BMPData {
	unsigned int width;    /* pixels per row */
	unsigned int height;   /* rows */
	uint8_t* red;          /* width * height bytes, row-major */
	uint8_t* green;
	uint8_t* blue;
} BMPData;

/* Writes `image` to `path` (54-byte header, BGR triplets, rows bottom-up and padded to 4 bytes, as the format demands).
 * Reference: lib/bmp.c write_bmp_data. */
void write_bmp_data(const char* path, BMPData* image);

#endif /* BLA_DROPIN_BMP_H */
