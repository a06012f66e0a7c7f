#ifndef __bmp_h__
#define __bmp_h__
/* Drop-in for the reference's lib/bmp.h: 24-bit uncompressed BMP dump of three 8-bit planes (host only). */
#include <stdint.h>

typedef struct BMPData {
	unsigned int width;
	unsigned int height;
	uint8_t* red;
	uint8_t* green;
	uint8_t* blue;
} BMPData;

void write_bmp_data(const char* filepath, BMPData* data);

#endif
