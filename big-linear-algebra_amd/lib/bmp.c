/* bmp.c -- BITMAPFILEHEADER (14 bytes) + BITMAPINFOHEADER (40 bytes) + bottom-up BGR rows padded to 4 bytes,
 * the layout the reference writes (lib/bmp.c:11-101): 72 px/m resolution fields, planes = 1, 24 bpp, no compression,
 * image-size field left 0. */
#include "bmp.h"
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void put_le32(uint8_t* p, uint32_t v) { p[0] = v & 0xFF; p[1] = (v >> 8) & 0xFF; p[2] = (v >> 16) & 0xFF; p[3] = (v >> 24) & 0xFF; }

void write_bmp_data(const char* filepath, BMPData* data) {
	const uint32_t row_bytes = ((24 * data->width + 31) / 32) * 4, pixel_bytes = row_bytes * data->height;
	uint8_t head[54];
	memset(head, 0, sizeof head);
	head[0] = 'B'; head[1] = 'M';
	put_le32(head + 2, 54 + pixel_bytes);
	put_le32(head + 10, 54);                       /* offset of the pixel array */
	put_le32(head + 14, 40);                       /* info header size */
	put_le32(head + 18, data->width & 0x7FFFFFFF);
	put_le32(head + 22, data->height & 0x7FFFFFFF);
	head[26] = 1;                                  /* colour planes */
	head[28] = 24;                                 /* bits per pixel */
	put_le32(head + 38, 72); put_le32(head + 42, 72);
	head[46] = 1;                                  /* the reference stores 1 in the low byte of the palette-size field (lib/bmp.c:71) */
	uint8_t* px = calloc(pixel_bytes ? pixel_bytes : 1, 1);
	for (uint32_t r = 0; r < data->height; r++)
		for (uint32_t c = 0; c < data->width; c++) {
			uint8_t* p = px + r * row_bytes + 3 * c;
			p[0] = data->blue[r * data->width + c];
			p[1] = data->green[r * data->width + c];
			p[2] = data->red[r * data->width + c];
		}
	int fd = open(filepath, O_WRONLY | O_CREAT, 0777);
	if (fd < 0 || write(fd, head, sizeof head) != (ssize_t)sizeof head) fprintf(stderr, "Error while writing bitmap header (errno=%d).\n", errno);
	else if (write(fd, px, pixel_bytes) != (ssize_t)pixel_bytes) fprintf(stderr, "Error while writing bitmap pixel data (errno=%d).\n", errno);
	if (fd >= 0) close(fd);
	free(px);
}
