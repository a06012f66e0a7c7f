/* cifar10.c -- host-side reader for the drop-in cifar10.h.
 *
 * One uniformly drawn record per call.  The record index comes from libc rand() scaled by 1 / (RAND_MAX + 1) in float, as in the
 * reference (lib/cifar10.c:13-32), so a fixed srand() reproduces the reference's example order; the label byte is skipped and every
 * colour plane is delivered with its rows reversed (the BMP writer stores rows bottom-up). */
#include "cifar10.h"
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

enum { kSide = 32, kPlane = kSide * kSide, kImage = 3 * kPlane, kRecord = kImage + 1, kRecords = 10000 };

/* the interface constants of cifar10.h */
const unsigned int CIFAR10_EXAMPLE_DIM = kSide;
const unsigned int CIFAR10_NUM_PIXELS = kPlane;
const unsigned int CIFAR10_DATA_LENGTH = kImage;
const unsigned int CIFAR10_LINE_LENGTH = kRecord;
const unsigned int CIFAR10_NUM_EXAMPLES_PER_FILE = kRecords;
const unsigned int CIFAR10_BATCH_FILE_SIZE = kRecords * kRecord;

void fill_random_data(int fd, uint8_t* pixels) {
	const float unit = (float)rand() / ((float)RAND_MAX + 1);          /* [0, 1) */
	const unsigned int pick = (unsigned int)(unit * kRecords);
	const long at = (long)pick * kRecord + 1;                           /* + 1: past the label */
	uint8_t raw[kImage];
	if (lseek(fd, at, SEEK_SET) != at)
		fprintf(stderr, "Error while seeking to CIFAR10 example %d (errno=%d).\n", pick, errno);
	if (read(fd, raw, sizeof raw) != (ssize_t)sizeof raw)
		fprintf(stderr, "Error while reading CIFAR10 example %d (errno=%d).\n", pick, errno);
	for (int colour = 0; colour < 3; colour++) {
		const uint8_t* from = raw + colour * kPlane;
		uint8_t* to = pixels + colour * kPlane;
		for (int y = 0; y < kSide; y++) memcpy(to + y * kSide, from + (kSide - 1 - y) * kSide, kSide);   /* top row last */
	}
}
