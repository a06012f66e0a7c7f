/* cifar10.c -- one random 3073-byte record (label + 3 x 32 x 32 bytes) per call, rows flipped top-to-bottom so that
 * a BMP dump shows the image upright; example index from libc rand() as in the reference (lib/cifar10.c:13-32). */
#include "cifar10.h"
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

const unsigned int CIFAR10_NUM_EXAMPLES_PER_FILE = 10000;
const unsigned int CIFAR10_LINE_LENGTH = 3073;
const unsigned int CIFAR10_DATA_LENGTH = 3072;
const unsigned int CIFAR10_BATCH_FILE_SIZE = 30730000;
const unsigned int CIFAR10_NUM_PIXELS = 1024;
const unsigned int CIFAR10_EXAMPLE_DIM = 32;

void fill_random_data(int fd, uint8_t* arr) {
	unsigned int example = (unsigned int)(((float)rand() / ((float)RAND_MAX + 1)) * CIFAR10_NUM_EXAMPLES_PER_FILE);
	long want = (long)example * CIFAR10_LINE_LENGTH + 1;   /* skip the label byte */
	if (lseek(fd, want, SEEK_SET) != want) fprintf(stderr, "Error while seeking to CIFAR10 example %d (errno=%d).\n", example, errno);
	uint8_t record[3072];
	if (read(fd, record, sizeof record) != (ssize_t)sizeof record) fprintf(stderr, "Error while reading CIFAR10 example %d (errno=%d).\n", example, errno);
	const unsigned int dim = CIFAR10_EXAMPLE_DIM;
	for (unsigned int plane = 0; plane < 3; plane++)
		for (unsigned int row = 0; row < dim; row++)
			memcpy(arr + plane * CIFAR10_NUM_PIXELS + row * dim, record + plane * CIFAR10_NUM_PIXELS + (dim - 1 - row) * dim, dim);
}
