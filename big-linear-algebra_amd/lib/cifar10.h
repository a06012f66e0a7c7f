#ifndef __cifar10_h__
#define __cifar10_h__
/* Drop-in for the reference's lib/cifar10.h: random-access reader of the CIFAR-10 binary batches (host only). */
#include <stdint.h>
#include <stdio.h>

extern const unsigned int CIFAR10_NUM_EXAMPLES_PER_FILE;
extern const unsigned int CIFAR10_LINE_LENGTH;
extern const unsigned int CIFAR10_DATA_LENGTH;
extern const unsigned int CIFAR10_BATCH_FILE_SIZE;
extern const unsigned int CIFAR10_NUM_PIXELS;
extern const unsigned int CIFAR10_EXAMPLE_DIM;

/* Fill arr with 3072 bytes of one random example: 1024 red, 1024 green, 1024 blue, rows flipped vertically */
void fill_random_data(int fd, uint8_t* arr);

#endif
