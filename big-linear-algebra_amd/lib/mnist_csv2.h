#ifndef __mnist_csv_h__
#define __mnist_csv_h__
/* Drop-in for the reference's lib/mnist_csv2.h (same include guard and type names as there -- it is mutually
 * exclusive with the legacy streaming reader lib/mnist_csv.h, which this build does not ship).  Host-side dataset
 * plumbing: nothing here touches the device. */
#include <stdio.h>

typedef struct MnistCSV {
	FILE* file;
	float* X;          /* feature-major: X[pixel * num_examples + example] */
	float* y;          /* labels */
	int num_examples;
	int num_sampled;
	char* sampled;
} MnistCSV;

typedef struct MnistExample {
	float* X;          /* points at the example's first pixel; consecutive pixels are num_examples apart */
	float y;
	int num_examples;
} MnistExample;

/* reads the whole file behind store->file (label + 784 pixels per row, every value comma-terminated) into the feature-major arrays
 * and closes it; prints the reference's progress line (lib/mnist_csv2.c:13-34) */
void mnist_csv_init(MnistCSV* store);
/* one example drawn uniformly WITH replacement from libc rand() (:36-39) */
MnistExample get_random_data_replace(MnistCSV* store);
/* one example drawn from those not yet taken; starts over when all were taken (:41-62) */
MnistExample get_random_data_take(MnistCSV* store);
/* 28 x 28 ASCII rendering (' ' < 80 <= ':' < 150 <= '#') between two rules (:64-80) */
void visualize_digit_data(MnistExample example);

#endif
