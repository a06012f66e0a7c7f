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

void mnist_csv_init(MnistCSV* csv);
MnistExample get_random_data_replace(MnistCSV* csv);   /* uniform, with replacement */
MnistExample get_random_data_take(MnistCSV* csv);      /* uniform, without replacement (restarts when exhausted) */
void visualize_digit_data(MnistExample ex);

#endif
