/* mnist_csv2.c -- in-memory MNIST CSV store + samplers with the reference's semantics (lib/mnist_csv2.c):
 * a row is label followed by 784 pixels, every value comma-terminated; the store is feature-major; sampling draws
 * from libc rand() exactly like the reference so that a fixed srand() gives the same example order. */
#include "mnist_csv2.h"
#include "csv.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { ROW = 785, PIXELS = 784 };

void mnist_csv_init(MnistCSV* csv) {                     /* reference lib/mnist_csv2.c:13-34 */
	int values = 0;
	float* flat = read_csv_contents_file(csv->file, &values);   /* closes csv->file */
	printf("MNIST CSV file contents read!\n");
	int n = values / ROW;
	csv->num_examples = n;
	csv->X = malloc((size_t)n * PIXELS * sizeof(float));
	csv->y = malloc((size_t)n * sizeof(float));
	csv->sampled = calloc(n > 0 ? n : 1, 1);
	csv->num_sampled = 0;
	for (int e = 0; e < n; e++) {
		const float* row = flat + (size_t)e * ROW;
		csv->y[e] = row[0];
		for (int px = 0; px < PIXELS; px++) csv->X[(size_t)px * n + e] = row[px + 1];
	}
	free(flat);
}

MnistExample get_random_data_replace(MnistCSV* csv) {    /* reference lib/mnist_csv2.c:36-39 */
	int n = (int)floor((float)csv->num_examples * (float)rand() / (float)RAND_MAX);
	MnistExample ex = {csv->X + n, csv->y[n], csv->num_examples};
	return ex;
}

MnistExample get_random_data_take(MnistCSV* csv) {       /* reference lib/mnist_csv2.c:41-62 */
	if (csv->num_sampled == csv->num_examples) {
		csv->num_sampled = 0;
		memset(csv->sampled, 0, csv->num_examples);
	}
	/* the n-th still-unsampled example, counted the way the reference counts it */
	int n = (int)floor((float)(csv->num_examples - csv->num_sampled) * (float)rand() / (float)RAND_MAX);
	int i = 0;
	while (i < csv->num_examples && n > 0) {
		if (!csv->sampled[i]) n--;
		i++;
	}
	csv->sampled[i] = 1;
	csv->num_sampled++;
	MnistExample ex = {csv->X + i, csv->y[i], csv->num_examples};
	return ex;
}

void visualize_digit_data(MnistExample ex) {              /* reference lib/mnist_csv2.c:64-80 */
	puts("============================");
	printf("Data for digit %f:\n", ex.y);
	for (int r = 0; r < 28; r++) {
		for (int c = 0; c < 28; c++) {
			float v = ex.X[(size_t)(r * 28 + c) * ex.num_examples];
			putchar(v < 80 ? ' ' : (v < 150 ? ':' : '#'));
		}
		putchar('\n');
	}
	puts("============================");
}
