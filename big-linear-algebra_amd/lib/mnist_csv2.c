/* mnist_csv2.c -- in-memory MNIST CSV store + samplers with the reference's semantics (lib/mnist_csv2.c):
 * a row is label followed by 784 pixels, every value comma-terminated; the store is feature-major; sampling draws
 * from libc rand() exactly like the reference so that a fixed srand() gives the same example order. */
#include "mnist_csv2.h"
#include "csv.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { ROW = 785, PIXELS = 784 };

void mnist_csv_init(MnistCSV* store) {                   /* reference lib/mnist_csv2.c:13-34 */
	int count = 0;
	float* flat = read_csv_contents_file(store->file, &count);   /* closes store->file */
	printf("MNIST CSV file contents read!\n");
	const int examples = count / ROW;
	store->num_examples = examples;
	store->num_sampled = 0;
	store->sampled = calloc(examples > 0 ? examples : 1, 1);
	store->y = malloc((size_t)examples * sizeof(float));
	store->X = malloc((size_t)examples * PIXELS * sizeof(float));
	/* file order is example-major (label, then its pixels); the store is feature-major */
	const float* cursor = flat;
	for (int e = 0; e < examples; e++, cursor += ROW) {
		store->y[e] = cursor[0];
		float* column = store->X + e;
		for (int px = 0; px < PIXELS; px++, column += examples) *column = cursor[1 + px];
	}
	free(flat);
}

/* both samplers scale rand() by 1 / RAND_MAX in float and floor, exactly as the reference does (so RAND_MAX itself maps one past the range --
 * a quirk kept for identical streams) */
static int draw_below(int bound) { return (int)floor((float)bound * (float)rand() / (float)RAND_MAX); }

MnistExample get_random_data_replace(MnistCSV* store) {  /* reference lib/mnist_csv2.c:36-39 */
	const int pick = draw_below(store->num_examples);
	MnistExample out = {store->X + pick, store->y[pick], store->num_examples};
	return out;
}

MnistExample get_random_data_take(MnistCSV* store) {     /* reference lib/mnist_csv2.c:41-62 */
	if (store->num_sampled == store->num_examples) {        /* everything taken: start over */
		memset(store->sampled, 0, store->num_examples);
		store->num_sampled = 0;
	}
	/* skip `skip` still-untaken examples, counted the way the reference counts them, and take where the walk stops */
	int skip = draw_below(store->num_examples - store->num_sampled);
	int at = 0;
	for (; at < store->num_examples && skip > 0; at++)
		if (!store->sampled[at]) skip--;
	if (at >= store->num_examples) at = store->num_examples - 1;   /* the reference marks one past its array here (only when rand() rounds to RAND_MAX in float) */
	store->sampled[at] = 1;
	store->num_sampled++;
	MnistExample out = {store->X + at, store->y[at], store->num_examples};
	return out;
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
