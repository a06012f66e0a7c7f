/* util.c -- host side of the drop-in lib/util.h (reference lib/util.c).  Kept in its own object, as the
 * reference does, because model/mnist_nn.c defines its own relu/softmax/load_matrix_from_csv. */
#include "util.h"
#include "bla_dev.h"
#include <math.h>
#include <stdlib.h>

void relu(matrix_float_t* data, int num) {                         /* reference lib/util.c:7-13 */
	bla_elem_t* d = dev_up(0, data, (size_t)num);
	BLA_TRY(DEV(relu)(NULL, d, (size_t)num));
	dev_down(data, d, (size_t)num);
}

void softmax(matrix_float_t* data, int rows, int cols) {            /* per column, reference lib/util.c:15-34 */
	size_t n = (size_t)rows * cols;
	bla_elem_t* d = dev_up(0, data, n);
	BLA_TRY(DEV(softmax_cols)(NULL, d, rows, cols));
	dev_down(data, d, n);
}

void softmax_row_wise(matrix_float_t* data, int rows, int cols) {   /* per row, reference lib/util.c:36-55 */
	size_t n = (size_t)rows * cols;
	bla_elem_t* d = dev_up(0, data, n);
	BLA_TRY(DEV(softmax_rows)(NULL, d, rows, cols));
	dev_down(data, d, n);
}

void load_matrix_from_csv(Matrix* m, const char* filepath, int rows, int cols) {   /* reference lib/util.c:57-65 */
	float* v = read_csv_contents(filepath);
	for (int i = 0; i < rows * cols; i++) m->data[i] = (matrix_float_t)v[i];
	free(v);
	m->rows = rows;
	m->cols = cols;
}

/* Box-Muller on libc rand(), two variates per pair of draws, second one cached in function-static state;
 * the seed argument is ignored (reference lib/util.c:68-95). */
double random_gaussian(unsigned int* seed) {
	(void)seed;
	static double spare;
	static int have_spare = 0;
	if (have_spare) {
		have_spare = 0;
		return spare;
	}
	double u1;
	do {
		u1 = (double)rand() / RAND_MAX;
	} while (u1 == 0);
	double u2 = (double)rand() / RAND_MAX;
	double r = sqrt(-2 * log(u1)), theta = 2 * 3.14159265358979323846 * u2;
	spare = r * sin(theta);
	have_spare = 1;
	return r * cos(theta);
}
