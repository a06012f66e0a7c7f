/* matrix.c -- host side of the drop-in lib/matrix.h: same functions, same argument meaning, same
 * ownership and error behaviour as the reference's lib/matrix.c, with the loops replaced by HIP kernels
 * reached through include/bla.h.  Each function cites the reference lines it stands in for. */
#include "matrix.h"
#include "bla_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* Element type on the device = matrix_float_t: float by default, double under -DBLA_FP64 (the reference's own type, lib/matrix.h:4).  The
 * staging buffers of bla_host.c are counted in floats: WORDS() converts. */
#ifdef BLA_FP64
typedef double bla_elem_t;
#define DEV(name) bla_##name##_f64
#else
typedef float bla_elem_t;
#define DEV(name) bla_##name##_f32
#endif
#define WORDS(n) ((size_t)(n) * (sizeof(matrix_float_t) / sizeof(float)))
static bla_elem_t* up(int slot, const matrix_float_t* h, size_t n) { return (bla_elem_t*)bla_host_up(slot, (const float*)h, WORDS(n)); }
static bla_elem_t* buf(int slot, size_t n) { return (bla_elem_t*)bla_host_buf(slot, WORDS(n)); }
static void down(matrix_float_t* h, const bla_elem_t* d, size_t n) { bla_host_down((float*)h, (const float*)d, WORDS(n)); }
static void dev_gemm(int m, int n, int k, const bla_elem_t* a, const bla_elem_t* b, bla_elem_t* c) {
#ifdef BLA_FP64
	BLA_TRY(bla_gemm_f64(NULL, 0, 0, m, n, k, a, k, b, n, c, n, 1.0, 0.0));
#else
	BLA_TRY(bla_gemm_f32(NULL, 0, 0, m, n, k, a, k, b, n, c, n, NULL));
#endif
}

/* reference lib/matrix.c:6-12: heap struct around caller-owned data */
struct Matrix* make_matrix(int rows, int cols, matrix_float_t* data) {
	struct Matrix* m = malloc(sizeof *m);
	m->rows = rows;
	m->cols = cols;
	m->data = data;
	return m;
}

/* reference lib/matrix.c:14-21: deep copy (host memory management, no arithmetic) */
struct Matrix* clone_matrix(struct Matrix m) {
	size_t n = (size_t)m.rows * m.cols;
	matrix_float_t* data = malloc(n * sizeof(matrix_float_t));
	memcpy(data, m.data, n * sizeof(matrix_float_t));
	return make_matrix(m.rows, m.cols, data);
}

void free_matrix_data(struct Matrix* m) { free(m->data); }      /* reference lib/matrix.c:24-26 */

void free_matrix(struct Matrix* m) {                              /* reference lib/matrix.c:29-32 */
	free_matrix_data(m);
	free(m);
}

/* reference lib/matrix.c:35-44: conformance check with the reference's exact message + exit(1), result is a
 * malloc'd struct with malloc'd data owned by the caller */
struct Matrix* matrix_multiply(struct Matrix a, struct Matrix b) {
	if (a.cols != b.rows) {
		printf("Attempted to multiply %dx%d matrix by %dx%d matrix, exiting\n", a.rows, a.cols, b.rows, b.cols);
		exit(1);
	}
	matrix_float_t* data = malloc((size_t)b.cols * a.rows * sizeof(matrix_float_t));
	struct Matrix* m = make_matrix(a.rows, b.cols, data);
	matrix_multiply_inplace(&a, &b, m);
	return m;
}

/* reference lib/matrix.c:47-57: c = a @ b, no shape checks, c's own dims ignored (output stride is b->cols).
 * MFMA GEMM on the device (fp32, or fp64 under -DBLA_FP64). */
void matrix_multiply_inplace(Matrix* a, Matrix* b, Matrix* c) {
	const int m = a->rows, k = a->cols, n = b->cols;
	if (m <= 0 || n <= 0) return;
	bla_elem_t* da = up(0, a->data, (size_t)m * k);
	bla_elem_t* db = up(1, b->data, (size_t)k * n);
	bla_elem_t* dc = buf(2, (size_t)m * n);
	dev_gemm(m, n, k, da, db, dc);
	down(c->data, dc, (size_t)m * n);
}

void matrix_scale(struct Matrix* m, matrix_float_t f) {          /* reference lib/matrix.c:59-63 */
	size_t n = (size_t)m->rows * m->cols;
	bla_elem_t* d = up(0, m->data, n);
	BLA_TRY(DEV(scale)(NULL, d, n, f));
	down(m->data, d, n);
}

void matrix_add(struct Matrix* a, struct Matrix* b) {            /* reference lib/matrix.c:65-69: a's size is trusted */
	size_t n = (size_t)a->rows * a->cols;
	bla_elem_t* da = up(0, a->data, n);
	bla_elem_t* db = up(1, b->data, n);
	BLA_TRY(DEV(add)(NULL, da, db, n));
	down(a->data, da, n);
}

/* reference lib/matrix.c:71-89.  Note `< 0.01` also sends negative values to %.2e. */
void print_matrix(struct Matrix m) {
	printf("%d x %d matrix\n", m.rows, m.cols);
	for (int i = 0; i < m.rows * m.cols; i++) {
		if (i % m.cols == 0) printf("[ ");
		if (m.data[i] == 0) printf("0 ");
		else if (m.data[i] < 0.01) printf("%.2e ", m.data[i]);
		else printf("%.2f ", m.data[i]);
		if ((i + 1) % m.cols == 0) printf("]\n");
	}
	printf("\n");
}

void print_matrix_dim(struct Matrix m) { printf("%d x %d matrix\n", m.rows, m.cols); }   /* reference lib/matrix.c:91-93 */

void matrix_multiply_elementwise(struct Matrix* a, struct Matrix* b) {   /* reference lib/matrix.c:95-103 */
	if (a->cols != b->cols || a->rows != b->rows) {
		printf("Attempted to multiply elements of %dx%d matrix by %dx%d matrix, exiting\n", a->rows, a->cols, b->rows, b->cols);
		exit(1);
	}
	size_t n = (size_t)a->rows * a->cols;
	bla_elem_t* da = up(0, a->data, n);
	bla_elem_t* db = up(1, b->data, n);
	BLA_TRY(DEV(hadamard)(NULL, da, db, n));
	down(a->data, da, n);
}

/* reference lib/matrix.c:105-118: dims swapped in place, data rewritten (LDS-tiled transpose on the device) */
void matrix_transpose(struct Matrix* m) {
	const int r = m->rows, c = m->cols;
	size_t n = (size_t)r * c;
	bla_elem_t* din = up(0, m->data, n);
	bla_elem_t* dout = buf(1, n);
	BLA_TRY(DEV(transpose)(NULL, din, dout, r, c));
	down(m->data, dout, n);
	m->rows = c;
	m->cols = r;
}

struct Matrix* matrix_row_sum(struct Matrix m) {                  /* reference lib/matrix.c:123-133 -> 1 x cols */
	matrix_float_t* data = malloc((size_t)m.cols * sizeof(matrix_float_t));
	struct Matrix* out = make_matrix(1, m.cols, data);
	bla_elem_t* d = up(0, m.data, (size_t)m.rows * m.cols);
	bla_elem_t* o = buf(1, (size_t)m.cols);
	BLA_TRY(DEV(row_sum)(NULL, d, m.rows, m.cols, o));
	down(data, o, (size_t)m.cols);
	return out;
}

/* reference lib/matrix.c:138-148 -> rows x 1.  As written the reference sums the flat window
 * data[i*rows .. i*rows+cols), which is in bounds only for rows <= cols.  Policy:
 * reproduce the as-written result wherever it is defined; where the reference reads out of bounds
 * (rows > cols) give the documented intent (true row sums), or refuse under BLA_STRICT_REFERENCE=1. */
struct Matrix* matrix_col_sum(struct Matrix m) {
	matrix_float_t* data = malloc((size_t)m.rows * sizeof(matrix_float_t));
	struct Matrix* out = make_matrix(m.rows, 1, data);
	bla_elem_t* d = up(0, m.data, (size_t)m.rows * m.cols);
	bla_elem_t* o = buf(1, (size_t)m.rows);
	int mode = m.rows <= m.cols ? BLA_COLSUM_AS_WRITTEN : BLA_COLSUM_INTENDED;
	if (bla_host_strict()) mode = BLA_COLSUM_AS_WRITTEN;
	BLA_TRY(DEV(col_sum)(NULL, d, m.rows, m.cols, o, mode));
	down(data, o, (size_t)m.rows);
	return out;
}

static matrix_float_t reduce_scalar(struct Matrix m, bla_status (*fn)(void*, const bla_elem_t*, size_t, bla_elem_t*)) {
	size_t n = (size_t)m.rows * m.cols;
	bla_elem_t* d = up(0, m.data, n);
	bla_elem_t* o = buf(1, 4);
	BLA_TRY(fn(NULL, d, n, o));
	matrix_float_t r;
	down(&r, o, 1);
	return r;
}

matrix_float_t frobenius_norm(struct Matrix m) { return reduce_scalar(m, DEV(frobenius)); }   /* reference lib/matrix.c:150-158 */
matrix_float_t max_value(struct Matrix m) { return reduce_scalar(m, DEV(max)); }               /* reference lib/matrix.c:160-168 */

void matrix_z_score_normalize(Matrix* m) {                        /* reference lib/matrix.c:170-185 */
	size_t n = (size_t)m->rows * m->cols;
	bla_elem_t* d = up(0, m->data, n);
	BLA_TRY(DEV(zscore)(NULL, d, n));
	down(m->data, d, n);
}

void matrix_add_tile_columns(struct Matrix* a, struct Matrix* b) {   /* reference lib/matrix.c:189-195 */
	size_t n = (size_t)a->rows * a->cols;
	bla_elem_t* da = up(0, a->data, n);
	bla_elem_t* db = up(1, b->data, (size_t)a->rows * b->cols);
	BLA_TRY(DEV(add_tile_columns)(NULL, da, a->rows, a->cols, db, b->cols));
	down(a->data, da, n);
}

void matrix_add_tile_rows(struct Matrix* a, struct Matrix* b) {      /* reference lib/matrix.c:199-205 */
	size_t n = (size_t)a->rows * a->cols;
	bla_elem_t* da = up(0, a->data, n);
	bla_elem_t* db = up(1, b->data, (size_t)a->cols);
	BLA_TRY(DEV(add_tile_rows)(NULL, da, a->rows, a->cols, db));
	down(a->data, da, n);
}
