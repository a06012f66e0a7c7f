#ifndef __matrix_h__
#define __matrix_h__

/*
 * Drop-in replacement for the reference's lib/matrix.h (same include guard, same type and function
 * names, same argument order and ownership rules) whose arithmetic runs on an MI355X through the
 * C-ABI of include/bla.h.  Programs written against the reference header (model/mnist_nn.c,
 * model/cifar_unet.c, main.c, ...) compile against this one unchanged.
 *
 * Element type: the reference declares `typedef double matrix_float_t` (lib/matrix.h:4); this build
 * computes in fp32 on the device and therefore declares float (DESIGN.md "element type").  Every
 * reference program except the already-stale model/mnist.c compiles warning-free with float, and the
 * float typedef is what lib/layer.h's `void (*)(float*, int)` activation pointers actually require.
 * Compiled with -DBLA_FP64 the type is double like the reference's and the functions of THIS header run in fp64 on the device (MFMA f64 GEMM;
 * lib/libbla_host_f64.so) -- for comparing against the reference's CPU results to 1e-12; conv.h / norm.h / util.h / layer.h are fp32 only.
 *
 * Every call is host-coherent like the reference: operands are staged to HBM, the kernel runs, results
 * are copied back before the function returns; no pointer is retained across calls.  There is no CPU
 * compute path: without a gfx950 device the first call prints the error and exits with status 1.
 */
#ifdef BLA_FP64
typedef double matrix_float_t;   /* the reference's own element type (lib/matrix.h:4): the <= 1e-12 comparison build, matrix.h functions only */
#else
typedef float matrix_float_t;
#endif

/* Data is stored in row major order (reference lib/matrix.h:6-11) */
typedef struct Matrix {
	int rows;
	int cols;
	matrix_float_t* data;
} Matrix;

/* ---- lifecycle (lib/matrix.c:6-32): plain malloc / free, released by the caller ---------------------------------------- */
struct Matrix* make_matrix(int rows, int cols, matrix_float_t* values);   /* adopts `values` (no copy) */
struct Matrix* clone_matrix(struct Matrix src);                           /* deep copy */
void free_matrix(struct Matrix* mat);                                     /* data, then the struct */
void free_matrix_data(struct Matrix* mat);                                /* data only */

/* ---- products (lib/matrix.c:35-57): fp32 MFMA GEMM on the device ----------------------------------------------------- */
/* out = lhs . rhs into a new matrix; on a shape mismatch prints the reference's message and exit(1)s, like :36-39 */
struct Matrix* matrix_multiply(struct Matrix lhs, struct Matrix rhs);
/* out[j * rhs->cols + i] = sum_k lhs[j][k] rhs[k][i]; no checks, `out`'s own dimensions are ignored (as in :47-57) */
void matrix_multiply_inplace(Matrix* lhs, Matrix* rhs, Matrix* out);

/* ---- elementwise, in place (lib/matrix.c:59-69,95-103) ---------------------------------------------------------------- */
void matrix_scale(struct Matrix* mat, matrix_float_t factor);
void matrix_add(struct Matrix* acc, struct Matrix* addend);                       /* acc += addend over acc's size, unchecked */
void matrix_multiply_elementwise(struct Matrix* acc, struct Matrix* other);       /* Hadamard; exit(1) on a shape mismatch (:96-99) */

/* ---- broadcasts (lib/matrix.c:189-205) -------------------------------------------------------------------------------- */
void matrix_add_tile_columns(struct Matrix* acc, struct Matrix* tile);   /* acc[r][c] += tile[r][c % tile->cols] (a bias column when tile is n x 1) */
void matrix_add_tile_rows(struct Matrix* acc, struct Matrix* tile);      /* acc[r][c] += tile[0][c] */

/* ---- layout and reductions (lib/matrix.c:105-185) ---------------------------------------------------------------------- */
void matrix_transpose(struct Matrix* mat);                  /* in place: swaps rows / cols and rewrites the data */
struct Matrix* matrix_row_sum(struct Matrix mat);           /* 1 x cols: sums down every column */
struct Matrix* matrix_col_sum(struct Matrix mat);           /* rows x 1 AS WRITTEN at :138-148 (flat windows; see matrix.c and SURVEY Q2) */
matrix_float_t frobenius_norm(struct Matrix mat);
matrix_float_t max_value(struct Matrix mat);
void matrix_z_score_normalize(Matrix* mat);                 /* (x - mean) / sqrtf(E[x^2] - mean^2) */

/* ---- printing (lib/matrix.c:71-93), host ------------------------------------------------------------------------------- */
void print_matrix(struct Matrix mat);
void print_matrix_dim(struct Matrix mat);

#endif
