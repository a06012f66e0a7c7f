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
 *
 * Every call is host-coherent like the reference: operands are staged to HBM, the kernel runs, results
 * are copied back before the function returns; no pointer is retained across calls.  There is no CPU
 * compute path: without a gfx950 device the first call prints the error and exits with status 1.
 */
typedef float matrix_float_t;

/* Data is stored in row major order (reference lib/matrix.h:6-11) */
typedef struct Matrix {
	int rows;
	int cols;
	matrix_float_t* data;
} Matrix;

struct Matrix* make_matrix(int rows, int cols, matrix_float_t* data);
struct Matrix* clone_matrix(struct Matrix m);
void free_matrix_data(struct Matrix* m);
void free_matrix(struct Matrix* m);
struct Matrix* matrix_multiply(struct Matrix a, struct Matrix b);
void matrix_scale(struct Matrix* m, matrix_float_t f);
void matrix_add(struct Matrix* a, struct Matrix* b);
void print_matrix(struct Matrix m);
void print_matrix_dim(struct Matrix m);
void matrix_multiply_elementwise(struct Matrix* a, struct Matrix* b);
void matrix_transpose(struct Matrix* m);
struct Matrix* matrix_row_sum(struct Matrix m);
struct Matrix* matrix_col_sum(struct Matrix m);
matrix_float_t frobenius_norm(struct Matrix m);
matrix_float_t max_value(struct Matrix m);
void matrix_z_score_normalize(Matrix* m);
void matrix_add_tile_columns(struct Matrix* a, struct Matrix* b);
void matrix_add_tile_rows(struct Matrix* a, struct Matrix* b);

void matrix_multiply_inplace(Matrix* a, Matrix* b, Matrix* c);

#endif
