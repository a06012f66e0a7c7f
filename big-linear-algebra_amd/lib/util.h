/* util.h -- drop-in for the reference's lib/util.h.
 *
 * The three activation routines work in place on a host array and run on the device here (bla_relu_f32,
 * bla_softmax_cols_f32, bla_softmax_rows_f32 behind the host-coherent staging of bla_host.c); the CSV loader and the
 * Gaussian sampler are host helpers with the reference's exact semantics (libc rand_r stream included).
 * model/mnist_nn.c carries its own relu / softmax / load_matrix_from_csv and must NOT be linked with util.c -- the
 * reference's link sets keep them apart and so does this build (SURVEY 8(b) "symbol collisions"). */
#ifndef BLA_DROPIN_UTIL_H
#define BLA_DROPIN_UTIL_H

#include "matrix.h"
#include "csv.h"

/* x < 0 -> 0, in place over `count` values (lib/util.c:7-13) */
void relu(matrix_float_t* values, int count);

/* softmax down every COLUMN of a rows x cols row-major array: subtract the column maximum, exponentiate, normalise (:15-34) */
void softmax(matrix_float_t* values, int rows, int cols);

/* the same along every ROW (:36-55) */
void softmax_row_wise(matrix_float_t* values, int rows, int cols);

/* fills `dst` (rows x cols, data allocated here) from a comma-terminated CSV (:57-66) */
void load_matrix_from_csv(Matrix* dst, const char* path, int rows, int cols);

/* one N(0,1) draw by Box-Muller on rand_r(state), caching the second value of each pair exactly like :68-90 */
double random_gaussian(unsigned int* state);

#endif /* BLA_DROPIN_UTIL_H */
