#ifndef __util_h__
#define __util_h__

/* Drop-in for the reference's lib/util.h: relu / softmax / softmax_row_wise run on the device,
 * load_matrix_from_csv and random_gaussian are host helpers with the reference's semantics. */
#include "matrix.h"
#include "csv.h"

void relu(matrix_float_t* data, int num);
void softmax(matrix_float_t* data, int rows, int cols);
void softmax_row_wise(matrix_float_t* data, int rows, int cols);
void load_matrix_from_csv(Matrix* m, const char* filepath, int rows, int cols);
double random_gaussian(unsigned int* seed);

#endif
