#ifndef __norm_h__
#define __norm_h__

/* Drop-in for the reference's lib/norm.h.  Argument orders are the reference's (note that they differ
 * between forward and backward): group_norm(in, out, stdevs, means, ...), group_norm_ddx(source, dest, data,
 * means, stdevs, ...).  Semantics as written there: epsilon is the integer 0 and `stdevs` receives
 * the VARIANCE; out = (x - mean) / variance. */
#include "matrix.h"

void group_norm(Matrix* in, Matrix* out, matrix_float_t* stdevs, matrix_float_t* means, int channels, int group_size);
void group_norm_ddx(Matrix* source, Matrix* dest, Matrix* data, matrix_float_t* means, matrix_float_t* stdevs, int channels, int group_size);

#endif
