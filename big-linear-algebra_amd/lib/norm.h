#ifndef __norm_h__
#define __norm_h__

/* Drop-in for the reference's lib/norm.h.  Argument orders are the reference's (note that they differ
 * between forward and backward): group_norm(in, out, stdevs, means, ...), group_norm_ddx(source, dest, data,
 * means, stdevs, ...).  Semantics as written there: epsilon is the integer 0 and `stdevs` receives
 * the VARIANCE; out = (x - mean) / variance. */
#include "matrix.h"

/* `input` / `result`: arrays of `channels` matrices (H x W each); one mean and one "stdev" slot per group of `group_size` channels
 * (the last group may be shorter).  lib/norm.c:5-50 */
void group_norm(Matrix* input, Matrix* result, matrix_float_t* group_stdevs, matrix_float_t* group_means, int channels, int group_size);

/* gradient w.r.t. the input: `upstream` = d loss / d result, `forward_input` = what group_norm was given; note means BEFORE stdevs here.
 * lib/norm.c:52-93 */
void group_norm_ddx(Matrix* upstream, Matrix* grad_input, Matrix* forward_input, matrix_float_t* group_means, matrix_float_t* group_stdevs,
                    int channels, int group_size);

#endif
