/*
 * oracle.c -- CPU restatement of the reference's dense linear-algebra hot path
 * (lib/matrix.c, lib/conv.c, lib/norm.c, lib/util.c, model/mnist_nn.c:218-315
 * of damians13/big-linear-algebra).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or call it, and only
 * as the checker.  The product (big-linear-algebra_amd/) never falls back here.
 *
 * Parity pin: the fp64 instantiation is checked bit-for-bit against the
 * reference itself (oracle/_ref/libref.so, built by oracle/Makefile straight
 * from /root/reference/lib/ *.c) in tests/test_oracle_vs_ref.py, and against
 * the golden vectors under tests/golden/ that oracle/gen_golden.py produced by
 * running that same reference build (plus main.c's printed known answers).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#define T double
#define FN(x) ora64_##x
#include "oracle_impl.inc"
#undef T
#undef FN

#define T float
#define FN(x) ora32_##x
#include "oracle_impl.inc"
#undef T
#undef FN

/* fp32-storage / fp64-accumulate product: the tight reference for the fp32
 * device GEMM's error bound (inputs are exactly the fp32 operands). */
void ora_matmul_f32_acc64(const float* a, const float* b, double* c, int a_rows, int a_cols, int b_cols) {
	for (int j = 0; j < a_rows; j++)
		for (int i = 0; i < b_cols; i++) c[(size_t)j * b_cols + i] = 0;
	for (int j = 0; j < a_rows; j++)
		for (int k = 0; k < a_cols; k++) {
			double av = a[(size_t)j * a_cols + k];
			const float* br = b + (size_t)k * b_cols;
			double* cr = c + (size_t)j * b_cols;
			for (int i = 0; i < b_cols; i++) cr[i] += av * (double)br[i];
		}
}

/* |A|.|B| in fp64: the scale of the elementwise GEMM tolerance. */
void ora_matmul_abs_f32(const float* a, const float* b, double* c, int a_rows, int a_cols, int b_cols) {
	for (int j = 0; j < a_rows; j++)
		for (int i = 0; i < b_cols; i++) c[(size_t)j * b_cols + i] = 0;
	for (int j = 0; j < a_rows; j++)
		for (int k = 0; k < a_cols; k++) {
			double av = fabs((double)a[(size_t)j * a_cols + k]);
			const float* br = b + (size_t)k * b_cols;
			double* cr = c + (size_t)j * b_cols;
			for (int i = 0; i < b_cols; i++) cr[i] += av * fabs((double)br[i]);
		}
}
