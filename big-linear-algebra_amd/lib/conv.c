/* conv.c -- host side of the drop-in lib/conv.h.
 *
 * The reference keeps an image as an array of per-channel Matrix structs and kernels as Matrix** [F][C]
 * (lib/conv.c:9-10,138-141); here they are gathered channel by channel into one contiguous [C][H][W] /
 * [F][C][k][k] device buffer, the stages run as HIP kernels + the MFMA GEMM, and every ConvData workspace
 * the reference fills is copied back so that callers observe the same side effects.
 *
 * Quirk policy:
 *  - reshape_channels_matrix / reshape_matrix_channels keep their AS-WRITTEN directions (lib/conv.c:174-203:
 *    the former writes channels <- matrix, the latter matrix <- channels) -- they are pure index maps.
 *  - conv(): as written the last step is reshape_matrix_channels(product, output), i.e. product is overwritten
 *    from the stale output and output is never produced.  Default here: the intended composition
 *    (GEMM result reaches output, product keeps the GEMM result).  BLA_STRICT_REFERENCE=1: literal behaviour.
 *  - conv_ddx(): likewise the first step as written overwrites del_Y from the stale del_Q; default feeds del_Y
 *    into del_Q.  The reference's _col2im indexes out of bounds for stride != 1: this layer then computes the
 *    intended gradient (the adjoint of _im2col); BLA_STRICT_REFERENCE=1 refuses instead.
 */
#include "conv.h"
#include "bla_dev.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { S_X = 3, S_KERN, S_IM2COL, S_KMAT, S_PRODUCT, S_OUTPUT, S_AUX0, S_AUX1, S_AUX2 };

/* channel arrays travel as one copy per operand (bla_host_up_planes / bla_host_down_planes); kernel sets [F][C] likewise */
#define up_channels dev_up_planes
#define down_channels dev_down_planes

static bla_elem_t* up_kernels(int slot, Matrix** kernels, int f_n, int c_n) {
	const size_t per = (size_t)kernels[0][0].rows * kernels[0][0].cols;
	matrix_float_t* block = (matrix_float_t*)bla_host_pack_block(0, WORDS(per * c_n * f_n));
	for (int f = 0; f < f_n; f++)
		for (int c = 0; c < c_n; c++) memcpy(block + ((size_t)f * c_n + c) * per, kernels[f][c].data, per * sizeof(matrix_float_t));
	bla_elem_t* d = dev_up(slot, block, per * c_n * f_n);
	BLA_TRY(bla_stream_sync(NULL));
	return d;
}

static void down_kernels(Matrix** kernels, int f_n, int c_n, const bla_elem_t* d) {
	const size_t per = (size_t)kernels[0][0].rows * kernels[0][0].cols;
	matrix_float_t* block = (matrix_float_t*)bla_host_pack_block(1, WORDS(per * c_n * f_n));
	dev_down(block, d, per * c_n * f_n);
	for (int f = 0; f < f_n; f++)
		for (int c = 0; c < c_n; c++) memcpy(kernels[f][c].data, block + ((size_t)f * c_n + c) * per, per * sizeof(matrix_float_t));
}

/* reference lib/conv.c:8-77 */
void _im2col(Matrix* in, Matrix* out, int kernel_size, int in_channels, int stride) {
	bla_elem_t* dx = up_channels(S_X, in, in_channels);
	size_t n = (size_t)out->rows * out->cols;
	bla_elem_t* dout = dev_buf(S_IM2COL, n);
	BLA_TRY(DEV(im2col)(NULL, dx, dout, in[0].rows, in[0].cols, kernel_size, in_channels, stride));
	dev_down(out->data, dout, n);
}

/* reference lib/conv.c:80-135 (defined for stride 1; other strides: the adjoint of _im2col, refused under BLA_STRICT_REFERENCE=1) */
void _col2im(Matrix* in, Matrix* out, int kernel_size, int out_channels, int stride) {
	bla_elem_t* dcols = dev_up(S_IM2COL, in->data, (size_t)in->rows * in->cols);
	bla_elem_t* dout = dev_buf(S_X, (size_t)out[0].rows * out[0].cols * out_channels);
	BLA_TRY(DEV(col2im)(NULL, dcols, dout, out[0].rows, out[0].cols, kernel_size, out_channels, stride));
	down_channels(out, out_channels, dout);
}

/* reference lib/conv.c:138-153: sizes are derived exactly as there (k from kernels[0][0].rows, F = matrix->cols) */
void _reshape_kernels_matrix(Matrix** kernels, Matrix* matrix) {
	int k = kernels[0][0].rows, f_n = matrix->cols, c_n = matrix->rows / (k * k);
	bla_elem_t* dk = up_kernels(S_KERN, kernels, f_n, c_n);
	size_t n = (size_t)matrix->rows * matrix->cols;
	bla_elem_t* dm = dev_buf(S_KMAT, n);
	BLA_TRY(DEV(kernels_to_matrix)(NULL, dk, dm, f_n, c_n, k));
	dev_down(matrix->data, dm, n);
}

/* reference lib/conv.c:156-171 */
void _reshape_matrix_kernels(Matrix* matrix, Matrix** kernels) {
	int k = kernels[0][0].rows, f_n = matrix->cols, c_n = matrix->rows / (k * k);
	bla_elem_t* dm = dev_up(S_KMAT, matrix->data, (size_t)matrix->rows * matrix->cols);
	bla_elem_t* dk = dev_buf(S_KERN, (size_t)f_n * c_n * k * k);
	BLA_TRY(DEV(matrix_to_kernels)(NULL, dm, dk, f_n, c_n, k));
	down_kernels(kernels, f_n, c_n, dk);
}

/* reference lib/conv.c:174-187, direction as written: channels[c][idx] = matrix[idx*C + c] */
void reshape_channels_matrix(Matrix* channels, Matrix* matrix) {
	int c_n = matrix->cols, hw = channels[0].rows * channels[0].cols;
	bla_elem_t* dm = dev_up(S_PRODUCT, matrix->data, (size_t)hw * c_n);
	bla_elem_t* dc = dev_buf(S_OUTPUT, (size_t)hw * c_n);
	BLA_TRY(DEV(reshape_channels_matrix)(NULL, dc, dm, c_n, hw));
	down_channels(channels, c_n, dc);
}

/* reference lib/conv.c:190-203, direction as written: matrix[idx*C + c] = channels[c][idx] */
void reshape_matrix_channels(Matrix* matrix, Matrix* channels) {
	int c_n = matrix->cols, hw = channels[0].rows * channels[0].cols;
	bla_elem_t* dc = up_channels(S_OUTPUT, channels, c_n);
	bla_elem_t* dm = dev_buf(S_PRODUCT, (size_t)hw * c_n);
	BLA_TRY(DEV(reshape_matrix_channels)(NULL, dm, dc, c_n, hw));
	dev_down(matrix->data, dm, (size_t)hw * c_n);
}

/* reference lib/conv.c:205-212.  X: in_channels matrices H x W; kernels[f][c]: k x k. */
void conv(Matrix* X, Matrix** kernels, ConvData* data, int in_channels, int out_channels, int stride) {
	(void)out_channels;  /* unused by the reference too: F comes from kernel_matrix->cols */
	const int k = kernels[0][0].cols, h = X[0].rows, w = X[0].cols;
	const int f_n = data->kernel_matrix->cols;
	int ho, wo;
	BLA_TRY(bla_conv_out_hw(h, w, stride, &ho, &wo));
	const size_t hw = (size_t)ho * wo, kkc = (size_t)k * k * in_channels;
	bla_elem_t* dx = up_channels(S_X, X, in_channels);
	bla_elem_t* dk = up_kernels(S_KERN, kernels, f_n, in_channels);
	bla_elem_t* dim = dev_buf(S_IM2COL, hw * kkc);
	bla_elem_t* dkm = dev_buf(S_KMAT, kkc * f_n);
	bla_elem_t* dpr = dev_buf(S_PRODUCT, hw * f_n);
	bla_elem_t* dout = dev_buf(S_OUTPUT, hw * f_n);
	if (!bla_host_strict()) {
		BLA_TRY(DEV(conv_forward)(NULL, dx, dk, dim, dkm, dpr, dout, h, w, k, in_channels, f_n, stride));
		dev_down(data->im2col->data, dim, hw * kkc);
		dev_down(data->kernel_matrix->data, dkm, kkc * f_n);
		dev_down(data->product->data, dpr, hw * f_n);
		down_channels(data->output, f_n, dout);
	} else {
		/* literal reference: im2col, kernel reshape, product ... then product <- stale output (lib/conv.c:211) */
		BLA_TRY(DEV(im2col)(NULL, dx, dim, h, w, k, in_channels, stride));
		BLA_TRY(DEV(kernels_to_matrix)(NULL, dk, dkm, f_n, in_channels, k));
		dev_down(data->im2col->data, dim, hw * kkc);
		dev_down(data->kernel_matrix->data, dkm, kkc * f_n);
		reshape_matrix_channels(data->product, data->output);
	}
}

/* reference lib/conv.c:214-229.  Workspaces come from grad_data exactly as there: product = del_Q,
 * kernel_matrix = del_kernels_matrix, im2col = del_input_matrix.
 * Stride 1 is the only case the reference defines (its _col2im walks out of bounds otherwise).  For other strides this layer
 * computes the INTENDED gradient -- del_input = adjoint of _im2col applied to del_Q . kernel_matrix^T, with the input's size taken from
 * del_input -- unless BLA_STRICT_REFERENCE=1, which refuses. */
void conv_ddx(Matrix* del_Y, ConvData* data, ConvData* grad_data, Matrix** del_kernels, Matrix* del_input, int in_channels, int stride) {
	const int k = del_kernels[0][0].cols;
	const int ho = del_Y[0].rows, wo = del_Y[0].cols;                      /* output pixels */
	const int h = stride == 1 ? ho : del_input[0].rows, w = stride == 1 ? wo : del_input[0].cols;
	const int f_n = grad_data->product->cols;
	const size_t hw = (size_t)ho * wo, kkc = (size_t)k * k * in_channels;
	if (stride != 1 && bla_host_strict()) {
		fflush(stdout);
		fprintf(stderr, "conv_ddx: stride %d is undefined in the reference (_col2im is only valid for stride 1, lib/conv.c:80-135)\n", stride);
		exit(1);
	}
	bla_elem_t* ddy;
	if (bla_host_strict()) {
		/* literal first step (lib/conv.c:220): del_Y is overwritten from the stale del_Q */
		reshape_channels_matrix(del_Y, grad_data->product);
	}
	ddy = up_channels(S_OUTPUT, del_Y, f_n);
	bla_elem_t* dim = dev_up(S_IM2COL, data->im2col->data, hw * kkc);
	bla_elem_t* dkm = dev_up(S_KMAT, data->kernel_matrix->data, kkc * f_n);
	bla_elem_t* ddq = dev_buf(S_PRODUCT, hw * f_n);
	bla_elem_t* ddkm = dev_buf(S_AUX0, kkc * f_n);
	bla_elem_t* ddk = dev_buf(S_KERN, kkc * f_n);
	bla_elem_t* ddcol = dev_buf(S_AUX1, hw * kkc);
	bla_elem_t* ddx = dev_buf(S_X, (size_t)h * w * in_channels);
	BLA_TRY(DEV(conv_backward)(NULL, ddy, dim, dkm, ddq, ddkm, ddk, ddcol, ddx, h, w, k, in_channels, f_n, stride));
	if (!bla_host_strict()) dev_down(grad_data->product->data, ddq, hw * f_n);   /* strict: del_Q is left as it was */
	dev_down(grad_data->kernel_matrix->data, ddkm, kkc * f_n);
	dev_down(grad_data->im2col->data, ddcol, hw * kkc);
	down_kernels(del_kernels, f_n, in_channels, ddk);
	down_channels(del_input, in_channels, ddx);
}
