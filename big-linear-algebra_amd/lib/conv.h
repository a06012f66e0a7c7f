#ifndef __conv_h__
#define __conv_h__

/* Drop-in for the reference's lib/conv.h: same ConvData workspace struct (caller allocated, recipe
 * model/cifar_unet.c:266-292) and the same four entry points; im2col, the kernel/channel reshapes, the
 * products and col2im run on the device (include/bla.h).  See conv.c for the quirk policy. */
#include "matrix.h"

typedef struct ConvData {
	Matrix* im2col;         /* [Ho*Wo] x [k*k*Cin]  */
	Matrix* kernel_matrix;  /* [k*k*Cin] x [Cout]   */
	Matrix* product;        /* [Ho*Wo] x [Cout]     */
	Matrix* output;         /* array of Cout matrices, Ho x Wo each */
} ConvData;

void conv(Matrix* X, Matrix** kernels, ConvData* data, int in_channels, int out_channels, int stride);
void reshape_channels_matrix(Matrix* channels, Matrix* matrix);
void reshape_matrix_channels(Matrix* matrix, Matrix* channels);
void conv_ddx(Matrix* del_Y, ConvData* data, ConvData* grad_data, Matrix** del_kernels, Matrix* del_input, int in_channels, int stride);

/* Not declared by the reference header but left with external linkage by lib/conv.c:8,80,138,156; kept
 * name-compatible for harnesses that link them directly. */
void _im2col(Matrix* in, Matrix* out, int kernel_size, int in_channels, int stride);
void _col2im(Matrix* in, Matrix* out, int kernel_size, int out_channels, int stride);
void _reshape_kernels_matrix(Matrix** kernels, Matrix* matrix);
void _reshape_matrix_kernels(Matrix* matrix, Matrix** kernels);

#endif
