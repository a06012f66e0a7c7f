// bla_unet.hip -- the elementwise / resize glue ops that sit on either side of every conv() in the reference's U-Net
// (model/cifar_unet.c:235-253,1024-1097,1168-1178,1229-1259), SURVEY 8(f) rank 1.  All HBM-bound, one pass each;
// channel arrays are contiguous [C][H*W] buffers.  The adds of _nearest_neighbours_ddx are done in the reference's
// order (row-major over the source block), so the fp32 result is bit-identical to the fp32 oracle.
#include "bla_internal.h"

namespace bla {
constexpr int kT = 256;
static inline unsigned blocks_for(size_t n) {
	size_t b = (n + kT - 1) / kT;
	return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// multi_channel_relu_ddx, model/cifar_unet.c:241-253 (dest may alias source, as in _backward_resnet :1203)
__global__ void __launch_bounds__(kT) relu_mask_kernel(float* dest, const float* source, const float* __restrict__ relu_result, size_t n) {
	for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) dest[i] = relu_result[i] <= 0.f ? 0.f : source[i];
}

// _dropout, model/cifar_unet.c:1032-1042 with the rand() draws supplied by the host (drop[i] != 0 <=> draw < DROPOUT_RATE)
__global__ void __launch_bounds__(kT) dropout_kernel(const float* __restrict__ x, float* __restrict__ y, const unsigned char* __restrict__ drop, size_t n) {
	for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) y[i] = drop[i] ? 0.f : x[i];
}

// _dropout_mask, model/cifar_unet.c:1168-1178
__global__ void __launch_bounds__(kT) dropout_mask_kernel(float* __restrict__ x, const float* __restrict__ dropout_result, size_t n) {
	for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT)
		if (dropout_result[i] == 0.f) x[i] = 0.f;
}

// _nearest_neighbours, model/cifar_unet.c:1074-1086
__global__ void __launch_bounds__(kT) nearest_kernel(const float* __restrict__ in, float* __restrict__ out, int channels, int in_w, int in_hw, int out_h,
                                                      int out_w, int scale) {
	size_t n = (size_t)channels * out_h * out_w;
	for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < n; e += (size_t)gridDim.x * kT) {
		int j = (int)(e % out_w), i = (int)((e / out_w) % out_h), c = (int)(e / ((size_t)out_w * out_h));
		out[e] = in[(size_t)c * in_hw + (i / scale) * in_w + j / scale];
	}
}

// _nearest_neighbours_ddx, model/cifar_unet.c:1229-1244, gather form: each destination pixel sums its scale x scale
// source block in the order the reference's scatter visits it (i ascending, then j ascending)
__global__ void __launch_bounds__(kT) nearest_ddx_kernel(const float* __restrict__ src, float* __restrict__ dest, int channels, int sh, int sw, int dh,
                                                          int dw, int scale) {
	size_t n = (size_t)channels * dh * dw;
	for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < n; e += (size_t)gridDim.x * kT) {
		int x = (int)(e % dw), y = (int)((e / dw) % dh), c = (int)(e / ((size_t)dw * dh));
		float acc = 0.f;
		for (int i = y * scale; i < min(sh, (y + 1) * scale); i++)
			for (int j = x * scale; j < min(sw, (x + 1) * scale); j++) acc += src[((size_t)c * sh + i) * sw + j];
		dest[e] = acc;
	}
}

// _softmax_ddx, model/cifar_unet.c:1246-1259: one wave per row; out = s * (g - <s, g>)
__global__ void __launch_bounds__(kT) softmax_ddx_kernel(const float* __restrict__ s, const float* __restrict__ g, float* __restrict__ out, int rows, int dim) {
	int r = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= rows) return;
	const float* sr = s + (size_t)r * dim; const float* gr = g + (size_t)r * dim;
	double dot = 0;
	for (int j = lane; j < dim; j += 64) dot += (double)sr[j] * gr[j];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
	float d = (float)__shfl(dot, 0, 64);
	for (int j = lane; j < dim; j += 64) out[(size_t)r * dim + j] = sr[j] * (gr[j] - d);
}
}  // namespace bla

using namespace bla;

#define BLA_ENTER()                       \
	bla_status st = require_ready();      \
	if (st) return st;

extern "C" {

bla_status bla_relu_mask_f32(void* stream, float* d_dest, const float* d_source, const float* d_relu_result, size_t n) {
	BLA_ENTER();
	if (n == 0) return BLA_OK;
	BLA_REQUIRE(d_dest && d_source && d_relu_result, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks_for(n)), dim3(kT), 0, pick_stream(stream), d_dest, d_source, d_relu_result, n);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_dropout_f32(void* stream, const float* d_x, float* d_y, const unsigned char* d_drop, size_t n) {
	BLA_ENTER();
	if (n == 0) return BLA_OK;
	BLA_REQUIRE(d_x && d_y && d_drop, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n)), dim3(kT), 0, pick_stream(stream), d_x, d_y, d_drop, n);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_dropout_mask_f32(void* stream, float* d_x, const float* d_dropout_result, size_t n) {
	BLA_ENTER();
	if (n == 0) return BLA_OK;
	BLA_REQUIRE(d_x && d_dropout_result, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for(n)), dim3(kT), 0, pick_stream(stream), d_x, d_dropout_result, n);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_nearest_neighbours_f32(void* stream, const float* d_in, float* d_out, int channels, int in_h, int in_w, int out_h, int out_w, int scale) {
	BLA_ENTER();
	BLA_REQUIRE(channels > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && scale > 0, BLA_ERR_INVALID, "bad resize shape");
	BLA_REQUIRE((out_h - 1) / scale < in_h && (out_w - 1) / scale < in_w, BLA_ERR_INVALID, "output %dx%d / scale %d exceeds input %dx%d", out_h, out_w, scale, in_h, in_w);
	BLA_REQUIRE(d_in && d_out, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(nearest_kernel, dim3(blocks_for((size_t)channels * out_h * out_w)), dim3(kT), 0, pick_stream(stream), d_in, d_out, channels, in_w,
	                   in_h * in_w, out_h, out_w, scale);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_nearest_neighbours_ddx_f32(void* stream, const float* d_source, float* d_dest, int channels, int src_h, int src_w, int dest_h, int dest_w, int scale) {
	BLA_ENTER();
	BLA_REQUIRE(channels > 0 && src_h > 0 && src_w > 0 && dest_h > 0 && dest_w > 0 && scale > 0, BLA_ERR_INVALID, "bad resize shape");
	BLA_REQUIRE((src_h - 1) / scale < dest_h && (src_w - 1) / scale < dest_w, BLA_ERR_INVALID, "source %dx%d / scale %d exceeds destination %dx%d", src_h, src_w, scale, dest_h, dest_w);
	BLA_REQUIRE(d_source && d_dest, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(nearest_ddx_kernel, dim3(blocks_for((size_t)channels * dest_h * dest_w)), dim3(kT), 0, pick_stream(stream), d_source, d_dest, channels,
	                   src_h, src_w, dest_h, dest_w, scale);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_softmax_ddx_f32(void* stream, const float* d_softmax_output, const float* d_gradient, float* d_out, int rows, int dim) {
	BLA_ENTER();
	BLA_REQUIRE(rows >= 0 && dim >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, dim);
	if (rows == 0 || dim == 0) return BLA_OK;
	BLA_REQUIRE(d_softmax_output && d_gradient && d_out, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(softmax_ddx_kernel, dim3((rows + 3) / 4), dim3(kT), 0, pick_stream(stream), d_softmax_output, d_gradient, d_out, rows, dim);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // extern "C"
