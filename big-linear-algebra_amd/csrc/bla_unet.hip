// bla_unet.hip -- the elementwise / resize glue ops that sit on either side of every conv() in the reference's U-Net
// (model/cifar_unet.c:235-253,1024-1097,1168-1178,1229-1259), SURVEY 8(f) rank 1.  All HBM-bound, one pass each;
// channel arrays are contiguous [C][H*W] buffers.  The adds of _nearest_neighbours_ddx are done in the reference's
// order (row-major over the source block), so the fp32 result is bit-identical to the fp32 oracle.
#include "bla_internal.h"
#include <cmath>

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

// scale 2 on rows of a multiple of two source pixels: one thread reads two source pixels (8 bytes) and writes the 2 x 4 destination pixels they
// cover (two 16-byte stores).  The general kernel above spends a division chain per 4 bytes (51.9 us for the 67 MB the U-Net's last up-sampling writes)
__global__ void __launch_bounds__(kT) nearest2_kernel(const float2* __restrict__ in, float4* __restrict__ out, size_t pairs, int in_w2) {
	for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < pairs; e += (size_t)gridDim.x * kT) {
		const size_t row = e / in_w2;                 // (channel, source row)
		const int jp = (int)(e - row * in_w2);
		const float2 v = in[e];
		const float4 o = make_float4(v.x, v.x, v.y, v.y);
		out[(2 * row) * in_w2 + jp] = o;
		out[(2 * row + 1) * in_w2 + jp] = o;
	}
}
// its gradient: a destination pixel = ((s00 + s01) + s10) + s11, the order of the reference's scatter; one thread two destination pixels
__global__ void __launch_bounds__(kT) nearest2_ddx_kernel(const float4* __restrict__ src, float2* __restrict__ dest, size_t pairs, int dw2) {
	for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < pairs; e += (size_t)gridDim.x * kT) {
		const size_t row = e / dw2;                   // (channel, destination row)
		const int jp = (int)(e - row * dw2);
		const float4 a = src[(2 * row) * dw2 + jp], b = src[(2 * row + 1) * dw2 + jp];
		dest[e] = make_float2(((a.x + a.y) + b.x) + b.y, ((a.z + a.w) + b.z) + b.w);
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

// out = a + b (the residual sum at model/cifar_unet.c:1067-1071)
__global__ void __launch_bounds__(kT) sum_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, size_t n) {
	for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) out[i] = a[i] + b[i];
}

// _softmax_ddx, model/cifar_unet.c:1246-1259: one wave per row; out = s * (g - <s, g>)
__global__ void __launch_bounds__(kT) softmax_ddx_kernel(const float* __restrict__ s, const float* __restrict__ g, float* __restrict__ out, int rows, int dim, float scale) {
	int r = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= rows) return;
	const float* sr = s + (size_t)r * dim; const float* gr = g + (size_t)r * dim;
	double dot = 0;
	for (int j = lane; j < dim; j += 64) dot += (double)sr[j] * gr[j];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
	float d = (float)__shfl(dot, 0, 64);
	if (scale == 1.f) { for (int j = lane; j < dim; j += 64) out[(size_t)r * dim + j] = sr[j] * (gr[j] - d); }
	else { for (int j = lane; j < dim; j += 64) out[(size_t)r * dim + j] = (sr[j] * (gr[j] - d)) * scale; }   // the matrix_scale behind it (:1308), same rounding
}
// dst[i] = sum over the images of src[b][i] (in image order): weight gradients of a batch from per-image products
__global__ void __launch_bounds__(kT) batch_sum_kernel(const float* __restrict__ src, float* __restrict__ dst, int batch, size_t n) {
	for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
		float acc = 0.f;
		int b = 0;
		for (; b + 8 <= batch; b += 8) {
			float v[8];
#pragma unroll
			for (int u = 0; u < 8; u++) v[u] = src[(size_t)(b + u) * n + i];
#pragma unroll
			for (int u = 0; u < 8; u++) acc += v[u];
		}
		for (; b < batch; b++) acc += src[(size_t)b * n + i];
		dst[i] = acc;
	}
}
static bla_status batch_sum(void* stream, const float* src, float* dst, int batch, size_t n) {
	hipLaunchKernelGGL(batch_sum_kernel, dim3(blocks_for(n)), dim3(kT), 0, pick_stream(stream), src, dst, batch, n);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
// Group norm over a batch [B][C][HW]: the groups of one image never reach into the next, so the batch folds into the channel count when the
// groups divide the channels (B*C channels, same groups) or when an image is one short group (groups of C).  false: ragged, image by image.
static bool fold_groups(int batch, int c, int group_size, int* channels, int* group) {
	if (c % group_size == 0) { *channels = batch * c; *group = group_size; return true; }
	if (c < group_size) { *channels = batch * c; *group = c; return true; }
	return batch == 1 ? (*channels = c, *group = group_size, true) : false;
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
	if (scale == 2 && out_h == 2 * in_h && out_w == 2 * in_w && in_w % 2 == 0 && (uintptr_t)d_in % 8 == 0 && (uintptr_t)d_out % 16 == 0) {
		const size_t pairs = (size_t)channels * in_h * in_w / 2;
		hipLaunchKernelGGL(nearest2_kernel, dim3(blocks_for(pairs)), dim3(kT), 0, pick_stream(stream), (const float2*)d_in, (float4*)d_out, pairs, in_w / 2);
	} else
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
	if (scale == 2 && src_h == 2 * dest_h && src_w == 2 * dest_w && dest_w % 2 == 0 && (uintptr_t)d_source % 16 == 0 && (uintptr_t)d_dest % 8 == 0) {
		const size_t pairs = (size_t)channels * dest_h * dest_w / 2;
		hipLaunchKernelGGL(nearest2_ddx_kernel, dim3(blocks_for(pairs)), dim3(kT), 0, pick_stream(stream), (const float4*)d_source, (float2*)d_dest, pairs, dest_w / 2);
	} else
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
	hipLaunchKernelGGL(softmax_ddx_kernel, dim3((rows + 3) / 4), dim3(kT), 0, pick_stream(stream), d_softmax_output, d_gradient, d_out, rows, dim, 1.f);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

/* ---- self-attention block, model/cifar_unet.c:999-1022 / :1261-1337, device-resident, intended composition --------
 * x, out, del_y, del_x: [C][S] (S = H*W); wq/wk/wv: [C][d]; w: [d][C]; bias: [C].  Every matrix_transpose the reference
 * materialises (9 of them) and both channel reshapes disappear into the GEMM's transa/transb: the [S][C] "input" matrix
 * is X^T, so Z.Wq is a TN product on X itself, and output = dense^T comes straight out of a TT product. */
static bla_status ep_gemm(void* s, int ta, int tb, int m, int n, int k, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float alpha,
                          float beta, const float* bias_row) {
	bla_gemm_epilogue ep = {};
	ep.alpha = alpha; ep.beta = beta; ep.bias_row = bias_row;
	return bla_gemm_f32(s, ta, tb, m, n, k, A, lda, B, ldb, C, ldc, &ep);
}

// two independent products in one launch where both are latency-bound shapes (bla_gemm_pair_f32); plain alpha/beta epilogues
static bla_status pair_gemm(void* s, int ta1, int tb1, int m1, int n1, int k1, const float* A1, int lda1, const float* B1, int ldb1, float* C1, int ldc1,
                            int ta2, int tb2, int m2, int n2, int k2, const float* A2, int lda2, const float* B2, int ldb2, float* C2, int ldc2, float alpha2 = 1.f) {
	bla_gemm_epilogue e1 = {}, e2 = {};
	e1.alpha = 1.f; e2.alpha = alpha2;
	bla_gemm_desc p = {ta1, tb1, m1, n1, k1, A1, lda1, B1, ldb1, C1, ldc1, &e1};
	bla_gemm_desc q = {ta2, tb2, m2, n2, k2, A2, lda2, B2, ldb2, C2, ldc2, &e2};
	return bla_gemm_pair_f32(s, &p, &q);
}

bla_status bla_attention_forward_f32(void* stream, const float* d_x, const float* d_wq, const float* d_wk, const float* d_wv, const float* d_w,
                                     const float* d_bias, const bla_attention_ws* ws, float* d_out, int c, int s, int d) {
	BLA_ENTER();
	BLA_REQUIRE(c > 0 && s > 0 && d > 0, BLA_ERR_INVALID, "bad attention shape C=%d S=%d d=%d", c, s, d);
	BLA_REQUIRE(d_x && d_wq && d_wk && d_wv && d_w && d_bias && d_out && ws && ws->q && ws->k && ws->v && ws->scores_raw && ws->weights && ws->attention,
	            BLA_ERR_INVALID, "null operand");
	const float inv = (float)(1.0 / sqrt((double)d));                                                  // :1010
	st = pair_gemm(stream, 1, 0, s, d, c, d_x, s, d_wq, d, ws->q, d,                                       // Q = Z Wq, Z = X^T   :1003-1004
	               1, 0, s, d, c, d_x, s, d_wk, d, ws->k, d); if (st) return st;                          // beside K = Z Wk (one launch)
	st = ep_gemm(stream, 1, 0, s, d, c, d_x, s, d_wv, d, ws->v, d, 1.f, 0.f, nullptr); if (st) return st;
	{   // (Q K^T) / sqrt(d) :1007-1014, stored twice by the product itself: scores_raw (kept) and weights (softmaxed in place, :1015)
		bla_gemm_epilogue ep = {};
		ep.alpha = inv; ep.pre_act = ws->scores_raw; ep.ld_pre = s;
		st = bla_gemm_f32(stream, 0, 1, s, s, d, ws->q, d, ws->k, d, ws->weights, s, &ep); if (st) return st;
	}
	st = bla_softmax_rows_f32(stream, ws->weights, s, s); if (st) return st;                               // :1015
	st = ep_gemm(stream, 0, 0, s, d, s, ws->weights, s, ws->v, d, ws->attention, d, 1.f, 0.f, nullptr); if (st) return st;   // :1018
	return ep_gemm(stream, 1, 1, c, s, d, d_w, c, ws->attention, d, d_out, s, 1.f, 0.f, d_bias);              // (P W + b)^T   :1019-1021
}

bla_status bla_attention_backward_f32(void* stream, const float* d_del_y, const float* d_x, const float* d_wq, const float* d_wk, const float* d_wv,
                                      const float* d_w, const bla_attention_ws* fw, const bla_attention_ws* g, float* d_del_wq, float* d_del_wk,
                                      float* d_del_wv, float* d_del_w, float* d_del_x, int c, int s, int d, int jacobian_from_raw) {
	BLA_ENTER();
	BLA_REQUIRE(c > 0 && s > 0 && d > 0, BLA_ERR_INVALID, "bad attention shape C=%d S=%d d=%d", c, s, d);
	BLA_REQUIRE(d_del_y && d_x && d_wq && d_wk && d_wv && d_w && fw && g && d_del_wq && d_del_wk && d_del_wv && d_del_w && d_del_x, BLA_ERR_INVALID, "null operand");
	BLA_REQUIRE(g->q && g->k && g->v && g->scores_raw && g->weights && g->attention, BLA_ERR_INVALID, "null gradient workspace");
	const float inv = (float)(1.0 / sqrt((double)d));
	float *del_q = g->q, *del_k = g->k, *del_v = g->v, *del_i = g->scores_raw, *del_s = g->weights, *del_p = g->attention;
	// independent products share a launch (the reference runs all fourteen one after the other)
	st = pair_gemm(stream, 1, 1, d, c, s, fw->attention, d, d_del_y, s, d_del_w, c,                                          // del_W = P^T del_Y'     :1291-1293
	               1, 1, s, d, c, d_del_y, s, d_w, c, del_p, d); if (st) return st;                                         // del_P = del_Y' W^T     :1295-1297
	st = pair_gemm(stream, 0, 1, s, s, d, del_p, d, fw->v, d, del_s, s,                                                     // del_S = del_P V^T      :1303-1305
	               1, 0, s, d, s, fw->weights, s, del_p, d, del_v, d); if (st) return st;                                   // del_V = S^T del_P      :1299-1301
	st = bla_softmax_ddx_f32(stream, jacobian_from_raw ? fw->scores_raw : fw->weights, del_s, del_i, s, s); if (st) return st;   // :1307
	st = bla_scale_f32(stream, del_i, (size_t)s * s, inv); if (st) return st;                                                // :1308
	st = pair_gemm(stream, 0, 0, s, d, s, del_i, s, fw->k, d, del_q, d,                                                     // :1310
	               1, 0, s, d, s, del_i, s, fw->q, d, del_k, d); if (st) return st;                                         // :1312-1314
	st = pair_gemm(stream, 0, 0, c, d, s, d_x, s, del_k, d, d_del_wk, d,                                                    // Z^T = X              :1316-1319
	               0, 0, c, d, s, d_x, s, del_q, d, d_del_wq, d); if (st) return st;
	st = pair_gemm(stream, 0, 0, c, d, s, d_x, s, del_v, d, d_del_wv, d,
	               0, 1, c, s, d, d_wq, d, del_q, d, d_del_x, s); if (st) return st;                                        // del_Z^T, same add order :1322-1334
	st = ep_gemm(stream, 0, 1, c, s, d, d_wk, d, del_k, d, d_del_x, s, 1.f, 1.f, nullptr); if (st) return st;
	return ep_gemm(stream, 0, 1, c, s, d, d_wv, d, del_v, d, d_del_x, s, 1.f, 1.f, nullptr);
}

/* The same block for a batch of images, x / out [B][C][S], every workspace buffer B times the single-image size.  Each per-image product becomes
 * one launch over the batch (bla_gemm_batched_f32); the four weight gradients are summed over the images from per-image products (d_partials:
 * [B][C*d] scratch). */
static bla_status bgemm(void* s, int ta, int tb, int m, int n, int k, const float* A, int lda, long sa, const float* B, int ldb, long sb, float* C, int ldc, long sc,
                        int batch, float alpha = 1.f, float beta = 0.f, const float* bias_row = nullptr, float* pre = nullptr, int ld_pre = 0, long spre = 0) {
	bla_gemm_epilogue ep = {};
	ep.alpha = alpha; ep.beta = beta; ep.bias_row = bias_row; ep.pre_act = pre; ep.ld_pre = ld_pre;
	return bla_gemm_batched_f32(s, ta, tb, m, n, k, A, lda, sa, B, ldb, sb, C, ldc, sc, batch, &ep, spre);
}

bla_status bla_attention_forward_batched_f32(void* stream, int batch, const float* d_x, const float* d_wq, const float* d_wk, const float* d_wv, const float* d_w,
                                             const float* d_bias, const bla_attention_ws* ws, float* d_out, int c, int s, int d) {
	if (batch == 1) return bla_attention_forward_f32(stream, d_x, d_wq, d_wk, d_wv, d_w, d_bias, ws, d_out, c, s, d);
	BLA_ENTER();
	BLA_REQUIRE(batch > 0 && c > 0 && s > 0 && d > 0, BLA_ERR_INVALID, "bad attention shape B=%d C=%d S=%d d=%d", batch, c, s, d);
	BLA_REQUIRE(d_x && d_wq && d_wk && d_wv && d_w && d_bias && d_out && ws && ws->q && ws->k && ws->v && ws->scores_raw && ws->weights && ws->attention,
	            BLA_ERR_INVALID, "null operand");
	const float inv = (float)(1.0 / sqrt((double)d));
	const long xs = (long)c * s, qs = (long)s * d, ss = (long)s * s;
	st = bgemm(stream, 1, 0, s, d, c, d_x, s, xs, d_wq, d, 0, ws->q, d, qs, batch); if (st) return st;                       // Q = Z Wq   :1003-1004
	st = bgemm(stream, 1, 0, s, d, c, d_x, s, xs, d_wk, d, 0, ws->k, d, qs, batch); if (st) return st;
	st = bgemm(stream, 1, 0, s, d, c, d_x, s, xs, d_wv, d, 0, ws->v, d, qs, batch); if (st) return st;
	st = bgemm(stream, 0, 1, s, s, d, ws->q, d, qs, ws->k, d, qs, ws->weights, s, ss, batch, inv, 0.f, nullptr, ws->scores_raw, s, ss); if (st) return st;   // :1007-1014
	st = bla_softmax_rows_f32(stream, ws->weights, batch * s, s); if (st) return st;                                       // :1015
	st = bgemm(stream, 0, 0, s, d, s, ws->weights, s, ss, ws->v, d, qs, ws->attention, d, qs, batch); if (st) return st;     // :1018
	return bgemm(stream, 1, 1, c, s, d, d_w, c, 0, ws->attention, d, qs, d_out, s, xs, batch, 1.f, 0.f, d_bias);             // :1019-1021
}

bla_status bla_attention_backward_batched_f32(void* stream, int batch, const float* d_del_y, const float* d_x, const float* d_wq, const float* d_wk,
                                              const float* d_wv, const float* d_w, const bla_attention_ws* fw, const bla_attention_ws* g, float* d_partials,
                                              float* d_del_wq, float* d_del_wk, float* d_del_wv, float* d_del_w, float* d_del_x, int c, int s, int d,
                                              int jacobian_from_raw) {
	if (batch == 1)
		return bla_attention_backward_f32(stream, d_del_y, d_x, d_wq, d_wk, d_wv, d_w, fw, g, d_del_wq, d_del_wk, d_del_wv, d_del_w, d_del_x, c, s, d, jacobian_from_raw);
	BLA_ENTER();
	BLA_REQUIRE(batch > 0 && c > 0 && s > 0 && d > 0, BLA_ERR_INVALID, "bad attention shape B=%d C=%d S=%d d=%d", batch, c, s, d);
	BLA_REQUIRE(d_del_y && d_x && d_wq && d_wk && d_wv && d_w && fw && g && d_partials && d_del_wq && d_del_wk && d_del_wv && d_del_w && d_del_x, BLA_ERR_INVALID, "null operand");
	BLA_REQUIRE(g->q && g->k && g->v && g->scores_raw && g->weights && g->attention, BLA_ERR_INVALID, "null gradient workspace");
	const float inv = (float)(1.0 / sqrt((double)d));
	const long xs = (long)c * s, qs = (long)s * d, ss = (long)s * s, ws_ = (long)c * d;
	float *del_q = g->q, *del_k = g->k, *del_v = g->v, *del_i = g->scores_raw, *del_s = g->weights, *del_p = g->attention;
	st = bgemm(stream, 1, 1, d, c, s, fw->attention, d, qs, d_del_y, s, xs, d_partials, c, ws_, batch); if (st) return st;      // del_W = sum P^T del_Y'   :1291-1293
	st = batch_sum(stream, d_partials, d_del_w, batch, (size_t)ws_); if (st) return st;
	st = bgemm(stream, 1, 1, s, d, c, d_del_y, s, xs, d_w, c, 0, del_p, d, qs, batch); if (st) return st;                       // del_P = del_Y' W^T       :1295-1297
	st = bgemm(stream, 0, 1, s, s, d, del_p, d, qs, fw->v, d, qs, del_s, s, ss, batch); if (st) return st;                      // del_S = del_P V^T        :1303-1305
	st = bgemm(stream, 1, 0, s, d, s, fw->weights, s, ss, del_p, d, qs, del_v, d, qs, batch); if (st) return st;                // del_V = S^T del_P        :1299-1301
	hipLaunchKernelGGL(softmax_ddx_kernel, dim3((batch * s + 3) / 4), dim3(kT), 0, pick_stream(stream), jacobian_from_raw ? fw->scores_raw : fw->weights, del_s, del_i,
	                   batch * s, s, inv);                                                                                    // :1307-1308 in one pass
	BLA_HIP(hipGetLastError());
	st = bgemm(stream, 0, 0, s, d, s, del_i, s, ss, fw->k, d, qs, del_q, d, qs, batch); if (st) return st;                      // :1310
	st = bgemm(stream, 1, 0, s, d, s, del_i, s, ss, fw->q, d, qs, del_k, d, qs, batch); if (st) return st;                      // :1312-1314
	st = bgemm(stream, 0, 0, c, d, s, d_x, s, xs, del_k, d, qs, d_partials, d, ws_, batch); if (st) return st;                  // Z^T = X                :1316-1319
	st = batch_sum(stream, d_partials, d_del_wk, batch, (size_t)ws_); if (st) return st;
	st = bgemm(stream, 0, 0, c, d, s, d_x, s, xs, del_q, d, qs, d_partials, d, ws_, batch); if (st) return st;
	st = batch_sum(stream, d_partials, d_del_wq, batch, (size_t)ws_); if (st) return st;
	st = bgemm(stream, 0, 0, c, d, s, d_x, s, xs, del_v, d, qs, d_partials, d, ws_, batch); if (st) return st;
	st = batch_sum(stream, d_partials, d_del_wv, batch, (size_t)ws_); if (st) return st;
	if (gemm_thin_applies(c, s, d, batch)) {   // the three d-deep terms of del_Z^T accumulate in registers (q, then k, then v) and are stored once  :1322-1334
		const ThinPart parts[3] = {{d_wq, del_q, 0, qs, d, d}, {d_wk, del_k, 0, qs, d, d}, {d_wv, del_v, 0, qs, d, d}};
		return gemm_thin_parts(stream, 0, 1, c, s, d, parts, 3, d_del_x, s, xs, batch, 1.f, 0.f, nullptr, nullptr, 0, 0);
	}
	st = bgemm(stream, 0, 1, c, s, d, d_wq, d, 0, del_q, d, qs, d_del_x, s, xs, batch); if (st) return st;                      // del_Z^T, same add order :1322-1334
	st = bgemm(stream, 0, 1, c, s, d, d_wk, d, 0, del_k, d, qs, d_del_x, s, xs, batch, 1.f, 1.f); if (st) return st;
	return bgemm(stream, 0, 1, c, s, d, d_wv, d, 0, del_v, d, qs, d_del_x, s, xs, batch, 1.f, 1.f);
}

bla_status bla_sum_f32(void* stream, float* d_out, const float* d_a, const float* d_b, size_t n) {
	BLA_ENTER();
	if (n == 0) return BLA_OK;
	BLA_REQUIRE(d_out && d_a && d_b, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(sum_kernel, dim3(blocks_for(n)), dim3(kT), 0, pick_stream(stream), d_out, d_a, d_b, n);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

/* ---- ResNet block, model/cifar_unet.c:1044-1072 / :1180-1227, device-resident, intended composition ------------------
 * group norm + ReLU fused; the 3x3 and 1x1 convolutions are implicit GEMMs (nothing of ConvData is materialised);
 * the time-embedding dense layer is a 1 x T . T x Cout product with the bias in the epilogue. */
bla_status bla_resnet_forward_f32(void* stream, const float* d_x, const float* d_temb, const bla_resnet_params* p, const unsigned char* d_drop,
                                  const bla_resnet_ws* ws, float* d_result, int h, int w, int cin, int cout, int k, int tdim, int group_size) {
	return resnet_forward_single(stream, d_x, d_temb, p, d_drop, ws, d_result, h, w, cin, cout, k, tdim, group_size, 0);
}
}  // extern "C"
bla_status bla::resnet_forward_single(void* stream, const float* d_x, const float* d_temb, const bla_resnet_params* p, const unsigned char* d_drop,
                                      const bla_resnet_ws* ws, float* d_result, int h, int w, int cin, int cout, int k, int tdim, int group_size, int flags) {
	BLA_ENTER();
	BLA_REQUIRE(h > 0 && w > 0 && cin > 0 && cout > 0 && k > 0 && tdim > 0 && group_size > 0, BLA_ERR_INVALID, "bad resnet shape");
	BLA_REQUIRE(d_x && d_temb && p && d_drop && ws && d_result && p->conv1 && p->conv2 && p->time_w && p->time_b, BLA_ERR_INVALID, "null operand");
	BLA_REQUIRE(cin == cout || (p->res && ws->res), BLA_ERR_INVALID, "Cin != Cout needs the residual 1x1 kernels and workspace");
	const int hw = h * w;
	st = bla_group_norm_relu_f32(stream, d_x, ws->relu1, ws->sd1, ws->mu1, cin, group_size, hw); if (st) return st;         // :1046-1047
	// the time-embedding projection first, so that its per-channel add (:1053) rides in the store of the first convolution (:1048)
	if (!(flags & RESNET_TDENSE_READY)) {
		bla_gemm_epilogue ep = {};
		ep.alpha = 1.f; ep.bias_col = p->time_b;
		st = bla_gemm_f32(stream, 0, 0, 1, cout, tdim, d_temb, tdim, p->time_w, cout, ws->tdense, cout, &ep); if (st) return st; // :1051-1052
	}
	st = conv2d_forward_epilogue(stream, ws->relu1, p->conv1, ws->c1, h, w, k, cin, cout, 1, ws->tdense, nullptr, nullptr); if (st) return st;   // :1048,1053
	st = group_norm_relu_dropout(stream, ws->c1, ws->relu2, d_drop, ws->dp, ws->sd2, ws->mu2, cout, group_size, hw); if (st) return st;   // :1056-1058, one pass
	const float* r = d_x;
	if (cin != cout) {
		st = bla_conv2d_forward_f32(stream, d_x, p->res, ws->res, h, w, 1, cin, cout, 1); if (st) return st;                // :1062-1066
		r = ws->res;
	}
	// second convolution (:1059) with the residual sum (:1067-1071) in its store: c2 = conv, result = c2 + r
	return conv2d_forward_epilogue(stream, ws->dp, p->conv2, ws->c2, h, w, k, cout, cout, 1, nullptr, r, d_result);
}
extern "C" {

bla_status bla_resnet_backward_f32(void* stream, const float* d_del_out, const float* d_x, const float* d_temb, const bla_resnet_params* p,
                                   const bla_resnet_ws* ws, const bla_resnet_grads* g, const bla_resnet_scratch* sc, float* d_del_x, int h, int w,
                                   int cin, int cout, int k, int tdim, int group_size) {
	BLA_ENTER();
	BLA_REQUIRE(h > 0 && w > 0 && cin > 0 && cout > 0 && k > 0 && tdim > 0 && group_size > 0, BLA_ERR_INVALID, "bad resnet shape");
	BLA_REQUIRE(d_del_out && d_x && d_temb && p && ws && g && sc && d_del_x && g->conv1 && g->conv2 && g->time_w && g->time_b && sc->g_out_a &&
	            sc->g_out_b && sc->g_in && sc->flip, BLA_ERR_INVALID, "null operand");
	BLA_REQUIRE(cin == cout || (p->res && g->res), BLA_ERR_INVALID, "Cin != Cout needs the residual 1x1 kernels and their gradient");
	const int hw = h * w;
	// second chunk, :1186-1189
	st = bla_conv2d_backward_f32(stream, d_del_out, ws->dp, p->conv2, g->conv2, sc->g_out_a, sc->flip, h, w, k, cout, cout, 1); if (st) return st;
	// _dropout_mask then multi_channel_relu_ddx (:1187-1188) as the gate of the group-norm gradient: dp = drop ? 0 : relu2 is zero wherever
	// relu2 is, so "dp <= 0" is both masks at once (dp >= 0 everywhere)
	st = group_norm_ddx_gated(stream, sc->g_out_a, sc->g_out_b, ws->c1, ws->mu2, ws->sd2, cout, group_size, hw, ws->dp, nullptr); if (st) return st;
	// time-embedding projection, :1191-1199: bias gradient = per-channel sums, weight gradient = temb^T . dtb (rank 1)
	st = bla_col_sum_f32(stream, sc->g_out_b, cout, hw, g->time_b, BLA_COLSUM_INTENDED); if (st) return st;
	st = bla_gemm_f32(stream, 1, 0, tdim, cout, 1, d_temb, tdim, g->time_b, cout, g->time_w, cout, nullptr); if (st) return st;
	// first chunk, :1202-1205 (gradient sink = the gradient struct; as written :1203 hands conv_ddx the parameters, Q8)
	st = bla_conv2d_backward_f32(stream, sc->g_out_b, ws->relu1, p->conv1, g->conv1, sc->g_in, sc->flip, h, w, k, cin, cout, 1); if (st) return st;
	// ReLU gate on the way in (:1204); with an identity residual its gradient (del_out itself, :1219) is added on the way out
	st = group_norm_ddx_gated(stream, sc->g_in, d_del_x, d_x, ws->mu1, ws->sd1, cin, group_size, hw, ws->relu1, cin == cout ? d_del_out : nullptr); if (st) return st;
	// residual connection through the 1x1 convolution, :1208-1220
	if (cin != cout) {
		st = bla_conv2d_backward_f32(stream, d_del_out, d_x, p->res, g->res, sc->g_in, sc->flip, h, w, 1, cin, cout, 1); if (st) return st;
		return bla_add_f32(stream, d_del_x, sc->g_in, (size_t)cin * hw);
	}
	return BLA_OK;
}

/* The ResNet block for a batch: x / del_x [B][Cin][HW], result / del_out [B][Cout][HW], temb [B][T] (every image its own time step), d_drop
 * [B][Cout*HW]; the workspace buffers are B times the single-image sizes (tdense [B][Cout], mu / sd [B][groups]).  The convolutions run as
 * batched implicit GEMMs, the norms over B*C channels; the gradients are summed over the images.  d_dtb: [B][Cout] scratch. */
// pad (optional): the padded copy of what this norm feeds the next convolution with, written in the same pass; *pad_done says whether it was (only when the
// batch folds into the channel count -- every shape of the U-Net does)
static bla_status gn_relu_b(void* stream, int batch, const float* in, float* out, float* sd, float* mu, int c, int gs, int hw, const unsigned char* drop, float* dropped,
                            const PadOut* pad = nullptr, bool* pad_done = nullptr) {
	int ch, grp;
	if (pad_done) *pad_done = false;
	if (fold_groups(batch, c, gs, &ch, &grp)) {
		if ((pad && pad->dst) || !out) { if (pad_done) *pad_done = pad && pad->dst; return group_norm_relu_dropout(stream, in, out, drop, dropped, sd, mu, ch, grp, hw, pad); }
		return drop ? group_norm_relu_dropout(stream, in, out, drop, dropped, sd, mu, ch, grp, hw) : bla_group_norm_relu_f32(stream, in, out, sd, mu, ch, grp, hw);
	}
	const int groups = (c + gs - 1) / gs;
	for (int b = 0; b < batch; b++) {
		const size_t o = (size_t)b * c * hw;
		bla_status st = drop ? group_norm_relu_dropout(stream, in + o, out + o, drop + o, dropped + o, sd + (size_t)b * groups, mu + (size_t)b * groups, c, gs, hw)
		                     : bla_group_norm_relu_f32(stream, in + o, out + o, sd + (size_t)b * groups, mu + (size_t)b * groups, c, gs, hw);
		if (st) return st;
	}
	return BLA_OK;
}
static bla_status gn_ddx_b(void* stream, int batch, const float* src, float* dst, const float* data, const float* mu, const float* sd, int c, int gs, int hw,
                           const float* gate, const float* addend, const PadOut* pad = nullptr, bool* pad_done = nullptr) {
	int ch, grp;
	if (pad_done) *pad_done = false;
	if (fold_groups(batch, c, gs, &ch, &grp)) {
		if (pad && pad->dst && pad_done) *pad_done = true;
		return group_norm_ddx_gated(stream, src, dst, data, mu, sd, ch, grp, hw, gate, addend, pad);
	}
	const int groups = (c + gs - 1) / gs;
	for (int b = 0; b < batch; b++) {
		const size_t o = (size_t)b * c * hw;
		bla_status st = group_norm_ddx_gated(stream, src + o, dst + o, data + o, mu + (size_t)b * groups, sd + (size_t)b * groups, c, gs, hw, gate ? gate + o : nullptr,
		                                     addend ? addend + o : nullptr);
		if (st) return st;
	}
	return BLA_OK;
}

bla_status bla_group_norm_relu_batched_f32(void* stream, int batch, const float* d_in, float* d_out, float* d_stdevs, float* d_means, int channels, int group_size, int hw) {
	return gn_relu_b(stream, batch, d_in, d_out, d_stdevs, d_means, channels, group_size, hw, nullptr, nullptr);
}
bla_status bla_group_norm_ddx_gated_batched_f32(void* stream, int batch, const float* d_source, float* d_dest, const float* d_data, const float* d_means,
                                                const float* d_stdevs, int channels, int group_size, int hw, const float* d_relu_gate, const float* d_addend) {
	return gn_ddx_b(stream, batch, d_source, d_dest, d_data, d_means, d_stdevs, channels, group_size, hw, d_relu_gate, d_addend);
}

bla_status bla_resnet_forward_batched_f32(void* stream, int batch, const float* d_x, const float* d_temb, const bla_resnet_params* p, const unsigned char* d_drop,
                                          const bla_resnet_ws* ws, float* d_result, int h, int w, int cin, int cout, int k, int tdim, int group_size) {
	return resnet_forward_batched(stream, batch, d_x, d_temb, p, d_drop, ws, d_result, h, w, cin, cout, k, tdim, group_size, 0, nullptr);
}
bla_status bla_resnet_backward_batched_f32(void* stream, int batch, const float* d_del_out, const float* d_x, const float* d_temb, const bla_resnet_params* p,
                                           const bla_resnet_ws* ws, const bla_resnet_grads* g, const bla_resnet_scratch* sc, float* d_dtb, float* d_del_x, int h,
                                           int w, int cin, int cout, int k, int tdim, int group_size) {
	return resnet_backward_batched(stream, batch, d_del_out, d_x, d_temb, p, ws, g, sc, d_dtb, d_del_x, h, w, cin, cout, k, tdim, group_size, 0, nullptr);
}

}  // extern "C"

// flags (bla_internal.h): RESNET_TDENSE_READY -- ws->tdense already holds temb . W_t + b_t (the U-Net forms all 18 blocks' projections in one launch before the
// first block); RESNET_DEFER_TIME_GRADS -- d_dtb receives the per-image channel sums and the caller forms time_w / time_b gradients from them later
bla_status bla::resnet_forward_batched(void* stream, int batch, const float* d_x, const float* d_temb, const bla_resnet_params* p, const unsigned char* d_drop,
                                       const bla_resnet_ws* ws, float* d_result, int h, int w, int cin, int cout, int k, int tdim, int group_size, int flags,
                                       ResnetPads* pads) {
	if (batch == 1) return resnet_forward_single(stream, d_x, d_temb, p, d_drop, ws, d_result, h, w, cin, cout, k, tdim, group_size, flags);
	BLA_ENTER();
	BLA_REQUIRE(batch > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && k > 0 && tdim > 0 && group_size > 0, BLA_ERR_INVALID, "bad resnet shape");
	BLA_REQUIRE(d_x && d_temb && p && d_drop && ws && d_result && p->conv1 && p->conv2 && p->time_w && p->time_b, BLA_ERR_INVALID, "null operand");
	BLA_REQUIRE(cin == cout || (p->res && ws->res), BLA_ERR_INVALID, "Cin != Cout needs the residual 1x1 kernels and workspace");
	const int hw = h * w;
	// the norm kernels leave the padded copies their convolutions (and, later, the weight gradients) gather from: no padding pass of relu1 / dp anywhere
	const PadLayout L = pads ? conv_padded_layout(h, w, k, 1) : PadLayout{};
	const PadOut po1 = {pads && L.plane ? pads->pad1 : nullptr, L}, po2 = {pads && L.plane ? pads->pad2 : nullptr, L};
	bool have1 = false, have2 = false;
	st = gn_relu_b(stream, batch, d_x, ws->relu1, ws->sd1, ws->mu1, cin, group_size, hw, nullptr, nullptr, &po1, &have1); if (st) return st;            // :1046-1047
	if (pads) { pads->have1 = have1; }
	if (!(flags & RESNET_TDENSE_READY)) {
		bla_gemm_epilogue ep = {};
		ep.alpha = 1.f; ep.bias_col = p->time_b;
		st = bla_gemm_f32(stream, 0, 0, batch, cout, tdim, d_temb, tdim, p->time_w, cout, ws->tdense, cout, &ep); if (st) return st;      // :1051-1052, one row per image
	}
	st = conv2d_forward_epilogue(stream, ws->relu1, p->conv1, ws->c1, h, w, k, cin, cout, 1, ws->tdense, nullptr, nullptr, batch, cout, have1 ? pads->pad1 : nullptr,
	                             pads ? pads->k1_fwd : nullptr);
	if (st) return st;                                                                                                                      // :1048,1053
	st = gn_relu_b(stream, batch, ws->c1, ws->relu2, ws->sd2, ws->mu2, cout, group_size, hw, d_drop, ws->dp, &po2, &have2); if (st) return st;          // :1056-1058
	if (pads) { pads->have2 = have2; }
	const float* r = d_x;
	if (cin != cout) {
		st = bla_conv2d_forward_batched_f32(stream, d_x, p->res, ws->res, batch, h, w, 1, cin, cout, 1); if (st) return st;           // :1062-1066
		r = ws->res;
	}
	return conv2d_forward_epilogue(stream, ws->dp, p->conv2, ws->c2, h, w, k, cout, cout, 1, nullptr, r, d_result, batch, 0, have2 ? pads->pad2 : nullptr,
	                               pads ? pads->k2_fwd : nullptr);                                                                          // :1059,1067-1071
}

bla_status bla::resnet_backward_batched(void* stream, int batch, const float* d_del_out, const float* d_x, const float* d_temb, const bla_resnet_params* p,
                                        const bla_resnet_ws* ws, const bla_resnet_grads* g, const bla_resnet_scratch* sc, float* d_dtb, float* d_del_x, int h,
                                        int w, int cin, int cout, int k, int tdim, int group_size, int flags, const ResnetPads* pads) {
	if (batch == 1 && d_del_x) return bla_resnet_backward_f32(stream, d_del_out, d_x, d_temb, p, ws, g, sc, d_del_x, h, w, cin, cout, k, tdim, group_size);
	BLA_ENTER();
	BLA_REQUIRE(batch > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && k > 0 && tdim > 0 && group_size > 0, BLA_ERR_INVALID, "bad resnet shape");
	BLA_REQUIRE(d_del_out && d_x && d_temb && p && ws && g && sc && d_dtb && g->conv1 && g->conv2 && g->time_w && g->time_b && sc->g_out_a &&
	            sc->g_out_b && sc->g_in && sc->flip, BLA_ERR_INVALID, "null operand");
	BLA_REQUIRE(cin == cout || (p->res && g->res), BLA_ERR_INVALID, "Cin != Cout needs the residual 1x1 kernels and their gradient");
	const int hw = h * w;
	// RESNET_WGRAD_SIDE: the three weight gradients go to the context's side lane (each after a fork: the lane waits for what this stream has issued so far),
	// the data gradients, norms and sums stay here -- MFMA-bound products beside HBM-bound passes instead of one after the other.  The caller joins the lane
	// once, at the end of its pass, and keeps what the lane reads (del_out, g_out_b, the activations and their padded copies) intact until then.
	const bool side = (flags & RESNET_WGRAD_SIDE) && pads && pads->g_out_b;
	float* const g_out_b = side ? pads->g_out_b : sc->g_out_b;
	hipStream_t main_s = pick_stream(stream);
	auto wgrad_on_lane = [&](const float* del_y, const float* x, const float* kern, float* gkern, int kk, int ci, int co, const float* x_padded) -> bla_status {
		hipStream_t lane;
		bla_status s2 = side_lane_fork(main_s, &lane);
		if (s2) return s2;
		s2 = conv2d_backward_batched(lane, del_y, x, kern, gkern, nullptr, sc->flip, batch, h, w, kk, ci, co, 1, x_padded, nullptr, nullptr);
		side_lane_done();
		return s2;
	};
	if (side) { st = wgrad_on_lane(d_del_out, ws->dp, p->conv2, g->conv2, k, cout, cout, pads->have2 ? pads->pad2 : nullptr); if (st) return st; }
	st = conv2d_backward_batched(stream, d_del_out, ws->dp, p->conv2, side ? nullptr : g->conv2, sc->g_out_a, sc->flip, batch, h, w, k, cout, cout, 1,
	                             pads && pads->have2 ? pads->pad2 : nullptr, pads ? pads->k2_bwd : nullptr);
	if (st) return st;                                                                                                                      // :1186-1189
	// the gradient that reaches the first convolution also lands in the padded copy its data gradient gathers from -- where that product runs on the padded-copy
	// kernel (prep mode 3) and someone wants it
	const PadLayout L = pads && pads->dy_pad && d_del_x && k % 2 == 1 && conv_kernel_prep_mode(batch, h, w, k, cin, cout, 1, true) == 3 ? conv_padded_layout(h, w, k, 1) : PadLayout{};
	const PadOut pod = {L.plane ? pads->dy_pad : nullptr, L};
	bool have_dy = false;
	st = gn_ddx_b(stream, batch, sc->g_out_a, g_out_b, ws->c1, ws->mu2, ws->sd2, cout, group_size, hw, ws->dp, nullptr, &pod, &have_dy); if (st) return st;
	// time-embedding projection, :1191-1199: per image the per-channel sums, then bias gradient = their sum over the images, weight gradient = temb^T . dtb
	st = bla_col_sum_f32(stream, g_out_b, batch * cout, hw, d_dtb, BLA_COLSUM_INTENDED); if (st) return st;
	if (!(flags & RESNET_DEFER_TIME_GRADS)) {
		st = batch_sum(stream, d_dtb, g->time_b, batch, (size_t)cout); if (st) return st;
		st = bla_gemm_f32(stream, 1, 0, tdim, cout, batch, d_temb, tdim, d_dtb, cout, g->time_w, cout, nullptr); if (st) return st;
	}
	// d_del_x == NULL: the gradient with respect to the block's input is not wanted (the first block of a network: nothing consumes it) -- the two data
	// gradients and the last norm gradient are not formed, the weight gradients are
	if (side) { st = wgrad_on_lane(g_out_b, ws->relu1, p->conv1, g->conv1, k, cin, cout, pads->have1 ? pads->pad1 : nullptr); if (st) return st; }
	if (!side || d_del_x) {
		st = conv2d_backward_batched(stream, g_out_b, ws->relu1, p->conv1, side ? nullptr : g->conv1, d_del_x ? sc->g_in : nullptr, sc->flip, batch, h, w, k, cin, cout, 1,
		                             pads && pads->have1 ? pads->pad1 : nullptr, pads && d_del_x ? pads->k1_bwd : nullptr, have_dy ? pads->dy_pad : nullptr);   // :1202-1205
		if (st) return st;
	}
	// residual connection through the 1x1 convolution, :1208-1220.  With a scratch of the caller's (pads->g_res) its data gradient is formed FIRST and rides
	// into del_x as the addend of the last norm gradient (the same d + addend per element as the separate add, :1219); without, one more pass adds it
	const bool res_first = cin != cout && d_del_x && pads && pads->g_res;
	if (side && cin != cout) { st = wgrad_on_lane(d_del_out, d_x, p->res, g->res, 1, cin, cout, nullptr); if (st) return st; }
	if (res_first) {
		// (sc->g_in still holds the first convolution's data gradient: the residual's goes to the scratch)
		st = bla_conv2d_backward_batched_f32(stream, d_del_out, d_x, p->res, side ? nullptr : g->res, pads->g_res, sc->flip, batch, h, w, 1, cin, cout, 1); if (st) return st;
	}
	if (d_del_x) {
		st = gn_ddx_b(stream, batch, sc->g_in, d_del_x, d_x, ws->mu1, ws->sd1, cin, group_size, hw, ws->relu1, cin == cout ? d_del_out : (res_first ? pads->g_res : nullptr));
		if (st) return st;
	}
	if (cin != cout && !res_first && !(side && !d_del_x)) {
		st = bla_conv2d_backward_batched_f32(stream, d_del_out, d_x, p->res, side ? nullptr : g->res, d_del_x ? sc->g_in : nullptr, sc->flip, batch, h, w, 1, cin, cout, 1); if (st) return st;
		if (d_del_x) return bla_add_f32(stream, d_del_x, sc->g_in, (size_t)batch * cin * hw);
	}
	return BLA_OK;
}
