// bla_conv.hip -- the data-movement stages of lib/conv.c (im2col / col2im / reshapes), the conv() and
// conv_ddx() compositions around the MFMA GEMM, and lib/norm.c's group norm.
//
// Layouts (device, fp32): an image is one contiguous [C][H][W] buffer (the reference's array of C Matrix
// structs, lib/conv.c:9-10, cifar_unet.c:255-262, holds the same values in the same order); kernels are
// [F][C][k][k]; the workspaces are the reference's ConvData members (lib/conv.h:6-11):
//     im2col [Ho*Wo][k*k*C], kernel_matrix [k*k*C][F], product [Ho*Wo][F], output [F][Ho][Wo].
// The index-only stages are bit-exact; col2im adds in the reference's order (so it is bit-identical to the
// fp32 instantiation of the oracle); the products go through bla_gemm_f32.
//
// Two identities remove data movement the reference pays for:
//   * kernel_matrix = kernels^T with kernels viewed as F x (k*k*C)  (lib/conv.c:138-153 is a transpose);
//   * reshape_channels_matrix / reshape_matrix_channels (lib/conv.c:174-203) are the (HW x C) <-> (C x HW)
//     transposes; both run on the LDS-tiled transpose kernel, and the matrix_transpose copies conv_ddx makes
//     around its two products (lib/conv.c:221-227) disappear into the GEMM's transa/transb.
#include "bla_internal.h"
#include <cmath>

namespace bla {

constexpr int kThreads = 256;

struct Geometry { int ho, wo, pt, pl; };

// TF "SAME" geometry exactly as lib/conv.c:13-28,55-56 computes it (ceil on a float quotient).
static Geometry same_geometry(int h, int w, int k, int s) {
	Geometry g;
	int vpad = (int)((ceil(((float)h) / s) - 1) * s + k - h);
	if (vpad < 0) vpad = 0;
	int hpad = (int)((ceil(((float)w) / s) - 1) * s + k - w);
	if (hpad < 0) hpad = 0;
	g.pt = vpad / 2; g.pl = hpad / 2;
	g.ho = (int)ceil((float)h / s); g.wo = (int)ceil((float)w / s);
	return g;
}

// out[(i*Wo+j)][c*k*k + p*k + q] = x[c][i*s+p-pt][j*s+q-pl] (0 outside)   -- lib/conv.c:58-74
__global__ void __launch_bounds__(kThreads) im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int h, int w, int k, int c_in,
                                                           int s, int ho, int wo, int pt, int pl) {
	const int kk = k * k, roww = kk * c_in;
	const size_t total = (size_t)ho * wo * roww;
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
		int col = (int)(e % roww);
		int r = (int)(e / roww);
		int c = col / kk, pq = col % kk, p = pq / k, q = pq % k;
		int i = r / wo, j = r % wo;
		int yy = i * s + p - pt, xx = j * s + q - pl;
		float v = 0.f;
		if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = x[((size_t)c * h + yy) * w + xx];
		out[e] = v;
	}
}

// Gather form of lib/conv.c:105-121,124-131 for stride 1: the reference scatters
// pad[c][i+p][j+q] += in[(i,j)][c,p,q] with i, j ascending, so a given output pixel receives its terms in
// order of ascending (i, j) = descending (p, q).  Summed in fp32 in that same order.
__global__ void __launch_bounds__(kThreads) col2im_s1_kernel(const float* __restrict__ cols, float* __restrict__ out, int h, int w, int k, int c_n,
                                                              int pt, int pl) {
	const int kk = k * k, roww = kk * c_n;
	const size_t total = (size_t)c_n * h * w;
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
		int xx = (int)(e % w), yy = (int)((e / w) % h), c = (int)(e / ((size_t)w * h));
		float acc = 0.f;
		for (int p = k - 1; p >= 0; p--) {
			int i = yy + pt - p;
			if (i < 0 || i >= h) continue;
			for (int q = k - 1; q >= 0; q--) {
				int j = xx + pl - q;
				if (j < 0 || j >= w) continue;
				acc += cols[((size_t)i * w + j) * roww + c * kk + p * k + q];
			}
		}
		out[e] = acc;
	}
}

// ---- group norm, lib/norm.c (quirk Q3 kept: epsilon is integer 0 and "stdev" is the variance) --------------
// One 1024-thread workgroup per group.  Groups of up to 1024*32 elements (the U-Net's 32 channels x 32x32) are read
// ONCE into registers and the mean / variance-about-the-mean / normalise passes of lib/norm.c:14-47 run on the
// register copy; larger groups re-read global memory per pass.  Sums are fp64.
constexpr int kGnThreads = 1024, kGnRegs = 32;

__device__ __forceinline__ double gn_block_sum(double v) {
	__shared__ double sh[kGnThreads / 64];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	__syncthreads();  // protect sh against the previous call's readers
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	double t = 0;
	for (int i = 0; i < kGnThreads / 64; i++) t += sh[i];
	return t;  // every thread gets the same total (same summation order)
}

__global__ void __launch_bounds__(kGnThreads) group_norm_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ stdevs,
                                                                 float* __restrict__ means, int channels, int group_size, int hw) {
	int g = blockIdx.x;
	int nch = min(group_size, channels - g * group_size);
	size_t off = (size_t)g * group_size * hw;
	int n = nch * hw;
	const int t = threadIdx.x;
	if (n <= kGnThreads * kGnRegs) {
		float v[kGnRegs];
		double s = 0;
#pragma unroll
		for (int i = 0; i < kGnRegs; i++) { int j = t + i * kGnThreads; v[i] = j < n ? in[off + j] : 0.f; s += v[i]; }
		float mean = (float)(gn_block_sum(s) / (double)n);
		double q = 0;
#pragma unroll
		for (int i = 0; i < kGnRegs; i++) { float d = t + i * kGnThreads < n ? v[i] - mean : 0.f; q += (double)d * d; }
		float var = (float)(gn_block_sum(q) / (double)n);
		if (t == 0) { means[g] = mean; stdevs[g] = var; }
#pragma unroll
		for (int i = 0; i < kGnRegs; i++) { int j = t + i * kGnThreads; if (j < n) out[off + j] = (v[i] - mean) / var; }   // (x - mean) / (stdev + 0), lib/norm.c:44
		return;
	}
	double s = 0;
	for (int i = t; i < n; i += kGnThreads) s += in[off + i];
	float mean = (float)(gn_block_sum(s) / (double)n);
	double q = 0;
	for (int i = t; i < n; i += kGnThreads) { float v = in[off + i] - mean; q += (double)v * v; }
	float var = (float)(gn_block_sum(q) / (double)n);
	if (t == 0) { means[g] = mean; stdevs[g] = var; }
	for (int i = t; i < n; i += kGnThreads) out[off + i] = (in[off + i] - mean) / var;
}

// lib/norm.c:52-93
__global__ void __launch_bounds__(kGnThreads) group_norm_ddx_kernel(const float* __restrict__ source, float* __restrict__ dest, const float* __restrict__ data,
                                                                     const float* __restrict__ means, const float* __restrict__ stdevs, int channels,
                                                                     int group_size, int hw) {
	int g = blockIdx.x;
	int nch = min(group_size, channels - g * group_size);
	size_t off = (size_t)g * group_size * hw;
	int n = nch * hw;
	const int t = threadIdx.x;
	float mean = means[g], sd = stdevs[g];
	if (n <= kGnThreads * (kGnRegs / 2)) {   // two register copies (source, normalised data): 16 each
		float sv[kGnRegs / 2], nv[kGnRegs / 2];
		double gs = 0, gws = 0;
#pragma unroll
		for (int i = 0; i < kGnRegs / 2; i++) {
			int j = t + i * kGnThreads;
			sv[i] = j < n ? source[off + j] : 0.f;
			nv[i] = j < n ? (data[off + j] - mean) / sd : 0.f;
			gs += sv[i]; gws += (double)nv[i] * sv[i];
		}
		float fgs = (float)(gn_block_sum(gs) / (double)n);
		float fgws = (float)(gn_block_sum(gws) / (double)n);
#pragma unroll
		for (int i = 0; i < kGnRegs / 2; i++) { int j = t + i * kGnThreads; if (j < n) dest[off + j] = (sv[i] - fgs - nv[i] * fgws) / sd; }
		return;
	}
	double gs = 0, gws = 0;
	for (int i = t; i < n; i += kGnThreads) {
		float wgt = (data[off + i] - mean) / sd;
		gs += source[off + i];
		gws += (double)wgt * source[off + i];
	}
	float fgs = (float)(gn_block_sum(gs) / (double)n);
	float fgws = (float)(gn_block_sum(gws) / (double)n);
	for (int i = t; i < n; i += kGnThreads) {
		float nv = (data[off + i] - mean) / sd;
		dest[off + i] = (source[off + i] - fgs - nv * fgws) / sd;
	}
}

static inline unsigned grid_for(size_t n) {
	size_t b = (n + kThreads - 1) / kThreads;
	if (b < 1) b = 1;
	if (b > 2048) b = 2048;
	return (unsigned)b;
}

}  // namespace bla

using namespace bla;

extern "C" {

bla_status bla_conv_out_hw(int h, int w, int stride, int* ho, int* wo) {
	BLA_REQUIRE(h > 0 && w > 0 && stride > 0 && ho && wo, BLA_ERR_INVALID, "bad conv geometry h=%d w=%d stride=%d", h, w, stride);
	Geometry g = same_geometry(h, w, 1, stride);
	*ho = g.ho; *wo = g.wo;
	return BLA_OK;
}

bla_status bla_im2col_f32(void* stream, const float* d_x, float* d_out, int h, int w, int k, int c_in, int stride) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(h > 0 && w > 0 && k > 0 && c_in > 0 && stride > 0, BLA_ERR_INVALID, "bad im2col shape h=%d w=%d k=%d c=%d s=%d", h, w, k, c_in, stride);
	BLA_REQUIRE(d_x && d_out, BLA_ERR_INVALID, "null operand");
	Geometry g = same_geometry(h, w, k, stride);
	size_t total = (size_t)g.ho * g.wo * k * k * c_in;
	hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(kThreads), 0, pick_stream(stream), d_x, d_out, h, w, k, c_in, stride, g.ho, g.wo, g.pt, g.pl);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_col2im_f32(void* stream, const float* d_cols, float* d_out, int h, int w, int k, int c_n, int stride) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(h > 0 && w > 0 && k > 0 && c_n > 0 && stride > 0, BLA_ERR_INVALID, "bad col2im shape h=%d w=%d k=%d c=%d s=%d", h, w, k, c_n, stride);
	if (stride != 1) {
		set_error("_col2im iterates the image grid with out_row = i*stride + k (lib/conv.c:80-135): out of bounds for stride %d, undefined in the reference", stride);
		return BLA_ERR_UNDEFINED;
	}
	BLA_REQUIRE(d_cols && d_out, BLA_ERR_INVALID, "null operand");
	Geometry g = same_geometry(h, w, k, 1);
	hipLaunchKernelGGL(col2im_s1_kernel, dim3(grid_for((size_t)c_n * h * w)), dim3(kThreads), 0, pick_stream(stream), d_cols, d_out, h, w, k, c_n, g.pt, g.pl);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

/* _reshape_kernels_matrix (lib/conv.c:138-153): [F][C*k*k] -> [C*k*k][F] is a transpose */
bla_status bla_kernels_to_matrix_f32(void* stream, const float* d_kern, float* d_mat, int f_n, int c_n, int k) {
	return bla_transpose_f32(stream, d_kern, d_mat, f_n, c_n * k * k);
}
/* _reshape_matrix_kernels (lib/conv.c:156-171) */
bla_status bla_matrix_to_kernels_f32(void* stream, const float* d_mat, float* d_kern, int f_n, int c_n, int k) {
	return bla_transpose_f32(stream, d_mat, d_kern, c_n * k * k, f_n);
}
/* reshape_channels_matrix AS WRITTEN (lib/conv.c:174-187): channels[c][idx] = matrix[idx*C + c] */
bla_status bla_reshape_channels_matrix_f32(void* stream, float* d_channels, const float* d_matrix, int c_n, int hw) {
	return bla_transpose_f32(stream, d_matrix, d_channels, hw, c_n);
}
/* reshape_matrix_channels AS WRITTEN (lib/conv.c:190-203): matrix[idx*C + c] = channels[c][idx] */
bla_status bla_reshape_matrix_channels_f32(void* stream, float* d_matrix, const float* d_channels, int c_n, int hw) {
	return bla_transpose_f32(stream, d_channels, d_matrix, c_n, hw);
}

/* conv(), lib/conv.c:205-212, with the intended last step (the GEMM result reaches `output`).
 * All four ConvData workspaces are filled like the reference fills them. */
bla_status bla_conv_forward_f32(void* stream, const float* d_x, const float* d_kern, float* d_im2col, float* d_kmat, float* d_product,
                                float* d_output, int h, int w, int k, int c_in, int f_n, int stride) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_x && d_kern && d_im2col && d_kmat && d_product && d_output, BLA_ERR_INVALID, "null operand");
	BLA_REQUIRE(f_n > 0, BLA_ERR_INVALID, "bad filter count %d", f_n);
	st = bla_im2col_f32(stream, d_x, d_im2col, h, w, k, c_in, stride);
	if (st) return st;
	Geometry g = same_geometry(h, w, k, stride);
	const int hw = g.ho * g.wo, kkc = k * k * c_in;
	st = bla_kernels_to_matrix_f32(stream, d_kern, d_kmat, f_n, c_in, k);
	if (st) return st;
	// product [HW x F] = im2col [HW x kkC] . kernels^T  (kernels viewed F x kkC: no pass over kernel_matrix needed)
	st = bla_gemm_f32(stream, 0, 1, hw, f_n, kkc, d_im2col, kkc, d_kern, kkc, d_product, f_n, nullptr);
	if (st) return st;
	// output [F x HW] <- product [HW x F]: the reference's final reshape (intended direction) is this transpose,
	// so output is bit-for-bit a re-indexing of product, as in the reference
	return bla_reshape_channels_matrix_f32(stream, d_output, d_product, f_n, hw);
}

/* conv_ddx(), lib/conv.c:214-229, with the intended first step (del_Y feeds del_Q); stride must be 1 (Q5).
 * im2col / kmat are the forward workspaces; del_q, del_kmat, del_col are the grad_data workspaces. */
bla_status bla_conv_backward_f32(void* stream, const float* d_del_y, const float* d_im2col, const float* d_kmat, float* d_del_q, float* d_del_kmat,
                                 float* d_del_kern, float* d_del_col, float* d_del_x, int h, int w, int k, int c_in, int f_n, int stride) {
	bla_status st = require_ready();
	if (st) return st;
	if (stride != 1) {
		set_error("conv_ddx is undefined for stride %d: _col2im is only valid for stride 1 (lib/conv.c:80-135)", stride);
		return BLA_ERR_UNDEFINED;
	}
	BLA_REQUIRE(d_del_y && d_im2col && d_kmat && d_del_q && d_del_kmat && d_del_kern && d_del_col && d_del_x, BLA_ERR_INVALID, "null operand");
	const int hw = h * w, kkc = k * k * c_in;
	st = bla_reshape_matrix_channels_f32(stream, d_del_q, d_del_y, f_n, hw);                 // del_Q [HW x F] <- del_Y [F x HW]
	if (st) return st;
	// del_kernel_matrix [kkC x F] = im2col^T . del_Q          (lib/conv.c:221-222 without the transpose copies)
	st = bla_gemm_f32(stream, 1, 0, kkc, f_n, hw, d_im2col, kkc, d_del_q, f_n, d_del_kmat, f_n, nullptr);
	if (st) return st;
	st = bla_matrix_to_kernels_f32(stream, d_del_kmat, d_del_kern, f_n, c_in, k);              // lib/conv.c:223 (a transpose)
	if (st) return st;
	// del_input_matrix [HW x kkC] = del_Q . kernel_matrix^T      (lib/conv.c:225-226)
	st = bla_gemm_f32(stream, 0, 1, hw, kkc, f_n, d_del_q, f_n, d_kmat, f_n, d_del_col, kkc, nullptr);
	if (st) return st;
	return bla_col2im_f32(stream, d_del_col, d_del_x, h, w, k, c_in, 1);                      // lib/conv.c:228
}

bla_status bla_group_norm_f32(void* stream, const float* d_in, float* d_out, float* d_stdevs, float* d_means, int channels, int group_size, int hw) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0, BLA_ERR_INVALID, "bad group_norm shape channels=%d group=%d hw=%d", channels, group_size, hw);
	BLA_REQUIRE(d_in && d_out && d_stdevs && d_means, BLA_ERR_INVALID, "null operand");
	int groups = (channels + group_size - 1) / group_size;
	hipLaunchKernelGGL(group_norm_kernel, dim3(groups), dim3(kGnThreads), 0, pick_stream(stream), d_in, d_out, d_stdevs, d_means, channels, group_size, hw);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_group_norm_ddx_f32(void* stream, const float* d_source, float* d_dest, const float* d_data, const float* d_means, const float* d_stdevs,
                                  int channels, int group_size, int hw) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0, BLA_ERR_INVALID, "bad group_norm shape channels=%d group=%d hw=%d", channels, group_size, hw);
	BLA_REQUIRE(d_source && d_dest && d_data && d_means && d_stdevs, BLA_ERR_INVALID, "null operand");
	int groups = (channels + group_size - 1) / group_size;
	hipLaunchKernelGGL(group_norm_ddx_kernel, dim3(groups), dim3(kGnThreads), 0, pick_stream(stream), d_source, d_dest, d_data, d_means, d_stdevs, channels,
	                   group_size, hw);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // extern "C"
