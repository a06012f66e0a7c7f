// bla_f64_conv.hip -- conv.h / norm.h / util.h in the reference's own element type (lib/matrix.h:4: double) for the -DBLA_FP64 build of the drop-in
// host layer: the index maps of lib/conv.c, the two products of conv() / conv_ddx() on the f64 MFMA GEMM of bla_f64.hip, group norm and its
// gradient (lib/norm.c), relu and the two softmaxes (lib/util.c).  Correctness build (<= 1e-12 against the reference's fp64 CPU results: only the
// order of additions differs); plain one-pass kernels, the fp32 kernels are the performance path.
#include "bla_internal.h"
#include <cmath>

using namespace bla;

namespace {
constexpr int kT = 256;
unsigned grid_of(size_t n) { size_t b = (n + kT - 1) / kT; return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }
struct Geo { int ho, wo, pt, pl; };
// TF "SAME" geometry exactly as lib/conv.c:13-28,55-56 computes it (ceil on a float quotient)
Geo same_geo(int h, int w, int k, int s) {
	Geo g;
	int vpad = (int)((ceil(((float)h) / s) - 1) * s + k - h); if (vpad < 0) vpad = 0;
	int hpad = (int)((ceil(((float)w) / s) - 1) * s + k - w); if (hpad < 0) hpad = 0;
	g.pt = vpad / 2; g.pl = hpad / 2;
	g.ho = (int)ceil((float)h / s); g.wo = (int)ceil((float)w / s);
	return g;
}

// lib/conv.c:58-74
__global__ void __launch_bounds__(kT) im2col_f64_kernel(const double* __restrict__ x, double* __restrict__ out, int h, int w, int k, int c_in, int s, int ho, int wo, int pt,
                                                        int pl) {
	const int kk = k * k, roww = kk * c_in;
	const size_t total = (size_t)ho * wo * roww;
	for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
		const int col = (int)(e % roww), r = (int)(e / roww);
		const int c = col / kk, pq = col % kk, p = pq / k, q = pq % k, i = r / wo, j = r % wo;
		const int yy = i * s + p - pt, xx = j * s + q - pl;
		out[e] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? x[((size_t)c * h + yy) * w + xx] : 0.0;
	}
}
// lib/conv.c:105-131 in gather form, terms in the reference's order (ascending (i, j)); stride != 1: the adjoint of _im2col (the reference is
// undefined there)
__global__ void __launch_bounds__(kT) col2im_f64_kernel(const double* __restrict__ cols, double* __restrict__ out, int h, int w, int k, int c_n, int s, int ho, int wo,
                                                        int pt, int pl) {
	const int kk = k * k, roww = kk * c_n;
	const size_t total = (size_t)c_n * h * w;
	for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
		const int xx = (int)(e % w), yy = (int)((e / w) % h), c = (int)(e / ((size_t)w * h));
		double acc = 0.0;
		for (int p = k - 1; p >= 0; p--) {
			const int iy = yy + pt - p;
			if (iy < 0 || iy % s) continue;
			const int i = iy / s;
			if (i >= ho) continue;
			for (int q = k - 1; q >= 0; q--) {
				const int jx = xx + pl - q;
				if (jx < 0 || jx % s) continue;
				const int j = jx / s;
				if (j >= wo) continue;
				acc += cols[((size_t)i * wo + j) * roww + c * kk + p * k + q];
			}
		}
		out[e] = acc;
	}
}

__device__ double block_sum(double v) {   // every thread gets the total (same order everywhere)
	__shared__ double sh[kT / 64];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	double t = 0;
	for (int i = 0; i < kT / 64; i++) t += sh[i];
	return t;
}
// lib/norm.c:5-50 (Q3 kept: epsilon is integer 0, "stdev" is the variance): one workgroup per group
__global__ void __launch_bounds__(kT) group_norm_f64_kernel(const double* __restrict__ in, double* __restrict__ out, double* __restrict__ stdevs, double* __restrict__ means,
                                                            int channels, int group_size, int hw) {
	const int g = blockIdx.x, nch = min(group_size, channels - g * group_size), n = nch * hw;
	const size_t off = (size_t)g * group_size * hw;
	double s = 0;
	for (int i = threadIdx.x; i < n; i += kT) s += in[off + i];
	const double mean = block_sum(s) / n;
	double q = 0;
	for (int i = threadIdx.x; i < n; i += kT) { const double v = in[off + i] - mean; q += v * v; }
	const double var = block_sum(q) / n;
	if (threadIdx.x == 0) { means[g] = mean; stdevs[g] = var; }
	for (int i = threadIdx.x; i < n; i += kT) out[off + i] = (in[off + i] - mean) / var;
}
// lib/norm.c:52-93
__global__ void __launch_bounds__(kT) group_norm_ddx_f64_kernel(const double* __restrict__ source, double* __restrict__ dest, const double* __restrict__ data,
                                                                const double* __restrict__ means, const double* __restrict__ stdevs, int channels, int group_size, int hw) {
	const int g = blockIdx.x, nch = min(group_size, channels - g * group_size), n = nch * hw;
	const size_t off = (size_t)g * group_size * hw;
	const double mean = means[g], sd = stdevs[g];
	double gs = 0, gws = 0;
	for (int i = threadIdx.x; i < n; i += kT) { const double sv = source[off + i]; gs += sv; gws += (data[off + i] - mean) / sd * sv; }
	gs = block_sum(gs) / n;
	gws = block_sum(gws) / n;
	for (int i = threadIdx.x; i < n; i += kT) dest[off + i] = (source[off + i] - gs - (data[off + i] - mean) / sd * gws) / sd;
}
__global__ void __launch_bounds__(kT) relu_f64_kernel(double* __restrict__ d, size_t n) {   // lib/util.c:7-13
	for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) if (d[i] < 0) d[i] = 0;
}
// lib/util.c:15-34 (per column; one thread walks a column) and :36-55 (per row; one wave per row)
__global__ void __launch_bounds__(kT) softmax_cols_f64_kernel(double* __restrict__ d, int rows, int cols) {
	const int c = blockIdx.x * kT + threadIdx.x;
	if (c >= cols) return;
	double mx = -INFINITY;
	for (int r = 0; r < rows; r++) mx = fmax(mx, d[(size_t)r * cols + c]);
	double s = 0;
	for (int r = 0; r < rows; r++) { const double e = exp(d[(size_t)r * cols + c] - mx); d[(size_t)r * cols + c] = e; s += e; }
	for (int r = 0; r < rows; r++) d[(size_t)r * cols + c] /= s;
}
__global__ void __launch_bounds__(kT) softmax_rows_f64_kernel(double* __restrict__ d, int rows, int cols) {
	const int r = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= rows) return;
	double* row = d + (size_t)r * cols;
	double mx = -INFINITY;
	for (int j = lane; j < cols; j += 64) mx = fmax(mx, row[j]);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
	double s = 0;
	for (int j = lane; j < cols; j += 64) { const double e = exp(row[j] - mx); row[j] = e; s += e; }
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
	for (int j = lane; j < cols; j += 64) row[j] /= s;
}
}  // namespace

#define BLA_ENTER()                  \
	bla_status st = require_ready(); \
	if (st) return st;

extern "C" {

bla_status bla_im2col_f64(void* stream, const double* d_x, double* d_out, int h, int w, int k, int c_in, int stride) {
	BLA_ENTER();
	BLA_REQUIRE(h > 0 && w > 0 && k > 0 && c_in > 0 && stride > 0 && d_x && d_out, BLA_ERR_INVALID, "bad im2col argument");
	const Geo g = same_geo(h, w, k, stride);
	hipLaunchKernelGGL(im2col_f64_kernel, dim3(grid_of((size_t)g.ho * g.wo * k * k * c_in)), dim3(kT), 0, pick_stream(stream), d_x, d_out, h, w, k, c_in, stride, g.ho, g.wo,
	                   g.pt, g.pl);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_col2im_f64(void* stream, const double* d_cols, double* d_out, int h, int w, int k, int c_n, int stride) {
	BLA_ENTER();
	BLA_REQUIRE(h > 0 && w > 0 && k > 0 && c_n > 0 && stride > 0 && d_cols && d_out, BLA_ERR_INVALID, "bad col2im argument");
	if (stride != 1) {
		const char* e = getenv("BLA_STRICT_REFERENCE");
		if (e && e[0] == '1') { set_error("_col2im is undefined for stride %d in the reference (lib/conv.c:80-135)", stride); return BLA_ERR_UNDEFINED; }
	}
	const Geo g = same_geo(h, w, k, stride);
	hipLaunchKernelGGL(col2im_f64_kernel, dim3(grid_of((size_t)c_n * h * w)), dim3(kT), 0, pick_stream(stream), d_cols, d_out, h, w, k, c_n, stride, g.ho, g.wo, g.pt, g.pl);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

/* the four reshapes of lib/conv.c:138-203 are transposes (directions as written) */
bla_status bla_kernels_to_matrix_f64(void* stream, const double* d_kern, double* d_mat, int f_n, int c_n, int k) { return bla_transpose_f64(stream, d_kern, d_mat, f_n, c_n * k * k); }
bla_status bla_matrix_to_kernels_f64(void* stream, const double* d_mat, double* d_kern, int f_n, int c_n, int k) { return bla_transpose_f64(stream, d_mat, d_kern, c_n * k * k, f_n); }
bla_status bla_reshape_channels_matrix_f64(void* stream, double* d_channels, const double* d_matrix, int c_n, int hw) { return bla_transpose_f64(stream, d_matrix, d_channels, hw, c_n); }
bla_status bla_reshape_matrix_channels_f64(void* stream, double* d_matrix, const double* d_channels, int c_n, int hw) { return bla_transpose_f64(stream, d_channels, d_matrix, c_n, hw); }

/* conv(), lib/conv.c:205-212 (intended last step), every ConvData workspace filled */
bla_status bla_conv_forward_f64(void* stream, const double* d_x, const double* d_kern, double* d_im2col, double* d_kmat, double* d_product, double* d_output, int h, int w,
                                int k, int c_in, int f_n, int stride) {
	BLA_ENTER();
	BLA_REQUIRE(d_x && d_kern && d_im2col && d_kmat && d_product && d_output && f_n > 0, BLA_ERR_INVALID, "bad conv argument");
	if ((st = bla_im2col_f64(stream, d_x, d_im2col, h, w, k, c_in, stride))) return st;
	const Geo g = same_geo(h, w, k, stride);
	const int hw = g.ho * g.wo, kkc = k * k * c_in;
	if ((st = bla_kernels_to_matrix_f64(stream, d_kern, d_kmat, f_n, c_in, k))) return st;
	if ((st = bla_gemm_f64(stream, 0, 0, hw, f_n, kkc, d_im2col, kkc, d_kmat, f_n, d_product, f_n, 1.0, 0.0))) return st;   /* im2col . kernel_matrix, :210 */
	return bla_reshape_channels_matrix_f64(stream, d_output, d_product, f_n, hw);
}

/* conv_ddx(), lib/conv.c:214-229 (intended first step); stride != 1: the adjoint (refused under BLA_STRICT_REFERENCE=1) */
bla_status bla_conv_backward_f64(void* stream, const double* d_del_y, const double* d_im2col, const double* d_kmat, double* d_del_q, double* d_del_kmat, double* d_del_kern,
                                 double* d_del_col, double* d_del_x, int h, int w, int k, int c_in, int f_n, int stride) {
	BLA_ENTER();
	BLA_REQUIRE(d_del_y && d_im2col && d_kmat && d_del_q && d_del_kmat && d_del_kern && d_del_col && d_del_x, BLA_ERR_INVALID, "null operand");
	const Geo g = same_geo(h, w, k, stride);
	const int hw = g.ho * g.wo, kkc = k * k * c_in;
	if ((st = bla_reshape_matrix_channels_f64(stream, d_del_q, d_del_y, f_n, hw))) return st;
	if ((st = bla_gemm_f64(stream, 1, 0, kkc, f_n, hw, d_im2col, kkc, d_del_q, f_n, d_del_kmat, f_n, 1.0, 0.0))) return st;   /* :221-222 */
	if ((st = bla_matrix_to_kernels_f64(stream, d_del_kmat, d_del_kern, f_n, c_in, k))) return st;                              /* :223 */
	if ((st = bla_gemm_f64(stream, 0, 1, hw, kkc, f_n, d_del_q, f_n, d_kmat, f_n, d_del_col, kkc, 1.0, 0.0))) return st;        /* :225-226 */
	return bla_col2im_f64(stream, d_del_col, d_del_x, h, w, k, c_in, stride);                                                   /* :228 */
}

bla_status bla_group_norm_f64(void* stream, const double* d_in, double* d_out, double* d_stdevs, double* d_means, int channels, int group_size, int hw) {
	BLA_ENTER();
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0 && d_in && d_out && d_stdevs && d_means, BLA_ERR_INVALID, "bad group_norm argument");
	hipLaunchKernelGGL(group_norm_f64_kernel, dim3((channels + group_size - 1) / group_size), dim3(kT), 0, pick_stream(stream), d_in, d_out, d_stdevs, d_means, channels,
	                   group_size, hw);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_group_norm_ddx_f64(void* stream, const double* d_source, double* d_dest, const double* d_data, const double* d_means, const double* d_stdevs, int channels,
                                  int group_size, int hw) {
	BLA_ENTER();
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0 && d_source && d_dest && d_data && d_means && d_stdevs, BLA_ERR_INVALID, "bad group_norm_ddx argument");
	hipLaunchKernelGGL(group_norm_ddx_f64_kernel, dim3((channels + group_size - 1) / group_size), dim3(kT), 0, pick_stream(stream), d_source, d_dest, d_data, d_means,
	                   d_stdevs, channels, group_size, hw);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_relu_f64(void* stream, double* d, size_t n) {
	BLA_ENTER();
	if (n == 0) return BLA_OK;
	BLA_REQUIRE(d, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(relu_f64_kernel, dim3(grid_of(n)), dim3(kT), 0, pick_stream(stream), d, n);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_softmax_cols_f64(void* stream, double* d, int rows, int cols) {
	BLA_ENTER();
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	if (rows == 0 || cols == 0) return BLA_OK;
	BLA_REQUIRE(d, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(softmax_cols_f64_kernel, dim3((cols + kT - 1) / kT), dim3(kT), 0, pick_stream(stream), d, rows, cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_softmax_rows_f64(void* stream, double* d, int rows, int cols) {
	BLA_ENTER();
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	if (rows == 0 || cols == 0) return BLA_OK;
	BLA_REQUIRE(d, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(softmax_rows_f64_kernel, dim3((rows + 3) / 4), dim3(kT), 0, pick_stream(stream), d, rows, cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // extern "C"
