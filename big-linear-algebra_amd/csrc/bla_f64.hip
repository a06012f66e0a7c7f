// bla_f64.hip -- the matrix.h path in the reference's own element type (lib/matrix.h:4: typedef double matrix_float_t), for the
// -DBLA_FP64 build of the drop-in host layer: GEMM on v_mfma_f64_16x16x4_f64 plus the elementwise / broadcast /
// transpose / reduction set of lib/matrix.c.  Purpose: the <= 1e-12 comparison mode against the reference's fp64 CPU results (only the
// order of additions differs) -- correctness first; the fp32 kernels are the performance path.
#include "bla_internal.h"
#include <cmath>

using namespace bla;

namespace {
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 64, TK = 16;   // 64 x 64 output tile per 256-thread workgroup, 16-deep slabs

// C[m x n] = alpha * op(A) . op(B) + beta * C.  Each of the four waves owns a 32 x 32 quarter = 2 x 2 blocks of 16 x 16;
// v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15], holds D[4 r + (l >> 4)][l & 15] in register r = 0..3.
__global__ void __launch_bounds__(256) gemm_f64_kernel(int transa, int transb, int M, int N, int K, const double* __restrict__ A, int lda,
                                                        const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc, double alpha, double beta) {
	__shared__ double As[TB][TK + 1];
	__shared__ double Bs[TK][TB + 1];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
	const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
	const int li = lane & 15, lq = lane >> 4;
	f64x4 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; i++)
#pragma unroll
		for (int j = 0; j < 2; j++)
#pragma unroll
			for (int r = 0; r < 4; r++) acc[i][j][r] = 0.0;
	for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
		for (int e = tid; e < TB * TK; e += 256) {
			const int r = e / TK, kk = e % TK;                       // A tile: k fastest (contiguous when !transa)
			const int gr = m0 + r, gk = k0 + kk;
			As[r][kk] = (gr < M && gk < K) ? (transa ? A[(size_t)gk * lda + gr] : A[(size_t)gr * lda + gk]) : 0.0;
			const int kb = e / TB, c = e % TB;                       // B tile: column fastest (contiguous when !transb)
			const int gc = n0 + c, gkb = k0 + kb;
			Bs[kb][c] = (gc < N && gkb < K) ? (transb ? B[(size_t)gc * ldb + gkb] : B[(size_t)gkb * ldb + gc]) : 0.0;
		}
		__syncthreads();
#pragma unroll
		for (int kk = 0; kk < TK; kk += 4) {
			double a[2], b[2];
#pragma unroll
			for (int i = 0; i < 2; i++) { a[i] = As[wm + i * 16 + li][kk + lq]; b[i] = Bs[kk + lq][wn + i * 16 + li]; }
#pragma unroll
			for (int i = 0; i < 2; i++)
#pragma unroll
				for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
		}
		__syncthreads();
	}
#pragma unroll
	for (int i = 0; i < 2; i++)
#pragma unroll
		for (int j = 0; j < 2; j++)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int row = m0 + wm + i * 16 + 4 * r + lq, col = n0 + wn + j * 16 + li;   // D register r of lane (li, lq) is row 4 r + lq
				if (row < M && col < N) {
					double* dst = C + (size_t)row * ldc + col;
					*dst = beta != 0.0 ? alpha * acc[i][j][r] + beta * *dst : alpha * acc[i][j][r];
				}
			}
}

enum { EW_SCALE, EW_ADD, EW_MUL };
template <int OP>
__global__ void __launch_bounds__(256) ew_f64_kernel(double* __restrict__ a, const double* __restrict__ b, double f, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		if (OP == EW_SCALE) a[i] *= f;
		else if (OP == EW_ADD) a[i] += b[i];
		else a[i] *= b[i];
	}
}
// a[r][c] += b[r][c % b_cols]  (lib/matrix.c:189-195) / a[r][c] += b[c]  (:199-205)
__global__ void __launch_bounds__(256) tile_f64_kernel(double* __restrict__ a, const double* __restrict__ b, int rows, int cols, int b_cols) {
	const size_t n = (size_t)rows * cols;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / cols), c = (int)(i % cols);
		a[i] += b_cols > 0 ? b[(size_t)r * b_cols + c % b_cols] : b[c];
	}
}
__global__ void __launch_bounds__(256) transpose_f64_kernel(const double* __restrict__ in, double* __restrict__ out, int rows, int cols) {
	__shared__ double t[32][33];
	const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	for (int j = ty; j < 32; j += 8)
		if (by + j < rows && bx + tx < cols) t[j][tx] = in[(size_t)(by + j) * cols + bx + tx];
	__syncthreads();
	for (int j = ty; j < 32; j += 8)
		if (bx + j < cols && by + tx < rows) out[(size_t)(bx + j) * rows + by + tx] = t[tx][j];
}
// out[i] = sum_{j < len} m[i * stride_i + j * stride_j], i < count: one workgroup per output, added in a fixed order
__global__ void __launch_bounds__(256) strided_sum_f64_kernel(const double* __restrict__ m, double* __restrict__ out, int len, size_t stride_i, size_t stride_j) {
	__shared__ double sh[256];
	const double* p = m + (size_t)blockIdx.x * stride_i;
	double s = 0.0;
	for (int j = threadIdx.x; j < len; j += 256) s += p[(size_t)j * stride_j];
	sh[threadIdx.x] = s;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
	if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}
// per-workgroup partials {sum, sum of squares, max} of a flat array (fixed order inside a workgroup)
__global__ void __launch_bounds__(256) stats_f64_kernel(const double* __restrict__ m, size_t n, double* __restrict__ part) {
	__shared__ double s1[256], s2[256], mx[256];
	double a = 0.0, b = 0.0, c = -INFINITY;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double v = m[i]; a += v; b += v * v; c = v > c ? v : c; }
	s1[threadIdx.x] = a; s2[threadIdx.x] = b; mx[threadIdx.x] = c;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; mx[threadIdx.x] = fmax(mx[threadIdx.x], mx[threadIdx.x + o]); }
		__syncthreads();
	}
	if (threadIdx.x == 0) { part[3 * blockIdx.x] = s1[0]; part[3 * blockIdx.x + 1] = s2[0]; part[3 * blockIdx.x + 2] = mx[0]; }
}
// what: 0 frobenius (sqrt of the sum of squares, lib/matrix.c:150-158), 1 max (:160-168), 2 z-score in place (:170-185: sigma = sqrtf(E[x^2] - mean^2))
__global__ void __launch_bounds__(256) finish_f64_kernel(const double* __restrict__ part, int blocks, int what, double* __restrict__ out, double* __restrict__ m, size_t n) {
	__shared__ double st[2];
	if (threadIdx.x == 0) {   // every workgroup folds the partials itself, in the same order: identical statistics everywhere
		double a = 0.0, b = 0.0, c = -INFINITY;
		for (int i = 0; i < blocks; i++) { a += part[3 * i]; b += part[3 * i + 1]; c = fmax(c, part[3 * i + 2]); }
		if (what == 0) out[0] = sqrt(b);
		else if (what == 1) out[0] = c;
		else { const double mean = a / (double)n; st[0] = mean; st[1] = (double)sqrtf((float)(b / (double)n - mean * mean)); }
	}
	if (what != 2) return;
	__syncthreads();
	const double mean = st[0], sd = st[1];
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m[i] = (m[i] - mean) / sd;
}
unsigned grid_for(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }
}  // namespace

extern "C" {

bla_status bla_gemm_f64(void* stream, int transa, int transb, int m, int n, int k, const double* d_a, int lda, const double* d_b, int ldb, double* d_c, int ldc,
                        double alpha, double beta) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(m >= 0 && n >= 0 && k >= 0, BLA_ERR_INVALID, "negative dimension m=%d n=%d k=%d", m, n, k);
	if (m == 0 || n == 0) return BLA_OK;
	BLA_REQUIRE(d_c && (k == 0 || (d_a && d_b)), BLA_ERR_INVALID, "null operand pointer");
	BLA_REQUIRE(lda >= (transa ? m : k) && ldb >= (transb ? k : n) && ldc >= n, BLA_ERR_INVALID, "leading dimension too small");
	hipLaunchKernelGGL(gemm_f64_kernel, dim3((n + TB - 1) / TB, (m + TB - 1) / TB), dim3(256), 0, pick_stream(stream), transa, transb, m, n, k, d_a, lda, d_b, ldb, d_c, ldc,
	                   alpha, beta);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

#define BLA_F64_EW(NAME, OP, BPTR, F)                                                                                   \
	bla_status st = require_ready();                                                                                   \
	if (st) return st;                                                                                                 \
	if (n == 0) return BLA_OK;                                                                                         \
	BLA_REQUIRE(d_a, BLA_ERR_INVALID, "null operand");                                                                 \
	hipLaunchKernelGGL((ew_f64_kernel<OP>), dim3(grid_for(n)), dim3(256), 0, pick_stream(stream), d_a, BPTR, F, n);    \
	BLA_HIP(hipGetLastError());                                                                                        \
	return BLA_OK;
bla_status bla_scale_f64(void* stream, double* d_a, size_t n, double f) { BLA_F64_EW(scale, EW_SCALE, nullptr, f) }
bla_status bla_add_f64(void* stream, double* d_a, const double* d_b, size_t n) { BLA_F64_EW(add, EW_ADD, d_b, 0.0) }
bla_status bla_hadamard_f64(void* stream, double* d_a, const double* d_b, size_t n) { BLA_F64_EW(hadamard, EW_MUL, d_b, 0.0) }
#undef BLA_F64_EW

bla_status bla_add_tile_columns_f64(void* stream, double* d_a, int a_rows, int a_cols, const double* d_b, int b_cols) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_a && d_b && a_rows >= 0 && a_cols >= 0 && b_cols > 0, BLA_ERR_INVALID, "bad argument");
	if ((size_t)a_rows * a_cols == 0) return BLA_OK;
	hipLaunchKernelGGL(tile_f64_kernel, dim3(grid_for((size_t)a_rows * a_cols)), dim3(256), 0, pick_stream(stream), d_a, d_b, a_rows, a_cols, b_cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
bla_status bla_add_tile_rows_f64(void* stream, double* d_a, int a_rows, int a_cols, const double* d_b) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_a && d_b && a_rows >= 0 && a_cols >= 0, BLA_ERR_INVALID, "bad argument");
	if ((size_t)a_rows * a_cols == 0) return BLA_OK;
	hipLaunchKernelGGL(tile_f64_kernel, dim3(grid_for((size_t)a_rows * a_cols)), dim3(256), 0, pick_stream(stream), d_a, d_b, a_rows, a_cols, 0);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
bla_status bla_transpose_f64(void* stream, const double* d_in, double* d_out, int rows, int cols) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_in && d_out && rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad argument");
	if ((size_t)rows * cols == 0) return BLA_OK;
	hipLaunchKernelGGL(transpose_f64_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, pick_stream(stream), d_in, d_out, rows, cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
/* matrix_row_sum, lib/matrix.c:123-133: 1 x cols, sums down every column */
bla_status bla_row_sum_f64(void* stream, const double* d_m, int rows, int cols, double* d_out) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_m && d_out && rows >= 0 && cols > 0, BLA_ERR_INVALID, "bad argument");
	hipLaunchKernelGGL(strided_sum_f64_kernel, dim3(cols), dim3(256), 0, pick_stream(stream), d_m, d_out, rows, (size_t)1, (size_t)cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
/* matrix_col_sum, lib/matrix.c:138-148: as written out[i] = sum_{j<cols} flat[i*rows + j] (BLA_ERR_UNDEFINED where that reads out of bounds), or true row sums */
bla_status bla_col_sum_f64(void* stream, const double* d_m, int rows, int cols, double* d_out, int mode) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_m && d_out && rows > 0 && cols >= 0, BLA_ERR_INVALID, "bad argument");
	if (mode == BLA_COLSUM_AS_WRITTEN && rows > cols) {
		set_error("matrix_col_sum as written reads out of bounds for a %d x %d matrix (rows > cols, lib/matrix.c:144)", rows, cols);
		return BLA_ERR_UNDEFINED;
	}
	hipLaunchKernelGGL(strided_sum_f64_kernel, dim3(rows), dim3(256), 0, pick_stream(stream), d_m, d_out, cols, (size_t)(mode == BLA_COLSUM_AS_WRITTEN ? rows : cols),
	                   (size_t)1);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
static bla_status stats_then(void* stream, double* d_m, size_t n, int what, double* d_out) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_m && (what == 2 || d_out), BLA_ERR_INVALID, "null operand");
	const int blocks = (int)grid_for(n > 0 ? n : 1) > 256 ? 256 : (int)grid_for(n > 0 ? n : 1);
	void* ws;
	st = ensure_workspace((size_t)blocks * 3 * sizeof(double), &ws);
	if (st) return st;
	hipStream_t s = pick_stream(stream);
	hipLaunchKernelGGL(stats_f64_kernel, dim3(blocks), dim3(256), 0, s, d_m, n, (double*)ws);
	hipLaunchKernelGGL(finish_f64_kernel, dim3(what == 2 ? blocks : 1), dim3(256), 0, s, (const double*)ws, blocks, what, d_out, d_m, n);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
bla_status bla_frobenius_f64(void* stream, const double* d_m, size_t n, double* d_out) { return stats_then(stream, const_cast<double*>(d_m), n, 0, d_out); }
bla_status bla_max_f64(void* stream, const double* d_m, size_t n, double* d_out) { return stats_then(stream, const_cast<double*>(d_m), n, 1, d_out); }
bla_status bla_zscore_f64(void* stream, double* d_m, size_t n) { return n ? stats_then(stream, d_m, n, 2, nullptr) : BLA_OK; }

}  // extern "C"
