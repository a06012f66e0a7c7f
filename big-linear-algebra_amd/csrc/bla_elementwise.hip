// bla_elementwise.hip -- HBM-bound kernels for the loops of lib/matrix.c:59-205 and lib/util.c:7-55
// (model/mnist_nn.c:38-73 carries identical copies of relu / softmax).
//
// All of these move 4..12 bytes per element and do O(1) flops per element: the roofline is HBM
// (8 TB/s spec, ~6.3 TB/s achievable), so the design rules are the coalescing ones:
//   * flat elementwise ops: 16 bytes per lane (float4) when the base is 16-byte aligned, grid capped at
//     8 workgroups per CU with a grid-stride loop;
//   * anything that walks a matrix "the short way" in the reference (column-major loops at
//     lib/matrix.c:150-158,189-195) is re-indexed so that consecutive lanes touch consecutive addresses;
//   * reductions accumulate in fp64 inside a thread, combine by wave shuffles (64 lanes) then LDS, and finish
//     in a fixed order (deterministic, no atomics); sums therefore sit closer to the fp64 reference than a
//     sequential fp32 chain would.
#include "bla_internal.h"
#include <cmath>

namespace bla {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8

static inline unsigned grid_for(size_t work_items) {
	size_t b = (work_items + kThreads - 1) / kThreads;
	if (b < 1) b = 1;
	if (b > kMaxBlocks) b = kMaxBlocks;
	return (unsigned)b;
}

// ---- flat elementwise ---------------------------------------------------------------------------
enum { OP_SCALE, OP_ADD, OP_MUL, OP_AXPY, OP_RELU, OP_RELU_DDX };

template <int OP>
__device__ __forceinline__ float ew_apply(float a, float b, float f) {
	switch (OP) {
		case OP_SCALE: return a * f;                   // lib/matrix.c:59-63
		case OP_ADD: return a + b;                     // lib/matrix.c:65-69
		case OP_MUL: return a * b;                     // lib/matrix.c:95-103
		case OP_AXPY: return a + f * b;                // matrix_scale(g, lr) then matrix_add(w, g): model/mnist_nn.c:303-315
		case OP_RELU: return a < 0.f ? 0.f : a;        // lib/util.c:7-13
		default: return a > 0.f ? 1.f : 0.f;           // model/mnist_nn.c:47-51
	}
}

template <int OP, bool BINARY>
__global__ void __launch_bounds__(kThreads) ew_kernel(float* __restrict__ a, const float* __restrict__ b, float f, size_t n, int vec_ok) {
	size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	size_t n4 = vec_ok ? n / 4 : 0;
	float4* a4 = reinterpret_cast<float4*>(a);
	const float4* b4 = reinterpret_cast<const float4*>(b);
	// one float4 per operand per iteration: measured faster than keeping 4 strided float4 in flight per lane
	// (3.7 vs 5.3 TB/s on the three-stream ops; occupancy already hides the latency)
	for (size_t i = tid; i < n4; i += stride) {
		float4 x = a4[i], y = BINARY ? b4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
		x.x = ew_apply<OP>(x.x, y.x, f); x.y = ew_apply<OP>(x.y, y.y, f);
		x.z = ew_apply<OP>(x.z, y.z, f); x.w = ew_apply<OP>(x.w, y.w, f);
		a4[i] = x;
	}
	for (size_t j = n4 * 4 + tid; j < n; j += stride) a[j] = ew_apply<OP>(a[j], BINARY ? b[j] : 0.f, f);
}

// Two reads and one write per element (matrix_add / hadamard / axpy, lib/matrix.c:65-69,95-103): two adjacent float4 per lane per stream (32 bytes) on a
// grid of two workgroups per CU.  Measured at 8192 x 8192 (round 3, tools/ew_bench.py): 5.32 TB/s algorithmic in the one-float4 grid-stride form below,
// 5.24 with non-temporal stores of the written stream, 5.36 with 32 bytes per lane, 5.40 with both, 5.52 with 32 bytes per lane on two workgroups per CU
// (this form); the one-read-one-write ops sit at 6.4 - 6.6 on the same part.
template <int OP>
__global__ void __launch_bounds__(kThreads) ew_binary_kernel(float* __restrict__ a, const float* __restrict__ b, float f, size_t n) {
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	float4* a4 = reinterpret_cast<float4*>(a);
	const float4* b4 = reinterpret_cast<const float4*>(b);
	const size_t n4 = n / 4, n8 = n4 / 2;
	auto op4 = [&](float4 x, float4 y) { return make_float4(ew_apply<OP>(x.x, y.x, f), ew_apply<OP>(x.y, y.y, f), ew_apply<OP>(x.z, y.z, f), ew_apply<OP>(x.w, y.w, f)); };
	for (size_t i = tid; i < n8; i += stride) {
		const float4 x0 = a4[2 * i], x1 = a4[2 * i + 1], y0 = b4[2 * i], y1 = b4[2 * i + 1];
		a4[2 * i] = op4(x0, y0); a4[2 * i + 1] = op4(x1, y1);
	}
	for (size_t i = n8 * 2 + tid; i < n4; i += stride) a4[i] = op4(a4[i], b4[i]);
	for (size_t j = n4 * 4 + tid; j < n; j += stride) a[j] = ew_apply<OP>(a[j], b[j], f);
}

template <int OP, bool BINARY>
static bla_status launch_ew(void* stream, float* a, const float* b, float f, size_t n) {
	bla_status st = require_ready();
	if (st) return st;
	if (n == 0) return BLA_OK;
	BLA_REQUIRE(a && (!BINARY || b), BLA_ERR_INVALID, "null operand");
	int vec_ok = ((uintptr_t)a % 16 == 0) && (!BINARY || (uintptr_t)b % 16 == 0);
	if (BINARY && vec_ok) {
		const size_t need = (n / 8 + kThreads - 1) / kThreads, cap = 2 * (size_t)(ctx().num_cus > 0 ? ctx().num_cus : 256);
		hipLaunchKernelGGL((ew_binary_kernel<OP>), dim3((unsigned)(need < 1 ? 1 : (need > cap ? cap : need))), dim3(kThreads), 0, pick_stream(stream), a, b, f, n);
		BLA_HIP(hipGetLastError());
		return BLA_OK;
	}
	hipLaunchKernelGGL((ew_kernel<OP, BINARY>), dim3(grid_for((n + 3) / 4)), dim3(kThreads), 0, pick_stream(stream), a, b, f, n, vec_ok);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

// ---- broadcasts: a[r][c] += b[r][c % b_cols] (tile columns) / a[r][c] += b[c] (tile rows) ----------------
__global__ void __launch_bounds__(kThreads) tile_columns_kernel(float* __restrict__ a, const float* __restrict__ b, int rows, int cols, int b_cols) {
	// blockIdx.y walks rows, blockIdx.x/threadIdx.x walk columns: no division per element; b_cols == 1 (a bias column,
	// model/mnist_nn.c:222) reads one scalar per row
	for (int r = blockIdx.y; r < rows; r += gridDim.y) {
		float* row = a + (size_t)r * cols;
		const float* brow = b + (size_t)r * b_cols;
		if (b_cols == 1) {
			float v = brow[0];
			for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) row[c] += v;
		} else {
			for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) row[c] += brow[c % b_cols];
		}
	}
}

// bias column (b_cols == 1), 16 bytes per lane: flat walk over float4 groups, one divide per group for the row
__global__ void __launch_bounds__(kThreads) bias_column_vec_kernel(float* __restrict__ a, const float* __restrict__ b, int rows, int cols) {
	const unsigned c4 = (unsigned)cols / 4;
	const size_t n4 = (size_t)rows * c4;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		float v = b[i / c4];
		float4 x = reinterpret_cast<float4*>(a)[i];
		x.x += v; x.y += v; x.z += v; x.w += v;
		reinterpret_cast<float4*>(a)[i] = x;
	}
}

__global__ void __launch_bounds__(kThreads) tile_rows_kernel(float* __restrict__ a, const float* __restrict__ b, int rows, int cols) {
	size_t n = (size_t)rows * cols;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i % cols];
}

// ---- transpose: 64x64 tile through LDS, stride 65 so both the row-wise write and the column-wise read are
// conflict-free for ds_*_b32 (bank = dword index mod 32 per 32-lane half) ---------------------------------
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
	__shared__ float tile[64][65];
	int tiles_c = (cols + 63) / 64;
	int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
	int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;  // 64 x 4
	for (int y = ly; y < 64; y += 4) {
		int r = tr * 64 + y, c = tc * 64 + lx;
		if (r < rows && c < cols) tile[y][lx] = in[(size_t)r * cols + c];
	}
	__syncthreads();
	for (int y = ly; y < 64; y += 4) {
		int c = tc * 64 + y, r = tr * 64 + lx;   // out is cols x rows
		if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[lx][y];
	}
}

// 16-byte global loads and stores on both sides (needs rows % 4 == 0, cols % 4 == 0, 16-byte aligned bases)
__global__ void __launch_bounds__(256) transpose_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
	__shared__ float tile[64][65];
	int tiles_c = (cols + 63) / 64, tiles_r = (rows + 63) / 64;
	// diagonal walk: consecutive workgroups move down AND right, so neither their reads nor their writes line up on one column of
	// tiles (at a power-of-two pitch a column of tiles sits on a few HBM channels)
	int tr = blockIdx.x % tiles_r, tc = (blockIdx.x / tiles_r + tr) % tiles_c;
	int q = threadIdx.x & 15, rr = threadIdx.x >> 4;  // 16 chunks of 4 floats x 16 rows per pass
#pragma unroll
	for (int i = 0; i < 4; i++) {
		int y = rr + 16 * i, r = tr * 64 + y, c = tc * 64 + q * 4;
		if (r < rows && c < cols) {
			float4 v = *reinterpret_cast<const float4*>(in + (size_t)r * cols + c);
			tile[y][q * 4 + 0] = v.x; tile[y][q * 4 + 1] = v.y; tile[y][q * 4 + 2] = v.z; tile[y][q * 4 + 3] = v.w;
		}
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < 4; i++) {
		int y = rr + 16 * i, c = tc * 64 + y, r = tr * 64 + q * 4;   // out row c, out columns r..r+3
		if (c < cols && r < rows) {
			float4 v = make_float4(tile[q * 4 + 0][y], tile[q * 4 + 1][y], tile[q * 4 + 2][y], tile[q * 4 + 3][y]);
			*reinterpret_cast<float4*>(out + (size_t)c * rows + r) = v;
		}
	}
}

// ---- reductions ------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
	return v;
}

// block-wide sum of (s0, s1), result valid in thread 0
__device__ __forceinline__ void block_sum2(double& s0, double& s1) {
	__shared__ double sh[2][kThreads / 64];
	s0 = wave_sum(s0); s1 = wave_sum(s1);
	int w = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) { sh[0][w] = s0; sh[1][w] = s1; }
	__syncthreads();
	if (threadIdx.x == 0) {
		s0 = 0; s1 = 0;
		for (int i = 0; i < kThreads / 64; i++) { s0 += sh[0][i]; s1 += sh[1][i]; }
	}
	__syncthreads();
}

enum { RED_SUM_SQ = 0, RED_MAX = 1, RED_SUM_AND_SQ = 2 };

// stage 1: per-block partials (sum, sum of squares, max); stage 2 (one block) folds them in block order
__global__ void __launch_bounds__(kThreads) reduce_stage1_kernel(const float* __restrict__ m, size_t n, double* __restrict__ part, int vec_ok) {
	double s = 0, q = 0;
	float mx = -INFINITY;
	size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	size_t n4 = vec_ok ? n / 4 : 0;
	const float4* m4 = reinterpret_cast<const float4*>(m);
	for (size_t i = tid; i < n4; i += stride) {   // 16 B per lane; 4 elements are combined in fp32 before joining the fp64 running sums
		float4 v = m4[i];
		s += (double)((v.x + v.y) + (v.z + v.w));
		q += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
		mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
	}
	for (size_t i = n4 * 4 + tid; i < n; i += stride) {
		float v = m[i];
		s += v; q += (double)v * v; mx = fmaxf(mx, v);
	}
	__shared__ float shm[kThreads / 64];
	mx = wave_max(mx);
	if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
	block_sum2(s, q);
	if (threadIdx.x == 0) {
		for (int i = 0; i < kThreads / 64; i++) mx = fmaxf(mx, shm[i]);
		part[3 * blockIdx.x + 0] = s; part[3 * blockIdx.x + 1] = q; part[3 * blockIdx.x + 2] = mx;
	}
}

// out[0] = result of `what`; for zscore also out[1], out[2] = mean, stdev (sqrtf on the variance as lib/matrix.c:179)
__global__ void __launch_bounds__(64) reduce_stage2_kernel(const double* __restrict__ part, int nparts, size_t n, int what, float* __restrict__ out) {
	double s = 0, q = 0;
	float mx = -INFINITY;
	for (int i = threadIdx.x; i < nparts; i += 64) { s += part[3 * i]; q += part[3 * i + 1]; mx = fmaxf(mx, (float)part[3 * i + 2]); }
	s = wave_sum(s); q = wave_sum(q); mx = wave_max(mx);
	if (threadIdx.x == 0) {
		if (what == RED_SUM_SQ) out[0] = (float)sqrt(q);          // frobenius_norm, lib/matrix.c:150-158
		else if (what == RED_MAX) out[0] = mx;                    // max_value, lib/matrix.c:160-168
		else {                                                    // matrix_z_score_normalize, lib/matrix.c:170-185
			double mean = s / (double)n;
			out[0] = (float)mean;
			out[1] = sqrtf((float)(q / (double)n - mean * mean));
		}
	}
}

__global__ void __launch_bounds__(kThreads) zscore_apply_kernel(float* __restrict__ m, size_t n, const float* __restrict__ stats) {
	float mean = stats[0], sd = stats[1];
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m[i] = (m[i] - mean) / sd;
}

// out[c] = sum_r m[r*cols + c]  (matrix_row_sum, lib/matrix.c:123-133): lanes across columns (coalesced), the rows are cut
// into gridDim.y chunks so that tall matrices fill the chip; chunk partials (fp64) are folded in chunk order.
__global__ void __launch_bounds__(kThreads) row_sum_partial_kernel(const float* __restrict__ m, int rows, int cols, int rows_per_chunk, double* __restrict__ part) {
	int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= cols) return;
	int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
	double s = 0;
	for (int r = r0; r < r1; r++) s += m[(size_t)r * cols + c];
	part[(size_t)blockIdx.y * cols + c] = s;
}

// 16-byte variant (cols % 4 == 0, 16-byte aligned base): a thread owns four columns, eight rows of loads in flight
__global__ void __launch_bounds__(kThreads) row_sum_partial_vec_kernel(const float* __restrict__ m, int rows, int cols, int rows_per_chunk, double* __restrict__ part) {
	int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (c >= cols) return;
	int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
	double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
	const float* col = m + c;
	int r = r0;
	for (; r + 8 <= r1; r += 8) {
		float4 v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4*>(col + (size_t)(r + u) * cols);
#pragma unroll
		for (int u = 0; u < 8; u++) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }   // row order, as the reference adds
	}
	for (; r < r1; r++) {
		float4 v = *reinterpret_cast<const float4*>(col + (size_t)r * cols);
		s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
	}
	double* o = part + (size_t)blockIdx.y * cols + c;
	o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
}

__global__ void __launch_bounds__(kThreads) row_sum_final_kernel(const double* __restrict__ part, int chunks, int cols, float* __restrict__ out) {
	int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= cols) return;
	double s = 0;
	for (int k = 0; k < chunks; k++) s += part[(size_t)k * cols + c];
	out[c] = (float)s;
}

// out[i] = sum_{j<len} m[i*stride + j]: one wave per output.
//   stride = cols -> true row sums (the documented intent of matrix_col_sum, lib/matrix.c:135-137)
//   stride = rows -> matrix_col_sum AS WRITTEN (lib/matrix.c:138-148), defined only when rows <= cols
__global__ void __launch_bounds__(kThreads) window_sum_kernel(const float* __restrict__ m, int count, int len, int stride, float* __restrict__ out) {
	int i = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (i >= count) return;
	double s = 0;
	for (int j = lane; j < len; j += 64) s += m[(size_t)i * stride + j];
	s = wave_sum(s);
	if (lane == 0) out[i] = (float)s;
}

// ---- softmax -----------------------------------------------------------------------------------------
// per column (lib/util.c:15-34).  A workgroup owns 32 columns; its 1024 threads are 32 column-lanes x 32 row-lanes
// (a wave covers 2 rows x 32 consecutive columns = two 128-byte segments).  Pass 1: each thread keeps an online
// (max, sum exp) over its rows, the 32 row-lanes of a column are merged through LDS; pass 2 writes exp(x-M)/S.
// 2 reads + 1 write per element instead of the reference's 3 reads + 2 writes.  Optional fused tail for the
// trainer: grad = (softmax - y) * scale (model/mnist_nn.c:263-268).
__global__ void __launch_bounds__(1024) softmax_cols_kernel(float* __restrict__ d, int rows, int cols, const float* __restrict__ y, float scale, float* __restrict__ grad) {
	__shared__ float sh_m[32][33], sh_s[32][33];
	const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
	const int c = blockIdx.x * 32 + cx;
	float mx = -INFINITY, sum = 0.f;
	if (c < cols) {
		for (int r = ry; r < rows; r += 32) {
			float v = d[(size_t)r * cols + c];
			float nm = fmaxf(mx, v);
			sum = sum * expf(mx - nm) + expf(v - nm);   // first term is 0 * exp(-inf - v) = 0 on the first row
			mx = nm;
		}
	}
	sh_m[ry][cx] = mx; sh_s[ry][cx] = sum;
	__syncthreads();
	float M = -INFINITY;
	for (int k = 0; k < 32; k++) M = fmaxf(M, sh_m[k][cx]);
	float S = 0.f;
	for (int k = 0; k < 32; k++) { float mk = sh_m[k][cx]; if (mk > -INFINITY) S += sh_s[k][cx] * expf(mk - M); }
	if (c < cols) {
		for (int r = ry; r < rows; r += 32) {
			size_t i = (size_t)r * cols + c;
			float p = expf(d[i] - M) / S;
			d[i] = p;
			if (grad) grad[i] = (p - y[i]) * scale;
		}
	}
}

// Tall matrices (rows >= 256, cols % 4 == 0): the strided walk above touches 128 bytes per row per workgroup.  Streaming
// variant in three launches, every access 16 bytes per lane along the rows:
//   stats   : a thread owns 4 columns over a chunk of rows, online (max, sum exp) -> partial[chunk][col]
//   merge   : partials of a column folded in chunk order -> (M, S)
//   apply   : p = exp(x - M) / S  (and the fused loss-gradient tail), plain elementwise stream
// Same 2 reads + 1 write per element, but coalesced: the bound is the HBM stream, not the access pattern.
__device__ __forceinline__ void online_update(float& mx, float& sum, float v) {
	float nm = fmaxf(mx, v);
	sum = sum * expf(mx - nm) + expf(v - nm);   // exp(-inf) = 0 covers the first element
	mx = nm;
}

__global__ void __launch_bounds__(kThreads) softmax_cols_stats_kernel(const float* __restrict__ d, int rows, int cols, int rows_per_chunk, float2* __restrict__ part) {
	int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (c >= cols) return;
	int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
	float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
	const float* col = d + c;
	int r = r0;
	// eight rows at a time: one rescale of the running sum per group (9 exponentials per 8 elements instead of 16)
	auto group = [](float& m, float& sum, const float (&x)[8]) {
		float g = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
		float nm = fmaxf(m, g);
		float acc = sum * expf(m - nm);
#pragma unroll
		for (int u = 0; u < 8; u++) acc += expf(x[u] - nm);
		sum = acc; m = nm;
	};
	for (; r + 8 <= r1; r += 8) {
		float4 v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4*>(col + (size_t)(r + u) * cols);
		float x[8];
#pragma unroll
		for (int u = 0; u < 8; u++) x[u] = v[u].x;
		group(m0, s0, x);
#pragma unroll
		for (int u = 0; u < 8; u++) x[u] = v[u].y;
		group(m1, s1, x);
#pragma unroll
		for (int u = 0; u < 8; u++) x[u] = v[u].z;
		group(m2, s2, x);
#pragma unroll
		for (int u = 0; u < 8; u++) x[u] = v[u].w;
		group(m3, s3, x);
	}
	for (; r < r1; r++) {
		float4 v = *reinterpret_cast<const float4*>(col + (size_t)r * cols);
		online_update(m0, s0, v.x); online_update(m1, s1, v.y); online_update(m2, s2, v.z); online_update(m3, s3, v.w);
	}
	float2* o = part + (size_t)blockIdx.y * cols + c;
	o[0] = make_float2(m0, s0); o[1] = make_float2(m1, s1); o[2] = make_float2(m2, s2); o[3] = make_float2(m3, s3);
}

__global__ void __launch_bounds__(kThreads) softmax_cols_merge_kernel(const float2* __restrict__ part, int chunks, int cols, float2* __restrict__ stats) {
	int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= cols) return;
	float M = -INFINITY;
	for (int k = 0; k < chunks; k++) M = fmaxf(M, part[(size_t)k * cols + c].x);
	float S = 0.f;
	for (int k = 0; k < chunks; k++) { float2 q = part[(size_t)k * cols + c]; if (q.x > -INFINITY) S += q.y * expf(q.x - M); }
	stats[c] = make_float2(M, 1.f / S);   // the apply pass multiplies (one rounding away from the reference's division)
}

__global__ void __launch_bounds__(kThreads) softmax_cols_apply_kernel(float* __restrict__ d, int rows, int cols, const float2* __restrict__ stats,
                                                                      const float* __restrict__ y, float scale, float* __restrict__ grad) {
	const int c4 = cols / 4;
	const size_t n4 = (size_t)rows * c4;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		int c = (int)(i % c4) * 4;
		float4 v = reinterpret_cast<float4*>(d)[i];
		const float4 a = *reinterpret_cast<const float4*>(stats + c), b = *reinterpret_cast<const float4*>(stats + c + 2);   // (M0,1/S0,M1,1/S1) (M2,1/S2,M3,1/S3)
		v.x = expf(v.x - a.x) * a.y; v.y = expf(v.y - a.z) * a.w; v.z = expf(v.z - b.x) * b.y; v.w = expf(v.w - b.z) * b.w;
		reinterpret_cast<float4*>(d)[i] = v;
		if (grad) {
			float4 t = reinterpret_cast<const float4*>(y)[i];
			t.x = (v.x - t.x) * scale; t.y = (v.y - t.y) * scale; t.z = (v.z - t.z) * scale; t.w = (v.w - t.w) * scale;
			reinterpret_cast<float4*>(grad)[i] = t;
		}
	}
}

// Up to 4096 rows (cols % 16 == 0): a 1024-thread workgroup keeps its strip of 16 columns IN REGISTERS -- thread (row group t / 4, column quad t % 4) holds
// rows t / 4 + 256 i as float4 -- so the matrix is read once and written once (8 bytes per element, the algorithmic count; the streaming form above
// reads it twice: 12).  Column maximum and sum of exponentials: each thread over its own rows, the 16 lanes of a wave that share a column quad by
// shuffles, the 16 waves through LDS in wave order.  The strips are dealt so that neighbours (which share 128-byte lines: a strip's row segment is 64 bytes)
// are neighbours in one XCD's dispatch order.
template <int RPT>
__global__ void __launch_bounds__(1024) softmax_cols_strip_kernel(float* __restrict__ d, int rows, int cols, const float* __restrict__ y, float scale,
                                                                   float* __restrict__ grad) {
	__shared__ float sh[2][16][16];   // [max | sum][wave][column of the strip]
	const int t = threadIdx.x, quad = t & 3, rg = t >> 2, wave = t >> 6, lane = t & 63;
	const int strips = cols / 16, per_xcd = strips / 8;
	const int strip = (strips % 8 == 0) ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
	float* base = d + (size_t)strip * 16 + quad * 4;
	float4 v[RPT];
#pragma unroll
	for (int i = 0; i < RPT; i++) {
		const int r = rg + 256 * i;
		v[i] = r < rows ? *reinterpret_cast<const float4*>(base + (size_t)r * cols) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
	}
	float4 m = v[0];
#pragma unroll
	for (int i = 1; i < RPT; i++) { m.x = fmaxf(m.x, v[i].x); m.y = fmaxf(m.y, v[i].y); m.z = fmaxf(m.z, v[i].z); m.w = fmaxf(m.w, v[i].w); }
#pragma unroll
	for (int o = 4; o < 64; o <<= 1) {   // the 16 lanes of this wave with the same column quad
		m.x = fmaxf(m.x, __shfl_xor(m.x, o, 64)); m.y = fmaxf(m.y, __shfl_xor(m.y, o, 64)); m.z = fmaxf(m.z, __shfl_xor(m.z, o, 64)); m.w = fmaxf(m.w, __shfl_xor(m.w, o, 64));
	}
	if (lane < 4) *reinterpret_cast<float4*>(&sh[0][wave][lane * 4]) = m;
	__syncthreads();
	float4 M = *reinterpret_cast<const float4*>(&sh[0][0][quad * 4]);
#pragma unroll
	for (int w = 1; w < 16; w++) {
		const float4 q = *reinterpret_cast<const float4*>(&sh[0][w][quad * 4]);
		M.x = fmaxf(M.x, q.x); M.y = fmaxf(M.y, q.y); M.z = fmaxf(M.z, q.z); M.w = fmaxf(M.w, q.w);
	}
	float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
	for (int i = 0; i < RPT; i++) {      // exp(-inf) = 0 for the rows past the end
		v[i].x = expf(v[i].x - M.x); v[i].y = expf(v[i].y - M.y); v[i].z = expf(v[i].z - M.z); v[i].w = expf(v[i].w - M.w);
		sum.x += v[i].x; sum.y += v[i].y; sum.z += v[i].z; sum.w += v[i].w;
	}
#pragma unroll
	for (int o = 4; o < 64; o <<= 1) {
		sum.x += __shfl_xor(sum.x, o, 64); sum.y += __shfl_xor(sum.y, o, 64); sum.z += __shfl_xor(sum.z, o, 64); sum.w += __shfl_xor(sum.w, o, 64);
	}
	if (lane < 4) *reinterpret_cast<float4*>(&sh[1][wave][lane * 4]) = sum;
	__syncthreads();
	float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
	for (int w = 0; w < 16; w++) {       // wave order: the same total in every thread
		const float4 q = *reinterpret_cast<const float4*>(&sh[1][w][quad * 4]);
		S.x += q.x; S.y += q.y; S.z += q.z; S.w += q.w;
	}
#pragma unroll
	for (int i = 0; i < RPT; i++) {
		const int r = rg + 256 * i;
		if (r < rows) {
			const float4 p = make_float4(v[i].x / S.x, v[i].y / S.y, v[i].z / S.z, v[i].w / S.w);   // exp(x - max) / sum, lib/util.c:26-31
			const size_t off = (size_t)r * cols;
			*reinterpret_cast<float4*>(base + off) = p;
			if (grad) {
				const float4 tt = *reinterpret_cast<const float4*>(y + (size_t)strip * 16 + quad * 4 + off);
				*reinterpret_cast<float4*>(grad + (size_t)strip * 16 + quad * 4 + off) = make_float4((p.x - tt.x) * scale, (p.y - tt.y) * scale, (p.z - tt.z) * scale, (p.w - tt.w) * scale);
			}
		}
	}
}

static bla_status softmax_cols_dispatch(void* stream, float* d, int rows, int cols, const float* y, float scale, float* grad) {
	hipStream_t s = pick_stream(stream);
	const bool al = ((uintptr_t)d | (uintptr_t)y | (uintptr_t)grad) % 16 == 0;
	if (rows >= 256 && rows <= 4096 && cols % 16 == 0 && cols >= 1024 && al) {   // (enough strips to fill the chip: 64 of them are a quarter of the CUs)
		const dim3 grid((unsigned)(cols / 16));
		if (rows <= 1024) hipLaunchKernelGGL(softmax_cols_strip_kernel<4>, grid, dim3(1024), 0, s, d, rows, cols, y, scale, grad);
		else if (rows <= 2048) hipLaunchKernelGGL(softmax_cols_strip_kernel<8>, grid, dim3(1024), 0, s, d, rows, cols, y, scale, grad);
		else hipLaunchKernelGGL(softmax_cols_strip_kernel<16>, grid, dim3(1024), 0, s, d, rows, cols, y, scale, grad);
		BLA_HIP(hipGetLastError());
		return BLA_OK;
	}
	if (rows >= 256 && cols % 4 == 0 && al) {
		unsigned vx = (unsigned)((cols / 4 + kThreads - 1) / kThreads);
		int chunks = (int)((ctx().num_cus > 0 ? ctx().num_cus : 256) / vx);   // one workgroup per CU, as for the column sums
		if (chunks > rows / 32) chunks = rows / 32;
		if (chunks < 1) chunks = 1;
		int rpc = (rows + chunks - 1) / chunks;
		chunks = (rows + rpc - 1) / rpc;
		void* ws;
		bla_status st = ensure_workspace(((size_t)chunks + 1) * cols * sizeof(float2), &ws);
		if (st) return st;
		float2* part = (float2*)ws; float2* stats = part + (size_t)chunks * cols;
		hipLaunchKernelGGL(softmax_cols_stats_kernel, dim3(vx, chunks), dim3(kThreads), 0, s, d, rows, cols, rpc, part);
		hipLaunchKernelGGL(softmax_cols_merge_kernel, dim3((cols + kThreads - 1) / kThreads), dim3(kThreads), 0, s, part, chunks, cols, stats);
		hipLaunchKernelGGL(softmax_cols_apply_kernel, dim3(grid_for((size_t)rows * (cols / 4))), dim3(kThreads), 0, s, d, rows, cols, stats, y, scale, grad);
	} else {
		hipLaunchKernelGGL(softmax_cols_kernel, dim3((cols + 31) / 32), dim3(1024), 0, s, d, rows, cols, y, scale, grad);
	}
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

// One workgroup per row, the row held in registers (T threads x 8 float4 = up to 32*T elements): one read + one write.
template <int T>
__global__ void __launch_bounds__(T) softmax_row_block_kernel(float* __restrict__ d, int rows, int cols) {
	__shared__ float sh_m[T / 64];
	__shared__ double sh_s[T / 64];
	float* row = d + (size_t)blockIdx.x * cols;
	const int c4 = cols / 4;   // cols % 4 == 0
	float4 v[8];
	float mx = -INFINITY;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		int j = threadIdx.x + T * i;
		if (j < c4) { v[i] = reinterpret_cast<const float4*>(row)[j]; mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w))); }
	}
	mx = wave_max(mx);
	if ((threadIdx.x & 63) == 0) sh_m[threadIdx.x >> 6] = mx;
	__syncthreads();
	mx = sh_m[0];
#pragma unroll
	for (int w = 1; w < T / 64; w++) mx = fmaxf(mx, sh_m[w]);
	double s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		int j = threadIdx.x + T * i;
		if (j < c4) {
			v[i].x = expf(v[i].x - mx); v[i].y = expf(v[i].y - mx); v[i].z = expf(v[i].z - mx); v[i].w = expf(v[i].w - mx);
			s += (double)v[i].x + (double)v[i].y + (double)v[i].z + (double)v[i].w;
		}
	}
	s = wave_sum(s);
	if ((threadIdx.x & 63) == 0) sh_s[threadIdx.x >> 6] = s;
	__syncthreads();
	double tot = 0;
#pragma unroll
	for (int w = 0; w < T / 64; w++) tot += sh_s[w];
	const float sf = (float)tot;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		int j = threadIdx.x + T * i;
		if (j < c4) { v[i].x /= sf; v[i].y /= sf; v[i].z /= sf; v[i].w /= sf; reinterpret_cast<float4*>(row)[j] = v[i]; }
	}
}

// per row (lib/util.c:36-55): one wave per row.  Rows of up to 64*32 = 2048 elements are held in registers
// (one read + one write per element); longer rows take the three-pass route of the reference.
template <int NI>   // rows of up to 64 * NI elements stay in registers
__device__ __forceinline__ void softmax_row_in_registers(float* __restrict__ row, int cols, int lane) {
	float v[NI];
	float mx = -INFINITY;
#pragma unroll
	for (int i = 0; i < NI; i++) {
		int j = lane + 64 * i;
		v[i] = j < cols ? row[j] : -INFINITY;
		mx = fmaxf(mx, v[i]);
	}
	mx = __shfl(wave_max(mx), 0, 64);
	double s = 0;
#pragma unroll
	for (int i = 0; i < NI; i++) {
		v[i] = lane + 64 * i < cols ? expf(v[i] - mx) : 0.f;
		s += v[i];
	}
	float sf = (float)__shfl(wave_sum(s), 0, 64);
#pragma unroll
	for (int i = 0; i < NI; i++) {
		int j = lane + 64 * i;
		if (j < cols) row[j] = v[i] / sf;
	}
}
__global__ void __launch_bounds__(kThreads) softmax_rows_kernel(float* __restrict__ d, int rows, int cols) {
	int r = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= rows) return;
	float* row = d + (size_t)r * cols;
	if (cols <= 256) { softmax_row_in_registers<4>(row, cols, lane); return; }     // the attention block's 256-token rows: no work on dead slots
	if (cols <= 512) { softmax_row_in_registers<8>(row, cols, lane); return; }
	if (cols <= 2048) { softmax_row_in_registers<32>(row, cols, lane); return; }
	float mx = -INFINITY;
	for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, row[j]);
	mx = wave_max(mx);
	mx = __shfl(mx, 0, 64);
	double s = 0;
	for (int j = lane; j < cols; j += 64) {
		float e = expf(row[j] - mx);
		row[j] = e;
		s += e;
	}
	s = wave_sum(s);
	float sf = (float)__shfl(s, 0, 64);
	for (int j = lane; j < cols; j += 64) row[j] /= sf;
}

bla_status window_sum(void* stream, const float* m, int count, int len, int stride, float* out) {
	if (count <= 0) return BLA_OK;
	hipLaunchKernelGGL(window_sum_kernel, dim3((count + 3) / 4), dim3(kThreads), 0, pick_stream(stream), m, count, len, stride, out);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

static bla_status reduce_common(void* stream, const float* m, size_t n, int what, float* d_out) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(d_out && (n == 0 || m), BLA_ERR_INVALID, "null operand");
	hipStream_t s = pick_stream(stream);
	unsigned blocks = grid_for(n ? (n + 3) / 4 : 1);
	if (blocks > 1024) blocks = 1024;
	void* ws;
	st = ensure_workspace((size_t)blocks * 3 * sizeof(double), &ws);
	if (st) return st;
	hipLaunchKernelGGL(reduce_stage1_kernel, dim3(blocks), dim3(kThreads), 0, s, m, n, (double*)ws, (int)((uintptr_t)m % 16 == 0));
	hipLaunchKernelGGL(reduce_stage2_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, (int)blocks, n, what, d_out);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // namespace bla

using namespace bla;

extern "C" {

bla_status bla_scale_f32(void* stream, float* d_m, size_t n, float f) { return launch_ew<OP_SCALE, false>(stream, d_m, nullptr, f, n); }
bla_status bla_add_f32(void* stream, float* d_a, const float* d_b, size_t n) { return launch_ew<OP_ADD, true>(stream, d_a, d_b, 0.f, n); }
bla_status bla_hadamard_f32(void* stream, float* d_a, const float* d_b, size_t n) { return launch_ew<OP_MUL, true>(stream, d_a, d_b, 0.f, n); }
bla_status bla_axpy_f32(void* stream, float* d_y, const float* d_x, float alpha, size_t n) { return launch_ew<OP_AXPY, true>(stream, d_y, d_x, alpha, n); }
bla_status bla_relu_f32(void* stream, float* d, size_t n) { return launch_ew<OP_RELU, false>(stream, d, nullptr, 0.f, n); }
bla_status bla_relu_ddx_f32(void* stream, float* d, size_t n) { return launch_ew<OP_RELU_DDX, false>(stream, d, nullptr, 0.f, n); }

bla_status bla_add_tile_columns_f32(void* stream, float* d_a, int a_rows, int a_cols, const float* d_b, int b_cols) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(a_rows >= 0 && a_cols >= 0 && b_cols > 0, BLA_ERR_INVALID, "bad shape %dx%d tiled by %d columns", a_rows, a_cols, b_cols);
	if (a_rows == 0 || a_cols == 0) return BLA_OK;
	BLA_REQUIRE(d_a && d_b, BLA_ERR_INVALID, "null operand");
	if (b_cols == 1 && a_cols % 4 == 0 && (uintptr_t)d_a % 16 == 0) {
		hipLaunchKernelGGL(bias_column_vec_kernel, dim3(grid_for((size_t)a_rows * (a_cols / 4))), dim3(kThreads), 0, pick_stream(stream), d_a, d_b, a_rows, a_cols);
		BLA_HIP(hipGetLastError());
		return BLA_OK;
	}
	unsigned gx = (unsigned)((a_cols + kThreads - 1) / kThreads);
	if (gx > 64) gx = 64;
	unsigned gy = (unsigned)a_rows;
	if (gy > 2048 / gx) gy = 2048 / gx;
	hipLaunchKernelGGL(tile_columns_kernel, dim3(gx, gy), dim3(kThreads), 0, pick_stream(stream), d_a, d_b, a_rows, a_cols, b_cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_add_tile_rows_f32(void* stream, float* d_a, int a_rows, int a_cols, const float* d_b) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(a_rows >= 0 && a_cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", a_rows, a_cols);
	if (a_rows == 0 || a_cols == 0) return BLA_OK;
	BLA_REQUIRE(d_a && d_b, BLA_ERR_INVALID, "null operand");
	hipLaunchKernelGGL(tile_rows_kernel, dim3(grid_for((size_t)a_rows * a_cols)), dim3(kThreads), 0, pick_stream(stream), d_a, d_b, a_rows, a_cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_transpose_f32(void* stream, const float* d_in, float* d_out, int rows, int cols) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	if (rows == 0 || cols == 0) return BLA_OK;
	BLA_REQUIRE(d_in && d_out && d_in != d_out, BLA_ERR_INVALID, "transpose needs distinct non-null in/out");
	unsigned tiles = (unsigned)(((rows + 63) / 64) * ((cols + 63) / 64));
	if (rows % 4 == 0 && cols % 4 == 0 && ((uintptr_t)d_in | (uintptr_t)d_out) % 16 == 0)
		hipLaunchKernelGGL(transpose_vec_kernel, dim3(tiles), dim3(256), 0, pick_stream(stream), d_in, d_out, rows, cols);
	else
		hipLaunchKernelGGL(transpose_kernel, dim3(tiles), dim3(256), 0, pick_stream(stream), d_in, d_out, rows, cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_row_sum_f32(void* stream, const float* d_m, int rows, int cols, float* d_out) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	if (cols == 0) return BLA_OK;
	BLA_REQUIRE(d_out && (rows == 0 || d_m), BLA_ERR_INVALID, "null operand");
	unsigned gx = (unsigned)((cols + kThreads - 1) / kThreads);
	int chunks = (int)(1024 / gx);
	if (chunks > (rows + 63) / 64) chunks = (rows + 63) / 64;   // at least 64 rows per chunk
	if (chunks < 1) chunks = 1;
	int rpc = (rows + chunks - 1) / chunks;
	chunks = rows > 0 ? (rows + rpc - 1) / rpc : 1;
	void* ws;
	st = ensure_workspace((size_t)chunks * cols * sizeof(double), &ws);
	if (st) return st;
	if (cols % 4 == 0 && (uintptr_t)d_m % 16 == 0 && rows >= 64) {
		// 16 bytes per lane: fewer, fatter workgroups (4 columns per thread), >= 32 rows per chunk
		unsigned vx = (unsigned)((cols / 4 + kThreads - 1) / kThreads);
		chunks = (int)((ctx().num_cus > 0 ? ctx().num_cus : 256) / vx);   // one workgroup per CU measured best (8192^2: 50.6 us; 2 per CU 61, 4 per CU 76)
		if (chunks > rows / 32) chunks = rows / 32;
		if (chunks < 1) chunks = 1;
		rpc = (rows + chunks - 1) / chunks;
		chunks = (rows + rpc - 1) / rpc;
		st = ensure_workspace((size_t)chunks * cols * sizeof(double), &ws);
		if (st) return st;
		hipLaunchKernelGGL(row_sum_partial_vec_kernel, dim3(vx, chunks), dim3(kThreads), 0, pick_stream(stream), d_m, rows, cols, rpc, (double*)ws);
	} else {
		hipLaunchKernelGGL(row_sum_partial_kernel, dim3(gx, chunks), dim3(kThreads), 0, pick_stream(stream), d_m, rows, cols, rpc, (double*)ws);
	}
	hipLaunchKernelGGL(row_sum_final_kernel, dim3(gx), dim3(kThreads), 0, pick_stream(stream), (const double*)ws, chunks, cols, d_out);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_col_sum_f32(void* stream, const float* d_m, int rows, int cols, float* d_out, int mode) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	BLA_REQUIRE(mode == BLA_COLSUM_AS_WRITTEN || mode == BLA_COLSUM_INTENDED, BLA_ERR_INVALID, "bad col_sum mode %d", mode);
	if (mode == BLA_COLSUM_AS_WRITTEN && rows > cols) {
		set_error("matrix_col_sum as written reads out of bounds for %dx%d (rows > cols, lib/matrix.c:144): undefined in the reference", rows, cols);
		return BLA_ERR_UNDEFINED;
	}
	if (rows == 0) return BLA_OK;
	BLA_REQUIRE(d_out && (cols == 0 || d_m), BLA_ERR_INVALID, "null operand");
	int stride = mode == BLA_COLSUM_AS_WRITTEN ? rows : cols;
	hipLaunchKernelGGL(window_sum_kernel, dim3((rows + 3) / 4), dim3(kThreads), 0, pick_stream(stream), d_m, rows, cols, stride, d_out);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_frobenius_f32(void* stream, const float* d_m, size_t n, float* d_out) { return reduce_common(stream, d_m, n, RED_SUM_SQ, d_out); }
bla_status bla_max_f32(void* stream, const float* d_m, size_t n, float* d_out) { return reduce_common(stream, d_m, n, RED_MAX, d_out); }

bla_status bla_zscore_f32(void* stream, float* d_m, size_t n) {
	bla_status st = require_ready();
	if (st) return st;
	if (n == 0) return BLA_OK;
	void* ws;
	st = ensure_workspace(1024 * 3 * sizeof(double) + 64, &ws);   // reduce_common's partials + 2 floats after them
	if (st) return st;
	float* stats = (float*)((char*)ws + 1024 * 3 * sizeof(double));
	st = reduce_common(stream, d_m, n, RED_SUM_AND_SQ, stats);
	if (st) return st;
	hipLaunchKernelGGL(zscore_apply_kernel, dim3(grid_for(n)), dim3(kThreads), 0, pick_stream(stream), d_m, n, stats);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status bla_softmax_cols_f32(void* stream, float* d, int rows, int cols) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	if (rows == 0 || cols == 0) return BLA_OK;
	BLA_REQUIRE(d, BLA_ERR_INVALID, "null operand");
	return softmax_cols_dispatch(stream, d, rows, cols, nullptr, 0.f, nullptr);
}

bla_status bla_softmax_cols_grad_f32(void* stream, float* d, int rows, int cols, const float* d_y, float scale, float* d_grad) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	if (rows == 0 || cols == 0) return BLA_OK;
	BLA_REQUIRE(d && d_y && d_grad, BLA_ERR_INVALID, "null operand");
	return softmax_cols_dispatch(stream, d, rows, cols, d_y, scale, d_grad);
}

bla_status bla_softmax_rows_f32(void* stream, float* d, int rows, int cols) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(rows >= 0 && cols >= 0, BLA_ERR_INVALID, "bad shape %dx%d", rows, cols);
	if (rows == 0 || cols == 0) return BLA_OK;
	BLA_REQUIRE(d, BLA_ERR_INVALID, "null operand");
	const bool vec = cols % 4 == 0 && (uintptr_t)d % 16 == 0;
	if (vec && cols > 2048 && cols <= 8192) hipLaunchKernelGGL(softmax_row_block_kernel<256>, dim3(rows), dim3(256), 0, pick_stream(stream), d, rows, cols);
	else if (vec && cols > 8192 && cols <= 32768) hipLaunchKernelGGL(softmax_row_block_kernel<1024>, dim3(rows), dim3(1024), 0, pick_stream(stream), d, rows, cols);
	else hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(kThreads), 0, pick_stream(stream), d, rows, cols);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // extern "C"
