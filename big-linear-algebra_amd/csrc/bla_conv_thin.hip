// bla_conv_thin.hip -- convolutions with at most four channels on ONE side: the U-Net's first block (3 -> 128, k 3 and the 1x1 residual,
// model/cifar_unet.c:1102) and its output convolution (128 -> 3, :1165) with their gradients (:1366, :1434-1435).  As GEMMs they are
// 128 x 27 x (B * 1024) and 3 x 1152 x (B * 1024): no MFMA tile fits a 3-wide side, and on the 32 x 32-tile gather kernel the three of them
// cost 0.7 ms of an 18.4 ms batch-64 pass (3.8 %) for 0.1 % of its FLOPs.  They are HBM-bound (one 33.5 MB activation tensor read or written),
// so they run here as direct convolutions on the vector ALUs: 27 or 9 image values per thread in registers, the kernels broadcast from LDS.
//   few inputs  (C <= 4): out[b][f][px] = sum_{c,t} K[f][c][t] x[b][c][px + t]          thread = pixel x 32 output channels
//   few outputs (F <= 4): the same sum, thread = pixel x a quarter of the input channels, the four quarters folded through LDS
//   weight gradient     : dK[f][c][t] = sum_{b,px} dy[b][f][px] x[b][c][px + t]          block = (wide-side channel, image chunk); 27 running sums
//                         per thread, folded over the block, then over the chunks in chunk order by a second launch (deterministic)
//   data gradient of the few-outputs form = the few-inputs form on del_y with the flipped kernels (pads mirrored)
// Stride 1, k in {1, 3}, "same" geometry.  Bounds: fp32 sums of K terms in a fixed order, well inside the GEMM tolerance of the tests.
#include "bla_internal.h"

namespace bla {

namespace {
constexpr int kFG = 32;                 // output channels per thread of the few-inputs form
constexpr int kMaxThin = 4;

inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// x [B][C][H][W], kern [F][C][K][K], out [B][F][H][W]; grid (pixel blocks of 256, F / 32, B)
template <int K>
__global__ void __launch_bounds__(256) thin_few_inputs_kernel(const float* __restrict__ x, const float* __restrict__ kern, float* __restrict__ out, int h, int w,
                                                               int c_n, int f_n, int pt, int pl, const float* __restrict__ bias, int bias_stride,
                                                               const float* __restrict__ add, float* __restrict__ out2) {
	constexpr int KK = K * K, ROW = kMaxThin * KK;
	__shared__ float wt[kFG][ROW];
	const int b = blockIdx.z, f0 = blockIdx.y * kFG, hw = h * w;
	for (int e = threadIdx.x; e < kFG * ROW; e += 256) {
		const int f = e / ROW, r = e - f * ROW;
		wt[f][r] = (f0 + f < f_n && r < c_n * KK) ? kern[(size_t)(f0 + f) * c_n * KK + r] : 0.f;
	}
	__syncthreads();
	const int px = blockIdx.x * 256 + threadIdx.x;
	if (px >= hw) return;
	const int i = px / w, j = px - i * w;
	float xv[ROW];
#pragma unroll
	for (int c = 0; c < kMaxThin; c++)
#pragma unroll
		for (int p = 0; p < K; p++)
#pragma unroll
			for (int q = 0; q < K; q++) {
				const int yy = i + p - pt, xx = j + q - pl;
				const bool ok = c < c_n && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
				xv[c * KK + p * K + q] = ok ? x[((size_t)b * c_n + c) * hw + yy * w + xx] : 0.f;
			}
	const int nf = min(kFG, f_n - f0);
	for (int f = 0; f < nf; f++) {
		float s = 0.f;
#pragma unroll
		for (int r = 0; r < ROW; r++) s = fmaf(wt[f][r], xv[r], s);
		const size_t at = ((size_t)b * f_n + f0 + f) * hw + px;
		if (bias) s += bias[(size_t)b * bias_stride + f0 + f];
		out[at] = s;
		if (out2) out2[at] = s + add[at];
	}
}

// x [B][C][H][W], kern [F][C][K][K] (F <= 4), out [B][F][H][W]; block = 64 pixels x 4 channel quarters; grid (pixel blocks of 64, B)
template <int K>
__global__ void __launch_bounds__(256) thin_few_outputs_kernel(const float* __restrict__ x, const float* __restrict__ kern, float* __restrict__ out, int h, int w,
                                                                int c_n, int f_n, int pt, int pl, const float* __restrict__ bias, int bias_stride,
                                                                const float* __restrict__ add, float* __restrict__ out2) {
	constexpr int KK = K * K;
	extern __shared__ float lds[];              // kMaxThin * c_n * KK kernels, then 4 x 64 x kMaxThin partial sums
	float* wt = lds;                            // [f][c][t]
	float* part = lds + kMaxThin * c_n * KK;
	const int b = blockIdx.y, hw = h * w;
	for (int e = threadIdx.x; e < kMaxThin * c_n * KK; e += 256) wt[e] = e < f_n * c_n * KK ? kern[e] : 0.f;
	__syncthreads();
	const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
	const int px = blockIdx.x * 64 + lane;
	const bool live = px < hw;
	const int i = live ? px / w : 0, j = live ? px - i * w : 0;
	int off[KK]; bool ok[KK];
#pragma unroll
	for (int p = 0; p < K; p++)
#pragma unroll
		for (int q = 0; q < K; q++) {
			const int yy = i + p - pt, xx = j + q - pl;
			ok[p * K + q] = live && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
			off[p * K + q] = ok[p * K + q] ? yy * w + xx : 0;
		}
	const int per = (c_n + 3) / 4, c0 = quarter * per, c1 = min(c_n, c0 + per);
	float acc[kMaxThin] = {0.f, 0.f, 0.f, 0.f};
	const float* xb = x + (size_t)b * c_n * hw;
	for (int c = c0; c < c1; c++) {
		float xv[KK];
#pragma unroll
		for (int t = 0; t < KK; t++) { const float v = xb[(size_t)c * hw + off[t]]; xv[t] = ok[t] ? v : 0.f; }
#pragma unroll
		for (int f = 0; f < kMaxThin; f++)
#pragma unroll
			for (int t = 0; t < KK; t++) acc[f] = fmaf(wt[(f * c_n + c) * KK + t], xv[t], acc[f]);
	}
#pragma unroll
	for (int f = 0; f < kMaxThin; f++) part[(quarter * 64 + lane) * kMaxThin + f] = acc[f];
	__syncthreads();
	for (int e = threadIdx.x; e < 64 * kMaxThin; e += 256) {       // e = f * 64 + pixel: consecutive lanes store consecutive pixels
		const int f = e >> 6, l = e & 63, p2 = blockIdx.x * 64 + l;
		if (f >= f_n || p2 >= hw) continue;
		float s = (part[(0 * 64 + l) * kMaxThin + f] + part[(1 * 64 + l) * kMaxThin + f]) + (part[(2 * 64 + l) * kMaxThin + f] + part[(3 * 64 + l) * kMaxThin + f]);
		const size_t at = ((size_t)b * f_n + f) * hw + p2;
		if (bias) s += bias[(size_t)b * bias_stride + f];
		out[at] = s;
		if (out2) out2[at] = s + add[at];
	}
}

// Weight gradient with a thin side.  THIN_IN: c < c_n <= 4, block.x = f;  else: f < f_n <= 4, block.x = c.  blockIdx.y = image chunk.
// part [chunks][F][C][KK] receives this block's sums over its images; thin_wgrad_fold_kernel adds the chunks in order.
template <int K, bool THIN_IN>
__global__ void __launch_bounds__(256) thin_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part, int batch, int h, int w,
                                                          int c_n, int f_n, int pt, int pl, int images_per_chunk) {
	constexpr int KK = K * K, NACC = kMaxThin * KK;
	__shared__ float red[4][NACC];
	const int wide = blockIdx.x, chunk = blockIdx.y, hw = h * w;
	const int thin_n = THIN_IN ? c_n : f_n;
	const int b0 = chunk * images_per_chunk, b1 = min(batch, b0 + images_per_chunk);
	float acc[NACC];
#pragma unroll
	for (int r = 0; r < NACC; r++) acc[r] = 0.f;
	for (int b = b0; b < b1; b++) {
		const float* xb = x + (size_t)b * c_n * hw;
		const float* yb = dy + (size_t)b * f_n * hw;
		for (int px = threadIdx.x; px < hw; px += 256) {
			const int i = px / w, j = px - i * w;
			float g[kMaxThin];        // THIN_IN: one del_y value (g[0]); else one per thin output channel
			if (THIN_IN) g[0] = yb[(size_t)wide * hw + px];
			else {
#pragma unroll
				for (int f = 0; f < kMaxThin; f++) g[f] = f < thin_n ? yb[(size_t)f * hw + px] : 0.f;
			}
#pragma unroll
			for (int p = 0; p < K; p++)
#pragma unroll
				for (int q = 0; q < K; q++) {
					const int yy = i + p - pt, xx = j + q - pl;
					const bool ok = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
					const int o = ok ? yy * w + xx : 0;
					if (THIN_IN) {
#pragma unroll
						for (int c = 0; c < kMaxThin; c++) {
							const float v = (ok && c < thin_n) ? xb[(size_t)c * hw + o] : 0.f;
							acc[c * KK + p * K + q] = fmaf(g[0], v, acc[c * KK + p * K + q]);
						}
					} else {
						const float v = ok ? xb[(size_t)wide * hw + o] : 0.f;
#pragma unroll
						for (int f = 0; f < kMaxThin; f++) acc[f * KK + p * K + q] = fmaf(g[f], v, acc[f * KK + p * K + q]);
					}
				}
		}
	}
	// fold over the block: lanes by xor-shuffles (a fixed tree), the four waves in wave order
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int r = 0; r < NACC; r++) {
		float v = acc[r];
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
		if (lane == 0) red[wave][r] = v;
	}
	__syncthreads();
	if (threadIdx.x < thin_n * KK) {
		const int t = threadIdx.x / KK, r = threadIdx.x - t * KK;      // t = thin channel
		const float s = (red[0][t * KK + r] + red[1][t * KK + r]) + (red[2][t * KK + r] + red[3][t * KK + r]);
		const size_t at = THIN_IN ? ((size_t)wide * c_n + t) * KK + r : ((size_t)t * c_n + wide) * KK + r;
		part[(size_t)chunk * f_n * c_n * KK + at] = s;
	}
}
__global__ void __launch_bounds__(256) thin_wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int total, int chunks) {
	const int e = blockIdx.x * 256 + threadIdx.x;
	if (e >= total) return;
	float s = 0.f;
	for (int z = 0; z < chunks; z++) s += part[(size_t)z * total + e];
	out[e] = s;
}
}  // namespace

bool thin_conv_applies(int k, int c_in, int f_n, int stride) {
	static const bool enabled = [] { const char* v = getenv("BLA_CONV_THIN"); return !(v && v[0] == '0'); }();
	return enabled && stride == 1 && (k == 1 || k == 3) && (c_in <= kMaxThin || (f_n <= kMaxThin && c_in * k * k <= 3584));   // (the few-outputs form keeps 4 x C x k x k kernels in LDS: 60 KB at most)
}

// out [B][F][H][W] = conv(x [B][C][H][W], kern [F][C][k][k]) with pads (pt, pl) on the top / left (stride 1, output H x W); optional epilogue
bla_status thin_conv_forward(hipStream_t s, const float* x, const float* kern, float* out, int batch, int h, int w, int k, int c_in, int f_n, int pt, int pl,
                             const float* ep_bias, int ep_bias_stride, const float* ep_add, float* ep_out2) {
	const int hw = h * w;
	if (c_in <= kMaxThin) {
		const dim3 grid(ceil_div(hw, 256), ceil_div(f_n, kFG), batch);
		if (k == 3) hipLaunchKernelGGL(thin_few_inputs_kernel<3>, grid, dim3(256), 0, s, x, kern, out, h, w, c_in, f_n, pt, pl, ep_bias, ep_bias_stride, ep_add, ep_out2);
		else hipLaunchKernelGGL(thin_few_inputs_kernel<1>, grid, dim3(256), 0, s, x, kern, out, h, w, c_in, f_n, pt, pl, ep_bias, ep_bias_stride, ep_add, ep_out2);
	} else {
		const dim3 grid(ceil_div(hw, 64), batch);
		const size_t lds = ((size_t)kMaxThin * c_in * k * k + 4 * 64 * kMaxThin) * sizeof(float);
		if (k == 3) hipLaunchKernelGGL(thin_few_outputs_kernel<3>, grid, dim3(256), lds, s, x, kern, out, h, w, c_in, f_n, pt, pl, ep_bias, ep_bias_stride, ep_add, ep_out2);
		else hipLaunchKernelGGL(thin_few_outputs_kernel<1>, grid, dim3(256), lds, s, x, kern, out, h, w, c_in, f_n, pt, pl, ep_bias, ep_bias_stride, ep_add, ep_out2);
	}
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

// del_kern [F][C][k][k] = sum over the batch of del_y (x) x
bla_status thin_conv_wgrad(hipStream_t s, const float* del_y, const float* x, float* del_kern, int batch, int h, int w, int k, int c_in, int f_n, int pt, int pl) {
	const bool thin_in = c_in <= kMaxThin;
	const int wide = thin_in ? f_n : c_in, total = f_n * c_in * k * k;
	// enough blocks to fill the chip a few times over; every block at least one image
	int chunks = (4 * ctx().num_cus + wide - 1) / wide;
	if (chunks > batch) chunks = batch;
	if (chunks < 1) chunks = 1;
	const int per = (batch + chunks - 1) / chunks;
	chunks = (batch + per - 1) / per;
	void* ws;
	bla_status st = ensure_workspace((size_t)chunks * total * sizeof(float), &ws);
	if (st) return st;
	const dim3 grid(wide, chunks);
#define BLA_THIN_W(K, TI) hipLaunchKernelGGL((thin_wgrad_kernel<K, TI>), grid, dim3(256), 0, s, del_y, x, (float*)ws, batch, h, w, c_in, f_n, pt, pl, per)
	if (k == 3) { if (thin_in) BLA_THIN_W(3, true); else BLA_THIN_W(3, false); }
	else { if (thin_in) BLA_THIN_W(1, true); else BLA_THIN_W(1, false); }
#undef BLA_THIN_W
	BLA_HIP(hipGetLastError());
	hipLaunchKernelGGL(thin_wgrad_fold_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)ws, del_kern, total, chunks);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // namespace bla
