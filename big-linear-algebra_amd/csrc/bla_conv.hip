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
#include <cstdlib>
#include <mutex>
#include <vector>

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

// Stride s > 1: the reference's _col2im is undefined there (it walks the IMAGE grid with out_row = i*stride + k: out of bounds).
// The INTENDED operation is the adjoint of _im2col (lib/conv.c:58-74): out[c][y][x] = sum over output pixels (i, j) and taps (p, q) with
// i*s + p - pt == y, j*s + q - pl == x of cols[(i, j)][c, p, q], pinned by <im2col(x), v> == <x, col2im(v)> (tests/test_conv_gpu.py).
// Same gather form and the same order of additions as the stride-1 kernel above (ascending (i, j)).
__global__ void __launch_bounds__(kThreads) col2im_adjoint_kernel(const float* __restrict__ cols, float* __restrict__ out, int h, int w, int k, int c_n,
                                                                   int s, int ho, int wo, int pt, int pl) {
	const int kk = k * k, roww = kk * c_n;
	const size_t total = (size_t)c_n * h * w;
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
		int xx = (int)(e % w), yy = (int)((e / w) % h), c = (int)(e / ((size_t)w * h));
		float acc = 0.f;
		for (int p = k - 1; p >= 0; p--) {
			int iy = yy + pt - p;
			if (iy < 0 || iy % s) continue;
			int i = iy / s;
			if (i >= ho) continue;
			for (int q = k - 1; q >= 0; q--) {
				int jx = xx + pl - q;
				if (jx < 0 || jx % s) continue;
				int j = jx / s;
				if (j >= wo) continue;
				acc += cols[((size_t)i * wo + j) * roww + c * kk + p * k + q];
			}
		}
		out[e] = acc;
	}
}

// dst [planes][(Ho-1)*s+1][(Wo-1)*s+1] = src [planes][Ho][Wo] with s-1 zeros between neighbours: the data gradient of a stride-s convolution
// is the stride-1 convolution of this dilated gradient with the flipped kernels
__global__ void __launch_bounds__(kThreads) dilate_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes, int ho, int wo, int s, int hd, int wd) {
	const size_t total = (size_t)planes * hd * wd;
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
		int x = (int)(e % wd); size_t t = e / wd;
		int y = (int)(t % hd); size_t pc = t / hd;
		dst[e] = (y % s == 0 && x % s == 0) ? src[(pc * ho + y / s) * wo + x / s] : 0.f;
	}
}

// BLA_STRICT_REFERENCE=1: operations the reference leaves undefined are refused instead of given their intended meaning
static bool strict_reference() {
	static const bool v = [] { const char* e = getenv("BLA_STRICT_REFERENCE"); return e && e[0] == '1'; }();
	return v;
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

template <bool RELU>   // RELU: multi_channel_relu fused behind the normalisation (model/cifar_unet.c:1046-1047,1056-1057)
__global__ void __launch_bounds__(kGnThreads) group_norm_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ stdevs,
                                                                 float* __restrict__ means, int channels, int group_size, int hw,
                                                                 const unsigned char* __restrict__ drop = nullptr, float* __restrict__ dropped = nullptr) {
	// drop / dropped: _dropout (model/cifar_unet.c:1032-1042, :1058) in the same pass -- dropped = drop ? 0 : out
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
		for (int i = 0; i < kGnRegs; i++) {   // (x - mean) / (stdev + 0), lib/norm.c:44
			int j = t + i * kGnThreads;
			float y = (v[i] - mean) / var;
			y = RELU && y < 0.f ? 0.f : y;
			if (j < n) { out[off + j] = y; if (dropped) dropped[off + j] = drop[off + j] ? 0.f : y; }
		}
		return;
	}
	double s = 0;
#pragma unroll 8
	for (int i = t; i < n; i += kGnThreads) s += in[off + i];
	float mean = (float)(gn_block_sum(s) / (double)n);
	double q = 0;
#pragma unroll 8
	for (int i = t; i < n; i += kGnThreads) { float v = in[off + i] - mean; q += (double)v * v; }
	float var = (float)(gn_block_sum(q) / (double)n);
	if (t == 0) { means[g] = mean; stdevs[g] = var; }
#pragma unroll 8
	for (int i = t; i < n; i += kGnThreads) {
		float y = (in[off + i] - mean) / var;
		y = RELU && y < 0.f ? 0.f : y;
		out[off + i] = y;
		if (dropped) dropped[off + i] = drop[off + i] ? 0.f : y;
	}
}

// The register path of the kernel above at 16 bytes per thread: groups of up to 16,384 elements (a multiple of 4, 16-byte aligned), four float4 per
// thread.  The U-Net's 16x16 / 8x8 / 4x4 maps at batch 64 are all of this kind (512 groups of 8,192 elements moved 3.7 TB/s in the scalar form).
constexpr int kGnVec = 4;
// a norm kernel's by-product: the four values it just stored, again at their place inside the zero-padded copy the convolution behind it gathers from
// (PadOut, bla_internal.h; e = flat element index, a multiple of 4; rows are multiples of four pixels, so the four stay in one row)
struct PadArgs { float* dst; int hw, w, wh, plane, ptl; };
__device__ __forceinline__ void pad_store4(const PadArgs& pa, size_t e, float a, float b, float c, float d) {
	const size_t pl = e / (unsigned)pa.hw;
	const int pix = (int)(e - pl * (unsigned)pa.hw), y = pix / pa.w, x = pix - y * pa.w;
	float* q = pa.dst + pl * (size_t)pa.plane + (size_t)y * pa.wh + x + pa.ptl;
	q[0] = a; q[3] = d;
	if (((uintptr_t)(q + 1) & 7) == 0) *reinterpret_cast<float2*>(q + 1) = make_float2(b, c);   // a one-pixel halo puts the middle pair on 8 bytes
	else { q[1] = b; q[2] = c; }
}
template <bool RELU>
__global__ void __launch_bounds__(kGnThreads) group_norm_vec_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ stdevs,
                                                                     float* __restrict__ means, int channels, int group_size, int hw,
                                                                     const unsigned char* __restrict__ drop, float* __restrict__ dropped, PadArgs pa) {
	const int g = blockIdx.x, t = threadIdx.x;
	const int nch = min(group_size, channels - g * group_size);
	const size_t off = (size_t)g * group_size * hw;
	const int n = nch * hw, n4 = n >> 2;
	const float4* in4 = reinterpret_cast<const float4*>(in + off);
	float4 v[kGnVec];
	double s = 0;
#pragma unroll
	for (int i = 0; i < kGnVec; i++) {
		const int j = t + i * kGnThreads;
		v[i] = j < n4 ? in4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
		s += (double)v[i].x; s += (double)v[i].y; s += (double)v[i].z; s += (double)v[i].w;
	}
	const float mean = (float)(gn_block_sum(s) / (double)n);
	double q = 0;
#pragma unroll
	for (int i = 0; i < kGnVec; i++)
		if (t + i * kGnThreads < n4) {
			const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
			q += (double)d0 * d0; q += (double)d1 * d1; q += (double)d2 * d2; q += (double)d3 * d3;
		}
	const float var = (float)(gn_block_sum(q) / (double)n);
	if (t == 0) { means[g] = mean; stdevs[g] = var; }
#pragma unroll
	for (int i = 0; i < kGnVec; i++) {
		const int j = t + i * kGnThreads;
		if (j >= n4) continue;
		float y[4] = {(v[i].x - mean) / var, (v[i].y - mean) / var, (v[i].z - mean) / var, (v[i].w - mean) / var};   // (x - mean) / (stdev + 0), lib/norm.c:44
#pragma unroll
		for (int e = 0; e < 4; e++) y[e] = RELU && y[e] < 0.f ? 0.f : y[e];
		if (out) reinterpret_cast<float4*>(out + off)[j] = make_float4(y[0], y[1], y[2], y[3]);   // (out == NULL: only the dropped form is kept)
		if (dropped) {
			const uchar4 d = reinterpret_cast<const uchar4*>(drop + off)[j];
			if (d.x) y[0] = 0.f;
			if (d.y) y[1] = 0.f;
			if (d.z) y[2] = 0.f;
			if (d.w) y[3] = 0.f;
			reinterpret_cast<float4*>(dropped + off)[j] = make_float4(y[0], y[1], y[2], y[3]);
		}
		if (pa.dst) pad_store4(pa, off + 4 * (size_t)j, y[0], y[1], y[2], y[3]);
	}
}
__global__ void __launch_bounds__(kGnThreads) group_norm_ddx_vec_kernel(const float* __restrict__ source, float* __restrict__ dest, const float* __restrict__ data,
                                                                         const float* __restrict__ means, const float* __restrict__ stdevs, int channels,
                                                                         int group_size, int hw, const float* __restrict__ relu_gate, const float* __restrict__ addend, PadArgs pa) {
	const int g = blockIdx.x, t = threadIdx.x;
	const int nch = min(group_size, channels - g * group_size);
	const size_t off = (size_t)g * group_size * hw;
	const int n = nch * hw, n4 = n >> 2;
	const float mean = means[g], sd = stdevs[g];
	float sv[kGnVec][4], nv[kGnVec][4];
	double gs = 0, gws = 0;
#pragma unroll
	for (int i = 0; i < kGnVec; i++) {
		const int j = t + i * kGnThreads;
		const bool live = j < n4;
		const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
		const float4 s4 = live ? reinterpret_cast<const float4*>(source + off)[j] : z, d4 = live ? reinterpret_cast<const float4*>(data + off)[j] : z;
		const float4 g4 = live && relu_gate ? reinterpret_cast<const float4*>(relu_gate + off)[j] : make_float4(1.f, 1.f, 1.f, 1.f);
		const float ss[4] = {s4.x, s4.y, s4.z, s4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
		for (int e = 0; e < 4; e++) {
			sv[i][e] = live && !(relu_gate && gg[e] <= 0.f) ? ss[e] : 0.f;
			nv[i][e] = live ? (dd[e] - mean) / sd : 0.f;
			gs += sv[i][e]; gws += (double)nv[i][e] * sv[i][e];
		}
	}
	const float fgs = (float)(gn_block_sum(gs) / (double)n);
	const float fgws = (float)(gn_block_sum(gws) / (double)n);
#pragma unroll
	for (int i = 0; i < kGnVec; i++) {
		const int j = t + i * kGnThreads;
		if (j >= n4) continue;
		float r[4];
#pragma unroll
		for (int e = 0; e < 4; e++) r[e] = (sv[i][e] - fgs - nv[i][e] * fgws) / sd;
		if (addend) { const float4 a = reinterpret_cast<const float4*>(addend + off)[j]; r[0] += a.x; r[1] += a.y; r[2] += a.z; r[3] += a.w; }
		reinterpret_cast<float4*>(dest + off)[j] = make_float4(r[0], r[1], r[2], r[3]);
		if (pa.dst) pad_store4(pa, off + 4 * (size_t)j, r[0], r[1], r[2], r[3]);
	}
}

// lib/norm.c:52-93
__global__ void __launch_bounds__(kGnThreads) group_norm_ddx_kernel(const float* __restrict__ source, float* __restrict__ dest, const float* __restrict__ data,
                                                                     const float* __restrict__ means, const float* __restrict__ stdevs, int channels,
                                                                     int group_size, int hw, const float* __restrict__ relu_gate = nullptr,
                                                                     const float* __restrict__ addend = nullptr) {
	// relu_gate: multi_channel_relu_ddx on the way in (source counts as 0 where the forward ReLU output was <= 0, model/cifar_unet.c:1204);
	// addend: the residual branch's gradient on the way out (dest = group_norm_ddx(...) + addend, :1219)
	int g = blockIdx.x;
	int nch = min(group_size, channels - g * group_size);
	size_t off = (size_t)g * group_size * hw;
	int n = nch * hw;
	const int t = threadIdx.x;
	float mean = means[g], sd = stdevs[g];
	if (n <= kGnThreads * (kGnRegs / 2)) {   // two register copies (source, normalised data): 16 each (32 each spills)
		float sv[kGnRegs / 2], nv[kGnRegs / 2];
		double gs = 0, gws = 0;
#pragma unroll
		for (int i = 0; i < kGnRegs / 2; i++) {
			int j = t + i * kGnThreads;
			sv[i] = j < n ? source[off + j] : 0.f;
			if (relu_gate && j < n && relu_gate[off + j] <= 0.f) sv[i] = 0.f;
			nv[i] = j < n ? (data[off + j] - mean) / sd : 0.f;
			gs += sv[i]; gws += (double)nv[i] * sv[i];
		}
		float fgs = (float)(gn_block_sum(gs) / (double)n);
		float fgws = (float)(gn_block_sum(gws) / (double)n);
#pragma unroll
		for (int i = 0; i < kGnRegs / 2; i++) {
			int j = t + i * kGnThreads;
			if (j < n) { float d = (sv[i] - fgs - nv[i] * fgws) / sd; dest[off + j] = addend ? d + addend[off + j] : d; }
		}
		return;
	}
	double gs = 0, gws = 0;
	// (unrolled: one workgroup walks a whole group, so the loads of several iterations must be in flight together)
#pragma unroll 8
	for (int i = t; i < n; i += kGnThreads) {
		float wgt = (data[off + i] - mean) / sd;
		float sv = relu_gate && relu_gate[off + i] <= 0.f ? 0.f : source[off + i];
		gs += sv;
		gws += (double)wgt * sv;
	}
	float fgs = (float)(gn_block_sum(gs) / (double)n);
	float fgws = (float)(gn_block_sum(gws) / (double)n);
#pragma unroll 8
	for (int i = t; i < n; i += kGnThreads) {
		float nv = (data[off + i] - mean) / sd;
		float sv = relu_gate && relu_gate[off + i] <= 0.f ? 0.f : source[off + i];
		float d = (sv - fgs - nv * fgws) / sd;
		dest[off + i] = addend ? d + addend[off + i] : d;
	}
}

// Groups too large for the one-pass path: one workgroup per group is bound by a single CU's instruction rate (24 us for 32 channels x 32x32),
// so the group is cut into slices of kGnSlice elements over blockIdx.x -- partial sums (fp64) per slice, then every slice's workgroup adds the
// partials in slice order (identical totals everywhere, deterministic) and writes its part.  Same gate / addend options as the kernel above.
constexpr int kGnSlice = 2048, kGnSliceThreads = 256;

// forward pass of a large group, same slicing: partial sum x and sum x^2 in fp64 (x^2 is exact in fp64), then every slice forms
//   mean = (float)(sum x / n)   and   variance about that float mean = (sum x^2 - 2 mean sum x + n mean^2) / n
// -- the two passes of lib/norm.c:26-37 from one pass over memory -- and normalises its part.
template <bool VEC>   // VEC: 16-byte loads / stores (hw a multiple of 4, 16-byte aligned tensors) -- the scalar form moved 3.3 TB/s
__global__ void __launch_bounds__(kGnSliceThreads) group_norm_stats_kernel(const float* __restrict__ in, int channels, int group_size, int hw, double2* partials) {
	const int g = blockIdx.y, slice = blockIdx.x;
	const int nch = min(group_size, channels - g * group_size);
	const size_t off = (size_t)g * group_size * hw;
	const int n = nch * hw, lo = slice * kGnSlice, hi = min(n, lo + kGnSlice);
	double a = 0, b = 0;
	if (VEC) {
		for (int i = lo + 4 * (int)threadIdx.x; i < hi; i += 4 * kGnSliceThreads) {
			const float4 v = *reinterpret_cast<const float4*>(in + off + i);
			const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
			a += x0; b += x0 * x0; a += x1; b += x1 * x1; a += x2; b += x2 * x2; a += x3; b += x3 * x3;
		}
	} else
		for (int i = lo + (int)threadIdx.x; i < hi; i += kGnSliceThreads) { double x = in[off + i]; a += x; b += x * x; }
	__shared__ double sh[2][kGnSliceThreads / 64];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
	if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
	__syncthreads();
	if (threadIdx.x == 0) {
		double ta = 0, tb = 0;
		for (int i = 0; i < kGnSliceThreads / 64; i++) { ta += sh[0][i]; tb += sh[1][i]; }
		partials[(size_t)g * gridDim.x + slice] = make_double2(ta, tb);
	}
}

template <bool RELU, bool VEC>
__global__ void __launch_bounds__(kGnSliceThreads) group_norm_apply_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ stdevs,
                                                                            float* __restrict__ means, int channels, int group_size, int hw,
                                                                            const unsigned char* __restrict__ drop, float* __restrict__ dropped,
                                                                            const double2* __restrict__ partials, PadArgs pa) {
	const int g = blockIdx.y, slice = blockIdx.x;
	const int nch = min(group_size, channels - g * group_size);
	const size_t off = (size_t)g * group_size * hw;
	const int n = nch * hw, lo = slice * kGnSlice, hi = min(n, lo + kGnSlice);
	double a = 0, b = 0;
	for (unsigned i = 0; i < gridDim.x; i++) { double2 q = partials[(size_t)g * gridDim.x + i]; a += q.x; b += q.y; }   // slice order: same totals everywhere
	const float mean = (float)(a / (double)n);
	const double m = (double)mean;
	const float var = (float)((b - 2.0 * m * a + (double)n * m * m) / (double)n);
	if (slice == 0 && threadIdx.x == 0) { means[g] = mean; stdevs[g] = var; }
	if (VEC) {
		for (int i = lo + 4 * (int)threadIdx.x; i < hi; i += 4 * kGnSliceThreads) {
			const float4 v = *reinterpret_cast<const float4*>(in + off + i);
			float y[4] = {(v.x - mean) / var, (v.y - mean) / var, (v.z - mean) / var, (v.w - mean) / var};
#pragma unroll
			for (int e = 0; e < 4; e++) y[e] = RELU && y[e] < 0.f ? 0.f : y[e];
			if (out) *reinterpret_cast<float4*>(out + off + i) = make_float4(y[0], y[1], y[2], y[3]);
			if (dropped) {
				const uchar4 d = *reinterpret_cast<const uchar4*>(drop + off + i);
				if (d.x) y[0] = 0.f;
				if (d.y) y[1] = 0.f;
				if (d.z) y[2] = 0.f;
				if (d.w) y[3] = 0.f;
				*reinterpret_cast<float4*>(dropped + off + i) = make_float4(y[0], y[1], y[2], y[3]);
			}
			if (pa.dst) pad_store4(pa, off + (size_t)i, y[0], y[1], y[2], y[3]);
		}
		return;
	}
	for (int i = lo + (int)threadIdx.x; i < hi; i += kGnSliceThreads) {
		float y = (in[off + i] - mean) / var;
		y = RELU && y < 0.f ? 0.f : y;
		out[off + i] = y;
		if (dropped) dropped[off + i] = drop[off + i] ? 0.f : y;
	}
}

template <bool VEC>
__global__ void __launch_bounds__(kGnSliceThreads) group_norm_ddx_stats_kernel(const float* __restrict__ source, const float* __restrict__ data,
                                                                                const float* __restrict__ means, const float* __restrict__ stdevs, int channels,
                                                                                int group_size, int hw, const float* __restrict__ relu_gate, double2* partials) {
	const int g = blockIdx.y, slice = blockIdx.x;
	const int nch = min(group_size, channels - g * group_size);
	const size_t off = (size_t)g * group_size * hw;
	const int n = nch * hw, lo = slice * kGnSlice, hi = min(n, lo + kGnSlice);
	const float mean = means[g], sd = stdevs[g];
	double gs = 0, gws = 0;
	if (VEC) {
		for (int i = lo + 4 * (int)threadIdx.x; i < hi; i += 4 * kGnSliceThreads) {
			const float4 dv = *reinterpret_cast<const float4*>(data + off + i), sv4 = *reinterpret_cast<const float4*>(source + off + i);
			float4 gt = make_float4(1.f, 1.f, 1.f, 1.f);
			if (relu_gate) gt = *reinterpret_cast<const float4*>(relu_gate + off + i);
			const float dd[4] = {dv.x, dv.y, dv.z, dv.w}, ss[4] = {sv4.x, sv4.y, sv4.z, sv4.w}, gg[4] = {gt.x, gt.y, gt.z, gt.w};
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const float wgt = (dd[e] - mean) / sd;
				const float sv = relu_gate && gg[e] <= 0.f ? 0.f : ss[e];
				gs += sv;
				gws += (double)wgt * sv;
			}
		}
	} else
	for (int i = lo + (int)threadIdx.x; i < hi; i += kGnSliceThreads) {
		float wgt = (data[off + i] - mean) / sd;
		float sv = relu_gate && relu_gate[off + i] <= 0.f ? 0.f : source[off + i];
		gs += sv;
		gws += (double)wgt * sv;
	}
	__shared__ double sh[2][kGnSliceThreads / 64];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { gs += __shfl_down(gs, o, 64); gws += __shfl_down(gws, o, 64); }
	if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = gs; sh[1][threadIdx.x >> 6] = gws; }
	__syncthreads();
	if (threadIdx.x == 0) {
		double a = 0, b = 0;
		for (int i = 0; i < kGnSliceThreads / 64; i++) { a += sh[0][i]; b += sh[1][i]; }
		partials[(size_t)g * gridDim.x + slice] = make_double2(a, b);
	}
}

template <bool VEC>
__global__ void __launch_bounds__(kGnSliceThreads) group_norm_ddx_apply_kernel(const float* __restrict__ source, float* __restrict__ dest, const float* __restrict__ data,
                                                                                const float* __restrict__ means, const float* __restrict__ stdevs, int channels,
                                                                                int group_size, int hw, const float* __restrict__ relu_gate,
                                                                                const float* __restrict__ addend, const double2* __restrict__ partials, PadArgs pa) {
	const int g = blockIdx.y, slice = blockIdx.x;
	const int nch = min(group_size, channels - g * group_size);
	const size_t off = (size_t)g * group_size * hw;
	const int n = nch * hw, lo = slice * kGnSlice, hi = min(n, lo + kGnSlice);
	const float mean = means[g], sd = stdevs[g];
	double a = 0, b = 0;
	for (unsigned i = 0; i < gridDim.x; i++) { double2 q = partials[(size_t)g * gridDim.x + i]; a += q.x; b += q.y; }   // slice order: same totals in every workgroup
	const float fgs = (float)(a / (double)n), fgws = (float)(b / (double)n);
	if (VEC) {
		for (int i = lo + 4 * (int)threadIdx.x; i < hi; i += 4 * kGnSliceThreads) {
			const float4 dv = *reinterpret_cast<const float4*>(data + off + i), sv4 = *reinterpret_cast<const float4*>(source + off + i);
			float4 gt = make_float4(1.f, 1.f, 1.f, 1.f), ad = make_float4(0.f, 0.f, 0.f, 0.f);
			if (relu_gate) gt = *reinterpret_cast<const float4*>(relu_gate + off + i);
			if (addend) ad = *reinterpret_cast<const float4*>(addend + off + i);
			const float dd[4] = {dv.x, dv.y, dv.z, dv.w}, ss[4] = {sv4.x, sv4.y, sv4.z, sv4.w}, gg[4] = {gt.x, gt.y, gt.z, gt.w}, aa[4] = {ad.x, ad.y, ad.z, ad.w};
			float r[4];
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const float nv = (dd[e] - mean) / sd;
				const float sv = relu_gate && gg[e] <= 0.f ? 0.f : ss[e];
				const float d = (sv - fgs - nv * fgws) / sd;
				r[e] = addend ? d + aa[e] : d;
			}
			*reinterpret_cast<float4*>(dest + off + i) = make_float4(r[0], r[1], r[2], r[3]);
			if (pa.dst) pad_store4(pa, off + (size_t)i, r[0], r[1], r[2], r[3]);
		}
		return;
	}
	for (int i = lo + (int)threadIdx.x; i < hi; i += kGnSliceThreads) {
		float nv = (data[off + i] - mean) / sd;
		float sv = relu_gate && relu_gate[off + i] <= 0.f ? 0.f : source[off + i];
		float d = (sv - fgs - nv * fgws) / sd;
		dest[off + i] = addend ? d + addend[off + i] : d;
	}
}

static inline unsigned grid_for(size_t n) {
	size_t b = (n + kThreads - 1) / kThreads;
	if (b < 1) b = 1;
	if (b > 2048) b = 2048;
	return (unsigned)b;
}

// ---- implicit-GEMM convolution ---------------------------------------------------------------------------
// The im2col matrix is never materialised: the MFMA kernel gathers it from the [C][H][W] image while it loads its
// B operand.  Same skeleton as the wave-split-K GEMM (bla_gemm.hip): one workgroup per 32x32 output tile, its 4 waves
// split K, fragments go global -> VGPR (the image is L2-resident), partials meet in LDS.
//   FWD   : out[F][Ho*Wo]   = kern[F][K]    . G[K][Ho*Wo],  K = C*k*k,  G[(c,p,q)][(i,j)] = x[c][i*s+p-pt][j*s+q-pl]
//           lanes run over consecutive output pixels -> consecutive image addresses (coalesced), zero outside.
//           The data gradient of a stride-1 conv is the same contraction with the kernels transposed and
//           flipped and del_Y as the image, so it reuses this mode (no del_col matrix, no col2im).
//   WGRAD : dkern[F][K]     = del_y[F][HW]  . G^T[HW][K]        (= _reshape_matrix_kernels(im2col^T . del_Q))
// A per-geometry table (built once on the device, cached) maps k -> {image offset of tap (c,p,q), dy, dx}.
struct ConvGeom { int h, w, k, c, s, ho, wo, pt, pl; };
struct ConvArgs {
	const float* A; int lda;     // FWD: kernels [F][K]; WGRAD: del_y [F][HW]
	const float* img;            // [C][H][W]
	const int2* tab;             // [C*k*k] {offset, dy | dx << 16}
	float* out; int ldo;
	int M, N, K;                 // output M x N, contraction K
	int tiles_n, kw;             // kw: K extent per wave (multiple of 8)
	int splits, k_per_split;     // K is also cut over blockIdx.y; partial products go to slab[split][M][N]
	float* slab;
	// batch of images sharing the kernels (the reference has no batch dimension: each image is one conv() call).
	// FWD: image index = blockIdx.z, A shared.  WGRAD: the contraction runs over (image, pixel): blockIdx.y = image * psplits + k-split,
	// A (= del_y) and the image advance per image, every (image, k-split) pair writes its own slab.
	int batch, psplits;
	size_t img_stride, out_stride, a_stride;
	ConvGeom g;
	// forward epilogue (the adds the U-Net puts behind a conv, model/cifar_unet.c:1053,1067-1071): out = conv + ep_bias[row];
	// ep_out2 = out + ep_add (same layout as out).  Applied where the output is stored: in the gather kernel or in the slab fold.
	const float* ep_bias = nullptr;
	const float* ep_add = nullptr;
	float* ep_out2 = nullptr;
	int ep_bias_stride = 0;      // batched tiled forward only: bias set per image
	bool ep_fused_tiled = false; // the tiled (padded-copy, half-slab) forward kernel applies the epilogue itself
	// data gradient on the window kernel: A is still to be made -- the forward kernels [F][C][3][3] (F = this product's K / 9, C = its M), flipped and
	// re-ordered in one pass (window_order_flipped_kernel) instead of flip_kernels_kernel + window_order_kernels_kernel
	const float* flip_src = nullptr;
	// the caller already holds the zero-padded copy of `img` in this geometry's layout (conv_padded_layout: the norm kernel in front wrote it as a by-product):
	// the padded-copy forward and the weight gradient read it instead of making their own
	const float* padded_src = nullptr;
	// the kernel matrix already in the form this product reads (conv_prepare_kernels: window order / flipped): no re-ordering launch in front
	const float* prepared_A = nullptr;
};

__device__ __forceinline__ void conv_store(const ConvArgs& p, float* out, size_t image_off, int row, int col, float s) {
	if (p.ep_bias) s += p.ep_bias[row];
	out[(size_t)row * p.ldo + col] = s;
	if (p.ep_out2) p.ep_out2[image_off + (size_t)row * p.ldo + col] = s + p.ep_add[image_off + (size_t)row * p.ldo + col];
}

enum { CONV_FWD = 0, CONV_WGRAD = 1 };

__global__ void __launch_bounds__(256) conv_table_kernel(int2* tab, ConvGeom g) {
	int kidx = blockIdx.x * blockDim.x + threadIdx.x;
	int kk = g.k * g.k;
	if (kidx >= g.c * kk) return;
	int c = kidx / kk, p = (kidx % kk) / g.k, q = kidx % g.k;
	int dy = p - g.pt, dx = q - g.pl;
	tab[kidx] = make_int2(c * g.h * g.w + dy * g.w + dx, (dy & 0xffff) | (dx << 16));
}

constexpr int kConvTabLds = 2048;   // tap-table entries a workgroup keeps in LDS (16 KB)
// bx = output tile, by = K-split (FWD) or image * psplits + K-split (WGRAD), bz = image (FWD)
template <int MODE, bool VEC>
__device__ __forceinline__ void conv_implicit_body(ConvArgs p, const int bx, const int by, const int bz, float (*red)[32 * 33], int2* s_tab) {
	constexpr int NW = 4, PF = 8;
	typedef float f32x16 __attribute__((ext_vector_type(16)));
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int l31 = lane & 31, h = lane >> 5;
	const int tile_m = bx / p.tiles_n, tile_n = bx % p.tiles_n;
	const int m0 = tile_m * 32, n0 = tile_n * 32;
	const int image = MODE == CONV_FWD ? bz : by / p.psplits;
	const int ks = MODE == CONV_FWD ? by : by % p.psplits;
	p.img += (size_t)image * p.img_stride; p.A += (size_t)image * p.a_stride;
	if (MODE == CONV_FWD) { p.out += (size_t)image * p.out_stride; if (p.slab) p.slab += (size_t)image * p.splits * p.M * p.N; }
	const int split_begin = ks * p.k_per_split, split_end = min(p.K, (ks + 1) * p.k_per_split);
	// FWD: the tap table of this workgroup's K range goes to LDS once (one coalesced fetch) -- looked up from global memory, every round
	// of operand loads would be two dependent round trips (table entry, then the pixel it points to)
	const bool tab_lds = MODE == CONV_FWD && split_end - split_begin <= kConvTabLds && split_end > split_begin;
	if (tab_lds) {
		for (int i = threadIdx.x; i < split_end - split_begin; i += 256) s_tab[i] = p.tab[split_begin + i];
		__syncthreads();
	}
	const int k_begin = min(split_end, ks * p.k_per_split + wave * p.kw), k_end = min(split_end, k_begin + p.kw);
	const int arow = min(m0 + l31, p.M - 1);
	const int ncol = min(n0 + l31, p.N - 1);
	const ConvGeom g = p.g;
	// lane-invariant part of the gather
	int base, y0, x0;
	if (MODE == CONV_FWD) {          // n = output pixel (i, j)
		int i = ncol / g.wo, j = ncol - i * g.wo;
		y0 = i * g.s; x0 = j * g.s; base = y0 * g.w + x0;
	} else {                         // n = tap (c, p, q)
		int2 t = p.tab[ncol];
		base = t.x; y0 = (short)(t.y & 0xffff); x0 = t.y >> 16;
	}

	auto load = [&](int k, float (&a)[4], float (&b)[4]) {
		const int kb = k + 4 * h;
		if (VEC) {
			float4 x = *reinterpret_cast<const float4*>(p.A + (size_t)arow * p.lda + min(kb, p.K - 4));
			bool ok = kb < k_end;
			a[0] = ok ? x.x : 0.f; a[1] = ok ? x.y : 0.f; a[2] = ok ? x.z : 0.f; a[3] = ok ? x.w : 0.f;
		} else {
#pragma unroll
			for (int j = 0; j < 4; j++) { float x = p.A[(size_t)arow * p.lda + min(kb + j, p.K - 1)]; a[j] = kb + j < k_end ? x : 0.f; }
		}
		if (MODE == CONV_FWD) {
#pragma unroll
			for (int j = 0; j < 4; j++) {
				int2 t = tab_lds ? s_tab[min(kb + j, split_end - 1) - split_begin] : p.tab[min(kb + j, p.K - 1)];
				int yy = y0 + (short)(t.y & 0xffff), xx = x0 + (t.y >> 16);
				bool ok = kb + j < k_end && (unsigned)yy < (unsigned)g.h && (unsigned)xx < (unsigned)g.w;
				float v = p.img[ok ? t.x + base : 0];
				b[j] = ok ? v : 0.f;
			}
		} else {   // k = output pixel r = (i, j): one division per group, then walk
			int r = min(kb, p.K - 1);
			int i = r / g.wo, j = r - i * g.wo;
#pragma unroll
			for (int e = 0; e < 4; e++) {
				int yy = i * g.s + y0, xx = j * g.s + x0;
				bool ok = kb + e < k_end && (unsigned)yy < (unsigned)g.h && (unsigned)xx < (unsigned)g.w;
				float v = p.img[ok ? base + i * g.s * g.w + j * g.s : 0];
				b[e] = ok ? v : 0.f;
				if (++j == g.wo) { j = 0; i++; }
			}
		}
	};

	f32x16 acc;
#pragma unroll
	for (int r = 0; r < 16; r++) acc[r] = 0.f;
	for (int k = k_begin; k < k_end; k += 8 * PF) {
		float fa[PF][4], fb[PF][4];
#pragma unroll
		for (int gidx = 0; gidx < PF; gidx++) load(k + 8 * gidx, fa[gidx], fb[gidx]);
#pragma unroll
		for (int gidx = 0; gidx < PF; gidx++)
#pragma unroll
			for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gidx][j], fb[gidx][j], acc, 0, 0, 0);
	}
#pragma unroll
	for (int r = 0; r < 16; r++) red[wave][((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = acc[r];
	__syncthreads();
	for (int e = tid; e < 1024; e += NW * 64) {
		int r = e >> 5, c = e & 31;
		float s = (red[0][r * 33 + c] + red[1][r * 33 + c]) + (red[2][r * 33 + c] + red[3][r * 33 + c]);
		if (m0 + r < p.M && n0 + c < p.N) {
			if (p.splits > 1) p.slab[((size_t)by * p.M + m0 + r) * p.N + n0 + c] = s;
			else conv_store(p, p.out, MODE == CONV_FWD ? (size_t)image * p.out_stride : 0, m0 + r, n0 + c, s);
		}
	}
}

template <int MODE, bool VEC>
__global__ void __launch_bounds__(256) conv_implicit_kernel(ConvArgs p) {
	__shared__ float red[4][32 * 33];
	__shared__ int2 s_tab[MODE == CONV_FWD ? kConvTabLds : 1];
	conv_implicit_body<MODE, VEC>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, red, s_tab);
}

// Both gradients of one convolution in ONE launch: the weight gradient (WGRAD on the forward input) and the data gradient (FWD on del_y
// with the flipped kernels) depend only on del_y, the reference runs them one after the other (lib/conv.c:221-228); a single image fills
// a fraction of the chip with either, so their workgroups share a grid.  Linear block index: [0, blocks_w) = weight gradient
// (tile fastest, then split), the rest = data gradient (tile, split, image).
template <bool VECW, bool VECD>
__global__ void __launch_bounds__(256) conv_backward_pair_kernel(ConvArgs w, ConvArgs d, unsigned blocks_w, unsigned tiles_w, unsigned tiles_d, unsigned splits_d) {
	__shared__ float red[4][32 * 33];
	__shared__ int2 s_tab[kConvTabLds];
	const unsigned b = blockIdx.x;
	if (b < blocks_w) conv_implicit_body<CONV_WGRAD, VECW>(w, (int)(b % tiles_w), (int)(b / tiles_w), 0, red, s_tab);
	else {
		const unsigned e = b - blocks_w;
		conv_implicit_body<CONV_FWD, VECD>(d, (int)(e % tiles_d), (int)(e / tiles_d % splits_d), (int)(e / tiles_d / splits_d), red, s_tab);
	}
}

// folds the K-split slabs in split order (deterministic); blockIdx.y = image for a batched forward.
// (Measured and not kept: folding inside the gather kernel -- arrival counter per tile, last workgroup sums the partials, as the
// wave-split-K GEMM can -- to save this launch.  The agent-scope release/acquire per workgroup costs more than the launch:
// 128->128 @32x32 forward 27.9 us instead of 18.9, both gradients 59.7 instead of 42.5.)
__device__ __forceinline__ void conv_slab_reduce_body(const ConvArgs& p, unsigned bx, unsigned nbx, unsigned image) {
	size_t total = (size_t)p.M * p.N;
	const float* slab = p.slab + (size_t)image * p.splits * total;
	float* out = p.out + (size_t)image * p.out_stride;
	for (size_t i = (size_t)bx * blockDim.x + threadIdx.x; i < total; i += (size_t)nbx * blockDim.x) {
		float s = 0.f;
		for (int z = 0; z < p.splits; z++) s += slab[(size_t)z * total + i];
		conv_store(p, out, (size_t)image * p.out_stride, (int)(i / p.N), (int)(i % p.N), s);
	}
}
__global__ void __launch_bounds__(kThreads) conv_slab_reduce_kernel(ConvArgs p) { conv_slab_reduce_body(p, blockIdx.x, gridDim.x, blockIdx.y); }
// the folds of both gradients in one launch: blocks [0, blocks_w) fold the weight-gradient slabs, the rest the data gradient's (per image)
__global__ void __launch_bounds__(kThreads) conv_slab_reduce_pair_kernel(ConvArgs w, ConvArgs d, unsigned blocks_w, unsigned blocks_d) {
	if (blockIdx.x < blocks_w) { if (w.splits > 1) conv_slab_reduce_body(w, blockIdx.x, blocks_w, 0); }
	else if (d.splits > 1) { const unsigned e = blockIdx.x - blocks_w; conv_slab_reduce_body(d, e % blocks_d, blocks_d, e / blocks_d); }
}

// kt[c][f][p][q] = kern[f][c][k-1-p][k-1-q]: the kernels of the data-gradient convolution
__global__ void __launch_bounds__(kThreads) flip_kernels_kernel(const float* __restrict__ kern, float* __restrict__ kt, int f_n, int c_n, int k) {
	int kk = k * k, total = f_n * c_n * kk;
	for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		int pq = e % kk, f = (e / kk) % f_n, c = e / (kk * f_n);
		int p = pq / k, q = pq % k;
		kt[e] = kern[((size_t)f * c_n + c) * kk + (k - 1 - p) * k + (k - 1 - q)];
	}
}

// Per-geometry gather tables, built once per device (device memory is never freed: a handful of KB per distinct layer shape).
// The caches are process-wide: guarded by one mutex, keyed by device.  A table is filled on the library's own stream and waited for
// BEFORE its cache entry is published, so a second caller on another stream never sees a half-built table; and a first use while the
// caller's stream is recording a graph is refused (allocating is not allowed there, and the fill would only run at replay time):
// run a sequence once eagerly before recording it (include/bla.h, bla_graph_begin).
static std::mutex g_table_mu;
static bla_status table_build_allowed(hipStream_t s) {
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
		set_error("first use of a convolution geometry while the stream is recording a graph: run the sequence once eagerly before bla_graph_begin");
		return BLA_ERR_INVALID;
	}
	(void)hipGetLastError();
	return BLA_OK;
}

struct TableEntry { int device; ConvGeom g; int2* tab; };
static std::vector<TableEntry> g_tables;

static bla_status get_table(hipStream_t s, const ConvGeom& g, const int2** out) {
	std::lock_guard<std::mutex> lk(g_table_mu);
	const int dev = ctx().device;
	for (const TableEntry& e : g_tables) {
		const ConvGeom& t = e.g;
		if (e.device == dev && t.h == g.h && t.w == g.w && t.k == g.k && t.c == g.c && t.s == g.s && t.pt == g.pt && t.pl == g.pl) { *out = e.tab; return BLA_OK; }
	}
	bla_status st = table_build_allowed(s);
	if (st) return st;
	int n = g.c * g.k * g.k;
	int2* tab;
	BLA_HIP(hipMalloc((void**)&tab, (size_t)n * sizeof(int2)));
	hipLaunchKernelGGL(conv_table_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx().stream, tab, g);
	BLA_HIP(hipGetLastError());
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	g_tables.push_back(TableEntry{dev, g, tab});
	*out = tab;
	return BLA_OK;
}

// per-geometry table of the output pixels: r = i*Wo + j -> {(i*s)*W + j*s, (i*s) | (j*s) << 16}
__global__ void __launch_bounds__(256) conv_pixel_table_kernel(int2* tab, ConvGeom g) {
	int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= g.ho * g.wo) return;
	int i = r / g.wo, j = r - i * g.wo;
	tab[r] = make_int2(i * g.s * g.w + j * g.s, ((i * g.s) & 0xffff) | ((j * g.s) << 16));
}

struct PixelTableEntry { int device, w, s, ho, wo; int2* tab; };
static std::vector<PixelTableEntry> g_ptables;

static bla_status get_pixel_table(hipStream_t s, const ConvGeom& g, const int2** out) {
	std::lock_guard<std::mutex> lk(g_table_mu);
	const int dev = ctx().device;
	for (const PixelTableEntry& t : g_ptables)
		if (t.device == dev && t.w == g.w && t.s == g.s && t.ho == g.ho && t.wo == g.wo) { *out = t.tab; return BLA_OK; }
	bla_status st = table_build_allowed(s);
	if (st) return st;
	int n = g.ho * g.wo;
	int2* tab;
	BLA_HIP(hipMalloc((void**)&tab, (size_t)n * sizeof(int2)));
	hipLaunchKernelGGL(conv_pixel_table_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx().stream, tab, g);
	BLA_HIP(hipGetLastError());
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	g_ptables.push_back(PixelTableEntry{dev, g.w, g.s, g.ho, g.wo, tab});
	*out = tab;
	return BLA_OK;
}

// ---- tiled gather on a zero-padded, stride-split copy -------------------------------------------------------------------------
// One pass rewrites the batch as dst [B*C][s][s][Hh][Wh]: the image with its SAME padding, cut into its s x s parity planes
// (dst[c][py][px][yy][xx] = padded[c][yy*s + py][xx*s + px], Hp = (Ho-1)s + k rows, Hh = ceil(Hp / s); 33 MB -> 38 MB for 64 x 128 x 32 x 32 at
// stride 1, where there is one plane and this is plain padding).  After it the gather needs no bounds checks, and tap (p, q) of output pixels
// (i, j .. j+3) is four CONSECUTIVE floats -- plane (p % s, q % s), row i + p / s, columns j + q / s .. -- for any stride, so the slab is
// fetched with the same 16-byte DMA as a dense row-contiguous operand.  (The 16-byte DMAs are mostly unaligned; forcing them aligned in an
// experiment changed nothing: 230.9 vs 229.9 us.)  32-bit index arithmetic: the copy is limited to 2 GiB anyway.
// One thread writes four consecutive floats of one plane row (Wh is rounded up to a multiple of 4, so every row starts on 16 bytes): two
// divisions per 16 bytes instead of five per float, 16-byte stores (23 -> 15 us for 64 x 128 x 32 x 32).
template <int S>   // compile-time stride (1 or 2), 0 = any
__global__ void __launch_bounds__(256) pad_split_kernel(const float* __restrict__ src, float* __restrict__ dst, unsigned rows, int h, int w, int pt, int pl, int s_rt,
                                                        unsigned hh, unsigned wh) {
	const int s = S ? S : s_rt;
	const unsigned wq = wh / 4, total = rows * wq;
	for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		const unsigned row = e / wq, cq = e - row * wq;
		const unsigned yy = row % hh;
		unsigned t = row / hh, px = 0, py = 0;
		if (S != 1) { px = t % s; t /= s; py = t % s; t /= s; }
		const int y = (int)(yy * s + py) - pt, x0 = (int)(cq * 4 * s + px) - pl;
		float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
		if ((unsigned)y < (unsigned)h) {
			const float* sp = src + ((size_t)t * h + y) * w;
			v.x = (unsigned)x0 < (unsigned)w ? sp[x0] : 0.f;
			v.y = (unsigned)(x0 + s) < (unsigned)w ? sp[x0 + s] : 0.f;
			v.z = (unsigned)(x0 + 2 * s) < (unsigned)w ? sp[x0 + 2 * s] : 0.f;
			v.w = (unsigned)(x0 + 3 * s) < (unsigned)w ? sp[x0 + 3 * s] : 0.f;
		}
		*reinterpret_cast<float4*>(dst + (size_t)e * 4) = v;
	}
}
static void launch_pad_split(hipStream_t st, const float* src, float* dst, unsigned planes, int h, int w, int pt, int pl, int s, unsigned hh, unsigned wh) {
	const unsigned rows = planes * s * s * hh;
	const dim3 grid(grid_for((size_t)rows * (wh / 4)));
	if (s == 1) hipLaunchKernelGGL(pad_split_kernel<1>, grid, dim3(256), 0, st, src, dst, rows, h, w, pt, pl, s, hh, wh);
	else if (s == 2) hipLaunchKernelGGL(pad_split_kernel<2>, grid, dim3(256), 0, st, src, dst, rows, h, w, pt, pl, s, hh, wh);
	else hipLaunchKernelGGL(pad_split_kernel<0>, grid, dim3(256), 0, st, src, dst, rows, h, w, pt, pl, s, hh, wh);
}
__global__ void __launch_bounds__(256) padded_tables_kernel(int2* taps, int2* pix, int c_n, int k, int s, int hh, int wh, int ho, int wo) {
	int e = blockIdx.x * blockDim.x + threadIdx.x;
	int kk = k * k;
	if (e < c_n * kk) { int c = e / kk, p = (e % kk) / k, q = e % k; taps[e] = make_int2(((c * s + p % s) * s + q % s) * hh * wh + (p / s) * wh + q / s, 0); }
	if (e < ho * wo) { int i = e / wo, j = e - i * wo; pix[e] = make_int2(i * wh + j, 0); }
}
struct PaddedGeom { int hh, wh; size_t plane_floats; };   // plane_floats: one channel's s*s*Hh*Wh
static PaddedGeom padded_geom(const ConvGeom& g) {
	const int hp = (g.ho - 1) * g.s + g.k, wp = (g.wo - 1) * g.s + g.k;
	PaddedGeom pg;
	pg.hh = (hp + g.s - 1) / g.s; pg.wh = ((wp + g.s - 1) / g.s + 3) / 4 * 4;   // rows of a plane start on 16 bytes (pad_split_kernel)
	pg.plane_floats = (size_t)g.s * g.s * pg.hh * pg.wh;
	return pg;
}
struct PaddedTables { int device; ConvGeom g; int2* taps; int2* pix; };
static std::vector<PaddedTables> g_padded;

static bla_status get_padded_tables(hipStream_t s, const ConvGeom& g, const int2** taps, const int2** pix) {
	std::lock_guard<std::mutex> lk(g_table_mu);
	const int dev = ctx().device;
	for (const PaddedTables& e : g_padded) {
		const ConvGeom& t = e.g;
		if (e.device == dev && t.k == g.k && t.c == g.c && t.s == g.s && t.ho == g.ho && t.wo == g.wo) { *taps = e.taps; *pix = e.pix; return BLA_OK; }
	}
	bla_status st = table_build_allowed(s);
	if (st) return st;
	const PaddedGeom pg = padded_geom(g);
	int nt = g.c * g.k * g.k, np = g.ho * g.wo;
	int2 *t, *q;
	BLA_HIP(hipMalloc((void**)&t, (size_t)nt * sizeof(int2)));
	BLA_HIP(hipMalloc((void**)&q, (size_t)np * sizeof(int2)));
	int n = nt > np ? nt : np;
	hipLaunchKernelGGL(padded_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx().stream, t, q, g.c, g.k, g.s, pg.hh, pg.wh, g.ho, g.wo);
	BLA_HIP(hipGetLastError());
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	g_padded.push_back(PaddedTables{dev, g, t, q});
	*taps = t; *pix = q;
	return BLA_OK;
}

// A batch large enough to fill the chip with 128x128 tiles goes to the LDS-tiled gather kernel (bla_gemm.hip: same pipeline as the
// dense GEMM, the B slab fetched by 4-byte direct-to-LDS loads from computed addresses); small batches and odd shapes stay on the
// 32x32 wave-split-K gather kernel above.
static bool use_tiled_gather(const ConvArgs& a, int batch, int mode) {
	const long cols = mode == 1 ? (long)a.N * batch : (long)a.N;
	const long kk = mode == 1 ? (long)a.K : (long)a.K * batch;
	const long tiles = ((a.M + 127) / 128) * ((cols + 127) / 128);
	const bool aligned = a.lda % 4 == 0 && (uintptr_t)a.A % 16 == 0;
	if (!aligned || a.M < 64) return false;
	// forward / data gradient: enough tiles to fill half the chip -- with the contraction cut over workgroups where the padded-copy form
	// applies (8x8 and 4x4 maps of a batch: 64 and 16 tiles)
	if (mode == 1) return a.K % 16 == 0 && (tiles >= 128 || (a.M % 128 == 0 && cols % 128 == 0 && a.g.wo % 4 == 0 && tiles * gather3_splits(a.M, (int)cols, a.K) >= 128));
	return a.K % 16 == 0 && kk >= 1024 && tiles * batch >= 128;   // a.K = output pixels per image here
}

// Which kernel a forward-shaped pass (forward, stride-1 data gradient) runs on, decided in ONE place: conv2d_forward asks it whether the tile store
// will apply the epilogue, launch_implicit follows it.
enum FwdPath { FWD_WSK = 0, FWD_TILED_CHECKED = 1, FWD_TILED_PADDED = 3, FWD_TILED_WINDOW = 7 };
// The image window in LDS (gather mode 7 of bla_gemm_kernel.h): 3x3, stride 1, SAME padding of one pixel, image rows of 16 or 32 pixels, whole 128-pixel
// tiles inside one image, channels in groups of 16, one pass over K with a workgroup for about every CU.  BLA_CONV_WINDOW=0 keeps the padded copy.
static bool window_geometry(const ConvArgs& a, int batch) {
	static const bool off = [] { const char* e = getenv("BLA_CONV_WINDOW"); return e && e[0] == '0'; }();
	const ConvGeom& g = a.g;
	return !off && g.s == 1 && g.k == 3 && g.pt == 1 && g.pl == 1 && (g.w == 16 || g.w == 32) && g.ho == g.h && g.wo == g.w && (g.h * g.w) % 128 == 0 && g.c % 16 == 0 &&
	       a.M % 128 == 0 && gather3_splits(a.M, a.N * batch, a.K) == 1 && (long)(a.M / 128) * ((long)a.N * batch / 128) >= 2L * (ctx().num_cus > 0 ? ctx().num_cus : 256);
	// (two workgroups for every CU: with one, 256 -> 256 @16x16 x64 measured 193 us against 186 on the padded copy; with two, 128 -> 128 @32x32 x64 179 against 186)
}
// the kernel matrix [M][(c, t)] -> [M][(g, t, c16)], c = 16 g + c16: the contraction order of gather mode 7
__global__ void __launch_bounds__(kThreads) window_order_kernels_kernel(const float* __restrict__ a, float* __restrict__ out, int rows, int c_n) {
	const int kdim = c_n * 9, total = rows * kdim;
	for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		const int m = e / kdim, r = e - m * kdim, g = r / 144, q = r - g * 144, t = q >> 4, c16 = q & 15;
		out[e] = a[(size_t)m * kdim + (g * 16 + c16) * 9 + t];
	}
}
// the same for the data gradient, straight from the forward kernels kern[f][m][3][3]: row m, contraction (g, t, f16) <- kern[16 g + f16][m][8 - t]
__global__ void __launch_bounds__(kThreads) window_order_flipped_kernel(const float* __restrict__ kern, float* __restrict__ out, int rows, int f_n) {
	const int kdim = f_n * 9, total = rows * kdim;
	for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		const int m = e / kdim, r = e - m * kdim, g = r / 144, q = r - g * 144, t = q >> 4, f16 = q & 15;
		out[e] = kern[((size_t)(g * 16 + f16) * rows + m) * 9 + 8 - t];
	}
}
struct FwdPlan { FwdPath path; bool fuses_epilogue; };
static FwdPlan plan_forward(const ConvArgs& a, int batch) {
	if (!use_tiled_gather(a, batch, 1)) return FwdPlan{FWD_WSK, batch == 1};   // (the 32x32 kernel applies one bias set where it stores: a single image)
	const PaddedGeom pg = padded_geom(a.g);
	const bool fits32 = (size_t)batch * a.g.c * pg.plane_floats < ((size_t)1 << 29) && (long)batch * a.M * a.g.ho * a.g.wo < (1L << 29);
	if (!fits32 || a.g.wo % 4 != 0) return FwdPlan{FWD_TILED_CHECKED, false};
	const bool fuses = gather3_fuses_epilogue(a.M, a.N * batch, a.K);
	if (window_geometry(a, batch)) return FwdPlan{FWD_TILED_WINDOW, true};
	return FwdPlan{FWD_TILED_PADDED, fuses};
}

// tiling / K-splitting of the 32x32 wave-split-K gather kernel for one pass; a.batch and the strides are set
template <int MODE>
static bla_status plan_wsk_gather(ConvArgs& a, int batch, size_t* slab_bytes, dim3* grid, bool* vec, int target_wgs = 768) {
	a.tiles_n = (a.N + 31) / 32;
	int tiles = ((a.M + 31) / 32) * a.tiles_n;
	// long contractions over few tiles are latency-bound: cut K over workgroups until ~3 of them sit on every CU,
	// keeping >= 128 k per workgroup (a batch multiplies the workgroups of a forward pass, and the K-splits of a weight gradient)
	int splits = (target_wgs + tiles * batch - 1) / (tiles * batch);
	if (splits > a.K / 128) splits = a.K / 128;
	if (splits < 1) splits = 1;
	if (splits > 32) splits = 32;
	a.k_per_split = ((a.K + splits - 1) / splits + 31) / 32 * 32;
	splits = (a.K + a.k_per_split - 1) / a.k_per_split;
	a.kw = a.k_per_split / 4;   // multiple of 8
	a.slab = nullptr;
	a.psplits = splits;
	const bool wgrad = MODE == CONV_WGRAD;
	a.splits = wgrad ? splits * batch : splits;          // slabs folded into one output
	*slab_bytes = a.splits > 1 ? (size_t)splits * batch * a.M * a.N * sizeof(float) : 0;
	BLA_REQUIRE((long)splits * batch <= 65535 && batch <= 65535, BLA_ERR_INVALID, "batch %d too large for one launch", batch);
	*grid = dim3((unsigned)tiles, (unsigned)(wgrad ? splits * batch : splits), (unsigned)(wgrad ? 1 : batch));
	*vec = a.K % 4 == 0 && a.K >= 4 && a.lda % 4 == 0 && (uintptr_t)a.A % 16 == 0 && a.a_stride % 4 == 0;
	return BLA_OK;
}

template <int MODE>
static bla_status launch_implicit(hipStream_t s, ConvArgs& a, int batch = 1, size_t img_stride = 0, size_t out_stride = 0, size_t a_stride = 0) {
	a.batch = batch; a.img_stride = img_stride; a.out_stride = out_stride; a.a_stride = a_stride;
	const FwdPlan plan = MODE == CONV_FWD ? plan_forward(a, batch) : FwdPlan{FWD_WSK, false};
	// ep_fused_tiled is conv2d_forward's word that the tile store applies the adds: it took that from the same plan (no second copy of the predicate)
	BLA_REQUIRE(!a.ep_fused_tiled || (plan.fuses_epilogue && plan.path != FWD_WSK), BLA_ERR_INVALID, "internal: a fused epilogue was planned for a path that has none");
	if (MODE == CONV_FWD && plan.path == FWD_TILED_WINDOW) {
		// straight from the image: the kernel matrix goes into the window kernel's contraction order in the workspace, then one product
		const float* ordered = a.prepared_A;
		if (!ordered) {
			void* ws;
			bla_status st = ensure_workspace((size_t)a.M * a.K * sizeof(float) + 64, &ws);
			if (st) return st;
			if (a.flip_src) hipLaunchKernelGGL(window_order_flipped_kernel, dim3(grid_for((size_t)a.M * a.K)), dim3(kThreads), 0, s, a.flip_src, (float*)ws, a.M, a.g.c);
			else hipLaunchKernelGGL(window_order_kernels_kernel, dim3(grid_for((size_t)a.M * a.K)), dim3(kThreads), 0, s, a.A, (float*)ws, a.M, a.g.c);
			BLA_HIP(hipGetLastError());
			ordered = (const float*)ws;
		}
		const GatherEpilogue gep = {a.ep_bias, a.ep_bias_stride, a.ep_add, a.ep_out2};
		return gather_gemm(s, 7, batch, a.M, a.N * batch, a.K, ordered, a.K, a.out, a.ldo, a.img, nullptr, nullptr, a.g.h, a.g.w, a.N, (int)img_stride,
		                   a.ep_fused_tiled ? &gep : nullptr);
	}
	if (MODE == CONV_FWD ? plan.path != FWD_WSK : use_tiled_gather(a, batch, 2)) {
		const int2* ptab;
		bla_status st = get_pixel_table(s, a.g, &ptab);
		if (st) return st;
		// (the padded-copy kernels address the copy with 32-bit byte offsets: at most 2 GiB of it, and of del_y)
		const PaddedGeom pg = padded_geom(a.g);
		const size_t copy_floats = (size_t)batch * a.g.c * pg.plane_floats;
		const bool fits32 = copy_floats < ((size_t)1 << 29) && (long)batch * a.M * a.g.ho * a.g.wo < (1L << 29);
		if (MODE == CONV_FWD && plan.path == FWD_TILED_PADDED) {
			// pad (and split by stride parity) once, then the B slab is fetched with the same 16-byte DMA as a dense operand
			const int splits3 = gather3_splits(a.M, a.N * batch, a.K);
			const size_t slab_bytes = splits3 > 1 ? ((size_t)splits3 * a.M * a.N * batch * sizeof(float) + 255) / 256 * 256 : 0;
			void* ws;
			// (a 1x1 kernel on rows of whole float4 has no halo and no row padding: the image IS its padded copy)
			const float* ready = a.padded_src ? a.padded_src : (a.g.k == 1 && a.g.s == 1 && a.g.w % 4 == 0 && (uintptr_t)a.img % 16 == 0 ? a.img : nullptr);
			st = ensure_workspace(slab_bytes + (ready ? 0 : copy_floats * sizeof(float)) + 64, &ws);   // [slabs][padded copy]
			if (st) return st;
			const float* padded = ready;
			const int2 *taps, *pix;
			st = get_padded_tables(s, a.g, &taps, &pix);
			if (st) return st;
			if (!padded) {
				float* mine = (float*)((char*)ws + slab_bytes);
				launch_pad_split(s, a.img, mine, (unsigned)(batch * a.g.c), a.g.h, a.g.w, a.g.pt, a.g.pl, a.g.s, (unsigned)pg.hh, (unsigned)pg.wh);
				BLA_HIP(hipGetLastError());
				padded = mine;
			}
			const GatherEpilogue gep = {a.ep_bias, a.ep_bias_stride, a.ep_add, a.ep_out2};
			return gather_gemm(s, 3, batch, a.M, a.N * batch, a.K, a.A, a.lda, a.out, a.ldo, padded, taps, pix, pg.hh, pg.wh, a.N, (int)(a.g.c * pg.plane_floats),
			                   a.ep_fused_tiled ? &gep : nullptr);
		}
		if (MODE == CONV_FWD)   // columns = (image, output pixel), contraction over the taps
			return gather_gemm(s, 1, batch, a.M, a.N * batch, a.K, a.A, a.lda, a.out, a.ldo, a.img, a.tab, ptab, a.g.h, a.g.w, a.N, (int)img_stride);
		if (MODE == CONV_WGRAD && fits32 && a.g.wo % 4 == 0 && a.N % 4 == 0) {
			// transposed product on the padded copy -- taps are the rows, both operands stream in 16-byte chunks
			const size_t slab_bytes = ((size_t)gather_gemm_splits(4, batch, a.N, a.M, a.K) * a.M * a.N * sizeof(float) + 255) / 256 * 256;
			void* ws;
			// (a 1x1 kernel on rows of whole float4 has no halo and no row padding: the image IS its padded copy)
			const float* ready = a.padded_src ? a.padded_src : (a.g.k == 1 && a.g.s == 1 && a.g.w % 4 == 0 && (uintptr_t)a.img % 16 == 0 ? a.img : nullptr);
			st = ensure_workspace(slab_bytes + (ready ? 0 : copy_floats * sizeof(float)) + 64, &ws);   // [slabs][padded copy]
			if (st) return st;
			const float* padded = ready;
			const int2 *taps, *pix;
			st = get_padded_tables(s, a.g, &taps, &pix);
			if (st) return st;
			if (!padded) {
				float* mine = (float*)((char*)ws + slab_bytes);
				launch_pad_split(s, a.img, mine, (unsigned)(batch * a.g.c), a.g.h, a.g.w, a.g.pt, a.g.pl, a.g.s, (unsigned)pg.hh, (unsigned)pg.wh);
				BLA_HIP(hipGetLastError());
				padded = mine;
			}
			return gather_gemm(s, 4, batch, a.N, a.M, a.K * batch, a.A, a.lda, a.out, a.ldo, padded, pix, taps, pg.hh, pg.wh, a.K, (int)(a.g.c * pg.plane_floats), nullptr, a.g.wo);
		}
		// weight gradient: columns = taps, contraction over (image, output pixel); A = del_y [image][M][HWo]
		return gather_gemm(s, 2, batch, a.M, a.N, a.K * batch, a.A, a.lda, a.out, a.ldo, a.img, ptab, a.tab, a.g.h, a.g.w, a.K, (int)img_stride);
	}
	size_t slab_bytes;
	dim3 grid;
	bool vec;
	bla_status st = plan_wsk_gather<MODE>(a, batch, &slab_bytes, &grid, &vec);
	if (st) return st;
	const bool wgrad = MODE == CONV_WGRAD;
	if (a.splits > 1) {
		void* ws;
		st = ensure_workspace(slab_bytes, &ws);
		if (st) return st;
		a.slab = (float*)ws;
	}
	if (vec) hipLaunchKernelGGL((conv_implicit_kernel<MODE, true>), grid, dim3(256), 0, s, a);
	else hipLaunchKernelGGL((conv_implicit_kernel<MODE, false>), grid, dim3(256), 0, s, a);
	if (a.splits > 1)
		hipLaunchKernelGGL(conv_slab_reduce_kernel, dim3(grid_for((size_t)a.M * a.N), (unsigned)(wgrad ? 1 : batch)), dim3(kThreads), 0, s, a);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

// Weight gradient (w: WGRAD on the forward input) and data gradient (d: FWD on del_y with the flipped kernels) of one convolution
// as one gather launch + one fold launch; only for shapes that stay on the 32x32 kernel (a single image, a small batch).
static bla_status launch_backward_pair(hipStream_t s, ConvArgs& w, ConvArgs& d, int batch) {
	size_t bytes_w, bytes_d;
	dim3 gw, gd;
	bool vw, vd;
	const int target = 384;   // workgroups per product: the two share the chip (768 for a product launched alone); measured 128 ... 768
	bla_status st = plan_wsk_gather<CONV_WGRAD>(w, batch, &bytes_w, &gw, &vw, target);
	if (st) return st;
	st = plan_wsk_gather<CONV_FWD>(d, batch, &bytes_d, &gd, &vd, target);
	if (st) return st;
	bytes_w = (bytes_w + 255) / 256 * 256;
	if (bytes_w + bytes_d > 0) {
		void* ws;
		st = ensure_workspace(bytes_w + bytes_d, &ws);
		if (st) return st;
		if (w.splits > 1) w.slab = (float*)ws;
		if (d.splits > 1) d.slab = (float*)((char*)ws + bytes_w);
	}
	const unsigned blocks_w = gw.x * gw.y, blocks_d = gd.x * gd.y * gd.z;
	dim3 grid(blocks_w + blocks_d);
#define BLA_PAIR(VW, VD) hipLaunchKernelGGL((conv_backward_pair_kernel<VW, VD>), grid, dim3(256), 0, s, w, d, blocks_w, gw.x, gd.x, gd.y)
	if (vw && vd) BLA_PAIR(true, true);
	else if (vw) BLA_PAIR(true, false);
	else if (vd) BLA_PAIR(false, true);
	else BLA_PAIR(false, false);
#undef BLA_PAIR
	if (w.splits > 1 || d.splits > 1) {
		const unsigned rw = grid_for((size_t)w.M * w.N), rd = grid_for((size_t)d.M * d.N);
		hipLaunchKernelGGL(conv_slab_reduce_pair_kernel, dim3(rw + rd * (unsigned)batch), dim3(kThreads), 0, s, w, d, rw, rd);
	}
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

template <bool RELU>
static bla_status launch_group_norm(hipStream_t s, const float* in, float* out, float* stdevs, float* means, int channels, int group_size, int hw,
                                    const unsigned char* drop, float* dropped, const PadOut* pad = nullptr) {
	const int groups = (channels + group_size - 1) / group_size;
	const long n_max = (long)(channels < group_size ? channels : group_size) * hw;
	// pad: the padded copy of what feeds the next convolution (the dropped output when there is one, else the ReLU output) -- written by the 16-byte kernels in
	// the same pass; the scalar forms do not carry it and a padding pass follows them
	PadArgs pa = {};
	bool pad_written = false;
	if (pad && pad->dst) {
		BLA_REQUIRE(pad->L.plane > 0 && pad->L.w > 0 && hw % pad->L.w == 0 && pad->L.w % 4 == 0, BLA_ERR_INVALID, "padded by-product: bad layout");
		pa = PadArgs{pad->dst, hw, pad->L.w, pad->L.wh, pad->L.plane, pad->L.pt * pad->L.wh + pad->L.pl};
	}
	const bool vec = hw % 4 == 0 && ((uintptr_t)in | (uintptr_t)out | (uintptr_t)dropped) % 16 == 0 && (uintptr_t)drop % 4 == 0;
	// out == NULL (the caller keeps only the dropped form: the backward pass gates on it alone): the 16-byte kernels take that, the scalar ones do not
	BLA_REQUIRE(out || (dropped && vec && (n_max > kGnThreads * (kGnRegs / 2) || n_max <= kGnThreads * kGnVec * 4)), BLA_ERR_INVALID, "group norm without its plain output needs the 16-byte kernels");
	if (n_max <= kGnThreads * (kGnRegs / 2)) {   // (the one-workgroup kernel holds up to twice that in registers, but at 12 us against 9 sliced)
		if (n_max <= kGnThreads * kGnVec * 4 && vec) {
			hipLaunchKernelGGL(group_norm_vec_kernel<RELU>, dim3(groups), dim3(kGnThreads), 0, s, in, out, stdevs, means, channels, group_size, hw, drop, dropped, pa);
			pad_written = true;
		} else
			hipLaunchKernelGGL(group_norm_kernel<RELU>, dim3(groups), dim3(kGnThreads), 0, s, in, out, stdevs, means, channels, group_size, hw, drop, dropped);
	} else {
		const unsigned slices = (unsigned)((n_max + kGnSlice - 1) / kGnSlice);
		BLA_REQUIRE(groups <= 65535, BLA_ERR_INVALID, "too many groups (%d)", groups);
		void* ws;
		bla_status st = ensure_workspace((size_t)groups * slices * sizeof(double2), &ws);
		if (st) return st;
		if (vec) {
			hipLaunchKernelGGL(group_norm_stats_kernel<true>, dim3(slices, groups), dim3(kGnSliceThreads), 0, s, in, channels, group_size, hw, (double2*)ws);
			hipLaunchKernelGGL((group_norm_apply_kernel<RELU, true>), dim3(slices, groups), dim3(kGnSliceThreads), 0, s, in, out, stdevs, means, channels, group_size, hw, drop,
			                   dropped, (const double2*)ws, pa);
			pad_written = true;
		} else {
			hipLaunchKernelGGL(group_norm_stats_kernel<false>, dim3(slices, groups), dim3(kGnSliceThreads), 0, s, in, channels, group_size, hw, (double2*)ws);
			hipLaunchKernelGGL((group_norm_apply_kernel<RELU, false>), dim3(slices, groups), dim3(kGnSliceThreads), 0, s, in, out, stdevs, means, channels, group_size, hw, drop,
			                   dropped, (const double2*)ws, PadArgs{});
		}
	}
	BLA_HIP(hipGetLastError());
	if (pa.dst && !pad_written) {   // rows of whole float4 are a precondition above, so this is the unaligned-pointer case only
		const int h = hw / pad->L.w;
		launch_pad_split(s, dropped ? dropped : out, pad->dst, (unsigned)channels, h, pad->L.w, pad->L.pt, pad->L.pl, 1, (unsigned)(pad->L.plane / pad->L.wh), (unsigned)pad->L.wh);
		BLA_HIP(hipGetLastError());
	}
	return BLA_OK;
}

static bla_status launch_group_norm_ddx(hipStream_t s, const float* source, float* dest, const float* data, const float* means, const float* stdevs, int channels,
                                        int group_size, int hw, const float* relu_gate, const float* addend, const PadOut* pad = nullptr) {
	const int groups = (channels + group_size - 1) / group_size;
	const long n_max = (long)(channels < group_size ? channels : group_size) * hw;
	PadArgs pa = {};       // the gradient again, inside the zero-padded copy the data-gradient convolution behind it gathers from (as launch_group_norm)
	bool pad_written = false;
	if (pad && pad->dst) {
		BLA_REQUIRE(pad->L.plane > 0 && pad->L.w > 0 && hw % pad->L.w == 0 && pad->L.w % 4 == 0, BLA_ERR_INVALID, "padded by-product: bad layout");
		pa = PadArgs{pad->dst, hw, pad->L.w, pad->L.wh, pad->L.plane, pad->L.pt * pad->L.wh + pad->L.pl};
	}
	const bool vec = hw % 4 == 0 && ((uintptr_t)source | (uintptr_t)dest | (uintptr_t)data | (uintptr_t)relu_gate | (uintptr_t)addend) % 16 == 0;
	if (n_max <= kGnThreads * (kGnRegs / 2)) {
		if (n_max <= kGnThreads * kGnVec * 4 && vec) {
			hipLaunchKernelGGL(group_norm_ddx_vec_kernel, dim3(groups), dim3(kGnThreads), 0, s, source, dest, data, means, stdevs, channels, group_size, hw, relu_gate, addend, pa);
			pad_written = true;
		} else
			hipLaunchKernelGGL(group_norm_ddx_kernel, dim3(groups), dim3(kGnThreads), 0, s, source, dest, data, means, stdevs, channels, group_size, hw, relu_gate, addend);
	} else {
		const unsigned slices = (unsigned)((n_max + kGnSlice - 1) / kGnSlice);
		BLA_REQUIRE(groups <= 65535, BLA_ERR_INVALID, "too many groups (%d)", groups);
		void* ws;
		bla_status st = ensure_workspace((size_t)groups * slices * sizeof(double2), &ws);
		if (st) return st;
		if (vec) {
			hipLaunchKernelGGL(group_norm_ddx_stats_kernel<true>, dim3(slices, groups), dim3(kGnSliceThreads), 0, s, source, data, means, stdevs, channels, group_size, hw,
			                   relu_gate, (double2*)ws);
			hipLaunchKernelGGL(group_norm_ddx_apply_kernel<true>, dim3(slices, groups), dim3(kGnSliceThreads), 0, s, source, dest, data, means, stdevs, channels, group_size, hw,
			                   relu_gate, addend, (const double2*)ws, pa);
			pad_written = true;
		} else {
			hipLaunchKernelGGL(group_norm_ddx_stats_kernel<false>, dim3(slices, groups), dim3(kGnSliceThreads), 0, s, source, data, means, stdevs, channels, group_size, hw,
			                   relu_gate, (double2*)ws);
			hipLaunchKernelGGL(group_norm_ddx_apply_kernel<false>, dim3(slices, groups), dim3(kGnSliceThreads), 0, s, source, dest, data, means, stdevs, channels, group_size, hw,
			                   relu_gate, addend, (const double2*)ws, PadArgs{});
		}
	}
	BLA_HIP(hipGetLastError());
	if (pa.dst && !pad_written) {
		launch_pad_split(s, dest, pad->dst, (unsigned)channels, hw / pad->L.w, pad->L.w, pad->L.pt, pad->L.pl, 1, (unsigned)(pad->L.plane / pad->L.wh), (unsigned)pad->L.wh);
		BLA_HIP(hipGetLastError());
	}
	return BLA_OK;
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
	if (stride != 1 && strict_reference()) {
		set_error("_col2im iterates the image grid with out_row = i*stride + k (lib/conv.c:80-135): out of bounds for stride %d, undefined in the reference", stride);
		return BLA_ERR_UNDEFINED;
	}
	BLA_REQUIRE(d_cols && d_out, BLA_ERR_INVALID, "null operand");
	Geometry g = same_geometry(h, w, k, stride);
	if (stride == 1)
		hipLaunchKernelGGL(col2im_s1_kernel, dim3(grid_for((size_t)c_n * h * w)), dim3(kThreads), 0, pick_stream(stream), d_cols, d_out, h, w, k, c_n, g.pt, g.pl);
	else   // the intended operation: the adjoint of _im2col
		hipLaunchKernelGGL(col2im_adjoint_kernel, dim3(grid_for((size_t)c_n * h * w)), dim3(kThreads), 0, pick_stream(stream), d_cols, d_out, h, w, k, c_n, stride,
		                   g.ho, g.wo, g.pt, g.pl);
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
	if (stride != 1 && strict_reference()) {
		set_error("conv_ddx is undefined for stride %d: _col2im is only valid for stride 1 (lib/conv.c:80-135)", stride);
		return BLA_ERR_UNDEFINED;
	}
	BLA_REQUIRE(d_del_y && d_im2col && d_kmat && d_del_q && d_del_kmat && d_del_kern && d_del_col && d_del_x, BLA_ERR_INVALID, "null operand");
	// h, w are the INPUT's; del_y is [F][Ho][Wo] (stride 1: the same size -- the only case the reference defines)
	const Geometry gm = same_geometry(h, w, k, stride);
	const int hw = gm.ho * gm.wo, kkc = k * k * c_in;
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
	return bla_col2im_f32(stream, d_del_col, d_del_x, h, w, k, c_in, stride);                 // lib/conv.c:228
}

/* Device-resident convolution without the ConvData workspaces: out [F][Ho][Wo] = conv(x [C][H][W], kern [F][C][k][k]),
 * same values as conv()'s `output` (lib/conv.c:205-212, intended composition), any stride. */
// The adds the U-Net puts behind a convolution, for a batch in one pass: out[b][c][:] += bias[b * bias_stride + c] (bias_stride 0: one bias set for
// every image), out2 = out + add.
__global__ void __launch_bounds__(kThreads) conv_epilogue_kernel(float* __restrict__ out, const float* __restrict__ bias, int bias_stride, const float* __restrict__ add,
                                                                  float* __restrict__ out2, int f_n, int hw, size_t total) {
	for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (size_t)gridDim.x * kThreads) {
		const size_t row = e / hw;
		const int b = (int)(row / f_n), c = (int)(row - (size_t)b * f_n);
		float v = out[e];
		if (bias) { v += bias[(size_t)b * bias_stride + c]; out[e] = v; }
		if (out2) out2[e] = v + add[e];
	}
}

static bla_status conv2d_forward(void* stream, const float* d_x, const float* d_kern, float* d_out, int batch, int h, int w, int k, int c_in, int f_n, int stride,
                                 const float* ep_bias = nullptr, const float* ep_add = nullptr, float* ep_out2 = nullptr, int ep_bias_stride = 0,
                                 const float* x_padded = nullptr, const float* prepared = nullptr) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(batch > 0 && h > 0 && w > 0 && k > 0 && c_in > 0 && f_n > 0 && stride > 0, BLA_ERR_INVALID, "bad conv shape");
	BLA_REQUIRE(d_x && d_kern && d_out, BLA_ERR_INVALID, "null operand");
	hipStream_t s = pick_stream(stream);
	Geometry gm = same_geometry(h, w, k, stride);
	BLA_REQUIRE((ep_add == nullptr) == (ep_out2 == nullptr), BLA_ERR_INVALID, "ep_add and ep_out2 go together");
	if (thin_conv_applies(k, c_in, f_n, stride)) return thin_conv_forward(s, d_x, d_kern, d_out, batch, h, w, k, c_in, f_n, gm.pt, gm.pl, ep_bias, ep_bias_stride, ep_add, ep_out2);
	ConvArgs a;
	a.g = ConvGeom{h, w, k, c_in, stride, gm.ho, gm.wo, gm.pt, gm.pl};
	st = get_table(s, a.g, &a.tab);
	if (st) return st;
	a.A = d_kern; a.lda = k * k * c_in; a.img = d_x; a.out = d_out; a.ldo = gm.ho * gm.wo;
	a.M = f_n; a.N = gm.ho * gm.wo; a.K = k * k * c_in;
	a.padded_src = x_padded;
	const bool ep = ep_bias || ep_out2;
	if (prepared && plan_forward(a, batch).path == FWD_TILED_WINDOW) a.prepared_A = prepared;   // (mode 1 of conv_kernel_prep_mode: the only prepared form a forward pass takes)
	const FwdPlan plan = plan_forward(a, batch);
	if (ep && !plan.fuses_epilogue) {
		// the half-slab forward kernels apply the adds where they store their tiles (one pass over K, whole tiles) and the 32x32 kernel does for a single
		// image; the other kernels carry no epilogue (the 32x32 kernel's knows one bias set): one pass behind them
		st = launch_implicit<CONV_FWD>(s, a, batch, (size_t)c_in * h * w, (size_t)f_n * gm.ho * gm.wo, 0);
		if (st) return st;
		const size_t total = (size_t)batch * f_n * a.N;
		hipLaunchKernelGGL(conv_epilogue_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, d_out, ep_bias, ep_bias_stride, ep_add, ep_out2, f_n, a.N, total);
		BLA_HIP(hipGetLastError());
		return BLA_OK;
	}
	a.ep_bias = ep_bias; a.ep_bias_stride = ep_bias_stride; a.ep_add = ep_add; a.ep_out2 = ep_out2;
	a.ep_fused_tiled = ep && plan.path != FWD_WSK;
	return launch_implicit<CONV_FWD>(s, a, batch, (size_t)c_in * h * w, (size_t)f_n * gm.ho * gm.wo, 0);
}


// ---- data gradient of a stride-2 convolution by output parity (no zero work) ------------------------------------------------------------------
// del_x[c][y][x] = sum over f and the taps (p, q) with i*2 + p - pt = y, j*2 + q - pl = x of del_y[f][i][j] * K[f][c][p][q].  For the output pixels of
// one parity class (y = 2u + ry, x = 2v + rx) the taps are a fixed subset (p = p0, p0 + 2, ...; i = u - dp with dp = (p - pt - ry) / 2 >= 0), so the class
// is a stride-1 correlation of del_y with a sub-kernel: a gathered product over K = F * |P| * |Q| instead of the F * k * k of the zero-dilated form, which
// multiplies three zeros for every value.  The four classes run as four gathered products on one padded copy of del_y (dp_max zero rows on top, dq_max zero
// columns on the left) and land in a class-planar buffer that one pass interleaves into del_x.
struct ParityTaps { int n; int tap[4]; int d[4]; };   // taps of one parity along one axis: kernel index and the (non-negative) source shift
static ParityTaps parity_taps(int k, int pad, int r) {
	ParityTaps t = {0, {0, 0, 0, 0}, {0, 0, 0, 0}};
	for (int p = 0; p < k; p++)
		if ((p - pad - r) % 2 == 0 && p - pad - r >= 0 && t.n < 4) { t.tap[t.n] = p; t.d[t.n] = (p - pad - r) / 2; t.n++; }
	return t;
}
// A_class[c][(f, a, b)] = K[f][c][P.tap[a]][Q.tap[b]] for the four classes, back to back in `out`
struct ClassTable { GatherClass c[4]; };
__global__ void __launch_bounds__(kThreads) parity_kernels_kernel(const float* __restrict__ kern, float* __restrict__ out, int f_n, int c_n, int k, ParityTaps p0, ParityTaps p1,
                                                                   ParityTaps q0, ParityTaps q1, ClassTable tab, GatherClass* d_tab) {
	if (d_tab && blockIdx.x == 0 && threadIdx.x < 4) d_tab[threadIdx.x] = tab.c[threadIdx.x];   // the class launch's table (gather_gemm_classes reads it on the device)
	const int total = f_n * c_n * k * k;
	for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		int cls = 0, base = 0, r = e;
		for (; cls < 4; cls++) {
			const int kc = f_n * ((cls >> 1) ? p1.n : p0.n) * ((cls & 1) ? q1.n : q0.n) * c_n;
			if (r < kc) break;
			r -= kc; base += kc;
		}
		const ParityTaps& P = (cls >> 1) ? p1 : p0;
		const ParityTaps& Q = (cls & 1) ? q1 : q0;
		const int kk = f_n * P.n * Q.n, c = r / kk, t = r - c * kk, f = t / (P.n * Q.n), ab = t - f * (P.n * Q.n), a = ab / Q.n, b = ab - a * Q.n;
		out[base + r] = kern[(((size_t)f * c_n + c) * k + P.tap[a]) * k + Q.tap[b]];
	}
}
__global__ void __launch_bounds__(kThreads) parity_tables_kernel(int2* taps, int2* pix, int f_n, ParityTaps P, ParityTaps Q, int dpmax, int dqmax, int hh, int wh, int hc, int wc) {
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	const int nt = f_n * P.n * Q.n;
	if (e < nt) { const int f = e / (P.n * Q.n), ab = e % (P.n * Q.n), a = ab / Q.n, b = ab % Q.n; taps[e] = make_int2((f * hh + dpmax - P.d[a]) * wh + dqmax - Q.d[b], 0); }
	if (e < hc * wc) { const int u = e / wc, v = e - u * wc; pix[e] = make_int2(u * wh + v, 0); }
}
// del_x[b][c][2u + ry][2v + rx] = cls[ry * 2 + rx][b][c][u][v]
__global__ void __launch_bounds__(kThreads) parity_interleave_kernel(const float* __restrict__ cls, float* __restrict__ out, unsigned planes, int h, int w) {
	const unsigned hc = h / 2, wc = w / 2;
	const size_t per = (size_t)planes * hc * wc, total = (size_t)planes * h * (w / 2);
	for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (size_t)gridDim.x * kThreads) {   // two neighbouring pixels per thread
		const unsigned v = (unsigned)(e % wc); const size_t t = e / wc; const unsigned y = (unsigned)(t % h); const size_t pc = t / h;
		const size_t src = (pc * hc + y / 2) * wc + v;
		const float2 o = make_float2(cls[(size_t)((y & 1) * 2) * per + src], cls[(size_t)((y & 1) * 2 + 1) * per + src]);
		*reinterpret_cast<float2*>(out + (pc * h + y) * w + 2 * v) = o;
	}
}
struct ParityTables { int device, f_n, k, pt, pl, ho, wo; int2* taps[4]; int2* pix; };
static std::vector<ParityTables> g_parity;

static bool parity_dgrad_applies(int batch, int h, int w, int k, int c_in, int f_n, int stride, const Geometry& gm) {
	static const bool off = [] { const char* e = getenv("BLA_CONV_PARITY"); return e && e[0] == '0'; }();
	if (off || stride != 2 || h % 2 || w % 8 || k > 4 || f_n % 16 || c_in % 128 || ((long)batch * (h / 2) * (w / 2)) % 128) return false;
	for (int r = 0; r < 2; r++) {   // every tap must belong to a class with a non-negative shift, and the shifted rows must stay inside del_y
		const ParityTaps P = parity_taps(k, gm.pt, r), Q = parity_taps(k, gm.pl, r);
		if (P.n == 0 || Q.n == 0) return false;
	}
	const ParityTaps a = parity_taps(k, gm.pt, 0), b = parity_taps(k, gm.pt, 1), c = parity_taps(k, gm.pl, 0), d = parity_taps(k, gm.pl, 1);
	// (32-bit gather offsets: the padded copy of del_y and a class plane must stay under 2 GiB, or the dilated form takes the call)
	const size_t copy_floats = (size_t)batch * f_n * (gm.ho + 2) * ((gm.wo + 2 + 3) / 4 * 4), cls_floats = (size_t)batch * c_in * (h / 2) * (w / 2);
	return a.n + b.n == k && c.n + d.n == k && gm.ho == h / 2 && gm.wo == w / 2 && copy_floats < ((size_t)1 << 29) && cls_floats < ((size_t)1 << 29);
}

static bla_status conv2d_backward_parity(hipStream_t s, const float* d_del_y, const float* d_kern, float* d_del_x, float* d_scratch, int batch, int h, int w, int k,
                                         int c_in, int f_n, const Geometry& gm) {
	const int hc = h / 2, wc = w / 2;
	ParityTaps P[2] = {parity_taps(k, gm.pt, 0), parity_taps(k, gm.pt, 1)}, Q[2] = {parity_taps(k, gm.pl, 0), parity_taps(k, gm.pl, 1)};
	int dpmax = 0, dqmax = 0;
	for (int r = 0; r < 2; r++)
		for (int i = 0; i < 4; i++) { if (i < P[r].n && P[r].d[i] > dpmax) dpmax = P[r].d[i]; if (i < Q[r].n && Q[r].d[i] > dqmax) dqmax = Q[r].d[i]; }
	const int hh = gm.ho + dpmax, wh = (gm.wo + dqmax + 3) / 4 * 4;
	// tables, once per geometry and device
	ParityTables t;   // (a copy: the cache vector may grow under another caller once the lock is released)
	{
		const ParityTables* tb = nullptr;
		std::lock_guard<std::mutex> lk(g_table_mu);
		const int dev = ctx().device;
		for (const ParityTables& e : g_parity)
			if (e.device == dev && e.f_n == f_n && e.k == k && e.pt == gm.pt && e.pl == gm.pl && e.ho == gm.ho && e.wo == gm.wo) { tb = &e; break; }
		if (!tb) {
			bla_status st = table_build_allowed(s);
			if (st) return st;
			ParityTables n = {dev, f_n, k, gm.pt, gm.pl, gm.ho, gm.wo, {nullptr, nullptr, nullptr, nullptr}, nullptr};
			BLA_HIP(hipMalloc((void**)&n.pix, (size_t)hc * wc * sizeof(int2)));
			for (int cls = 0; cls < 4; cls++) {
				const ParityTaps& p = P[cls >> 1]; const ParityTaps& q = Q[cls & 1];
				const int nt = f_n * p.n * q.n;
				BLA_HIP(hipMalloc((void**)&n.taps[cls], (size_t)nt * sizeof(int2)));
				const int cnt = nt > hc * wc ? nt : hc * wc;
				hipLaunchKernelGGL(parity_tables_kernel, dim3((cnt + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx().stream, n.taps[cls], n.pix, f_n, p, q, dpmax, dqmax, hh,
				                   wh, hc, wc);
				BLA_HIP(hipGetLastError());
			}
			BLA_HIP(hipStreamSynchronize(ctx().stream));
			g_parity.push_back(n);
			tb = &g_parity.back();
		}
		t = *tb;
	}
	const int N = batch * hc * wc;
	size_t slab_bytes = 0;
	for (int cls = 0; cls < 4; cls++) {
		const int kc = f_n * P[cls >> 1].n * Q[cls & 1].n, sp = gather3_splits(c_in, N, kc);
		if (sp > 1) slab_bytes = std::max(slab_bytes, ((size_t)sp * c_in * N * sizeof(float) + 255) / 256 * 256);
	}
	const size_t copy_floats = (size_t)batch * f_n * hh * wh, cls_floats = (size_t)batch * c_in * hc * wc;
	BLA_REQUIRE(copy_floats < ((size_t)1 << 29) && cls_floats < ((size_t)1 << 29), BLA_ERR_INVALID, "batch too large for 32-bit gather offsets");
	void* ws;
	bla_status st = ensure_workspace(slab_bytes + (copy_floats + 4 * cls_floats) * sizeof(float) + 64 + 256, &ws);   // [slabs][padded del_y][four class planes][class table]
	if (st) return st;
	float* padded = (float*)((char*)ws + slab_bytes);
	float* planes = padded + (copy_floats + 3) / 4 * 4;
	GatherClass* d_tab = (GatherClass*)(((uintptr_t)(planes + 4 * cls_floats) + 63) / 64 * 64);
	launch_pad_split(s, d_del_y, padded, (unsigned)(batch * f_n), gm.ho, gm.wo, dpmax, dqmax, 1, (unsigned)hh, (unsigned)wh);
	size_t a_off = 0;
	GatherClass gc[4];
	for (int cls = 0; cls < 4; cls++) {
		const int kc = f_n * P[cls >> 1].n * Q[cls & 1].n;
		gc[cls] = GatherClass{d_scratch + a_off, kc, t.taps[cls], planes + (size_t)cls * cls_floats};
		a_off += (size_t)c_in * kc;
	}
	// one launch for the four classes where they fill the chip: longest contraction beside shortest on a CU (the workgroups are dealt in launch order:
	// classes sorted [longest, middle, shortest, middle] puts workgroup j and j + half the grid -- the two a CU holds -- on a long and a short one)
	const bool one_launch = gather_classes_fit(4, c_in, N);
	ClassTable sorted = {{gc[0], gc[1], gc[2], gc[3]}};
	if (one_launch) {
		int order[4] = {0, 1, 2, 3};
		for (int i = 0; i < 4; i++) for (int j = i + 1; j < 4; j++) if (gc[order[j]].K > gc[order[i]].K) { int x = order[i]; order[i] = order[j]; order[j] = x; }
		sorted = ClassTable{{gc[order[0]], gc[order[1]], gc[order[3]], gc[order[2]]}};
	}
	hipLaunchKernelGGL(parity_kernels_kernel, dim3(grid_for((size_t)f_n * c_in * k * k)), dim3(kThreads), 0, s, d_kern, d_scratch, f_n, c_in, k, P[0], P[1], Q[0], Q[1], sorted,
	                   one_launch ? d_tab : nullptr);
	BLA_HIP(hipGetLastError());
	if (one_launch) {
		st = gather_gemm_classes(s, batch, c_in, N, sorted.c, d_tab, 4, hc * wc, padded, t.pix, hh, wh, hc * wc, f_n * hh * wh);
		if (st) return st;
	} else {
		for (int cls = 0; cls < 4; cls++) {
			st = gather_gemm(s, 3, batch, c_in, N, gc[cls].K, gc[cls].A, gc[cls].K, gc[cls].C, hc * wc, padded, gc[cls].ktab, t.pix, hh, wh, hc * wc, f_n * hh * wh);
			if (st) return st;
		}
	}
	hipLaunchKernelGGL(parity_interleave_kernel, dim3(grid_for((size_t)batch * c_in * h * (w / 2))), dim3(kThreads), 0, s, planes, d_del_x, (unsigned)(batch * c_in), h, w);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

static bla_status conv2d_backward(void* stream, const float* d_del_y, const float* d_x, const float* d_kern, float* d_del_kern, float* d_del_x,
                                  float* d_scratch, int batch, int h, int w, int k, int c_in, int f_n, int stride, const float* x_padded = nullptr,
                                  const float* prepared = nullptr, const float* dy_padded = nullptr) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(batch > 0 && h > 0 && w > 0 && k > 0 && c_in > 0 && f_n > 0 && stride > 0, BLA_ERR_INVALID, "bad conv shape");
	BLA_REQUIRE(d_del_y, BLA_ERR_INVALID, "null operand");
	hipStream_t s = pick_stream(stream);
	Geometry gm = same_geometry(h, w, k, stride);
	const size_t x_sz = (size_t)c_in * h * w, y_sz = (size_t)f_n * gm.ho * gm.wo;
	if (thin_conv_applies(k, c_in, f_n, stride)) {
		// a side of at most four channels: direct kernels (bla_conv_thin.hip); the data gradient is the forward form on del_y with the flipped kernels
		if (d_del_kern) {
			BLA_REQUIRE(d_x, BLA_ERR_INVALID, "weight gradient needs the forward input");
			st = thin_conv_wgrad(s, d_del_y, d_x, d_del_kern, batch, h, w, k, c_in, f_n, gm.pt, gm.pl);
			if (st) return st;
		}
		if (d_del_x) {
			BLA_REQUIRE(d_kern && d_scratch, BLA_ERR_INVALID, "data gradient needs the kernels and a scratch buffer of F*C*k*k floats");
			hipLaunchKernelGGL(flip_kernels_kernel, dim3(grid_for((size_t)f_n * c_in * k * k)), dim3(kThreads), 0, s, d_kern, d_scratch, f_n, c_in, k);
			BLA_HIP(hipGetLastError());
			return thin_conv_forward(s, d_del_y, d_scratch, d_del_x, batch, h, w, k, f_n, c_in, k - 1 - gm.pt, k - 1 - gm.pl, nullptr, 0, nullptr, nullptr);
		}
		return BLA_OK;
	}
	if (d_del_kern && d_del_x && stride == 1) {
		// both gradients, both on the latency-bound kernel: one gather launch and one fold launch for the two
		BLA_REQUIRE(d_x && d_kern && d_scratch, BLA_ERR_INVALID, "the gradients need the forward input, the kernels and a scratch buffer of F*C*k*k floats");
		ConvArgs aw, ad;
		aw.g = ConvGeom{h, w, k, c_in, stride, gm.ho, gm.wo, gm.pt, gm.pl};
		aw.A = d_del_y; aw.lda = gm.ho * gm.wo; aw.img = d_x; aw.out = d_del_kern; aw.ldo = k * k * c_in;
		aw.M = f_n; aw.N = k * k * c_in; aw.K = gm.ho * gm.wo;
		aw.batch = batch; aw.img_stride = x_sz; aw.out_stride = 0; aw.a_stride = y_sz;
		ad.g = ConvGeom{h, w, k, f_n, 1, h, w, k - 1 - gm.pt, k - 1 - gm.pl};
		ad.A = d_scratch; ad.lda = k * k * f_n; ad.img = d_del_y; ad.out = d_del_x; ad.ldo = h * w;
		ad.M = c_in; ad.N = h * w; ad.K = k * k * f_n;
		ad.batch = batch; ad.img_stride = y_sz; ad.out_stride = x_sz; ad.a_stride = 0;
		if (!use_tiled_gather(aw, batch, 2) && !use_tiled_gather(ad, batch, 1)) {
			st = get_table(s, aw.g, &aw.tab);
			if (st) return st;
			st = get_table(s, ad.g, &ad.tab);
			if (st) return st;
			hipLaunchKernelGGL(flip_kernels_kernel, dim3(grid_for((size_t)f_n * c_in * k * k)), dim3(kThreads), 0, s, d_kern, d_scratch, f_n, c_in, k);
			BLA_HIP(hipGetLastError());
			return launch_backward_pair(s, aw, ad, batch);
		}
		// both on the tiled gather kernels (a batch that fills the chip): ONE launch for the two products -- the data gradient's workgroups move in as the weight
		// gradient's finish, each product's prologue / drain / tail under the other's body (gather_pair_kernel).  BLA_CONV_PAIR=0: one launch each.
		static const bool pair_on = [] { const char* e = getenv("BLA_CONV_PAIR"); return !(e && e[0] == '0'); }();
		const PaddedGeom pgw = padded_geom(aw.g), pgd = padded_geom(ad.g);
		const size_t copy_w = (size_t)batch * aw.g.c * pgw.plane_floats, copy_d = (size_t)batch * ad.g.c * pgd.plane_floats;
		const bool fits_w = copy_w < ((size_t)1 << 29) && (long)batch * aw.M * aw.g.ho * aw.g.wo < (1L << 29);
		ad.padded_src = k % 2 == 1 ? dy_padded : nullptr;
		const FwdPath dpath = plan_forward(ad, batch).path;
		static const long pair_max_cols = [] { const char* e = getenv("BLA_CONV_PAIR_COLS"); return e && *e ? atol(e) : 2048L; }();
		if (pair_on && (long)ad.N * batch <= pair_max_cols && use_tiled_gather(aw, batch, 2) && fits_w && (aw.g.wo == 4 || aw.g.wo == 8 || aw.g.wo % 16 == 0) && aw.N % 4 == 0 && gather_pair_fits(4, aw.N, aw.M) &&
		    (dpath == FWD_TILED_WINDOW || (dpath == FWD_TILED_PADDED && gather_pair_fits(3, ad.M, ad.N * batch)))) {
			const int2 *taps_w, *pix_w, *taps_d = nullptr, *pix_d = nullptr;
			st = get_padded_tables(s, aw.g, &taps_w, &pix_w);
			if (st) return st;
			if (dpath == FWD_TILED_PADDED) { st = get_padded_tables(s, ad.g, &taps_d, &pix_d); if (st) return st; }
			// the two products as gather_gemm would take them (launch_implicit's weight-gradient and forward branches)
			GatherProduct gw = {4, aw.N, aw.M, aw.K * batch, aw.A, aw.lda, aw.out, aw.ldo, nullptr, pix_w, taps_w, pgw.hh, pgw.wh, aw.K, (int)(aw.g.c * pgw.plane_floats), GatherEpilogue{}, aw.g.wo};
			GatherProduct gd = dpath == FWD_TILED_WINDOW
				? GatherProduct{7, ad.M, ad.N * batch, ad.K, nullptr, ad.K, ad.out, ad.ldo, ad.img, nullptr, nullptr, ad.g.h, ad.g.w, ad.N, (int)y_sz, GatherEpilogue{}}
				: GatherProduct{3, ad.M, ad.N * batch, ad.K, nullptr, ad.lda, ad.out, ad.ldo, nullptr, taps_d, pix_d, pgd.hh, pgd.wh, ad.N, (int)(ad.g.c * pgd.plane_floats), GatherEpilogue{}};
			// one workspace: [weight-gradient slabs][data-gradient slabs][padded x][padded del_y][kernel matrix in the data gradient's form], each only where needed
			const float* xp = x_padded ? x_padded : (aw.g.k == 1 && aw.g.w % 4 == 0 && (uintptr_t)d_x % 16 == 0 ? d_x : nullptr);
			const float* yp = dpath == FWD_TILED_PADDED ? (ad.padded_src ? ad.padded_src : (ad.g.k == 1 && ad.g.w % 4 == 0 && (uintptr_t)d_del_y % 16 == 0 ? d_del_y : nullptr)) : d_del_y;
			const bool need_kern = !prepared;
			auto up = [](size_t floats) { return (floats + 63) / 64 * 64; };
			const size_t slab_w = up(gather_product_slab_floats(gw, batch)), slab_d = up(gather_product_slab_floats(gd, batch));
			const size_t pad_w = xp ? 0 : up(copy_w), pad_d = yp ? 0 : up(copy_d), kern_f = need_kern && dpath == FWD_TILED_WINDOW ? up((size_t)ad.M * ad.K) : 0;
			void* ws;
			st = ensure_workspace((slab_w + slab_d + pad_w + pad_d + kern_f) * sizeof(float) + 64, &ws);
			if (st) return st;
			float* base = (float*)ws;
			float *w_slab = base, *d_slab = base + slab_w, *w_pad = d_slab + slab_d, *d_pad = w_pad + pad_w, *kbuf = d_pad + pad_d;
			if (!xp) { launch_pad_split(s, d_x, w_pad, (unsigned)(batch * aw.g.c), aw.g.h, aw.g.w, aw.g.pt, aw.g.pl, 1, (unsigned)pgw.hh, (unsigned)pgw.wh); xp = w_pad; }
			if (!yp) { launch_pad_split(s, d_del_y, d_pad, (unsigned)(batch * ad.g.c), ad.g.h, ad.g.w, ad.g.pt, ad.g.pl, 1, (unsigned)pgd.hh, (unsigned)pgd.wh); yp = d_pad; }
			if (prepared) gd.A = prepared;      // (conv_kernel_prep_mode 2 / 3: flipped, and window-ordered where the window kernel runs)
			else if (dpath == FWD_TILED_WINDOW) {
				hipLaunchKernelGGL(window_order_flipped_kernel, dim3(grid_for((size_t)ad.M * ad.K)), dim3(kThreads), 0, s, d_kern, kbuf, ad.M, ad.g.c);
				gd.A = kbuf;
			} else {
				hipLaunchKernelGGL(flip_kernels_kernel, dim3(grid_for((size_t)f_n * c_in * k * k)), dim3(kThreads), 0, s, d_kern, d_scratch, f_n, c_in, k);
				gd.A = d_scratch;
			}
			BLA_HIP(hipGetLastError());
			gw.img = xp;
			if (dpath == FWD_TILED_PADDED) gd.img = yp;
			return gather_pair_products(s, batch, gw, w_slab, gd, d_slab);
		}
		ad.padded_src = nullptr;
	}
	if (d_del_kern) {
		BLA_REQUIRE(d_x, BLA_ERR_INVALID, "weight gradient needs the forward input");
		ConvArgs a;
		a.g = ConvGeom{h, w, k, c_in, stride, gm.ho, gm.wo, gm.pt, gm.pl};
		st = get_table(s, a.g, &a.tab);
		if (st) return st;
		a.A = d_del_y; a.lda = gm.ho * gm.wo; a.img = d_x; a.out = d_del_kern; a.ldo = k * k * c_in;
		a.M = f_n; a.N = k * k * c_in; a.K = gm.ho * gm.wo;
		a.padded_src = x_padded;
		st = launch_implicit<CONV_WGRAD>(s, a, batch, x_sz, 0, y_sz);
		if (st) return st;
	}
	if (d_del_x) {
		if (stride != 1 && strict_reference()) {
			set_error("the data gradient is undefined in the reference for stride %d (_col2im, lib/conv.c:80-135)", stride);
			return BLA_ERR_UNDEFINED;
		}
		BLA_REQUIRE(d_kern && d_scratch, BLA_ERR_INVALID, "data gradient needs the kernels and a scratch buffer of F*C*k*k floats");
		if (parity_dgrad_applies(batch, h, w, k, c_in, f_n, stride, gm)) return conv2d_backward_parity(s, d_del_y, d_kern, d_del_x, d_scratch, batch, h, w, k, c_in, f_n, gm);
		// Stride s > 1 (the intended adjoint; the reference is undefined there): the same stride-1 convolution over del_y with s-1 zeros
		// put between its pixels, [F][(Ho-1)s+1][(Wo-1)s+1] -- the U-Net's three down-convolutions (model/cifar_unet.c:1105,1111,1115).
		const float* src = d_del_y;
		int hd = h, wd = w;
		size_t src_sz = y_sz;
		if (stride != 1) {
			hd = (gm.ho - 1) * stride + 1; wd = (gm.wo - 1) * stride + 1;
			src_sz = (size_t)f_n * hd * wd;
			void* ws;
			st = ensure_workspace2(src_sz * batch * sizeof(float), &ws);
			if (st) return st;
			hipLaunchKernelGGL(dilate_kernel, dim3(grid_for(src_sz * batch)), dim3(kThreads), 0, s, d_del_y, (float*)ws, batch * f_n, gm.ho, gm.wo, stride, hd, wd);
			BLA_HIP(hipGetLastError());
			src = (const float*)ws;
		}
		ConvArgs a;   // image = del_y [F][H][W] (or its dilated form), "input channels" = F, pads mirrored: k-1-pt, k-1-pl
		a.g = ConvGeom{hd, wd, k, f_n, 1, h, w, k - 1 - gm.pt, k - 1 - gm.pl};
		st = get_table(s, a.g, &a.tab);
		if (st) return st;
		a.A = d_scratch; a.lda = k * k * f_n; a.img = src; a.out = d_del_x; a.ldo = h * w;
		a.M = c_in; a.N = h * w; a.K = k * k * f_n;
		if (stride == 1 && k % 2 == 1) a.padded_src = dy_padded;   // (odd k: the mirrored pads are the forward's, so the copy has conv_padded_layout(h, w, k, 1))
		const bool window = plan_forward(a, batch).path == FWD_TILED_WINDOW;
		if (prepared && stride == 1) { if (window) a.prepared_A = prepared; else a.A = prepared; }   // conv_kernel_prep_mode 2 / 3: already flipped (and window-ordered)
		else if (window) a.flip_src = d_kern;      // flipped and window-ordered in one pass, inside launch_implicit
		else {
			hipLaunchKernelGGL(flip_kernels_kernel, dim3(grid_for((size_t)f_n * c_in * k * k)), dim3(kThreads), 0, s, d_kern, d_scratch, f_n, c_in, k);
			BLA_HIP(hipGetLastError());
		}
		st = launch_implicit<CONV_FWD>(s, a, batch, src_sz, x_sz, 0);
		if (st) return st;
	}
	return BLA_OK;
}

/* Device-resident convolution without the ConvData workspaces: out [F][Ho][Wo] = conv(x [C][H][W], kern [F][C][k][k]),
 * same values as conv()'s `output` (lib/conv.c:205-212, intended composition), any stride. */
bla_status bla_conv2d_forward_f32(void* stream, const float* d_x, const float* d_kern, float* d_out, int h, int w, int k, int c_in, int f_n, int stride) {
	return conv2d_forward(stream, d_x, d_kern, d_out, 1, h, w, k, c_in, f_n, stride);
}

/* Gradients of the same convolution (conv_ddx, lib/conv.c:214-229, intended composition) without del_Q / del_col:
 *   d_del_kern [F][C][k][k] = sum over output pixels of del_y x patches     (any stride)
 *   d_del_x    [C][H][W]    = conv(del_y, kernels transposed + flipped)     (stride 1 only, as in the reference)
 * Either output may be NULL.  d_scratch holds the flipped kernels (F*C*k*k floats) when d_del_x is requested. */
bla_status bla_conv2d_backward_f32(void* stream, const float* d_del_y, const float* d_x, const float* d_kern, float* d_del_kern, float* d_del_x,
                                   float* d_scratch, int h, int w, int k, int c_in, int f_n, int stride) {
	return conv2d_backward(stream, d_del_y, d_x, d_kern, d_del_kern, d_del_x, d_scratch, 1, h, w, k, c_in, f_n, stride);
}

/* `batch` images through the same kernels in one launch (SURVEY 8(d) cfg 5 "batch of 64"): x [B][C][H][W] -> out [B][F][Ho][Wo],
 * every image exactly what bla_conv2d_forward_f32 gives for it.  The reference has no batch dimension (one conv() per image). */
bla_status bla_conv2d_forward_batched_f32(void* stream, const float* d_x, const float* d_kern, float* d_out, int batch, int h, int w, int k, int c_in,
                                          int f_n, int stride) {
	return conv2d_forward(stream, d_x, d_kern, d_out, batch, h, w, k, c_in, f_n, stride);
}

/* Batched gradients: d_del_x [B][C][H][W] per image; d_del_kern = SUM over the images of the per-image weight gradient
 * (slabs per (image, k-split) folded in image order -- deterministic), i.e. what `batch` conv_ddx calls accumulate to. */
bla_status bla_conv2d_backward_batched_f32(void* stream, const float* d_del_y, const float* d_x, const float* d_kern, float* d_del_kern, float* d_del_x,
                                           float* d_scratch, int batch, int h, int w, int k, int c_in, int f_n, int stride) {
	return conv2d_backward(stream, d_del_y, d_x, d_kern, d_del_kern, d_del_x, d_scratch, batch, h, w, k, c_in, f_n, stride);
}

bla_status bla_group_norm_f32(void* stream, const float* d_in, float* d_out, float* d_stdevs, float* d_means, int channels, int group_size, int hw) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0, BLA_ERR_INVALID, "bad group_norm shape channels=%d group=%d hw=%d", channels, group_size, hw);
	BLA_REQUIRE(d_in && d_out && d_stdevs && d_means, BLA_ERR_INVALID, "null operand");
	return launch_group_norm<false>(pick_stream(stream), d_in, d_out, d_stdevs, d_means, channels, group_size, hw, nullptr, nullptr);
}

bla_status bla_group_norm_relu_f32(void* stream, const float* d_in, float* d_out, float* d_stdevs, float* d_means, int channels, int group_size, int hw) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0, BLA_ERR_INVALID, "bad group_norm shape channels=%d group=%d hw=%d", channels, group_size, hw);
	BLA_REQUIRE(d_in && d_out && d_stdevs && d_means, BLA_ERR_INVALID, "null operand");
	return launch_group_norm<true>(pick_stream(stream), d_in, d_out, d_stdevs, d_means, channels, group_size, hw, nullptr, nullptr);
}

bla_status bla_group_norm_ddx_f32(void* stream, const float* d_source, float* d_dest, const float* d_data, const float* d_means, const float* d_stdevs,
                                  int channels, int group_size, int hw) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0, BLA_ERR_INVALID, "bad group_norm shape channels=%d group=%d hw=%d", channels, group_size, hw);
	BLA_REQUIRE(d_source && d_dest && d_data && d_means && d_stdevs, BLA_ERR_INVALID, "null operand");
	return launch_group_norm_ddx(pick_stream(stream), d_source, d_dest, d_data, d_means, d_stdevs, channels, group_size, hw, nullptr, nullptr);
}

}  // extern "C"

namespace bla {
bla_status conv2d_forward_epilogue(void* stream, const float* d_x, const float* d_kern, float* d_out, int h, int w, int k, int c_in, int f_n, int stride,
                                   const float* ep_bias, const float* ep_add, float* ep_out2, int batch, int ep_bias_stride, const float* x_padded, const float* prepared) {
	return conv2d_forward(stream, d_x, d_kern, d_out, batch, h, w, k, c_in, f_n, stride, ep_bias, ep_add, ep_out2, ep_bias_stride, x_padded, prepared);
}
bla_status conv2d_backward_batched(void* stream, const float* d_del_y, const float* d_x, const float* d_kern, float* d_del_kern, float* d_del_x, float* d_scratch, int batch,
                                   int h, int w, int k, int c_in, int f_n, int stride, const float* x_padded, const float* prepared, const float* dy_padded) {
	return conv2d_backward(stream, d_del_y, d_x, d_kern, d_del_kern, d_del_x, d_scratch, batch, h, w, k, c_in, f_n, stride, x_padded, prepared, dy_padded);
}
// Which prepared form (KernelPrepJob::mode) this convolution's forward / stride-1 data-gradient product reads; the same planning as conv2d_forward /
// conv2d_backward (a 16-byte aligned kernel matrix is assumed: the U-Net's parameter bucket aligns every tensor)
int conv_kernel_prep_mode(int batch, int h, int w, int k, int c_in, int f_n, int stride, bool data_gradient) {
	if (batch < 2 || thin_conv_applies(k, c_in, f_n, stride) || !ctx().ready) return 0;
	const Geometry gm = same_geometry(h, w, k, stride);
	ConvArgs a;
	a.A = reinterpret_cast<const float*>(uintptr_t(256));
	if (!data_gradient) {
		a.g = ConvGeom{h, w, k, c_in, stride, gm.ho, gm.wo, gm.pt, gm.pl};
		a.lda = k * k * c_in; a.M = f_n; a.N = gm.ho * gm.wo; a.K = k * k * c_in;
		return plan_forward(a, batch).path == FWD_TILED_WINDOW ? 1 : 0;
	}
	if (stride != 1) return 0;                 // parity classes / zero dilation build their own sub-kernels
	a.g = ConvGeom{h, w, k, f_n, 1, h, w, k - 1 - gm.pt, k - 1 - gm.pl};
	a.lda = k * k * f_n; a.M = c_in; a.N = h * w; a.K = k * k * f_n;
	// (conv2d_backward's one-launch pair for shapes that stay on the 32x32 kernel flips for itself: no prepared form there)
	ConvArgs aw;
	aw.g = ConvGeom{h, w, k, c_in, stride, gm.ho, gm.wo, gm.pt, gm.pl};
	aw.A = a.A; aw.lda = gm.ho * gm.wo; aw.M = f_n; aw.N = k * k * c_in; aw.K = gm.ho * gm.wo;
	if (!use_tiled_gather(aw, batch, 2) && !use_tiled_gather(a, batch, 1)) return 0;
	return plan_forward(a, batch).path == FWD_TILED_WINDOW ? 2 : 3;
}
// Modes 2 and 3 transpose the (f, c) axes of [F][C][k*k]: read and written through a 16 x 16 tile of k*k-vectors in LDS so that both sides move whole
// 16 * k*k-float runs (element by element the reads were 36-byte pieces a row apart: 49 us for the U-Net's 35 matrices).  Mode 1 only permutes inside a row.
__global__ void __launch_bounds__(kThreads) prepare_kernels_kernel(const KernelPrepJob* __restrict__ jobs) {
	const KernelPrepJob j = jobs[blockIdx.y];
	const int kk = j.k * j.k;
	if (j.mode == 1 || j.f_n % 16 || j.c_n % 16 || kk > 9) {
		const int total = j.f_n * j.c_n * kk;
		for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
			if (j.mode == 1) {          // window_order_kernels_kernel: rows f, contraction (g, t, c16)
				const int kdim = j.c_n * 9, m = e / kdim, r = e - m * kdim, g = r / 144, q = r - g * 144, t = q >> 4, c16 = q & 15;
				j.dst[e] = j.src[(size_t)m * kdim + (g * 16 + c16) * 9 + t];
			} else if (j.mode == 2) {   // window_order_flipped_kernel: rows c, contraction (g, t, f16)
				const int kdim = j.f_n * 9, m = e / kdim, r = e - m * kdim, g = r / 144, q = r - g * 144, t = q >> 4, f16 = q & 15;
				j.dst[e] = j.src[((size_t)(g * 16 + f16) * j.c_n + m) * 9 + 8 - t];
			} else {                    // flip_kernels_kernel
				const int pq = e % kk, f = (e / kk) % j.f_n, c = e / (kk * j.f_n);
				j.dst[e] = j.src[((size_t)f * j.c_n + c) * kk + kk - 1 - pq];
			}
		}
		return;
	}
	__shared__ float tile[16][16 * 9 + 1];          // [f16][c16 * kk + pq]
	const int tiles_c = j.c_n / 16, tiles = (j.f_n / 16) * tiles_c, run = 16 * kk;
	for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
		const int f0 = (t / tiles_c) * 16, c0 = (t % tiles_c) * 16;
		for (int e = threadIdx.x; e < 16 * run; e += kThreads) { const int f = e / run, r = e - f * run; tile[f][r] = j.src[((size_t)(f0 + f) * j.c_n + c0) * kk + r]; }
		__syncthreads();
		for (int e = threadIdx.x; e < 16 * run; e += kThreads) {
			const int c = e / run, r = e - c * run;
			if (j.mode == 3) {          // dst[c][f][pq] = src[f][c][kk - 1 - pq]
				const int f = r / kk, pq = r - f * kk;
				j.dst[((size_t)(c0 + c) * j.f_n + f0) * kk + r] = tile[f][c * kk + kk - 1 - pq];
			} else {                    // dst[c][(g, t, f16)] = src[16 g + f16][c][8 - t]   (k = 3; g = f0 / 16)
				const int tp = r >> 4, f = r & 15;
				j.dst[(size_t)(c0 + c) * j.f_n * 9 + (size_t)(f0 / 16) * 144 + r] = tile[f][c * 9 + 8 - tp];
			}
		}
		__syncthreads();
	}
}
bla_status conv_prepare_kernels(void* stream, const KernelPrepJob* d_jobs, int njobs, size_t max_elements) {
	if (njobs <= 0) return BLA_OK;
	BLA_REQUIRE(d_jobs && njobs <= 65535, BLA_ERR_INVALID, "bad kernel preparation table");
	const unsigned bx = (unsigned)std::min<size_t>(512, std::max<size_t>(1, max_elements / (16 * 16 * 9)));   // about one workgroup per 16 x 16 tile of the largest matrix
	hipLaunchKernelGGL(prepare_kernels_kernel, dim3(bx, (unsigned)njobs), dim3(kThreads), 0, pick_stream(stream), d_jobs);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}
// layout of the zero-padded copy a stride-1 convolution of this geometry gathers from (padded_geom / pad_split_kernel<1>): per plane hh rows of wh floats, the
// image at (pt, pl); 0 = this geometry has no padded copy a producer could write (stride != 1, or rows that are no multiple of four pixels)
PadLayout conv_padded_layout(int h, int w, int k, int stride) {
	PadLayout L = {};
	if (stride != 1 || w % 4 != 0) return L;
	const Geometry gm = same_geometry(h, w, k, 1);
	const PaddedGeom pg = padded_geom(ConvGeom{h, w, k, 1, 1, gm.ho, gm.wo, gm.pt, gm.pl});
	L.w = w; L.wh = pg.wh; L.plane = (int)pg.plane_floats; L.pt = gm.pt; L.pl = gm.pl;
	return L;
}
bla_status group_norm_relu_dropout(void* stream, const float* d_in, float* d_relu, const unsigned char* d_drop, float* d_dropped, float* d_stdevs, float* d_means,
                                   int channels, int group_size, int hw, const PadOut* pad) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0, BLA_ERR_INVALID, "bad group_norm shape channels=%d group=%d hw=%d", channels, group_size, hw);
	BLA_REQUIRE(d_in && (d_relu || d_dropped) && d_stdevs && d_means && (d_drop == nullptr) == (d_dropped == nullptr) && (d_drop || pad), BLA_ERR_INVALID, "null operand");
	return launch_group_norm<true>(pick_stream(stream), d_in, d_relu, d_stdevs, d_means, channels, group_size, hw, d_drop, d_dropped, pad);
}

bla_status group_norm_ddx_gated(void* stream, const float* d_source, float* d_dest, const float* d_data, const float* d_means, const float* d_stdevs,
                                int channels, int group_size, int hw, const float* d_relu_gate, const float* d_addend, const PadOut* pad) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(channels > 0 && group_size > 0 && hw > 0, BLA_ERR_INVALID, "bad group_norm shape channels=%d group=%d hw=%d", channels, group_size, hw);
	BLA_REQUIRE(d_source && d_dest && d_data && d_means && d_stdevs, BLA_ERR_INVALID, "null operand");
	return launch_group_norm_ddx(pick_stream(stream), d_source, d_dest, d_data, d_means, d_stdevs, channels, group_size, hw, d_relu_gate, d_addend, pad);
}
}  // namespace bla
