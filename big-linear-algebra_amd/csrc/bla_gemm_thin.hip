// bla_gemm_thin.hip -- batched products with a SHORT contraction (k <= 64) and a large output: the self-attention block's S x S scores
// Q K^T and del_S = del_P V^T (k = key_dim = 16), its dense output (P W + b)^T and the three d-deep terms of del_X
// (model/cifar_unet.c:1007-1021, 1303-1334).  At batch 64 each writes 16.8 MB for 0.13 GFLOP: HBM-bound, and on the 32 x 32-tile wave-split-K
// kernel (four waves share 16 k-values, LDS fold, one barrier per 4 KB of output) they ran at ~0.6 TB/s.  Here one WAVE owns a 32 x (32 NT)
// block of the output for the whole contraction: operand fragments come straight from global memory into MFMA registers (no LDS, no barrier),
// up to three (A, B) pairs accumulate into the same tile before it is stored once.
//   C[b] = beta * C[b] + alpha * sum_j op(A_j[b]) op(B_j[b]) (+ bias_row), optional copy of the pre-beta value to `pre`.
// v_mfma_f32_32x32x2_f32 sums two k-values per step, one from each half of the wave: half h takes k in [h k/2, (h+1) k/2), so a lane's values are
// consecutive in memory for a K-contiguous operand (16-byte loads).  The order of the k-sum differs from the LDS kernels'; it is fixed.
#include "bla_internal.h"

namespace bla {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ThinArgs {
	ThinPart part[3];
	int nparts, m, n, k, tiles_n;     // tiles_n: wave jobs along n (each 32 NT wide)
	float* C; int ldc; long sc;
	float alpha, beta;
	const float* bias_row;
	float* pre; int ld_pre; long spre;
};

// 4 consecutive k-values of one operand row/column for this lane: KC = contiguous along k
template <bool KC, bool VEC>
__device__ __forceinline__ void thin_load4(const float* __restrict__ base, int ld, int idx, int k0, float (&v)[4]) {
	if (KC) {
		const float* p = base + (size_t)idx * ld + k0;
		if (VEC) { const float4 x = *reinterpret_cast<const float4*>(p); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
		else { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; }
	} else {
#pragma unroll
		for (int t = 0; t < 4; t++) v[t] = base[(size_t)(k0 + t) * ld + idx];
	}
}

template <bool AKC, bool BKC, bool VEC, int NT>
__global__ void __launch_bounds__(256) gemm_thin_kernel(ThinArgs p) {
	const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
	const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int tile_m = job / p.tiles_n, tile_n = job - tile_m * p.tiles_n;
	if (tile_m * 32 >= p.m) return;                                    // whole waves leave together: no barrier in this kernel
	const int b = blockIdx.y, m0 = tile_m * 32, n0 = tile_n * 32 * NT, ks = p.k >> 1;
	f32x16 acc[NT];
#pragma unroll
	for (int j = 0; j < NT; j++)
#pragma unroll
		for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
	for (int q = 0; q < p.nparts; q++) {
		const float* A = p.part[q].A + (size_t)b * p.part[q].sa;
		const float* B = p.part[q].B + (size_t)b * p.part[q].sb;
		const int lda = p.part[q].lda, ldb = p.part[q].ldb;
		for (int t0 = 0; t0 < ks; t0 += 4) {
			float a[4], bv[NT][4];
			thin_load4<AKC, VEC>(A, lda, m0 + l31, h * ks + t0, a);
#pragma unroll
			for (int j = 0; j < NT; j++) thin_load4<BKC, VEC>(B, ldb, n0 + 32 * j + l31, h * ks + t0, bv[j]);
#pragma unroll
			for (int t = 0; t < 4; t++)
#pragma unroll
				for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv[j][t], acc[j], 0, 0, 0);
		}
	}
	float* C = p.C + (size_t)b * p.sc;
	float* pre = p.pre ? p.pre + (size_t)b * p.spre : nullptr;
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const int row = m0 + (r >> 2) * 8 + h * 4 + (r & 3);
		const float bias = p.bias_row ? p.bias_row[row] : 0.f;
#pragma unroll
		for (int j = 0; j < NT; j++) {
			const int col = n0 + 32 * j + l31;
			float v = p.alpha * acc[j][r] + bias;
			if (pre) pre[(size_t)row * p.ld_pre + col] = v;
			float* dst = C + (size_t)row * p.ldc + col;
			if (p.beta != 0.f) v += p.beta * *dst;
			*dst = v;
		}
	}
}

template <bool AKC, bool BKC, bool VEC>
void launch_thin(hipStream_t s, const ThinArgs& a, int batch, int nt) {
	const int jobs = (a.m / 32) * a.tiles_n;
	const dim3 grid((unsigned)((jobs + 3) / 4), (unsigned)batch);
	if (nt == 4) hipLaunchKernelGGL((gemm_thin_kernel<AKC, BKC, VEC, 4>), grid, dim3(256), 0, s, a);
	else if (nt == 2) hipLaunchKernelGGL((gemm_thin_kernel<AKC, BKC, VEC, 2>), grid, dim3(256), 0, s, a);
	else hipLaunchKernelGGL((gemm_thin_kernel<AKC, BKC, VEC, 1>), grid, dim3(256), 0, s, a);
}
}  // namespace

bool gemm_thin_applies(int m, int n, int k, int batch) {
	static const bool enabled = [] { const char* v = getenv("BLA_GEMM_THIN"); return !(v && v[0] == '0'); }();
	return enabled && k >= 8 && k <= 64 && k % 8 == 0 && m % 32 == 0 && n % 32 == 0 && (long)m * n * batch >= (1L << 20);
}

bla_status gemm_thin_parts(void* stream, int transa, int transb, int m, int n, int k, const ThinPart* parts, int nparts, float* C, int ldc, long stride_c, int batch,
                           float alpha, float beta, const float* bias_row, float* pre, int ld_pre, long stride_pre) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(parts && nparts >= 1 && nparts <= 3 && C && batch >= 1, BLA_ERR_INVALID, "bad thin product");
	BLA_REQUIRE(gemm_thin_applies(m, n, k, batch), BLA_ERR_INVALID, "thin product: k %d must be a multiple of 8 up to 64, m %d and n %d multiples of 32", k, m, n);
	ThinArgs a = {};
	bool vec = true;
	const bool akc = !transa, bkc = transb != 0;
	for (int j = 0; j < nparts; j++) {
		a.part[j] = parts[j];
		BLA_REQUIRE(parts[j].A && parts[j].B, BLA_ERR_INVALID, "null operand");
		if (akc) vec = vec && parts[j].lda % 4 == 0 && parts[j].sa % 4 == 0 && (uintptr_t)parts[j].A % 16 == 0;
		if (bkc) vec = vec && parts[j].ldb % 4 == 0 && parts[j].sb % 4 == 0 && (uintptr_t)parts[j].B % 16 == 0;
	}
	a.nparts = nparts; a.m = m; a.n = n; a.k = k;
	a.C = C; a.ldc = ldc; a.sc = stride_c; a.alpha = alpha; a.beta = beta; a.bias_row = bias_row; a.pre = pre; a.ld_pre = ld_pre; a.spre = stride_pre;
	// widest wave job that still leaves every wave slot of the chip a job (4 SIMDs x 8 waves per CU)
	int nt = 4;
	while (nt > 1 && (n % (32 * nt) != 0 || (long)(m / 32) * (n / (32 * nt)) * batch < 8L * ctx().num_cus)) nt >>= 1;
	a.tiles_n = n / (32 * nt);
	hipStream_t s = pick_stream(stream);
	if (akc && bkc) { if (vec) launch_thin<true, true, true>(s, a, batch, nt); else launch_thin<true, true, false>(s, a, batch, nt); }
	else if (akc && !bkc) { if (vec) launch_thin<true, false, true>(s, a, batch, nt); else launch_thin<true, false, false>(s, a, batch, nt); }
	else if (!akc && bkc) { if (vec) launch_thin<false, true, true>(s, a, batch, nt); else launch_thin<false, true, false>(s, a, batch, nt); }
	else launch_thin<false, false, false>(s, a, batch, nt);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // namespace bla
