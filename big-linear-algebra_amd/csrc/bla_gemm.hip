// bla_gemm.hip -- fp32 MFMA GEMM for gfx950 (CDNA4), written for 64-wide wavefronts.
//
// Replaces the reference's triple loop matrix_multiply_inplace (lib/matrix.c:47-57) and,
// through transa/transb, every transpose-multiply-transpose sandwich around it
// (model/mnist_nn.c:267-292, lib/conv.c:221-227).
//
// Structure (one workgroup = WM x WN waves, one BM x BN output tile, K walked in BK slabs):
//   * operands are staged global -> VGPR -> LDS with 16-byte loads; the loads of slab t+1 are
//     issued before the MFMAs of slab t and written to the other LDS buffer after them
//     (issue-early / write-late, one barrier per slab, two LDS buffers);
//   * an operand whose K index is contiguous in memory ("KC": A of NN/NT, B of NT) lives in LDS
//     as [row][BK+4] and a lane fetches 4 consecutive k with one ds_read_b128 (row stride
//     BK+4 floats = odd multiple of 16 B -> the 16 lanes of a b128 group hit 16 distinct slots);
//     an operand whose row index is contiguous ("RC": B of NN/TN, A of TN) lives as [k][rows]
//     and a lane fetches its 4 k with four ds_read_b32 (32 consecutive floats per half-wave);
//   * v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31].
//     Register j (0..3) of a fragment holds k = 8*kk + 4*(l>>5) + j, so MFMA j of a group
//     contracts the k pair {8kk + j, 8kk + 4 + j}; A and B use the same map, hence every k is
//     visited exactly once (order differs from the reference's ascending chain: fp32 rounding
//     only, see DESIGN.md "tolerances");
//   * accumulators: TM x TN blocks of 32x32 per wave (16 VGPRs each), C/D map
//     col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5);
//   * blockIdx -> tile map is XCD-aware (the 8 XCDs take contiguous chunks of a grouped tile
//     order so tiles that share A rows / B columns share an L2);
//   * small problems use a 64x64 tile and split K over blockIdx.z into fp32 slabs that a second
//     kernel reduces in a fixed order (deterministic, no atomics) and runs the epilogue on.
#include "bla_internal.h"
#include <cstdlib>

namespace bla {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
	const float* A; const float* B; float* C;
	int M, N, K, lda, ldb, ldc;
	int tiles_m, tiles_n, k_per_split, splits;
	float* slab;  // splits > 1: partial products [split][M][N]
	float alpha, beta;
	const float* bias_row; const float* bias_col;
	float* pre_act; int ld_pre; int act;
	const float* relu_mask; int ld_mask;
#ifdef BLA_WSK_DIAG
	unsigned long long* stamps;    // diagnostics build only
#endif
	unsigned* counters;            // wsk kernels with splits > 1: one arrival counter per output tile (zero between launches)
	float* row_sum_a;              // fused bias gradient: row_sum_a[r] = sum_k op(A)[r][k]   (wsk kernels, A K-contiguous)
	float rs_alpha, rs_beta;       // ... stored as rs_beta * old + rs_alpha * sum (1, 0 = plain)
	const float* softmax_y; float softmax_scale; float* softmax_grad;   // fused column softmax + (p - y)*scale (wsk kernels, M <= 32)
	double* sm_loss; unsigned* sm_correct;   // optional per-column accumulators of the loss / accuracy bookkeeping (model/mnist_nn.c:237-257)
	// implicit-GEMM convolution over a batch of images (gather variants of the direct-to-LDS kernel only): the B operand is
	// never stored, element (k, n) is img[g_off(k) + g_off(n)] when (y(k) + y(n), x(k) + x(n)) lies inside the H x W image, else 0.
	//   mode 1 (forward / data gradient): n = (image, output pixel), k = tap (c, p, q);  C is written as [image][M][HWo]
	//   mode 2 (weight gradient):         n = tap,                   k = (image, output pixel);  A = del_y [image][M][HWo]
	//   mode 3 (forward, stride 1, zero-PADDED image copy): as mode 1 without bounds checks, and four consecutive output pixels of a
	//           row are four consecutive floats of the padded image: 16-byte DMA exactly like a dense row-contiguous operand
	//   mode 4 (weight gradient, stride 1, padded copy), transposed: C'[tap][f] = sum_(image,pixel) P[tap][(image,pixel)] . del_y[f][(image,pixel)]:
	//           both operands K-contiguous (the gathered one in 16-byte chunks of four pixels), B = del_y [image][N][HWo], the tile is
	//           stored transposed (dkern [f][tap], or slab [split][f][tap])
	const float* g_img; const float* g_zero;
	const int2* g_ktab; const int2* g_ntab;   // {element offset, y | x << 16} per tap / per output pixel
	int g_mode, g_H, g_W, g_HWo, g_img_stride;
	// mode 3 on the half-slab pipeline, one pass over K: the adds the U-Net puts behind a convolution, applied where the tile is stored
	// (out = product + g_bias[image * g_bias_stride + row]; g_out2 = out + g_add, same layout as out; each optional)
	const float* g_bias; int g_bias_stride; const float* g_add; float* g_out2;
	int rc_global;   // host-side only: pick the instantiation that fetches row-contiguous operands with global_load_lds
	int wsk_tile;    // wave-split-K kernels: 32 (32x32 tiles, MFMA 32x32x2) or 16 (16x16 tiles, MFMA 16x16x4)
};

__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int r, int c, float acc) {
	float v = p.alpha * acc;
	if (p.bias_row) v += p.bias_row[r];
	if (p.bias_col) v += p.bias_col[c];
	if (p.pre_act) p.pre_act[(size_t)r * p.ld_pre + c] = v;
	if (p.act == BLA_ACT_RELU) v = v < 0.f ? 0.f : v;
	if (p.relu_mask) v = p.relu_mask[(size_t)r * p.ld_mask + c] > 0.f ? v : 0.f * v;
	float* dst = p.C + (size_t)r * p.ldc + c;
	if (p.beta != 0.f) v += p.beta * *dst;
	*dst = v;
}

// How a tile is fetched.  All three are branch-free (loads are always issued, at clamped in-bounds
// addresses, and out-of-range lanes are zeroed by a select) so the K loop stays one basic block.
//   LOAD_FULL   : the whole tile is in range -- no clamps, 16-byte loads
//   LOAD_VEC    : 16-byte loads; needs ld % 4 == 0, 16-byte aligned base and contiguous extents
//                 that are multiples of 4 (a chunk is then either fully in or fully out of range)
//   LOAD_SCALAR : any shape / alignment, four 4-byte loads per chunk
enum { LOAD_FULL = 0, LOAD_VEC = 1, LOAD_SCALAR = 2 };

// ROWS x COLS tile (COLS contiguous in memory) held in registers between the global load and the LDS write.
template <int ROWS, int COLS, int NT>
struct TileRegs {
	static constexpr int CPR = COLS / 4;            // 16-byte chunks per tile row
	static constexpr int N = ROWS * COLS / 4 / NT;  // chunks per thread
	static_assert(ROWS * COLS / 4 % NT == 0, "tile must divide over the workgroup");
	float4 v[N];

	// element (r,c) of the tile is g[(row0+r)*ld + col0+c]; rows >= row_end / cols >= col_end read as 0
	template <int MODE>
	__device__ __forceinline__ void load(const float* __restrict__ g, int ld, int row0, int col0, int row_end, int col_end, int tid) {
#pragma unroll
		for (int i = 0; i < N; i++) {
			int f = tid + i * NT;
			int gr = row0 + f / CPR, gc = col0 + (f % CPR) * 4;
			if (MODE == LOAD_FULL) {
				v[i] = *reinterpret_cast<const float4*>(g + (size_t)gr * ld + gc);
			} else if (MODE == LOAD_VEC) {
				bool ok = gr < row_end && gc < col_end;
				const float* q = g + (size_t)min(gr, row_end - 1) * ld + min(gc, col_end - 4);
				float4 x = *reinterpret_cast<const float4*>(q);
				v[i] = make_float4(ok ? x.x : 0.f, ok ? x.y : 0.f, ok ? x.z : 0.f, ok ? x.w : 0.f);
			} else {
				const float* q = g + (size_t)min(gr, row_end - 1) * ld;
				bool okr = gr < row_end;
				float x0 = q[min(gc, col_end - 1)], x1 = q[min(gc + 1, col_end - 1)];
				float x2 = q[min(gc + 2, col_end - 1)], x3 = q[min(gc + 3, col_end - 1)];
				v[i] = make_float4(okr && gc < col_end ? x0 : 0.f, okr && gc + 1 < col_end ? x1 : 0.f,
				                   okr && gc + 2 < col_end ? x2 : 0.f, okr && gc + 3 < col_end ? x3 : 0.f);
			}
		}
	}

	template <int STRIDE>
	__device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
		for (int i = 0; i < N; i++) {
			int f = tid + i * NT;
			*reinterpret_cast<float4*>(lds + (f / CPR) * STRIDE + (f % CPR) * 4) = v[i];
		}
	}
};

// XCD-aware tile id: hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so give
// XCD x the contiguous chunk [x*q, (x+1)*q) of the (grouped) tile order.  Bijective for any count.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
	int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
	return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, int MODE>
__global__ void __launch_bounds__(WM * WN * 64) gemm_f32_kernel(GemmArgs p) {
	constexpr int NT = WM * WN * 64;
	constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
	constexpr int A_ROWS = AKC ? BM : BK, A_COLS = AKC ? BK : BM, A_STRIDE = AKC ? BK + 4 : BM;
	constexpr int B_ROWS = BKC ? BN : BK, B_COLS = BKC ? BK : BN, B_STRIDE = BKC ? BK + 4 : BN;
	constexpr int A_SZ = A_ROWS * A_STRIDE, B_SZ = B_ROWS * B_STRIDE;
	extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][A_SZ + B_SZ]

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int l31 = lane & 31, h = lane >> 5;
	const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

	// tile coordinates: XCD remap, then groups of 8 tile-rows walked column by column
	int pid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
	constexpr int GROUP_M = 8;
	int per_group = GROUP_M * p.tiles_n;
	int first_m = (pid / per_group) * GROUP_M;
	int gsz = min(p.tiles_m - first_m, GROUP_M);
	int tile_m = first_m + (pid % per_group) % gsz, tile_n = (pid % per_group) / gsz;
	const int m0 = tile_m * BM, n0 = tile_n * BN;

	const int k_begin = blockIdx.z * p.k_per_split;
	const int k_end = min(p.K, k_begin + p.k_per_split);
	const int nkt = (k_end - k_begin + BK - 1) / BK;

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; i++)
#pragma unroll
		for (int j = 0; j < TN; j++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

	TileRegs<A_ROWS, A_COLS, NT> ta;
	TileRegs<B_ROWS, B_COLS, NT> tb;

	auto gload = [&](int kt) {
		int k0 = k_begin + kt * BK;
		if (AKC) ta.template load<MODE>(p.A, p.lda, m0, k0, p.M, k_end, tid);
		else     ta.template load<MODE>(p.A, p.lda, k0, m0, k_end, p.M, tid);
		if (BKC) tb.template load<MODE>(p.B, p.ldb, n0, k0, p.N, k_end, tid);
		else     tb.template load<MODE>(p.B, p.ldb, k0, n0, k_end, p.N, tid);
	};
	auto lstore = [&](int buf) {
		float* base = lds + buf * (A_SZ + B_SZ);
		ta.template store<A_STRIDE>(base, tid);
		tb.template store<B_STRIDE>(base + A_SZ, tid);
	};

	if (nkt > 0) {
		gload(0);
		lstore(0);
	}
	__syncthreads();

	// fragment fetch for k-group kk (8 consecutive k) of the current LDS buffer
	auto frags = [&](const float* As, const float* Bs, int kk, float (&a)[TM][4], float (&b)[TN][4]) {
#pragma unroll
		for (int i = 0; i < TM; i++) {
			if (AKC) {
				float4 x = *reinterpret_cast<const float4*>(As + (wm0 + i * 32 + l31) * A_STRIDE + kk * 8 + 4 * h);
				a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w;
			} else {
#pragma unroll
				for (int j = 0; j < 4; j++) a[i][j] = As[(kk * 8 + 4 * h + j) * A_STRIDE + wm0 + i * 32 + l31];
			}
		}
#pragma unroll
		for (int i = 0; i < TN; i++) {
			if (BKC) {
				float4 x = *reinterpret_cast<const float4*>(Bs + (wn0 + i * 32 + l31) * B_STRIDE + kk * 8 + 4 * h);
				b[i][0] = x.x; b[i][1] = x.y; b[i][2] = x.z; b[i][3] = x.w;
			} else {
#pragma unroll
				for (int j = 0; j < 4; j++) b[i][j] = Bs[(kk * 8 + 4 * h + j) * B_STRIDE + wn0 + i * 32 + l31];
			}
		}
	};

	constexpr int KK = BK / 8;
	for (int kt = 0; kt < nkt; kt++) {
		const int cur = kt & 1;
		const float* As = lds + cur * (A_SZ + B_SZ);
		const float* Bs = As + A_SZ;
		float fa[2][TM][4], fb[2][TN][4];
		frags(As, Bs, 0, fa[0], fb[0]);
		gload(min(kt + 1, nkt - 1));  // in flight during the MFMAs below (the last one is a harmless re-read)
#pragma unroll
		for (int kk = 0; kk < KK; kk++) {
			if (kk + 1 < KK) frags(As, Bs, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);  // fetch the next group under these MFMAs
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int im = 0; im < TM; im++)
#pragma unroll
					for (int in = 0; in < TN; in++)
						acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][im][j], fb[kk & 1][in][j], acc[im][in], 0, 0, 0);
		}
		if (kt + 1 < nkt) lstore(cur ^ 1);
		__syncthreads();
	}

	// C/D map of the 32x32 MFMA: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
#pragma unroll
	for (int im = 0; im < TM; im++)
#pragma unroll
		for (int in = 0; in < TN; in++) {
			int col = n0 + wn0 + in * 32 + l31;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				int row = m0 + wm0 + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
				if (MODE == LOAD_FULL || (row < p.M && col < p.N)) {
					if (p.splits > 1) p.slab[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[im][in][r];
					else epilogue_store(p, row, col, acc[im][in][r]);
				}
			}
		}
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS variant (the fast path): global_load_lds_dwordx4 moves 16 B per lane straight from
// global memory into LDS (no VGPR staging, no ds_write), so the only thing between a K slab's
// arrival and its MFMAs is one s_waitcnt vmcnt(0) + s_barrier per slab, and the DMA of slab t+1
// runs under the MFMAs of slab t.  One wave-instruction writes 1 KiB of LDS contiguously
// (wave-uniform base + lane*16), so the LDS images are lane-linear and the swizzle that keeps
// ds_read_b128 conflict-free is applied to the per-lane SOURCE address (and again on the read):
//   KC operand: image [rows][BK] (no padding), 16-byte chunk c of row r sits at chunk position
//               c ^ ((r >> S) & (BK/4-1)), S = log2(16 / (BK/4)): the 16 rows a b128 lane group
//               touches then cover all 16 slots of the 256-byte bank row;
//   RC operand: image [BK][rows], read with ds_read_b32 of 32 consecutive floats -- no swizzle.
// Needs K % BK == 0, 16-byte aligned rows and contiguous extents that are multiples of 4; rows /
// columns past M / N are fetched from clamped addresses (their products are never stored).
template <int ROWS, int BK>
struct KcImage {  // ROWS x BK floats, K contiguous
	static constexpr int CPR = BK / 4;                       // chunks per row: 4 (BK=16) or 8 (BK=32)
	static constexpr int RPI = 64 / CPR;                     // rows per wave-instruction
	static constexpr int SH = CPR == 4 ? 2 : 1;              // swizzle uses row bits [SH, SH+log2 CPR)
	static constexpr int NINST = ROWS / RPI;                 // wave-instructions per image
	__device__ static __forceinline__ int swz(int r, int c) { return c ^ ((r >> SH) & (CPR - 1)); }
	// float offset of logical (row r, chunk c)
	__device__ static __forceinline__ int off(int r, int c) { return r * BK + swz(r, c) * 4; }
};

// (Forcing 3 workgroups per CU through __launch_bounds__ -- 167 VGPRs, accumulators out of the AGPRs -- was measured
// at 100 vs 141 TFLOP/s on 4096^3: two resident workgroups with AGPR accumulators is the operating point.)
// WK > 1: WK groups of WM x WN waves; group g multiplies k-parts [g KK/WK, (g+1) KK/WK) of every slab for the WHOLE tile, the groups' accumulators
// meet in LDS after the K loop (group order).  Two waves per SIMD on a tile that would otherwise give every SIMD one wave with one accumulator
// block (64x64: a lone wave's waits and barrier skew leave the matrix pipe idle).
template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, int MINW = 1, int NBUF = 2, bool PERSIST = false, int GATHER = 0, bool RCG = false, bool HS = false,
          int WK = 1>
__global__ void __launch_bounds__(WM * WN * WK * 64, MINW) gemm_f32_glds_kernel(GemmArgs p) {
	static_assert(GATHER == 0 || (AKC && BKC == (GATHER == 4) && NBUF == 2 && !PERSIST && BM == 128 && (BN == 128 || (BN == 256 && GATHER == 3 && HS)) && BK == 16 && WM * WN == 4),
	              "gather variants: A K-contiguous; modes 1-3 gather B as a [16][128] image (mode 3 on the half-slab pipeline: [16][256] too), mode 4 gathers A and takes a K-contiguous B");
	// mode 3: one wave-instruction of the B image (1 KiB = 256 floats) covers G3_RPI k-rows of BN / 4 sixteen-byte chunks each
	constexpr int G3_CPR = BN / 4, G3_RPI = 64 / (G3_CPR < 64 ? G3_CPR : 64);
	constexpr int NW = WM * WN * WK;
	constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
	constexpr int A_SZ = BM * BK, B_SZ = BN * BK, KK = BK / 8, KKW = KK / WK;   // KKW: k-parts of a slab this wave multiplies
	static_assert(WK == 1 || (KK % WK == 0 && NBUF == 2 && !PERSIST && GATHER == 0 && !HS && !(TM == 4 && TN == 4)), "waves along K: the plain two-buffer pipeline only");
	typedef KcImage<BM, BK> AI;
	typedef KcImage<BN, BK> BI;
	extern __shared__ __attribute__((aligned(16))) float lds[];  // [NBUF][A_SZ + B_SZ], all LDS in this one array

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, h = lane >> 5;
	const int wk = wave / (WM * WN), wsp = wave % (WM * WN);   // group along K, position in the tile
	const int wm0 = (wsp / WN) * (BM / WM), wn0 = (wsp % WN) * (BN / WN);

	// virtual block id -> tile origin: XCD remap, then groups of 8 tile-rows walked column by column
	auto tile_origin = [&](int vb, int& tm0, int& tn0) {
		int pid = xcd_remap(vb, p.tiles_m * p.tiles_n);
		constexpr int GROUP_M = 8;
		int per_group = GROUP_M * p.tiles_n;
		int first_m = (pid / per_group) * GROUP_M;
		int gsz = min(p.tiles_m - first_m, GROUP_M);
		tm0 = (first_m + (pid % per_group) % gsz) * BM; tn0 = ((pid % per_group) / gsz) * BN;
	};
	int m0, n0;   // origin of the tile being COMPUTED (the persistent variant fetches one tile ahead)
	tile_origin(blockIdx.x, m0, n0);
	const int k_begin = blockIdx.z * p.k_per_split;
	const int k_end = min(p.K, k_begin + p.k_per_split);
	const int nkt = (k_end - k_begin) / BK;

	// Per-lane source pointers of this wave's DMA instructions for slab 0 of a tile (advance by BK per slab).
	constexpr int A_NI = (AKC ? AI::NINST : BK * BM / 256) / NW;   // wave-instructions per wave per slab
	constexpr int B_NI = (GATHER == 1 || GATHER == 2) ? BK * BN / 64 / NW : (BKC ? BI::NINST : BK * BN / 256) / NW;   // checked gather: dword DMA, 64 columns of one k-row per instruction
	static_assert(A_NI >= 1 && B_NI >= 1, "tile too small for the wave count");
	const float* ga[A_NI];
	const float* gb[B_NI];
	const size_t a_step = AKC ? (size_t)BK : (size_t)BK * p.lda, b_step = BKC ? (size_t)BK : (size_t)BK * p.ldb;
	auto open_tile = [&](int tm0, int tn0) {
		if (AKC) {
#pragma unroll
			for (int i = 0; i < A_NI; i++) {
				int inst = wave * A_NI + i;
				int r = inst * AI::RPI + lane / AI::CPR, pos = lane % AI::CPR;
				ga[i] = p.A + (size_t)min(tm0 + r, p.M - 1) * p.lda + k_begin + AI::swz(r, pos) * 4;   // swz is an involution
			}
		} else {  // A stored [k][m]: image [BK][BM]
#pragma unroll
			for (int i = 0; i < A_NI; i++) {
				int f = (wave * A_NI + i) * 64 + lane;   // chunk index in the image
				int kr = f / (BM / 4), c = (f % (BM / 4)) * 4;
				ga[i] = p.A + (size_t)(k_begin + kr) * p.lda + min(tm0 + c, p.M - 4);
			}
		}
		if (GATHER >= 1 && GATHER <= 3) {
			// nothing per tile for B: the gather addresses are rebuilt per slab from the lane / scalar table entries below
		} else if (BKC) {
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				int inst = wave * B_NI + i;
				int r = inst * BI::RPI + lane / BI::CPR, pos = lane % BI::CPR;
				gb[i] = p.B + (size_t)min(tn0 + r, p.N - 1) * p.ldb + k_begin + BI::swz(r, pos) * 4;
			}
		} else {
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				int f = (wave * B_NI + i) * 64 + lane;
				int kr = f / (BN / 4), c = (f % (BN / 4)) * 4;
				gb[i] = p.B + (size_t)(k_begin + kr) * p.ldb + min(tn0 + c, p.N - 4);
			}
		}
	};
	open_tile(m0, n0);
	// gather state: this lane's two columns (n0 + lane, n0 + 64 + lane) -> {element offset, y | x << 16}, and the scalar cursor
	int g_loff[2] = {0, 0}, g_lyx[2] = {0, 0};
	bool g_lval[2] = {false, false};
	int g_k = k_begin;            // k of the next slab to fetch
	int g_img = 0, g_r = 0;       // mode 2: image and pixel of g_k
	int g4_tap[A_NI], g4_chunk[A_NI];   // mode 4: padded offset of this lane's tap row, and its (swizzled) pixel chunk, per A instruction
	if (GATHER == 4) {
#pragma unroll
		for (int i = 0; i < A_NI; i++) {
			int inst = wave * A_NI + i;
			int r = inst * AI::RPI + lane / AI::CPR, pos = lane % AI::CPR;
			g4_tap[i] = p.g_ntab[min(m0 + r, p.M - 1)].x;
			g4_chunk[i] = AI::swz(r, pos) * 4;
		}
		g_img = k_begin / p.g_HWo; g_r = k_begin - g_img * p.g_HWo;
#pragma unroll
		for (int i = 0; i < B_NI; i++) gb[i] += (ptrdiff_t)g_img * p.N * p.g_HWo + (g_r - k_begin);   // B = del_y [image][N][HWo], ldb = HWo
	}
	int g3_base = 0;              // mode 3: padded-image offset of this lane's four columns
	if (GATHER == 3) {
		int n = min(n0 + (lane % G3_CPR) * 4, p.N - 4);
		int b = n / p.g_HWo, r = n - b * p.g_HWo;
		g3_base = p.g_ntab[r].x + b * p.g_img_stride;
	}
	if (GATHER == 1 || GATHER == 2) {
#pragma unroll
		for (int hf = 0; hf < 2; hf++) {
			int n = n0 + hf * 64 + lane;
			g_lval[hf] = n < p.N;
			n = min(n, p.N - 1);
			if (GATHER == 1) {
				int b = n / p.g_HWo, r = n - b * p.g_HWo;
				int2 t = p.g_ntab[r];
				g_loff[hf] = t.x + b * p.g_img_stride; g_lyx[hf] = t.y;
			} else {
				int2 t = p.g_ntab[n];
				g_loff[hf] = t.x; g_lyx[hf] = t.y;
			}
		}
		if (GATHER == 2) {
			g_img = k_begin / p.g_HWo; g_r = k_begin - g_img * p.g_HWo;
#pragma unroll
			for (int i = 0; i < A_NI; i++) ga[i] += (ptrdiff_t)g_img * p.M * p.g_HWo + (g_r - k_begin);   // A = del_y [image][M][HWo]: row stride lda = HWo
		}
	}

	typedef __attribute__((address_space(3))) void* lds_ptr_t;
	typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
	// Which operands go through buffer_load ... lds (SGPR descriptor + 32-bit lane offset + scalar slab offset) instead of
	// global_load_lds (64-bit pointer per lane): always the K-contiguous ones (NT 4096^3: 142 vs 115 TFLOP/s); the row-contiguous
	// ones too unless the host picks the RCG instantiation -- the buffer form holds 139-147 TFLOP/s across 5120^3 / 6144^3 / 8192^3
	// where the global form drops to 128, but on power-of-two pitches up to 16 KiB the global form is 1-4 % ahead (4096^3: 142.7 vs 141.0).
	// Compile-time: choosing between the two forms at run time inside the loop costs 4 %.
	constexpr bool BUF_OK = GATHER == 0 && !PERSIST;   // (the persistent variant re-bases its pointers per tile)
	constexpr bool A_BUF = (BUF_OK && (AKC || !RCG)) || GATHER == 3, B_BUF = (BUF_OK && (BKC || !RCG)) || GATHER == 4;   // padded-copy conv modes: their dense operand too
#if defined(__HIP_DEVICE_COMPILE__)
	// raw descriptors, no bounds (rows / columns past the matrix are fetched from clamped offsets); lane offsets in bytes
	__amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7fffffff, 0x00020000);
	__amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, 0x7fffffff, 0x00020000);
	int voff_a[A_NI], voff_b[B_NI], soff_a = 0, soff_b = 0;
#pragma unroll
	for (int i = 0; i < A_NI; i++) voff_a[i] = A_BUF ? (int)((ga[i] - p.A) * 4) : 0;
#pragma unroll
	for (int i = 0; i < B_NI; i++) voff_b[i] = B_BUF ? (int)((gb[i] - p.B) * 4) : 0;
#endif
	auto dma = [&](int buf) {
		float* base = lds + buf * (A_SZ + B_SZ);
#if defined(__HIP_DEVICE_COMPILE__)
		if (GATHER == 4) {   // dense operand (del_y) through buffer_load ... lds
			const float* ib = p.g_img + g_img * p.g_img_stride;   // gathered operand: global_load_lds (its 16-byte chunks are mostly unaligned;
#pragma unroll                                          // the buffer form measured slower for them: 254 vs 230 us forward at 128->128 @32x32 x64)
			for (int i = 0; i < A_NI; i++) {
				const float* src = ib + (g4_tap[i] + p.g_ktab[g_r + g4_chunk[i]].x);
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, 0, 0);
			}
#pragma unroll
			for (int i = 0; i < B_NI; i++)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(base + A_SZ + (wave * B_NI + i) * 256), 16, voff_b[i], soff_b, 0, 0);
			soff_b += BK * 4;
			g_r += BK;
			if (g_r >= p.g_HWo) {   // next image: del_y [image][N][HWo]
				g_r = 0; g_img++;
				soff_b += (p.N - 1) * p.g_HWo * 4;
			}
			return;
		}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
		if (GATHER == 0) {   // dense operands
#pragma unroll
			for (int i = 0; i < A_NI; i++) {
				if (A_BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, voff_a[i], soff_a, 0, 0);
				else { __builtin_amdgcn_global_load_lds((gbl_ptr_t)ga[i], (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, 0, 0); ga[i] += a_step; }
			}
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				if (B_BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(base + A_SZ + (wave * B_NI + i) * 256), 16, voff_b[i], soff_b, 0, 0);
				else { __builtin_amdgcn_global_load_lds((gbl_ptr_t)gb[i], (lds_ptr_t)(base + A_SZ + (wave * B_NI + i) * 256), 16, 0, 0); gb[i] += b_step; }
			}
			if (A_BUF) soff_a += (int)(a_step * 4);
			if (B_BUF) soff_b += (int)(b_step * 4);
			return;
		}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
		if (A_BUF) {   // mode 3: the dense kernels operand
#pragma unroll
			for (int i = 0; i < A_NI; i++)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, voff_a[i], soff_a, 0, 0);
			soff_a += (int)(a_step * 4);
		} else
#endif
		{
#pragma unroll
			for (int i = 0; i < A_NI; i++) {
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)ga[i], (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, 0, 0);
				ga[i] += a_step;
			}
		}
		if (GATHER == 3) {
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				const int idx = wave * B_NI + i, kr = idx * 2 + (lane >> 5);
				const float* src = p.g_img + (g3_base + p.g_ktab[g_k + kr].x);
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + A_SZ + idx * 256), 16, 0, 0);
			}
			g_k += BK;
		} else if (GATHER) {
			const int simg = GATHER == 2 ? g_img * p.g_img_stride : 0;
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				const int idx = wave * B_NI + i, krow = idx >> 1, hf = idx & 1;            // wave-uniform
				const int2 ts = p.g_ktab[GATHER == 1 ? g_k + krow : g_r + krow];            // scalar load
				const int yy = (short)(g_lyx[hf] & 0xffff) + (short)(ts.y & 0xffff), xx = (g_lyx[hf] >> 16) + (ts.y >> 16);
				const bool ok = g_lval[hf] && (unsigned)yy < (unsigned)p.g_H && (unsigned)xx < (unsigned)p.g_W;
				const float* src = ok ? p.g_img + (g_loff[hf] + ts.x + simg) : p.g_zero;
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + A_SZ + krow * BN + hf * 64), 4, 0, 0);
			}
			g_k += BK;
			if (GATHER == 2) {
				g_r += BK;
				if (g_r >= p.g_HWo) {   // next image (HWo % 16 == 0: a slab never straddles two)
					g_r = 0; g_img++;
#pragma unroll
					for (int i = 0; i < A_NI; i++) ga[i] += (size_t)(p.M - 1) * p.g_HWo;
				}
			}
		} else {
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)gb[i], (lds_ptr_t)(base + A_SZ + (wave * B_NI + i) * 256), 16, 0, 0);
				gb[i] += b_step;
			}
		}
	};

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; i++)
#pragma unroll
		for (int j = 0; j < TN; j++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

	auto frags = [&](const float* As, const float* Bs, int kk, float (&a)[TM][4], float (&b)[TN][4]) {
#pragma unroll
		for (int i = 0; i < TM; i++) {
			if (AKC) {
				float4 x = *reinterpret_cast<const float4*>(As + AI::off(wm0 + i * 32 + l31, kk * 2 + h));
				a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w;
			} else {
#pragma unroll
				for (int j = 0; j < 4; j++) a[i][j] = As[(kk * 8 + 4 * h + j) * BM + wm0 + i * 32 + l31];
			}
		}
#pragma unroll
		for (int i = 0; i < TN; i++) {
			if (BKC) {
				float4 x = *reinterpret_cast<const float4*>(Bs + BI::off(wn0 + i * 32 + l31, kk * 2 + h));
				b[i][0] = x.x; b[i][1] = x.y; b[i][2] = x.z; b[i][3] = x.w;
			} else {
#pragma unroll
				for (int j = 0; j < 4; j++) b[i][j] = Bs[(kk * 8 + 4 * h + j) * BN + wn0 + i * 32 + l31];
			}
		}
	};

	// Software pipeline (two LDS buffers, two fragment register sets P/Q):
	//   step(kt): [first MFMA group of slab kt from P]
	//             s_waitcnt vmcnt(0); s_barrier       -> slab kt+1 has landed for every wave, and every wave has
	//                                                     finished reading slab kt (its reads fed MFMAs already issued)
	//             ds_read slab kt+1 -> Q               (latency hidden under the remaining MFMAs of slab kt)
	//             DMA slab kt+2 -> the buffer slab kt lived in   (issued AFTER the reads: hipcc puts an
	//                                                     s_waitcnt vmcnt(0) before any ds_read that follows an LDS-DMA)
	//             [remaining MFMA groups of slab kt from P]
	// so a wave's MFMA stream only pauses for the barrier skew, never for LDS or HBM latency.
	auto mfma_group = [&](float (&a)[KKW][TM][4], float (&b)[KKW][TN][4], int kk, int j) {
#pragma unroll
		for (int im = 0; im < TM; im++)
#pragma unroll
			for (int in = 0; in < TN; in++)
				acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][im][j], b[kk][in][j], acc[im][in], 0, 0, 0);
	};
	auto rest = [&](float (&pa)[KKW][TM][4], float (&pb)[KKW][TN][4]) {
#pragma unroll
		for (int kk = 0; kk < KKW; kk++)
#pragma unroll
			for (int j = 0; j < 4; j++)
				if (kk != 0 || j != 0) mfma_group(pa, pb, kk, j);
	};
	// Measured and not kept (4096^3 / 8192^3 stayed at 143 / 145 TFLOP/s, 2048^3 lost 10 %): (a) a one-off s_sleep of half a
	// slab period for the wave in the odd hardware slot, to de-phase co-resident workgroups; (b) sched_group_barrier
	// interleaving of one ds_read / one DMA per MFMA gap instead of the clump after the barrier; (c) s_setprio 3 around the
	// read / DMA clump so that the wave issuing memory operations wins arbitration over its co-resident wave's MFMAs; (d) one-wave
	// workgroups on 64x64 tiles (no cross-wave barrier at all, every wave streams its own operands): 74 TFLOP/s.
	// A launch that issues nothing but independent 32x32x2 MFMAs reaches 155.2 TFLOP/s (tools/mfma_peak.py): that, not 157.3, is the ceiling.
	// Where the rest goes (this kernel rebuilt with pieces compiled out, 4096^3 / 8192^3): as is 143 / 145; without the DMA 146 / 151;
	// without DMA and barrier 149 / 152; without LDS reads as well 150 / 153 -- the exposed cost is the DMA, not the barrier or the reads.
	// (e) Fetching TWO slabs ahead (three buffers, s_waitcnt vmcnt(4) for the older batch only, fragment reads written as asm so that
	// hipcc does not put vmcnt(0) in front of them): correct, 113 / 117 TFLOP/s -- also with vmcnt(0) and with compiler-visible reads,
	// i.e. it is the second batch in flight per wave that hurts (64 outstanding 1-KiB DMA instructions per CU instead of 32), not the wait.  PMC: MFMA pipe 93 % busy at
	// 2.38 GHz with two workgroups per CU, 87 % with one -- the residue tracks the LDS-DMA issue cost (4 per 32 MFMAs per wave).
	// One pipeline step on slab kt (fragments in P); slab kt+1 must exist.  No branch touches the fragment
	// registers (a conditional around the reads would make hipcc copy them at the join and wait for them).
	// One-workgroup-per-CU tile (256x256: a wave holds 16 accumulator blocks = 256 AGPRs, and there is no second workgroup to fill
	// the matrix pipe while this one reads and fetches).  Two things change against the pipeline below:
	//  * fragments are held per k-HALF of a slab, not per slab: P = k-half 0, Q = k-half 1 (32 registers each instead of 2 x 64);
	//    Q of slab t is read under the MFMAs of P, P of slab t+1 under the MFMAs of Q -- so the wait + barrier for slab t+1 sits in
	//    the MIDDLE of slab t, and the roles of P and Q never swap (no unrolling by two);
	//  * every LDS read and DMA instruction is dealt out BETWEEN MFMAs (one read unit per 8 or 2 MFMAs, one DMA per 2) instead of
	//    standing in a clump behind the barrier.  All LDS reads of a phase come before its first DMA: hipcc waits for vmcnt(0) in
	//    front of any LDS read that follows an LDS-DMA.
	// the half-slab interleaved pipeline: always for 4x4 blocks per wave (256x256, 128x512), on request (HS) for 2x2 (128x128)
	// (the padded-copy convolution modes 3 / 4 run on it too: their gather is one more address per DMA instruction, dealt out between MFMAs like the rest)
	constexpr bool HALFSLAB = ((TM == 4 && TN == 4) || (HS && TM >= 2 && TM <= 4 && TN >= 2 && TN <= 4)) && (KK == 2 || KK == 4) &&
	                          (GATHER == 0 || ((GATHER == 3 || GATHER == 4) && HS)) && !PERSIST && NBUF == 2;
	constexpr int NDMA = A_NI + B_NI;   // DMA instructions per wave per slab (8 at BK = 16, 16 at BK = 32)
	size_t g_adv_a = 0, g_adv_b = 0;   // global-form operands of the half-slab pipeline: scalar advance added to the per-lane pointers
#if defined(__HIP_DEVICE_COMPILE__)
	// Gather tables of the half-slab pipeline: the entries a slab's DMA instructions need are wave-uniform (tap rows of mode 3, the four
	// pixel chunks of mode 4), so they are SCALAR loads, issued when the cursor moves -- a whole slab before the DMA that uses them.  (A per-lane
	// table load in front of each DMA waits on vmcnt, i.e. for every LDS-DMA issued before it: 256->256 @16x16 ran 11 % slower that way.)
	int hs_tap[B_NI][2], hs_pix[4] = {0, 0, 0, 0};
	auto hs_prefetch = [&]() {
		if (GATHER == 3) {
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				const int row = __builtin_amdgcn_readfirstlane(g_k + (wave * B_NI + i) * G3_RPI);
				hs_tap[i][0] = p.g_ktab[row].x; hs_tap[i][1] = G3_RPI == 2 ? p.g_ktab[row + 1].x : 0;
			}
		}
		if (GATHER == 4) {
			const int r = __builtin_amdgcn_readfirstlane(g_r);
#pragma unroll
			for (int c = 0; c < 4; c++) hs_pix[c] = p.g_ktab[r + 4 * c].x;
		}
	};
	if (HALFSLAB && (GATHER == 3 || GATHER == 4)) hs_prefetch();
	auto dma_one = [&](int buf, int d) {   // d-th DMA instruction of a slab; the offsets / gather cursors advance in dma_advance()
		float* base = lds + buf * (A_SZ + B_SZ);
		if (d < A_NI) {
			const int i = d;
			if (GATHER == 4) {   // gathered operand: 16-byte chunk of four pixels of this lane's tap row (padded image copy)
				const int c = g4_chunk[i] >> 2;     // which of the slab's four pixel chunks this lane fetches: its offset was loaded a slab ahead (hs_pix)
				const int pix = c == 0 ? hs_pix[0] : c == 1 ? hs_pix[1] : c == 2 ? hs_pix[2] : hs_pix[3];
				const float* src = p.g_img + (size_t)g_img * p.g_img_stride + (g4_tap[i] + pix);
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, 0, 0);
			} else if (A_BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, voff_a[i], soff_a, 0, 0);
			else __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ga[i] + g_adv_a), (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, 0, 0);
		} else {
			const int i = d - A_NI;
			if (GATHER == 3) {   // gathered operand: four consecutive output pixels = four consecutive floats of the padded copy, tap from the scalar table
				const int idx = wave * B_NI + i;   // the instruction covers k-rows 2 idx (lanes 0-31) and 2 idx + 1 (BN = 256: the one row idx): their tap offsets were loaded a slab ahead (hs_tap)
				const float* src = p.g_img + (g3_base + ((G3_RPI == 2 && (lane >> 5)) ? hs_tap[i][1] : hs_tap[i][0]));
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + A_SZ + idx * 256), 16, 0, 0);
			} else if (B_BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(base + A_SZ + (wave * B_NI + i) * 256), 16, voff_b[i], soff_b, 0, 0);
			else __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gb[i] + g_adv_b), (lds_ptr_t)(base + A_SZ + (wave * B_NI + i) * 256), 16, 0, 0);
		}
	};
	auto dma_advance = [&](bool really) {   // uniform select, no branch: past the last slab the cursor stays on it (harmless re-fetch)
		const int sa = really ? (int)(a_step * 4) : 0, sb = really ? (int)(b_step * 4) : 0;
		if (GATHER == 3) { soff_a += sa; g_k += really ? BK : 0; hs_prefetch(); return; }
		if (GATHER == 4) {   // B = del_y [image][N][HWo]: the pixel cursor wraps into the next image (HWo % 16 == 0: a slab never straddles two)
			const int r1 = g_r + (really ? BK : 0);
			const bool wrap = r1 >= p.g_HWo;
			soff_b += (really ? BK * 4 : 0) + (wrap ? (p.N - 1) * p.g_HWo * 4 : 0);
			g_r = wrap ? 0 : r1; g_img += wrap ? 1 : 0;
			hs_prefetch();
			return;
		}
		if (A_BUF) soff_a += sa; else g_adv_a += really ? a_step : 0;
		if (B_BUF) soff_b += sb; else g_adv_b += really ? b_step : 0;
	};
#else
	auto dma_one = [&](int, int) {};
	auto dma_advance = [&](bool) {};
#endif
	auto step = [&](int kt, bool do_dma, float (&pa)[KKW][TM][4], float (&pb)[KKW][TN][4], float (&qa)[KKW][TM][4], float (&qb)[KKW][TN][4]) {
		// sched_barrier(0) pins the order: hipcc otherwise floats the MFMAs (which touch no memory) across the
		// barrier and the waits, e.g. hoisting the NEXT step's vmcnt(0)+barrier above this step's MFMAs.
		mfma_group(pa, pb, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		const float* As = lds + ((kt + 1) & 1) * (A_SZ + B_SZ);
#pragma unroll
		for (int kk = 0; kk < KKW; kk++) frags(As, As + A_SZ, wk * KKW + kk, qa[kk], qb[kk]);
		if (do_dma) dma(kt & 1);
		__builtin_amdgcn_sched_barrier(0);
		rest(pa, pb);
		__builtin_amdgcn_sched_barrier(0);
	};

	auto store_tile = [&]() {
		if (GATHER == 4) {   // transposed: tile element (row = tap, col = f) -> out[f][tap]; a lane's registers q&3 are 4 consecutive taps
			float* dst = p.splits > 1 ? p.slab + (size_t)blockIdx.z * p.M * p.N : p.C;
			const int ld = p.splits > 1 ? p.M : p.ldc;
#pragma unroll
			for (int in = 0; in < TN; in++) {
				int col = n0 + wn0 + in * 32 + l31;
				if (col >= p.N) continue;
#pragma unroll
				for (int im = 0; im < TM; im++)
#pragma unroll
					for (int qq = 0; qq < 4; qq++) {
						int row = m0 + wm0 + im * 32 + 8 * qq + 4 * h;
						if (row + 3 < p.M)
							*reinterpret_cast<float4*>(dst + (size_t)col * ld + row) =
								make_float4(acc[im][in][4 * qq], acc[im][in][4 * qq + 1], acc[im][in][4 * qq + 2], acc[im][in][4 * qq + 3]);
					}
			}
			return;
		}
		if constexpr (GATHER == 3 && HALFSLAB) {   // whole tiles; a lane owns TN consecutive columns (the row-contiguous operand's interleaved blocks)
			const int col = n0 + wn0 + TN * l31;              // TN consecutive pixels of one image (HWo % 4 == 0)
			const int b = col / p.g_HWo, rr = col - b * p.g_HWo;
			const size_t img_off = (size_t)b * p.M * p.g_HWo + rr;
			float* cbase = (p.splits > 1 ? p.slab + (size_t)blockIdx.z * p.M * p.N : p.C) + img_off;   // slabs are C-shaped
			const float* bias = p.g_bias ? p.g_bias + (size_t)b * p.g_bias_stride : nullptr;
#pragma unroll
			for (int im = 0; im < TM; im++)
#pragma unroll
				for (int r = 0; r < 16; r++) {
					const int row = m0 + wm0 + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
					const size_t o = (size_t)row * p.g_HWo;
					const float bv = bias ? bias[row] : 0.f;
					if (TN == 4) {
						const float4 v = make_float4(acc[im][0][r] + bv, acc[im][1][r] + bv, acc[im][TN > 2 ? 2 : 0][r] + bv, acc[im][TN > 3 ? 3 : 0][r] + bv);
						*reinterpret_cast<float4*>(cbase + o) = v;
						if (p.g_out2) {
							const float4 a4 = *reinterpret_cast<const float4*>(p.g_add + img_off + o);
							*reinterpret_cast<float4*>(p.g_out2 + img_off + o) = make_float4(v.x + a4.x, v.y + a4.y, v.z + a4.z, v.w + a4.w);
						}
					} else {
						const float2 v = make_float2(acc[im][0][r] + bv, acc[im][1][r] + bv);
						*reinterpret_cast<float2*>(cbase + o) = v;
						if (p.g_out2) {
							const float2 a2 = *reinterpret_cast<const float2*>(p.g_add + img_off + o);
							*reinterpret_cast<float2*>(p.g_out2 + img_off + o) = make_float2(v.x + a2.x, v.y + a2.y);
						}
					}
				}
			return;
		}
		if (GATHER == 1 || GATHER == 3) {   // C is [image][M][HWo]: column n = (image, pixel)
#pragma unroll
			for (int in = 0; in < TN; in++) {
				int col = n0 + wn0 + in * 32 + l31;
				if (col >= p.N) continue;
				int b = col / p.g_HWo, r = col - b * p.g_HWo;
				float* cbase = p.C + (size_t)b * p.M * p.g_HWo + r;
#pragma unroll
				for (int im = 0; im < TM; im++)
#pragma unroll
					for (int q = 0; q < 16; q++) {
						int row = m0 + wm0 + im * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
						if (row < p.M) cbase[(size_t)row * p.g_HWo] = acc[im][in][q];
					}
			}
			return;
		}
		if (HALFSLAB || (m0 + BM <= p.M && n0 + BN <= p.N && p.splits == 1)) {   // (the 256x256 variant is only launched on whole tiles)
			// interior tile: no per-element bounds branch (with 256 accumulators per lane hipcc otherwise parks them all in scratch
			// and reloads them one by one behind each branch); one block at a time
			if constexpr (HALFSLAB) {
				// block (im, in), register r of lane (l31, h): row index inside the block i = (r&3) + 8*(r>>2) + 4h, column index l31.
				// K-contiguous operand: block b owns rows / columns w0 + 32 b + index; row-contiguous: w0 + T*index + b (T blocks interleaved).
#pragma unroll
				for (int im = 0; im < TM; im++)
#pragma unroll
					for (int r = 0; r < 16; r++) {
						const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
						const int row = m0 + wm0 + ((AKC || TM == 3) ? im * 32 + i : TM * i + im);   // (three blocks: not interleaved, see the fragment reads)
						float* cp = p.C + (size_t)row * p.ldc + n0 + wn0;
						if (BKC || TN == 3) {
#pragma unroll
							for (int in = 0; in < TN; in++) cp[in * 32 + l31] = p.alpha * acc[im][in][r];
						} else if (TN == 4) {   // four consecutive columns per lane
							*reinterpret_cast<float4*>(cp + 4 * l31) =
								make_float4(p.alpha * acc[im][0][r], p.alpha * acc[im][1][r], p.alpha * acc[im][TN > 2 ? 2 : 0][r], p.alpha * acc[im][TN > 3 ? 3 : 0][r]);
						} else {
							*reinterpret_cast<float2*>(cp + 2 * l31) = make_float2(p.alpha * acc[im][0][r], p.alpha * acc[im][1][r]);
						}
					}
				return;
			}
			if (!p.bias_row && !p.bias_col && !p.pre_act && p.act == BLA_ACT_NONE && !p.relu_mask && p.beta == 0.f) {   // plain C = alpha * acc
				// (the 256x256 variant is only launched with a plain epilogue: any other path in this function makes hipcc park its 256
				// accumulators in scratch at the loop exit)
#pragma unroll
				for (int im = 0; im < TM; im++)
#pragma unroll
					for (int in = 0; in < TN; in++) {
						float* cp = p.C + (size_t)(m0 + wm0 + im * 32 + 4 * h) * p.ldc + n0 + wn0 + in * 32 + l31;
#pragma unroll
						for (int r = 0; r < 16; r++) cp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = p.alpha * acc[im][in][r];
						__builtin_amdgcn_sched_barrier(0);
					}
				return;
			}
#pragma unroll
			for (int im = 0; im < TM; im++)
#pragma unroll
				for (int in = 0; in < TN; in++) {
					const int col = n0 + wn0 + in * 32 + l31;
#pragma unroll
					for (int r = 0; r < 16; r++) epilogue_store(p, m0 + wm0 + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col, acc[im][in][r]);
					__builtin_amdgcn_sched_barrier(0);
				}
			return;
		}
#pragma unroll
		for (int im = 0; im < TM; im++)
#pragma unroll
			for (int in = 0; in < TN; in++) {
				int col = n0 + wn0 + in * 32 + l31;
#pragma unroll
				for (int r = 0; r < 16; r++) {
					int row = m0 + wm0 + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
					if (row < p.M && col < p.N) {
						if (p.splits > 1) p.slab[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[im][in][r];
						else epilogue_store(p, row, col, acc[im][in][r]);
					}
				}
			}
	};

	if constexpr (HALFSLAB) {
		// Fragment reads are written as asm: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it can see after an LDS-DMA, and
		// in a loop the reads of a slab's first phase always follow the DMA instructions of the previous slab's second phase -- every
		// slab would start by waiting for a fetch issued half a microsecond earlier.  The asm reads are invisible to that rule; their
		// results are tied to an explicit s_waitcnt lgkmcnt(0) ("land") in front of their first MFMA.
		typedef float v4f __attribute__((ext_vector_type(4)));
		typedef float vra __attribute__((ext_vector_type(TM)));   // row-contiguous A: one element per block
		typedef float vrb __attribute__((ext_vector_type(TN)));
		typedef __attribute__((address_space(3))) float* lds_f;
		const unsigned lds0 = (unsigned)(size_t)(lds_f)lds;
		constexpr unsigned BUF_BYTES = (A_SZ + B_SZ) * 4;
		// K-contiguous operand: block x of k-part kk = 16 bytes at row w0 + x*32 + l31, chunk kk*2 + h (x only adds a multiple of 32 rows).
		// Row-contiguous operand: its T blocks do not own 32 consecutive rows each but the rows w0 + T*lane + block -- then ONE read of
		// 4T bytes at (k, w0 + T*l31) delivers a lane's element of all T blocks for that k (instead of T dwords), and in the output a lane
		// holds T consecutive columns (wide stores).  Which rows / columns a block owns is a free choice: only the epilogue's index map changes.
		unsigned a_ad[KK], b_ad[KK];
#pragma unroll
		for (int kk = 0; kk < KK; kk++) {
			a_ad[kk] = lds0 + (AKC ? AI::off(wm0 + l31, kk * 2 + h) : (kk * 8 + 4 * h) * BM + wm0 + (TM == 3 ? 1 : TM) * l31) * 4;
			b_ad[kk] = lds0 + (A_SZ + (BKC ? BI::off(wn0 + l31, kk * 2 + h) : (kk * 8 + 4 * h) * BN + wn0 + (TN == 3 ? 1 : TN) * l31)) * 4;
		}
		struct Frag {
			v4f ka[TM], kb[TN];   // K-contiguous: [block], elements = k-offset j
			vra ra[4]; vrb rb[4]; // row-contiguous: [k-offset j], elements = block
			float sa[4][TM], sb[4][TN];   // row-contiguous with three blocks (192-wide tiles): 12-byte reads would be unaligned, so the blocks keep
			                              // 32 consecutive rows each and a lane's three elements come as three dword reads
		};
		Frag P, Q;
		constexpr int UA = AKC ? TM : 4, UB = BKC ? TN : 4, NU = UA + UB;   // fragment-read units per k-part
		constexpr int NM = 4 * TM * TN;                                       // MFMAs per k-part
		auto opa = [&](const Frag& f, int im, int j) -> float { return AKC ? f.ka[im][j] : TM == 3 ? f.sa[j][im] : f.ra[j][im]; };
		auto opb = [&](const Frag& f, int in, int j) -> float { return BKC ? f.kb[in][j] : TN == 3 ? f.sb[j][in] : f.rb[j][in]; };
		auto mf1 = [&](const Frag& f, int idx) {   // idx-th MFMA of a k-part: j-major, then im, in
			const int j = idx / (TM * TN), im = (idx % (TM * TN)) / TN, in = idx % TN;
			acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa(f, im, j), opb(f, in, j), acc[im][in], 0, 0, 0);
		};
		auto read_unit = [&](unsigned buf, int kk, int u, Frag& f) {   // units 0..UA-1: A side, then B side
			if (u < UA) {
				const int x = u;
				const unsigned ad = buf + a_ad[kk];
				if constexpr (AKC) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.ka[x]) : "v"(ad), "n"(x * 32 * BK * 4));
				else if constexpr (TM == 3) {
#pragma unroll
					for (int b = 0; b < 3; b++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.sa[x][b]) : "v"(ad), "n"((x * BM + b * 32) * 4));
				} else if constexpr (TM == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.ra[x]) : "v"(ad), "n"(x * BM * 4));
				else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.ra[x]) : "v"(ad), "n"(x * BM * 4));
			} else {
				const int x = u - UA;
				const unsigned ad = buf + b_ad[kk];
				if constexpr (BKC) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.kb[x]) : "v"(ad), "n"(x * 32 * BK * 4));
				else if constexpr (TN == 3) {
#pragma unroll
					for (int b = 0; b < 3; b++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.sb[x][b]) : "v"(ad), "n"((x * BN + b * 32) * 4));
				} else if constexpr (TN == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.rb[x]) : "v"(ad), "n"(x * BN * 4));
				else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.rb[x]) : "v"(ad), "n"(x * BN * 4));
			}
		};
		auto land = [&](Frag& f) {   // every read into f has returned; the empty statements make each register's later uses depend on the wait
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int x = 0; x < UA; x++) {
				if constexpr (AKC) asm volatile("" : "+v"(f.ka[x]));
				else if constexpr (TM == 3) { asm volatile("" : "+v"(f.sa[x][0])); asm volatile("" : "+v"(f.sa[x][1])); asm volatile("" : "+v"(f.sa[x][2])); }
				else asm volatile("" : "+v"(f.ra[x]));
			}
#pragma unroll
			for (int x = 0; x < UB; x++) {
				if constexpr (BKC) asm volatile("" : "+v"(f.kb[x]));
				else if constexpr (TN == 3) { asm volatile("" : "+v"(f.sb[x][0])); asm volatile("" : "+v"(f.sb[x][1])); asm volatile("" : "+v"(f.sb[x][2])); }
				else asm volatile("" : "+v"(f.rb[x]));
			}
		};
		// slab t in buffer t&1.  One uniform body for every slab: past the end the fetch cursor stays on the last slab (re-fetched
		// into a buffer nobody reads again) and the "next" fragments are stale LDS that is never multiplied -- so there is no tail
		// code, no branch in the loop, and the 256 accumulators never leave their registers.
		int fetched = 0;                       // slabs the cursor has been advanced past
		auto fetch_slab = [&](int buf) {       // prologue form (clumped)
#pragma unroll
			for (int d = 0; d < NDMA; d++) dma_one(buf, d);
			const bool adv = fetched + 1 < nkt;
			dma_advance(adv); fetched += adv ? 1 : 0;
		};
		auto slab = [&](int t) {
			const unsigned cur = (t & 1) * BUF_BYTES, nxt = ((t + 1) & 1) * BUF_BYTES;
			// phases 0 .. KK-2: k-part q from one set while k-part q+1 of this slab is read into the other, the read units spread evenly
#pragma unroll
			for (int q = 0; q + 1 < KK; q++) {
				Frag& use = (q & 1) ? Q : P;
				Frag& fill = (q & 1) ? P : Q;
				land(use);
#pragma unroll
				for (int u = 0; u < NU; u++) {
					read_unit(cur, q + 1, u, fill);
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int m = u * NM / NU; m < (u + 1) * NM / NU; m++) mf1(use, m);
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			// last phase (k-part KK-1 from Q): slab t+1 has landed for everyone, and everyone is done reading slab t
			land(Q);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			constexpr int PER = 2 * (NU + NDMA) <= NM ? 2 : 1;   // MFMAs after each read unit / DMA instruction
			static_assert(PER * (NU + NDMA) <= NM, "too many memory instructions for the MFMAs of one k-part");
#pragma unroll
			for (int u = 0; u < NU; u++) {    // k-part 0 of slab t+1 -> P (all LDS reads before the first DMA)
				read_unit(nxt, 0, u, P);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int m = 0; m < PER; m++) mf1(Q, PER * u + m);
				__builtin_amdgcn_sched_barrier(0);
			}
#pragma unroll
			for (int d = 0; d < NDMA; d++) {  // slab t+2 -> this slab's buffer
				dma_one(t & 1, d);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int m = 0; m < PER; m++) mf1(Q, PER * (NU + d) + m);
				__builtin_amdgcn_sched_barrier(0);
			}
			{ const bool adv = fetched + 1 < nkt; dma_advance(adv); fetched += adv ? 1 : 0; }
#pragma unroll
			for (int m = PER * (NU + NDMA); m < NM; m++) mf1(Q, m);
			__builtin_amdgcn_sched_barrier(0);
		};
		if (nkt > 0) {
			fetch_slab(0);
			fetch_slab(1);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
#pragma unroll
			for (int u = 0; u < NU; u++) read_unit(0, 0, u, P);
			for (int t = 0; t < nkt; t++) slab(t);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetches of the last slab
		}
		store_tile();
		return;
	}

	if constexpr (PERSIST) {
		// Persistent variant: gridDim.x workgroups (two per CU) walk the tile list with stride gridDim.x and treat the K slabs
		// of their consecutive tiles as ONE stream -- the DMA cursor runs two slabs ahead of the MFMAs straight across tile
		// boundaries, so a new tile starts with its first slabs already in LDS, and the C stores of the finished tile drain
		// under the next tile's MFMAs instead of in front of a fresh workgroup's cold prologue (a 4096^2 output cost a fixed
		// 29 us that way).  Needs an even slab count per tile (fragment sets and LDS buffers alternate by slab parity).
		static_assert(NBUF == 2, "persistent pipeline uses the two-buffer scheme");
		const int total = p.tiles_m * p.tiles_n;
		// DMA cursor: ga/gb walk the slabs of the tile being fetched; gan/gbn hold slab 0 of the tile after it (computed once per
		// tile, outside the steps).  The switch is a select, not a branch: a branch between the fragment reads and their MFMAs
		// makes hipcc wait for the reads and copy them at the join.
		const float* gan[A_NI];
		const float* gbn[B_NI];
		int dma_left = nkt;
		auto dma_next = [&](int buf) {
			const bool sw = dma_left == 0;
#pragma unroll
			for (int i = 0; i < A_NI; i++) ga[i] = sw ? gan[i] : ga[i];
#pragma unroll
			for (int i = 0; i < B_NI; i++) gb[i] = sw ? gbn[i] : gb[i];
			dma_left = (sw ? nkt : dma_left) - 1;
			dma(buf);
		};
		auto plan_next = [&](int vb_next) {   // slab-0 pointers of the tile after the one being computed (past the end: this tile
			int tm0, tn0;                      // again -- a harmless re-fetch into a buffer nobody reads)
			tile_origin(vb_next < total ? vb_next : (int)blockIdx.x, tm0, tn0);
			const float* sa[A_NI]; const float* sb[B_NI];
#pragma unroll
			for (int i = 0; i < A_NI; i++) sa[i] = ga[i];
#pragma unroll
			for (int i = 0; i < B_NI; i++) sb[i] = gb[i];
			open_tile(tm0, tn0);
#pragma unroll
			for (int i = 0; i < A_NI; i++) { gan[i] = ga[i]; ga[i] = sa[i]; }
#pragma unroll
			for (int i = 0; i < B_NI; i++) { gbn[i] = gb[i]; gb[i] = sb[i]; }
		};
		float fa0[KK][TM][4], fb0[KK][TN][4], fa1[KK][TM][4], fb1[KK][TN][4];
		// step on the slab in buffer `cur` (fragments in P): first MFMA group, wait + barrier, read the next slab of the stream
		// into Q, fetch the slab after that into `cur`, remaining MFMAs
		auto pstep = [&](int cur, float (&pa)[KK][TM][4], float (&pb)[KK][TN][4], float (&qa)[KK][TM][4], float (&qb)[KK][TN][4]) {
			mfma_group(pa, pb, 0, 0);
			__builtin_amdgcn_sched_barrier(0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			const float* As = lds + (cur ^ 1) * (A_SZ + B_SZ);
#pragma unroll
			for (int kk = 0; kk < KK; kk++) frags(As, As + A_SZ, kk, qa[kk], qb[kk]);
			dma_next(cur);
			__builtin_amdgcn_sched_barrier(0);
			rest(pa, pb);
			__builtin_amdgcn_sched_barrier(0);
		};
		if (nkt > 0 && (int)blockIdx.x < total) {
			dma_next(0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
#pragma unroll
			for (int kk = 0; kk < KK; kk++) frags(lds, lds + A_SZ, kk, fa0[kk], fb0[kk]);
			dma_next(1);
			for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
				tile_origin(vb, m0, n0);
				plan_next(vb + gridDim.x);
				__builtin_amdgcn_sched_barrier(0);
				for (int kt = 0; kt < nkt; kt += 2) {
					pstep(0, fa0, fb0, fa1, fb1);
					pstep(1, fa1, fb1, fa0, fb0);
				}
				store_tile();
#pragma unroll
				for (int i = 0; i < TM; i++)
#pragma unroll
					for (int j = 0; j < TN; j++)
#pragma unroll
						for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		return;
	}

	if constexpr (NBUF == 3) {
		// Big-tile variant (256x128: half the DMA / barrier / LDS-read traffic per MFMA of the 128x128 tile, one residency
		// round at 4096^3) under the 256-register budget of two waves per SIMD: only HALF a slab of fragments is
		// prefetched across the barrier.  Sets X/Y alternate as "k-half 0 of the current / next slab", Z is k-half 1 of the
		// current slab, read right after the barrier (2,000+ cycles of MFMAs before its first use).  Because slab t is then
		// still being read after barrier t, the DMA of slab t+2 must not reuse its buffer: three LDS buffers.
		static_assert(KK == 2, "split-fragment pipeline is written for BK = 16");
		// Under a 256-register budget hipcc selects the all-VGPR MFMA forms (accumulators in architected VGPRs, measured
		// 10-25 % slower here) unless the function visibly uses AGPRs; an "a"-constrained operand is that signal.
		{ float agpr_hint = 0.f; asm volatile("; keep accumulators in AGPRs %0" ::"a"(agpr_hint)); }
		float xa[TM][4], xb[TN][4], ya[TM][4], yb[TN][4], za[TM][4], zb[TN][4];
		auto mf = [&](float (&a)[TM][4], float (&b)[TN][4], int j) {
#pragma unroll
			for (int im = 0; im < TM; im++)
#pragma unroll
				for (int in = 0; in < TN; in++) acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[im][j], b[in][j], acc[im][in], 0, 0, 0);
		};
		auto buf_of = [&](int c) { return lds + c * (A_SZ + B_SZ); };
		// slab kt (k-half 0 in ca/cb, LDS buffer c); slab kt+1 must exist (buffer c1); DMA of slab kt+2 goes to buffer c2
		auto step3 = [&](int c, int c1, int c2, bool do_dma, float (&ca)[TM][4], float (&cb)[TN][4], float (&na)[TM][4], float (&nb)[TN][4]) {
			mf(ca, cb, 0);
			__builtin_amdgcn_sched_barrier(0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			frags(buf_of(c), buf_of(c) + A_SZ, 1, za, zb);
			frags(buf_of(c1), buf_of(c1) + A_SZ, 0, na, nb);
			if (do_dma) dma(c2);
			__builtin_amdgcn_sched_barrier(0);
			mf(ca, cb, 1); mf(ca, cb, 2); mf(ca, cb, 3);
			mf(za, zb, 0); mf(za, zb, 1); mf(za, zb, 2); mf(za, zb, 3);
			__builtin_amdgcn_sched_barrier(0);
		};
		auto last3 = [&](int c, float (&ca)[TM][4], float (&cb)[TN][4]) {
			frags(buf_of(c), buf_of(c) + A_SZ, 1, za, zb);
			mf(ca, cb, 0); mf(ca, cb, 1); mf(ca, cb, 2); mf(ca, cb, 3);
			mf(za, zb, 0); mf(za, zb, 1); mf(za, zb, 2); mf(za, zb, 3);
		};
		if (nkt > 0) {
			dma(0);
			if (nkt > 1) dma(1);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			frags(buf_of(0), buf_of(0) + A_SZ, 0, xa, xb);
			int kt = 0, c = 0;
			auto nx = [](int v) { return v == 2 ? 0 : v + 1; };
			// Two slabs per trip so the X/Y roles are static.  The prefetch of a slab past the end is skipped by a scalar
			// branch (the DMA has no register results, so the branch costs no copies); the fragment reads of a missing slab
			// fetch stale LDS that is never multiplied.
			for (; kt + 1 < nkt; kt += 2) {
				step3(c, nx(c), nx(nx(c)), kt + 2 < nkt, xa, xb, ya, yb); c = nx(c);
				step3(c, nx(c), nx(nx(c)), kt + 3 < nkt, ya, yb, xa, xb); c = nx(c);
			}
			if (nkt & 1) last3(c, xa, xb);        // odd slab count: k-half 0 of the last slab is in X
			__builtin_amdgcn_sched_barrier(0);
		}
	} else {
	float fa0[KKW][TM][4], fb0[KKW][TN][4], fa1[KKW][TM][4], fb1[KKW][TN][4];
	if (nkt > 0) {
		dma(0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
#pragma unroll
		for (int kk = 0; kk < KKW; kk++) frags(lds, lds + A_SZ, wk * KKW + kk, fa0[kk], fb0[kk]);
		if (nkt > 1) dma(1);
		int kt = 0;
		for (; kt + 2 < nkt; kt += 2) {
			step(kt, true, fa0, fb0, fa1, fb1);               // slab kt+2 exists
			step(kt + 1, kt + 3 < nkt, fa1, fb1, fa0, fb0);
		}
		if (nkt - kt == 2) {
			step(kt, false, fa0, fb0, fa1, fb1);
			mfma_group(fa1, fb1, 0, 0);
			rest(fa1, fb1);
		} else {
			mfma_group(fa0, fb0, 0, 0);
			rest(fa0, fb0);
		}
	}
	}   // NBUF == 2

	if constexpr (WK > 1) {   // the groups' partial tiles meet in LDS (the slab buffers are free now), summed in group order
		constexpr int PER = TM * TN * 16 * 64;   // floats of one wave's accumulators, [block][register][lane]
		static_assert((WK - 1) * WM * WN * PER <= NBUF * (A_SZ + B_SZ), "the partial tiles must fit the slab buffers");
		__syncthreads();
		if (wk > 0) {
			float* mine = lds + ((wk - 1) * (WM * WN) + wsp) * PER;
#pragma unroll
			for (int i = 0; i < TM; i++)
#pragma unroll
				for (int j = 0; j < TN; j++)
#pragma unroll
					for (int r = 0; r < 16; r++) mine[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
		}
		__syncthreads();
		if (wk > 0) return;
#pragma unroll
		for (int g = 1; g < WK; g++) {
			const float* theirs = lds + ((g - 1) * (WM * WN) + wsp) * PER;
#pragma unroll
			for (int i = 0; i < TM; i++)
#pragma unroll
				for (int j = 0; j < TN; j++)
#pragma unroll
					for (int r = 0; r < 16; r++) acc[i][j][r] += theirs[((i * TN + j) * 16 + r) * 64 + lane];
		}
	}
	if constexpr (!PERSIST) store_tile();
}

// ---------------------------------------------------------------------------------------------
// Latency-bound shapes (few output tiles, long K: the MNIST-NN layers at batch 256, single-image conv
// products): one workgroup owns ONE 32x32 output tile and its NW waves split K among themselves.  No
// operand is shared between waves, so fragments go straight from global memory to VGPRs (no LDS, no
// barrier in the K loop; each wave prefetches PF k-groups ahead), the NW partial accumulators meet in LDS
// once at the end and are summed in wave order (deterministic), and the epilogue is fused -- no slab
// kernel, no second launch.  K-contiguous operands load 16 B per lane (row l&31, k-half l>>5),
// row-contiguous operands four coalesced dwords per lane.
constexpr int kWskLdsFloats = 4 * 2 * 2 * 32 * 32;   // [wave][set][operand][k of a half-chunk][T] (32 k x 32 or 64 k x 16)
struct WskShared {   // the LDS of one workgroup, declared once by the kernel (the pair kernel runs either body on it)
	float smem[kWskLdsFloats];
	float red_rs[4][64];
	int is_last;
};
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int T> struct WskAcc { typedef f32x16 type; };
template <> struct WskAcc<16> { typedef f32x4 type; };

// T = 32: one 32x32 tile per workgroup on v_mfma_f32_32x32x2_f32 (lane l: index l&31, k-slot l>>5, 8 k per load group).
// T = 16: one 16x16 tile per workgroup on v_mfma_f32_16x16x4_f32 (lane l: index l&15, k-slot l>>4, 16 k per load group): four times the
// workgroups for the same product -- the MNIST-NN layers give 8-64 tiles of 32x32 on a 256-CU chip, and every workgroup then moves half
// the operand bytes through its CU's memory queue, which is what bounds these launches (DESIGN 3.1 "latency-bound shapes").
// In both shapes a lane holds 4 consecutive k of its row / column (one 16-byte load) and MFMA j of a group multiplies element j of every
// lane, i.e. k = {group base + 4*slot + j}: the groups' k are permuted against the reference's ascending order, each is used once.
// bx = tile index (row-major over tiles_m x tiles_n), by = K-split index (0 when splits == 1)
template <int T, bool AKC, bool BKC, bool AVEC, bool BVEC>
__device__ __forceinline__ void wsk_body(GemmArgs p, const int bx, const int by, WskShared& sh) {
	constexpr int NW = 4;  // one wave per SIMD: a CU retires 256 fp32-MFMA FLOP/clk however many waves it hosts (8 / 16 measured slower)
	constexpr int PF = 4;  // load groups per half-chunk; two half-chunks (register sets) are in flight
	constexpr int QN = 64 / T;       // k-slots: lanes that share a row / column index
	constexpr int GK = 4 * QN;       // k per load group (8 or 16)
	constexpr int KH = GK * PF;      // k per half-chunk (32 or 64)
	constexpr int TT = T * T, RS = T + 1;
	constexpr int SEG = T / 4;       // lanes per staged row segment (16 bytes each)
	constexpr int RPI = 64 / SEG;    // k-rows one staging instruction covers (8 or 16); KH / RPI == PF
	typedef typename WskAcc<T>::type acc_t;
	constexpr int NR = T == 32 ? 16 : 4;
	// One LDS array: per-wave staging of row-contiguous operands during the K loop ([wave][set][operand][KH k][T]),
	// the cross-wave reduction afterwards.
	float* smem = sh.smem;
	float* red = smem;                 // [wave][T][RS]
	float (*red_rs)[64] = sh.red_rs;
	int& is_last = sh.is_last;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int li = lane & (T - 1), h = lane / T;
	const int tile_m = bx / p.tiles_n, tile_n = bx % p.tiles_n;
	const int m0 = tile_m * T, n0 = tile_n * T;
	const int kw = p.k_per_split;  // per-wave K extent, multiple of GK; a workgroup covers 4*kw, by selects which
	const int blk_end = min(p.K, (by + 1) * NW * kw);
	const int k_begin = min(blk_end, (by * NW + wave) * kw), k_end = min(blk_end, k_begin + kw);
	const int arow = min(m0 + li, p.M - 1), bcol = min(n0 + li, p.N - 1);  // clamped: out-of-range rows/cols are never stored
	float* stage = smem + wave * (2 * 2 * KH * T);   // [set][operand][KH k][T]

	// K-contiguous operand (row r of X, 4 consecutive k per lane): 16 B per lane when aligned
	auto load_kc = [&](const float* X, int ld, int r, bool vec, int k, float (&f)[4]) {
		const int kb = k + 4 * h;
		if (vec) {
			float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * ld + min(kb, p.K - 4));
			bool ok = kb < k_end;
			f[0] = ok ? x.x : 0.f; f[1] = ok ? x.y : 0.f; f[2] = ok ? x.z : 0.f; f[3] = ok ? x.w : 0.f;
		} else {
#pragma unroll
			for (int j = 0; j < 4; j++) { float x = X[(size_t)r * ld + min(kb + j, p.K - 1)]; f[j] = kb + j < k_end ? x : 0.f; }
		}
	};
	// Row-contiguous operand X[k][r], unaligned: four dword loads per lane and group
	auto load_rc_scalar = [&](const float* X, int ld, int r, int k, float (&f)[4]) {
		const int kb = k + 4 * h;
#pragma unroll
		for (int j = 0; j < 4; j++) { float x = X[(size_t)min(kb + j, p.K - 1) * ld + r]; f[j] = kb + j < k_end ? x : 0.f; }
	};
	// Row-contiguous operand, aligned: the wave fetches the whole KH x T chunk with PF x 16-byte loads per lane (SEG lanes
	// cover one row segment) into its private LDS slab, then every lane picks its fragment dwords from there.
	// 4x fewer vector-memory instructions than the dword form -- the texture addresser, not the MFMA pipe, bounds these kernels.
	auto stage_rc = [&](const float* X, int ld, int r0, int rdim, int k, float* stage) {
		const int c4 = (lane % SEG) * 4, kr = lane / SEG;
		float4 v[PF];
#pragma unroll
		for (int i = 0; i < PF; i++) {
			int kk = k + i * RPI + kr;
			bool ok = kk < k_end && r0 + c4 < rdim;
			float4 x = *reinterpret_cast<const float4*>(X + (size_t)min(kk, p.K - 1) * ld + min(r0 + c4, rdim - 4));
			v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
		}
#pragma unroll
		for (int i = 0; i < PF; i++) *reinterpret_cast<float4*>(stage + (i * RPI + kr) * T + c4) = v[i];
	};

#ifdef BLA_WSK_DIAG   // diagnostics build only (tools/wsk_stamps.py): s_memtime stamps of workgroup 0, wave 0
#define BLA_STAMP(i) do { if (bx == 0 && by == 0 && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (p.stamps) p.stamps[i] = t_; } } while (0)
#else
#define BLA_STAMP(i) do {} while (0)
#endif
	BLA_STAMP(0);
	acc_t acc;
#pragma unroll
	for (int r = 0; r < NR; r++) acc[r] = 0.f;
	float rs = 0.f;   // this lane's share of sum_k A[row][k] (fused bias gradient)
	const bool want_rs = p.row_sum_a != nullptr && tile_n == 0;
	// Half-chunks of PF load groups, software-pipelined over two register sets: the loads of half-chunk t+1 are issued
	// before the MFMAs of half-chunk t, so from the second half-chunk on the 1,000-3,000-cycle operand latency (measured
	// with s_memtime stamps) hides under 16 MFMAs instead of adding to them.
	auto fetch = [&](int k, int set, float (&fa)[PF][4], float (&fb)[PF][4]) {      // issue the loads of one half-chunk
		float* st_a = stage + (set * 2 + 0) * KH * T;
		float* st_b = stage + (set * 2 + 1) * KH * T;
		if (!AKC && AVEC) stage_rc(p.A, p.lda, m0, p.M, k, st_a);
		if (!BKC && BVEC) stage_rc(p.B, p.ldb, n0, p.N, k, st_b);
#pragma unroll
		for (int g = 0; g < PF; g++) {
			if (AKC) load_kc(p.A, p.lda, arow, AVEC, k + GK * g, fa[g]);
			else if (!AVEC) load_rc_scalar(p.A, p.lda, arow, k + GK * g, fa[g]);
			if (BKC) load_kc(p.B, p.ldb, bcol, BVEC, k + GK * g, fb[g]);
			else if (!BVEC) load_rc_scalar(p.B, p.ldb, bcol, k + GK * g, fb[g]);
		}
	};
	auto consume = [&](int set, float (&fa)[PF][4], float (&fb)[PF][4]) {           // fragments from LDS (staged operands), then MFMAs
		if ((!AKC && AVEC) || (!BKC && BVEC)) {
			const float* st_a = stage + (set * 2 + 0) * KH * T;
			const float* st_b = stage + (set * 2 + 1) * KH * T;
			__builtin_amdgcn_wave_barrier();   // LDS ops of one wave execute in order; this only pins the compiler's order
#pragma unroll
			for (int g = 0; g < PF; g++)
#pragma unroll
				for (int j = 0; j < 4; j++) {
					if (!AKC && AVEC) fa[g][j] = st_a[(GK * g + 4 * h + j) * T + li];
					if (!BKC && BVEC) fb[g][j] = st_b[(GK * g + 4 * h + j) * T + li];
				}
			__builtin_amdgcn_wave_barrier();
		}
		if (want_rs) {
#pragma unroll
			for (int g = 0; g < PF; g++) rs += (fa[g][0] + fa[g][1]) + (fa[g][2] + fa[g][3]);
		}
#pragma unroll
		for (int g = 0; g < PF; g++)
#pragma unroll
			for (int j = 0; j < 4; j++) {
				if constexpr (T == 32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][j], fb[g][j], acc, 0, 0, 0);
				else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[g][j], fb[g][j], acc, 0, 0, 0);
			}
	};
	{
		float fa0[PF][4], fb0[PF][4], fa1[PF][4], fb1[PF][4];
		if (k_begin < k_end) fetch(k_begin, 0, fa0, fb0);
		for (int k = k_begin; k < k_end; k += 2 * KH) {
			if (k + KH < k_end) fetch(k + KH, 1, fa1, fb1);
			consume(0, fa0, fb0);
			if (k + KH < k_end) {
				if (k + 2 * KH < k_end) fetch(k + 2 * KH, 0, fa0, fb0);
				consume(1, fa1, fb1);
			}
		}
	}
	BLA_STAMP(20);
	__syncthreads();   // every wave is done with its staging slabs before they are reused as the reduction buffer
	BLA_STAMP(21);
	// partial tiles -> LDS (stride T+1: the C/D map writes T consecutive columns per register), sum in wave order
#pragma unroll
	for (int r = 0; r < NR; r++) {
		const int row = T == 32 ? (r & 3) + 8 * (r >> 2) + 4 * h : 4 * h + r;
		red[wave * (T * RS) + row * RS + li] = acc[r];
	}
	if (want_rs) red_rs[wave][lane] = rs;
	__syncthreads();
	if (want_rs && tid < T && m0 + tid < p.M) {
		float t = 0.f;
#pragma unroll
		for (int w = 0; w < NW; w++)
#pragma unroll
			for (int qq = 0; qq < QN; qq++) t += red_rs[w][tid + T * qq];
		p.row_sum_a[m0 + tid] = p.rs_beta != 0.f ? p.rs_beta * p.row_sum_a[m0 + tid] + p.rs_alpha * t : p.rs_alpha * t;
	}
	auto fold = [&](int e) {   // element e of the tile, partials of the four waves added in wave order
		const int r = e / T, c = e % T;
		float s = 0.f;
#pragma unroll
		for (int w = 0; w < NW; w++) s += red[w * (T * RS) + r * RS + c];
		return s;
	};
	if (p.splits > 1) {
		// K is also cut over by (few tiles, long K: otherwise most CUs idle).  Each workgroup publishes its partial
		// tile, draws a ticket on the tile's counter, and the LAST arriver folds the partials in split order (deterministic)
		// and runs the epilogue -- no slab kernel, no second launch.  Hand-off per cdna_hip_programming.md "In-launch split-K
		// reduction": plain stores -> every wave s_waitcnt vmcnt(0) -> barrier -> lane 0 agent release fence -> vmcnt(0) ->
		// relaxed agent fetch_add; last arriver: agent acquire fence -> vmcnt(0) -> barrier -> plain loads.  Correct for any
		// placement of a tile's workgroups over CUs / XCDs.
		float* mine = p.slab + ((size_t)bx * p.splits + by) * TT;
		for (int e = tid; e < TT; e += NW * 64) mine[e] = fold(e);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (tid == 0) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			unsigned ticket = __hip_atomic_fetch_add(&p.counters[bx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			is_last = ticket == (unsigned)p.splits - 1;
			if (is_last) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__hip_atomic_store(&p.counters[bx], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
			}
		}
		__syncthreads();
		if (!is_last) return;
		const float* all = p.slab + (size_t)bx * p.splits * TT;
		for (int e = tid; e < TT; e += NW * 64) {
			int r = e / T, c = e % T;
			float s = 0.f;
			for (int z = 0; z < p.splits; z++) s += all[(size_t)z * TT + e];
			if (m0 + r < p.M && n0 + c < p.N) epilogue_store(p, m0 + r, n0 + c, s);
		}
		return;
	}
	if (p.softmax_grad == nullptr) {
		for (int e = tid; e < TT; e += NW * 64) {
			int r = e / T, c = e % T;
			float s = fold(e);
			if (m0 + r < p.M && n0 + c < p.N) epilogue_store(p, m0 + r, n0 + c, s);
		}
		return;
	}
	// fused tail for the output layer (M <= T: this tile holds whole columns): Z = alpha*acc + bias -> pre_act,
	// P = softmax over the rows of each column -> C, grad = (P - Y) * scale        (model/mnist_nn.c:231-234,260-268)
	for (int e = tid; e < TT; e += NW * 64) {
		int r = e / T, c = e % T;
		float s = fold(e);
		s *= p.alpha;
		if (p.bias_row && r < p.M) s += p.bias_row[r];
		red[r * RS + c] = s;   // element e is read and rewritten by this thread only: no barrier needed here
		if (p.pre_act && r < p.M && n0 + c < p.N) p.pre_act[(size_t)r * p.ld_pre + n0 + c] = s;
	}
	__syncthreads();
	if (tid < T && n0 + tid < p.N) {
		const int c = tid, col = n0 + tid;
		float mx = -INFINITY;
		for (int r = 0; r < p.M; r++) mx = fmaxf(mx, red[r * RS + c]);
		float sum = 0.f;
		for (int r = 0; r < p.M; r++) { float e = expf(red[r * RS + c] - mx); red[r * RS + c] = e; sum += e; }
		// bookkeeping of model/mnist_nn.c:237-257 for this column: prediction = first row whose probability exceeds every earlier one
		// (`> max_confidence` from 0), correct when the one-hot label has a 1 there; loss = -sum_r y * log(p + LOSS_EPSILON) in double.  Rows
		// with y == 0 contribute an exact -0.0 there (the logarithm is finite) and are skipped.  The reference walks the flat 10 x B arrays
		// in chunks of 10 (:252-254): over a whole batch that is the same set of terms, so the batch totals agree.
		int pred = 0; float best = 0.f; double loss = 0.0;
		for (int r = 0; r < p.M; r++) {
			float pr = red[r * RS + c] / sum;
			const float yv = p.softmax_y[(size_t)r * p.ldc + col];
			p.C[(size_t)r * p.ldc + col] = pr;
			p.softmax_grad[(size_t)r * p.ldc + col] = (pr - yv) * p.softmax_scale;
			if (p.sm_loss) {
				if (pr > best) { best = pr; pred = r; }
				if (yv != 0.f) loss += -1.0 * ((double)yv * log((double)pr + 1e-15));
			}
		}
		if (p.sm_loss) {   // one thread owns a column's slots: plain read-modify-write, deterministic
			p.sm_loss[col] += loss;
			p.sm_correct[col] += p.softmax_y[(size_t)pred * p.ldc + col] == 1.f ? 1u : 0u;
		}
	}
}

template <int T, bool AKC, bool BKC, bool AVEC, bool BVEC>
__global__ void __launch_bounds__(256) gemm_f32_wsk_kernel(GemmArgs p) {
	__shared__ __attribute__((aligned(16))) WskShared sh;
	wsk_body<T, AKC, BKC, AVEC, BVEC>(p, (int)blockIdx.x, (int)blockIdx.y, sh);
}

// Two independent latency-bound products in ONE launch: workgroups [0, tiles_p) run product p, the rest product q.  In the
// MNIST-NN backward pass dW_l = dZ_l . A_{l-1}^T (NT) and dZ_{l-1} = W_l^T . dZ_l (TN) both depend only on dZ_l: launched
// together they overlap instead of queueing, which takes two ~5 us launches off the step's critical path.
// Each product brings its own tile size (GemmArgs::wsk_tile).
template <bool AKC1, bool BKC1, bool AKC2, bool BKC2>
__global__ void __launch_bounds__(256) gemm_f32_wsk_pair_kernel(GemmArgs p, GemmArgs q, int tiles_p) {
	__shared__ __attribute__((aligned(16))) WskShared sh;
	if ((int)blockIdx.x < tiles_p) {
		if (p.wsk_tile == 16) wsk_body<16, AKC1, BKC1, true, true>(p, (int)blockIdx.x, 0, sh);
		else wsk_body<32, AKC1, BKC1, true, true>(p, (int)blockIdx.x, 0, sh);
	} else {
		if (q.wsk_tile == 16) wsk_body<16, AKC2, BKC2, true, true>(q, (int)blockIdx.x - tiles_p, 0, sh);
		else wsk_body<32, AKC2, BKC2, true, true>(q, (int)blockIdx.x - tiles_p, 0, sh);
	}
}

// The same product for `batch` operand sets at fixed strides (the per-image products of a batched attention block): blockIdx.y = set.
template <int T, bool AKC, bool BKC>
__global__ void __launch_bounds__(256) gemm_f32_wsk_batched_kernel(GemmArgs p, long stride_a, long stride_b, long stride_c, long stride_pre) {
	__shared__ __attribute__((aligned(16))) WskShared sh;
	const long z = blockIdx.y;
	p.A += z * stride_a; p.B += z * stride_b; p.C += z * stride_c;
	if (p.pre_act) p.pre_act += z * stride_pre;
	wsk_body<T, AKC, BKC, true, true>(p, (int)blockIdx.x, 0, sh);
}

// Three independent products of the same layout class (A and B both K-contiguous: the three weight gradients dW_l = dZ_l . A_{l-1}^T of one
// MNIST-NN step, model/mnist_nn.c:267-292) in ONE launch: workgroups [0, t1) run p, [t1, t2) run q, the rest r; each brings its own tile size.
__global__ void __launch_bounds__(256) gemm_f32_wsk_triple_nt_kernel(GemmArgs p, GemmArgs q, GemmArgs r, int t1, int t2) {
	__shared__ __attribute__((aligned(16))) WskShared sh;
	const int b = (int)blockIdx.x;
	const GemmArgs& a = b < t1 ? p : (b < t2 ? q : r);
	const int bx = b < t1 ? b : (b < t2 ? b - t1 : b - t2);
	if (a.wsk_tile == 16) wsk_body<16, true, true, true, true>(a, bx, 0, sh);
	else wsk_body<32, true, true, true, true>(a, bx, 0, sh);
}

// Sums the split-K slabs in split order (deterministic) and applies the epilogue.
// Plain epilogue on a contiguous C: 16 bytes per thread, eight slab loads in flight (the sum still runs in split order).
__global__ void __launch_bounds__(64) gemm_splitk_reduce4_kernel(GemmArgs p) {
	const size_t total4 = (size_t)p.M * p.N / 4;
	const float4* slab = reinterpret_cast<const float4*>(p.slab);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
		float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
		int z = 0;
		for (; z + 8 <= p.splits; z += 8) {
			float4 v[8];
#pragma unroll
			for (int u = 0; u < 8; u++) v[u] = slab[(size_t)(z + u) * total4 + i];
#pragma unroll
			for (int u = 0; u < 8; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
		}
		for (; z < p.splits; z++) { const float4 v = slab[(size_t)z * total4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
		reinterpret_cast<float4*>(p.C)[i] = make_float4(p.alpha * s.x, p.alpha * s.y, p.alpha * s.z, p.alpha * s.w);
	}
}
static hipError_t launch_splitk_reduce(const GemmArgs& r, hipStream_t s);
__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(GemmArgs p) {
	size_t total = (size_t)p.M * p.N;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		float s = 0.f;
		for (int z = 0; z < p.splits; z++) s += p.slab[(size_t)z * total + i];
		epilogue_store(p, (int)(i / p.N), (int)(i % p.N), s);
	}
}

static hipError_t launch_splitk_reduce(const GemmArgs& r, hipStream_t s) {
	const size_t total = (size_t)r.M * r.N;
	const bool plain = !r.bias_row && !r.bias_col && !r.pre_act && r.act == BLA_ACT_NONE && !r.relu_mask && r.beta == 0.f;
	if (plain && r.ldc == r.N && total % 4 == 0 && (uintptr_t)r.C % 16 == 0 && (uintptr_t)r.slab % 16 == 0) {
		size_t blocks = (total / 4 + 63) / 64;
		hipLaunchKernelGGL(gemm_splitk_reduce4_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(64), 0, s, r);
	} else {
		size_t blocks = (total + 255) / 256;
		hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, r);
	}
	return hipGetLastError();
}

struct Config { int bm, bn, bk, threads; bool glds; const char* name; };
static const Config kConfigs[] = {
	{128, 128, 16, 256, false, "t128x128x16"},
	{64, 64, 16, 256, false, "t64x64x16"},
	{128, 128, 32, 256, false, "t128x128x32"},
	{128, 128, 16, 256, true, "glds128x128x16"},
	{64, 64, 16, 256, true, "glds64x64x16"},
	{128, 128, 32, 256, true, "glds128x128x32"},
	{32, 32, 8, 256, false, "wsk32x32"},        // wave-split-K: 4 waves per 32x32 tile, latency-bound shapes
	{128, 64, 16, 256, true, "glds128x64x16"},
	{128, 256, 16, 256, true, "glds128x256x16"},
	{256, 128, 16, 256, true, "glds256x128x16"},
	{128, 128, 16, 256, true, "glds128x128x16p"},   // persistent: 2 workgroups per CU walk the tile list, slab stream continuous across tiles
	{256, 256, 16, 256, true, "glds256x256x16"},    // one workgroup per CU, each wave a 128x128 sub-tile (256 accumulator registers)
	{256, 256, 32, 256, true, "glds256x256x32"},    // same with 32-deep slabs: one barrier per 256 MFMAs, 128 KB of LDS
	{128, 512, 16, 256, true, "glds128x512x16"},    // the same pipeline for products with 128 rows: four waves side by side, each 128x128
	{128, 128, 16, 256, true, "glds128x128x16h"},   // the same pipeline on the 128x128 tile (whole tiles, plain epilogue)
	{128, 256, 16, 256, true, "glds128x256x16h"},   // ... and on 128x256 (waves 2x2, each 64x128): +8 % on 128 x 1152 x 65536, behind elsewhere
	{16, 16, 16, 256, false, "wsk16x16"},       // wave-split-K on 16x16 tiles (MFMA 16x16x4): forced form of what config 6 picks by itself for few tiles
	{192, 192, 16, 256, true, "glds192x192x16h"},   // the half-slab pipeline on 192x192 (waves 2x2, each 96x96 = 3x3 blocks): 3072^2 is exactly 256 of them
	{64, 64, 32, 512, true, "glds64x64x32k2"},      // 64x64 tiles, 32-deep slabs, two groups of 2x2 waves along K (two waves per SIMD on a one-tile-per-CU problem: 1024^3)
};
static constexpr int kCfgHs192 = 17;
static constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

static int g_force_config = -1, g_force_split = 0;
static thread_local int t_plan_batch = 1;   // bla_gemm_batched_f32: the tile-size choice of the wave-split-K kernel counts the tiles of all sets
#ifdef BLA_WSK_DIAG
static unsigned long long* g_diag_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) void bla_diag_set_stamps(void* p) { g_diag_stamps = (unsigned long long*)p; }
#endif
static char g_last_kernel[96] = "none";

template <int BM, int BN, int BK, int WM, int WN>
static hipError_t launch_variant(const GemmArgs& a, bool akc, bool bkc, int mode, dim3 grid, hipStream_t s) {
	constexpr int A_SZ_KC = BM * (BK + 4), A_SZ_RC = BK * BM, B_SZ_KC = BN * (BK + 4), B_SZ_RC = BK * BN;
	size_t lds_bytes = 2 * ((akc ? A_SZ_KC : A_SZ_RC) + (bkc ? B_SZ_KC : B_SZ_RC)) * sizeof(float);
	dim3 block(WM * WN * 64);
#define BLA_LAUNCH(AK, BK_, MODE_)                                                                          \
	do {                                                                                                    \
		auto kern = gemm_f32_kernel<BM, BN, BK, WM, WN, AK, BK_, MODE_>;                                    \
		if (lds_bytes > 48 * 1024) {                                                                        \
			hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
			if (e != hipSuccess) return e;                                                                  \
		}                                                                                                   \
		hipLaunchKernelGGL(kern, grid, block, lds_bytes, s, a);                                             \
		return hipGetLastError();                                                                           \
	} while (0)
#define BLA_LAUNCH_MODE(AK, BK_)                                            \
	do {                                                                    \
		if (mode == LOAD_FULL) BLA_LAUNCH(AK, BK_, LOAD_FULL);              \
		else if (mode == LOAD_VEC) BLA_LAUNCH(AK, BK_, LOAD_VEC);           \
		else BLA_LAUNCH(AK, BK_, LOAD_SCALAR);                              \
	} while (0)
	if (akc && !bkc) BLA_LAUNCH_MODE(true, false);
	if (akc && bkc) BLA_LAUNCH_MODE(true, true);
	if (!akc && !bkc) BLA_LAUNCH_MODE(false, false);
	BLA_LAUNCH_MODE(false, true);
#undef BLA_LAUNCH_MODE
#undef BLA_LAUNCH
}

template <int BM, int BN, int BK, int WM, int WN, int MINW = 1, int NBUF = 2, bool PERSIST = false, bool HS = false, int WK = 1>
static hipError_t launch_glds(const GemmArgs& a, bool akc, bool bkc, dim3 grid, hipStream_t s) {
	size_t lds_bytes = NBUF * (BM + BN) * BK * sizeof(float);
	dim3 block(WM * WN * WK * 64);
#define BLA_LAUNCH2(AK, BK_, RG)                                                                            \
	do {                                                                                                    \
		auto kern = gemm_f32_glds_kernel<BM, BN, BK, WM, WN, AK, BK_, MINW, NBUF, PERSIST, 0, RG, HS, WK>;  \
		if (lds_bytes > 48 * 1024) {                                                                        \
			hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
			if (e != hipSuccess) return e;                                                                  \
		}                                                                                                   \
		hipLaunchKernelGGL(kern, grid, block, lds_bytes, s, a);                                             \
		return hipGetLastError();                                                                           \
	} while (0)
#define BLA_LAUNCH(AK, BK_)                                                                                 \
	do {                                                                                                    \
		if (!((AK) && (BK_)) && a.rc_global && NBUF == 2 && !PERSIST) BLA_LAUNCH2(AK, BK_, true);           \
		BLA_LAUNCH2(AK, BK_, false);                                                                        \
	} while (0)
	if (akc && !bkc) BLA_LAUNCH(true, false);
	if (akc && bkc) BLA_LAUNCH(true, true);
	if (!akc && !bkc) BLA_LAUNCH(false, false);
	BLA_LAUNCH(false, true);
#undef BLA_LAUNCH2
#undef BLA_LAUNCH
}

static hipError_t launch_wsk(const GemmArgs& a, bool akc, bool bkc, bool avec, bool bvec, dim3 grid, hipStream_t s) {
	dim3 block(256);
#define BLA_W(AK, BK_, AV, BV) do { hipLaunchKernelGGL((gemm_f32_wsk_kernel<32, AK, BK_, AV, BV>), grid, block, 0, s, a); return hipGetLastError(); } while (0)
#define BLA_W16(AK, BK_) do { hipLaunchKernelGGL((gemm_f32_wsk_kernel<16, AK, BK_, true, true>), grid, block, 0, s, a); return hipGetLastError(); } while (0)
#define BLA_WV(AK, BK_) do { if (avec && bvec && a.wsk_tile == 16) BLA_W16(AK, BK_); if (avec && bvec) BLA_W(AK, BK_, true, true); if (avec) BLA_W(AK, BK_, true, false); if (bvec) BLA_W(AK, BK_, false, true); BLA_W(AK, BK_, false, false); } while (0)
	if (akc && !bkc) BLA_WV(true, false);
	if (akc && bkc) BLA_WV(true, true);
	if (!akc && !bkc) BLA_WV(false, false);
	BLA_WV(false, true);
#undef BLA_WV
#undef BLA_W16
#undef BLA_W
}

// K splits of a gathered weight-gradient product (contraction over (image, pixel), K = batch * HWo, few output tiles).  The tiles are MFMA-bound
// and equally long, so what matters is that every CU gets the SAME number of workgroups: tiles * splits is rounded DOWN to a whole number of
// workgroups per CU (two) -- 9 tiles x 32 splits on 256 CUs
// left 32 CUs with two workgroups and took twice the time of 9 x 28.  A split is a whole number of 16-deep slabs, not of images.
static bool gather_hs(int mode, int M, int N) {
	static const bool use_hs = [] { const char* e = getenv("BLA_CONV_HS"); return !(e && e[0] == '0'); }();
	// mode 3 (forward / data gradient): the half-slab form (172 vs 198 us at 128->128 @32x32 x64).  Mode 4 (weight gradient): also, with its K cut for
	// TWO workgroups per CU (they fit: 32 KB of LDS, under half the registers) -- 183 against 191 us on the older form; cut for one per CU it
	// measured slower (208).  BLA_CONV_HS=1 keeps the weight gradient on the older form, BLA_CONV_HS=0 everything.
	static const bool hs4 = [] { const char* e = getenv("BLA_CONV_HS"); return !(e && e[0] == '1'); }();
	return use_hs && M % 128 == 0 && N % 128 == 0 && (mode == 3 || (mode == 4 && hs4));
}
static int gather_k_per_split(int mode, int batch, int M, int N, int HWo) {
	const long K = (long)batch * HWo;
	if (mode != 2 && mode != 4) return (int)K;
	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
	const long slots = 2L * cus;   // two workgroups per CU on either form
	long splits = slots / tiles;
	const long slabs = K / 16;
	if (splits > slabs / 8) splits = slabs / 8;      // at least 8 slabs per split
	if (splits < 1) splits = 1;
	return (int)((slabs + splits - 1) / splits) * 16;
}
// Forward / data gradient on small feature maps (8x8, 4x4: fewer 128x128 tiles than CUs): the contraction over the taps is cut so that about
// one workgroup sits on every CU, at least 8 slabs each; the slabs have the shape of the output and are summed flat.
int gather3_splits(int M, int N, int K) {
	if (!gather_hs(3, M, N)) return 1;
	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	const long tiles = (long)(M / 128) * (N / 128), slabs = K / 16;
	long splits = cus / tiles;
	if (splits > slabs / 8) splits = slabs / 8;
	if (splits < 1) splits = 1;
	const long per = (slabs + splits - 1) / splits;
	return (int)((slabs + per - 1) / per);
}
bool gather3_fuses_epilogue(int M, int N, int K) { return gather_hs(3, M, N) && gather3_splits(M, N, K) == 1; }
int gather_gemm_splits(int mode, int batch, int M, int N, int HWo) {
	if (mode != 2 && mode != 4) return 1;
	const long K = (long)batch * HWo;
	const int kps = gather_k_per_split(mode, batch, M, N, HWo);
	return (int)((K + kps - 1) / kps);
}

bla_status gather_gemm(hipStream_t s, int mode, int batch, int M, int N, int K, const float* A, int lda, float* C, int ldc, const float* img,
                       const int2* ktab, const int2* ntab, int H, int W, int HWo, int img_stride, const GatherEpilogue* ep) {
	BLA_REQUIRE(mode >= 1 && mode <= 4, BLA_ERR_INVALID, "gather mode %d", mode);
	BLA_REQUIRE(mode != 3 || (N % 4 == 0 && HWo % 4 == 0 && N >= 4), BLA_ERR_INVALID, "mode 3 needs pixel counts that are multiples of 4");
	BLA_REQUIRE(mode != 4 || M % 4 == 0, BLA_ERR_INVALID, "mode 4 needs a tap count that is a multiple of 4");
	BLA_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0 && lda % 4 == 0 && (uintptr_t)A % 16 == 0 && ((mode != 2 && mode != 4) || HWo % 16 == 0), BLA_ERR_INVALID,
	            "gathered product needs K %% 16 == 0 and a 16-byte aligned dense operand (M=%d N=%d K=%d lda=%d)", M, N, K, lda);
	BLA_REQUIRE((long)batch * img_stride < (mode >= 3 ? (1L << 29) : (1L << 31)) && (long)N < (1L << 31), BLA_ERR_INVALID, "batch too large for 32-bit gather offsets");
	GemmArgs a = {};
	a.C = C; a.M = M; a.N = N; a.K = K; a.ldc = ldc;
	if (mode == 4) { a.A = nullptr; a.lda = 0; a.B = A; a.ldb = lda; }      // the dense operand (del_y) is the K-contiguous B
	else { a.A = A; a.lda = lda; a.B = nullptr; a.ldb = 0; }
	a.alpha = 1.f; a.beta = 0.f; a.act = BLA_ACT_NONE;
	a.g_img = img; a.g_zero = zero_word(); a.g_ktab = ktab; a.g_ntab = ntab; a.g_mode = mode; a.g_H = H; a.g_W = W; a.g_HWo = HWo; a.g_img_stride = img_stride;
	a.tiles_m = (M + 127) / 128; a.tiles_n = (N + 127) / 128;
	// mode 3 on the half-slab pipeline: 128 x 256 tiles (waves 2 x 2, each 64 x 128: half the LDS reads and DMA instructions per MFMA of the 128 x 128 tile)
	// once they give every CU a workgroup; BLA_CONV_BN=128 keeps the 128 x 128 tile
	static const int force_bn = [] { const char* e = getenv("BLA_CONV_BN"); return e ? atoi(e) : 0; }();
	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	const bool bn256 = mode == 3 && gather_hs(mode, M, N) && N % 256 == 0 && force_bn != 128 && ((long)(M / 128) * (N / 256) >= cus || force_bn == 256) && gather3_splits(M, N, K) == 1;
	if (bn256) a.tiles_n = N / 256;
	const int splits = mode == 3 ? gather3_splits(M, N, K) : gather_gemm_splits(mode, batch, M, N, HWo);
	a.k_per_split = (mode == 2 || mode == 4) ? gather_k_per_split(mode, batch, M, N, HWo) : mode == 3 ? (K / 16 + splits - 1) / splits * 16 : K;
	a.splits = splits; a.slab = nullptr;
	if (splits > 1) {
		void* ws;
		bla_status st = ensure_workspace((size_t)splits * M * N * sizeof(float), &ws);
		if (st) return st;
		a.slab = (float*)ws;
	}
	dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)splits), block(256);
	size_t lds_bytes = 2 * (128 + 128) * 16 * sizeof(float);
	// whole tiles: the half-slab pipeline (fragment sets per k-half, every LDS read and DMA dealt out between MFMAs) -- BLA_CONV_HS=0 keeps the older form
	const bool hs = gather_hs(mode, M, N);
	if (ep && (ep->bias || ep->out2)) {
		BLA_REQUIRE(mode == 3 && hs && splits == 1, BLA_ERR_INVALID, "the fused convolution epilogue needs the half-slab forward kernel in one pass over K (gather3_fuses_epilogue)");
		a.g_bias = ep->bias; a.g_bias_stride = ep->bias_stride; a.g_add = ep->add; a.g_out2 = ep->out2;
	}
	if (bn256) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 256, 16, 2, 2, true, false, 1, 2, false, 3, false, true>), grid, block, 2 * (128 + 256) * 16 * sizeof(float), s, a);
	else if (hs && mode == 3) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 3, false, true>), grid, block, lds_bytes, s, a);
	else if (hs) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, true, 1, 2, false, 4, false, true>), grid, block, lds_bytes, s, a);
	else if (mode == 1) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 1>), grid, block, lds_bytes, s, a);
	else if (mode == 2) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 2>), grid, block, lds_bytes, s, a);
	else if (mode == 3) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 3>), grid, block, lds_bytes, s, a);
	else hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, true, 1, 2, false, 4>), grid, block, lds_bytes, s, a);
	BLA_HIP(hipGetLastError());
	if (splits > 1) {
		GemmArgs r = a;
		if (mode == 4) { r.M = N; r.N = M; }   // the slabs hold the transposed tile: [split][N][M] -> C [N][M]
		if (mode == 3) r.ldc = r.N;            // C-shaped slabs ([image][M][HWo]): a flat sum
		BLA_HIP(launch_splitk_reduce(r, s));
	}
	return BLA_OK;
}

}  // namespace bla

using namespace bla;

extern "C" {

bla_status bla_gemm_set_config(int config, int split_k) {
	BLA_REQUIRE(config >= -1 && config < kNumConfigs, BLA_ERR_INVALID, "config %d out of range [-1,%d)", config, kNumConfigs);
	BLA_REQUIRE(split_k >= 0 && split_k <= 64, BLA_ERR_INVALID, "split_k %d out of range [0,64]", split_k);
	g_force_config = config;
	g_force_split = split_k;
	return BLA_OK;
}

const char* bla_gemm_last_kernel(void) { return g_last_kernel; }

}  // extern "C"

namespace {
struct WskPlan { GemmArgs a; bool akc, bkc, valid; };
}

// plan != nullptr: a product that resolves to the un-split, fully vectorised wave-split-K kernel is NOT launched but handed
// back (plan->valid) so that bla_gemm_pair_f32 can put two of them into one launch; anything else launches as usual.
static bla_status gemm_impl(void* stream, int transa, int transb, int m, int n, int k,
                            const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                            const bla_gemm_epilogue* ep, WskPlan* plan) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(m >= 0 && n >= 0 && k >= 0, BLA_ERR_INVALID, "negative dimension m=%d n=%d k=%d", m, n, k);
	if (m == 0 || n == 0) return BLA_OK;
	BLA_REQUIRE(C && (k == 0 || (A && B)), BLA_ERR_INVALID, "null operand pointer");
	BLA_REQUIRE(lda >= (transa ? m : k) && ldb >= (transb ? k : n) && ldc >= n, BLA_ERR_INVALID,
	            "leading dimension too small (lda=%d ldb=%d ldc=%d for m=%d n=%d k=%d ta=%d tb=%d)", lda, ldb, ldc, m, n, k, transa, transb);
	hipStream_t s = pick_stream(stream);

	GemmArgs a;
	a.A = A; a.B = B; a.C = C; a.M = m; a.N = n; a.K = k; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
	a.alpha = ep ? ep->alpha : 1.f; a.beta = ep ? ep->beta : 0.f;
	a.bias_row = ep ? ep->bias_row : nullptr; a.bias_col = ep ? ep->bias_col : nullptr;
	a.pre_act = ep ? ep->pre_act : nullptr; a.ld_pre = ep ? ep->ld_pre : 0; a.act = ep ? ep->act : BLA_ACT_NONE;
	a.relu_mask = ep ? ep->relu_mask : nullptr; a.ld_mask = ep ? ep->ld_mask : 0;
	a.row_sum_a = ep ? ep->row_sum_a : nullptr;
	a.rs_alpha = ep ? ep->row_sum_alpha : 0.f; a.rs_beta = ep ? ep->row_sum_beta : 0.f;
	if (a.rs_alpha == 0.f && a.rs_beta == 0.f) a.rs_alpha = 1.f;
	{   // row-contiguous operand pitches that are powers of two up to 16 KiB: global_load_lds measured ahead (see the kernel)
		auto pow2_small = [](int ld) { return ld > 0 && (ld & (ld - 1)) == 0 && ld <= 4096; };
		a.rc_global = (transa ? pow2_small(lda) : true) && (!transb ? pow2_small(ldb) : true);
	}
	a.softmax_y = ep ? ep->softmax_y : nullptr; a.softmax_scale = ep ? ep->softmax_scale : 0.f; a.softmax_grad = ep ? ep->softmax_grad : nullptr;
	a.sm_loss = ep ? ep->softmax_loss_acc : nullptr; a.sm_correct = ep ? ep->softmax_correct_acc : nullptr;
	BLA_REQUIRE((a.sm_loss == nullptr) == (a.sm_correct == nullptr) && (!a.sm_loss || a.softmax_grad), BLA_ERR_INVALID,
	            "softmax_loss_acc and softmax_correct_acc go together and need the fused softmax tail");
	BLA_REQUIRE(!a.row_sum_a || !transa, BLA_ERR_INVALID, "row_sum_a needs a non-transposed A");
	BLA_REQUIRE((a.softmax_y == nullptr) == (a.softmax_grad == nullptr), BLA_ERR_INVALID, "softmax_y and softmax_grad go together");
	BLA_REQUIRE(!a.softmax_grad || (m <= 32 && a.beta == 0.f && !a.relu_mask && a.act == BLA_ACT_NONE && !a.bias_col && k > 0), BLA_ERR_INVALID,
	            "fused softmax needs m <= 32, k > 0 and no other post-ops than bias_row/pre_act");
	BLA_REQUIRE(!a.pre_act || a.ld_pre >= n, BLA_ERR_INVALID, "ld_pre %d < n %d", a.ld_pre, n);
	BLA_REQUIRE(!a.relu_mask || a.ld_mask >= n, BLA_ERR_INVALID, "ld_mask %d < n %d", a.ld_mask, n);

	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	const bool akc = !transa, bkc = transb != 0;
	const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
	const int a_contig = akc ? k : m, b_contig = bkc ? k : n;   // extents along the contiguous axis of each operand
	const bool vec_ok = aligned && a_contig % 4 == 0 && b_contig % 4 == 0 && a_contig >= 4 && b_contig >= 4;
	int cfg = g_force_config;
	if (cfg < 0) {
		long big_tiles = (long)((m + 127) / 128) * ((n + 127) / 128);
		long tiles32 = (long)((m + 31) / 32) * ((n + 31) / 32);
		bool big = big_tiles >= cus / 2;
		// Latency-bound shapes: one 32x32 tile per workgroup, K split over its 4 waves (one per SIMD), no slabs.
		// A CU retires 256 fp32 MFMA FLOP/clk whatever the wave count (measured: 8 or 16 waves per tile only add
		// overhead), so a tile costs ~8*K cycles; beyond K ~ 1024 splitting K over workgroups (tiled path) wins.
		if (((!big && tiles32 <= 512 && k <= 1280) || a.softmax_grad) && k > 0) cfg = 6;
		else if (vec_ok && k % 16 == 0 && k > 0) {
			// Direct-to-LDS family; tile by a small cost model of the residency rounds: a CU works through ceil(tiles / CUs) tiles, a tile
			// costs its area over the tile's in-loop efficiency (128x128 1.0, 128x64 0.95, 64x64 0.87), and a CU left with fewer than two
			// workgroups loses the overlap between them (x 0.88).  Reproduces the measured order at 1024^3 (64x64: 87 vs 69 / 58 TFLOP/s),
			// 2048^3 (128x64: 136 vs 124 / 123), 3072^3 (64x64: 125 vs 118 / 104 -- 576 big tiles are 2.25 rounds) and 4096^3 (128x128).
			const int cand[8] = {3, 7, 4, 11, 13, 14, kCfgHs192, 18};
			const double eff[8] = {1.0, 0.95, 0.87, 1.03, 1.03, 1.0, 1.03, 0.93};
			// Half-slab pipeline (configs 11, 13, 14, 17): one workgroup per CU by design (no x 0.88), whole tiles, plain epilogue, 16-byte aligned C
			// only.  The tile is picked so that the tile count is a whole number of rounds over the CUs: 4096^2 = 256 tiles of 256x256, 3072^2 = 256
			// of 192x192 (147.6 TFLOP/s against 128 on 64x64 tiles), 2048^2 = 256 of 128x128 (134.6 against 125.7 on 128x64).
			const bool big_ok = k >= 32 && ldc % 4 == 0 && (uintptr_t)C % 16 == 0 && !a.bias_row && !a.bias_col &&
			                    !a.pre_act && a.act == BLA_ACT_NONE && !a.relu_mask && a.beta == 0.f && !a.row_sum_a;
			double best = 0;
			for (int i = 0; i < 8; i++) {
				const Config& cc = kConfigs[cand[i]];
				if (i >= 3 && i < 7 && !(big_ok && m % cc.bm == 0 && n % cc.bn == 0)) continue;
				long t = (long)((m + cc.bm - 1) / cc.bm) * ((n + cc.bn - 1) / cc.bn);
				// 64x64 tiles with two wave groups along K (config 18): the small problems where a CU holds one or two tiles (1024^3: 19.8 against 20.6 us,
				// 1280^3: 41.9 against 45.3, 1536^3: 71 against 77); needs whole 32-deep slabs
				if (i == 7 && !(k % 32 == 0 && t <= 3L * cus)) continue;
				long rounds = (t + cus - 1) / cus;
				double cost = (double)rounds * cc.bm * cc.bn / eff[i];
				if (t < 2L * cus && (i < 3 || i == 7)) cost /= 0.88;
				if (t < cus && i < 3) cost *= 1.25;   // ... and will have its K cut over workgroups: slabs and a fold launch (1280^3 on 128x64 tiles: 51 us)
				if (i >= 3 && i < 7 && t < cus) cost *= 2;   // a partly filled chip: leave it to the smaller tiles / split-K
				if (i == 0 || cost < best) { best = cost; cfg = cand[i]; }
			}
		}
		else cfg = big ? 0 : 1;
	}
	const bool force_wsk16 = cfg == 16;
	if (cfg == 16) cfg = 6;
	if (cfg == 6) {   // wave-split-K kernel: one 32x32 (or 16x16) tile per workgroup, K divided over its waves
		BLA_REQUIRE(k > 0, BLA_ERR_INVALID, "gemm config %d needs k > 0", cfg);
		const int nw = 4;
		// 16-byte loads per operand: along K for a K-contiguous operand, along the rows/columns for a row-contiguous one
		const bool a_al = lda % 4 == 0 && (uintptr_t)A % 16 == 0, b_al = ldb % 4 == 0 && (uintptr_t)B % 16 == 0;
		const bool avec = a_al && (akc ? (k % 4 == 0 && k >= 4) : (m % 4 == 0 && m >= 4));
		const bool bvec = b_al && (bkc ? (k % 4 == 0 && k >= 4) : (n % 4 == 0 && n >= 4));
		const long tiles32 = (long)((m + 31) / 32) * ((n + 31) / 32);
		int ksplit = g_force_split > 0 ? g_force_split : 1;
		// (automatic only for long contractions: at K <= 1280 the release/acquire hand-off (~2 us per doubling) costs what the
		// shorter K loop saves -- measured 256x784x256: 11.0 / 10.8 / 11.2 us at 1 / 2 / 4 splits)
		if (g_force_split <= 0 && tiles32 < cus && k > 1280 && !a.row_sum_a && !a.softmax_grad) {   // idle CUs and a long K: cut K over workgroups too
			long want = (cus + tiles32 - 1) / tiles32, maxs = k / 128;
			ksplit = (int)(want < maxs ? want : maxs);
			if (ksplit > 8) ksplit = 8;
			if (ksplit < 1) ksplit = 1;
		}
		if (a.row_sum_a || a.softmax_grad || tiles32 > 16384) ksplit = 1;
		// Tile size: 16x16 tiles (four times the workgroups, half the operand bytes through each CU's memory queue) while the 32x32 tiling
		// leaves CUs idle; BLA_WSK_TILE=16|32 forces it (experiments).
		static const int forced_tile = [] { const char* e = getenv("BLA_WSK_TILE"); return e ? atoi(e) : 0; }();
		static const long tile16_below = [] { const char* e = getenv("BLA_WSK_TILE16_BELOW"); return e ? atol(e) : 129L; }();
		int tile = 32;
		const bool can16 = avec && bvec && ksplit == 1 && (!a.softmax_grad || m <= 16);
		if (force_wsk16) {
			BLA_REQUIRE(can16, BLA_ERR_INVALID, "gemm config 16 (wsk16x16) needs 16-byte loads on both operands, no K split over workgroups and m <= 16 with the fused softmax");
			tile = 16;
		} else if (g_force_config == 6) tile = 32;
		else if (can16 && (forced_tile == 16 || (forced_tile == 0 && tiles32 * t_plan_batch < tile16_below))) tile = 16;
		a.wsk_tile = tile;
		a.tiles_m = (m + tile - 1) / tile; a.tiles_n = (n + tile - 1) / tile;
		const long wtiles = (long)a.tiles_m * a.tiles_n;
		const int gk = tile == 16 ? 16 : 8;   // k per load group: the per-wave extent is a multiple of it
		a.k_per_split = (((k + ksplit - 1) / ksplit + nw - 1) / nw + gk - 1) / gk * gk;
		ksplit = (k + nw * a.k_per_split - 1) / (nw * a.k_per_split);
		a.splits = ksplit; a.slab = nullptr; a.counters = ctx().tile_counters;
#ifdef BLA_WSK_DIAG
		a.stamps = g_diag_stamps;
#endif
		if (ksplit > 1) {
			void* ws;
			st = ensure_workspace((size_t)wtiles * ksplit * 1024 * sizeof(float), &ws);
			if (st) return st;
			a.slab = (float*)ws;
		}
		dim3 grid((unsigned)wtiles, (unsigned)ksplit);
		if (plan && ksplit == 1 && avec && bvec) { plan->a = a; plan->akc = akc; plan->bkc = bkc; plan->valid = true; return BLA_OK; }
		hipError_t e = launch_wsk(a, akc, bkc, avec, bvec, grid, s);
		if (e != hipSuccess) return hip_fail(e, "gemm_f32_wsk_kernel launch");
		snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_wsk%dx%d_%c%c_%s%s_ksplit%d", tile, tile, transa ? 't' : 'n', transb ? 't' : 'n',
		         avec ? "v" : "s", bvec ? "v" : "s", ksplit);
		return BLA_OK;
	}
	BLA_REQUIRE(!a.softmax_grad || cfg == 6, BLA_ERR_INVALID, "fused softmax is only available on the wave-split-K config (6)");
	float* deferred_row_sum = nullptr;   // tiled kernels do not fuse the row sum: run it as a separate pass below
	BLA_REQUIRE(cfg == 6 || !a.row_sum_a || (a.rs_alpha == 1.f && a.rs_beta == 0.f), BLA_ERR_INVALID, "a scaled / accumulated row sum is only available on the wave-split-K config (6)");
	if (cfg != 6 && a.row_sum_a) { deferred_row_sum = a.row_sum_a; a.row_sum_a = nullptr; }
	if (kConfigs[cfg].glds && !(vec_ok && k > 0 && k % kConfigs[cfg].bk == 0)) {
		set_error("gemm config %d (%s) needs 16-byte aligned operands, contiguous extents %% 4 == 0 and k %% %d == 0", cfg,
		          kConfigs[cfg].name, kConfigs[cfg].bk);
		return BLA_ERR_INVALID;
	}
	const Config& c = kConfigs[cfg];
	a.tiles_m = (m + c.bm - 1) / c.bm;
	a.tiles_n = (n + c.bn - 1) / c.bn;
	long tiles = (long)a.tiles_m * a.tiles_n;
	int splits = g_force_split;
	if (splits <= 0) {
		splits = 1;
		if (tiles < cus && k >= 256) {  // fill the chip: aim at ~2 workgroups per CU, keep >= 128 k per split
			long want = (2L * cus + tiles - 1) / tiles;
			long maxs = k / 128;
			splits = (int)(want < maxs ? want : maxs);
			if (splits < 1) splits = 1;
			if (splits > 32) splits = 32;
		}
	}
	if ((cfg >= 11 && cfg <= 15) || cfg == kCfgHs192) {   // half-slab pipeline: whole tiles only (the epilogue has no bounds checks), one pass over K
		BLA_REQUIRE(m % c.bm == 0 && n % c.bn == 0 && k >= 2 * c.bk && ldc % 4 == 0 && (uintptr_t)C % 16 == 0, BLA_ERR_INVALID,
		            "gemm config %d (%s) needs m, n multiples of the tile, k >= %d and a 16-byte aligned C", cfg, c.name, 2 * c.bk);
		a.rc_global = 0;   // buffer_load ... lds for every operand: 149.7 vs 146.6 TFLOP/s on NN 4096^3 in this kernel
		BLA_REQUIRE(!a.bias_row && !a.bias_col && !a.pre_act && a.act == BLA_ACT_NONE && !a.relu_mask && a.beta == 0.f && !deferred_row_sum, BLA_ERR_INVALID,
		            "gemm config %d (%s) takes a plain epilogue (alpha only)", cfg, c.name);
		splits = 1;
	}
	if (cfg == 10) {   // persistent variant: one pass over K per tile, slab parity must restart with every tile
		BLA_REQUIRE(k % (2 * c.bk) == 0, BLA_ERR_INVALID, "gemm config 10 (%s) needs k %% %d == 0", c.name, 2 * c.bk);
		splits = 1;
	}
	int kps = (k + splits - 1) / splits;
	kps = (kps + c.bk - 1) / c.bk * c.bk;
	if (kps == 0) kps = c.bk;
	splits = k > 0 ? (k + kps - 1) / kps : 1;
	a.k_per_split = kps;
	a.splits = splits;
	a.slab = nullptr;
	if (splits > 1) {
		void* ws;
		st = ensure_workspace((size_t)splits * m * n * sizeof(float), &ws);
		if (st) return st;
		a.slab = (float*)ws;
	}
	int mode = LOAD_SCALAR;
	if (vec_ok) {
		mode = LOAD_VEC;
		if (m % c.bm == 0 && n % c.bn == 0 && k % c.bk == 0) mode = LOAD_FULL;   // kps is a multiple of bk, so every slab is whole
	}
	dim3 grid((unsigned)tiles, 1, (unsigned)splits);
	hipError_t e;
	switch (cfg) {
		case 0: e = launch_variant<128, 128, 16, 2, 2>(a, akc, bkc, mode, grid, s); break;
		case 1: e = launch_variant<64, 64, 16, 2, 2>(a, akc, bkc, mode, grid, s); break;
		case 2: e = launch_variant<128, 128, 32, 2, 2>(a, akc, bkc, mode, grid, s); break;
		case 3: e = launch_glds<128, 128, 16, 2, 2>(a, akc, bkc, grid, s); break;
		case 4: e = launch_glds<64, 64, 16, 2, 2>(a, akc, bkc, grid, s); break;
		case 5: e = launch_glds<128, 128, 32, 2, 2>(a, akc, bkc, grid, s); break;
		case 7: e = launch_glds<128, 64, 16, 2, 2>(a, akc, bkc, grid, s); break;
		case 8: e = launch_glds<128, 256, 16, 2, 2, 2, 3>(a, akc, bkc, grid, s); break;
		case 9: e = launch_glds<256, 128, 16, 2, 2, 2, 3>(a, akc, bkc, grid, s); break;
		case 10: {
			unsigned tiles = grid.x, cap = 2u * (unsigned)cus;
			e = launch_glds<128, 128, 16, 2, 2, 1, 2, true>(a, akc, bkc, dim3(tiles < cap ? tiles : cap, 1, 1), s);
		} break;
		case 11: e = launch_glds<256, 256, 16, 2, 2>(a, akc, bkc, grid, s); break;
		case 12: e = launch_glds<256, 256, 32, 2, 2>(a, akc, bkc, grid, s); break;
		case 13: e = launch_glds<128, 512, 16, 1, 4>(a, akc, bkc, grid, s); break;
		case 14: e = launch_glds<128, 128, 16, 2, 2, 1, 2, false, true>(a, akc, bkc, grid, s); break;
		case 15: e = launch_glds<128, 256, 16, 2, 2, 1, 2, false, true>(a, akc, bkc, grid, s); break;
		case 18: e = launch_glds<64, 64, 32, 2, 2, 1, 2, false, false, 2>(a, akc, bkc, grid, s); break;
		default: e = launch_glds<192, 192, 16, 2, 2, 1, 2, false, true>(a, akc, bkc, grid, s); break;   // 17
	}
	if (e != hipSuccess) return hip_fail(e, "gemm_f32_kernel launch");
	static const char* kModeName[] = {"full", "vec", "scalar"};
	snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_%s_%c%c_%s_splitk%d", c.name, transa ? 't' : 'n', transb ? 't' : 'n',
	         c.glds ? "dma" : kModeName[mode], splits);
	if (splits > 1) {
		e = launch_splitk_reduce(a, s);
		if (e != hipSuccess) return hip_fail(e, "gemm_splitk_reduce_kernel launch");
	}
	if (deferred_row_sum) return window_sum(stream, A, m, k, lda, deferred_row_sum);
	return BLA_OK;
}

extern "C" {

bla_status bla_gemm_f32(void* stream, int transa, int transb, int m, int n, int k,
                        const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        const bla_gemm_epilogue* ep) {
	return gemm_impl(stream, transa, transb, m, n, k, A, lda, B, ldb, C, ldc, ep, nullptr);
}

/* Two INDEPENDENT products (neither reads what the other writes) issued together.  When both are latency-bound shapes of the
 * wave-split-K kernel they share one launch and overlap; otherwise this is two bla_gemm_f32 calls. */
bla_status bla_gemm_pair_f32(void* stream, const bla_gemm_desc* p, const bla_gemm_desc* q) {
	BLA_REQUIRE(p && q, BLA_ERR_INVALID, "null descriptor");
	WskPlan pp = {}, pq = {};
	const bool try_pair = g_force_config < 0 && g_force_split <= 0;
	bla_status st = gemm_impl(stream, p->transa, p->transb, p->m, p->n, p->k, p->A, p->lda, p->B, p->ldb, p->C, p->ldc, p->ep, try_pair ? &pp : nullptr);
	if (st) return st;
	st = gemm_impl(stream, q->transa, q->transb, q->m, q->n, q->k, q->A, q->lda, q->B, q->ldb, q->C, q->ldc, q->ep, pp.valid ? &pq : nullptr);
	if (st) return st;
	hipStream_t s = pick_stream(stream);
	const int tp = pp.valid ? pp.a.tiles_m * pp.a.tiles_n : 0, tq = pq.valid ? pq.a.tiles_m * pq.a.tiles_n : 0;
	hipError_t e = hipSuccess;
	if (pp.valid && pq.valid && !pp.a.softmax_grad && !pq.a.softmax_grad) {
		// every layout combination has its instantiation (NT beside TN = dW_l beside dZ_{l-1}, NT beside NT = two weight gradients,
		// TN beside TN = the attention block's Q and K projections, ...)
		const dim3 grid((unsigned)(tp + tq));
		const int combo = (pp.akc ? 8 : 0) | (pp.bkc ? 4 : 0) | (pq.akc ? 2 : 0) | (pq.bkc ? 1 : 0);
#define BLA_PAIR_CASE(I) case I: hipLaunchKernelGGL((gemm_f32_wsk_pair_kernel<((I) & 8) != 0, ((I) & 4) != 0, ((I) & 2) != 0, ((I) & 1) != 0>), grid, dim3(256), 0, s, pp.a, pq.a, tp); break
		switch (combo) {
			BLA_PAIR_CASE(0); BLA_PAIR_CASE(1); BLA_PAIR_CASE(2); BLA_PAIR_CASE(3); BLA_PAIR_CASE(4); BLA_PAIR_CASE(5); BLA_PAIR_CASE(6); BLA_PAIR_CASE(7);
			BLA_PAIR_CASE(8); BLA_PAIR_CASE(9); BLA_PAIR_CASE(10); BLA_PAIR_CASE(11); BLA_PAIR_CASE(12); BLA_PAIR_CASE(13); BLA_PAIR_CASE(14); BLA_PAIR_CASE(15);
		}
#undef BLA_PAIR_CASE
		e = hipGetLastError();
		snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_wsk_pair_%c%c+%c%c_%dx%d+%dx%d", pp.akc ? 'n' : 't', pp.bkc ? 't' : 'n', pq.akc ? 'n' : 't',
		         pq.bkc ? 't' : 'n', tp, pp.a.wsk_tile, tq, pq.a.wsk_tile);
	} else {
		if (pp.valid) e = launch_wsk(pp.a, pp.akc, pp.bkc, true, true, dim3((unsigned)tp, 1), s);
		if (e == hipSuccess && pq.valid) e = launch_wsk(pq.a, pq.akc, pq.bkc, true, true, dim3((unsigned)tq, 1), s);
	}
	if (e != hipSuccess) return hip_fail(e, "gemm_f32_wsk pair launch");
	return BLA_OK;
}

/* C_i = op(A_i) op(B_i) for i < batch, operand i at base + i * stride (stride 0 = shared).  Latency-bound shapes (what a self-attention block over
 * a batch of images is made of) run as ONE launch of the wave-split-K kernel; anything else is issued set by set.  ep (alpha / beta, bias_row,
 * bias_col, act; pre_act at stride_pre) is shared by the sets. */
bla_status bla_gemm_batched_f32(void* stream, int transa, int transb, int m, int n, int k, const float* A, int lda, long stride_a, const float* B, int ldb,
                                long stride_b, float* C, int ldc, long stride_c, int batch, const bla_gemm_epilogue* ep, long stride_pre) {
	BLA_REQUIRE(batch >= 1, BLA_ERR_INVALID, "batch %d", batch);
	BLA_REQUIRE(!ep || (!ep->relu_mask && !ep->row_sum_a && !ep->softmax_grad), BLA_ERR_INVALID, "batched products take alpha / beta, the biases, act and pre_act only");
	if (batch == 1) return gemm_impl(stream, transa, transb, m, n, k, A, lda, B, ldb, C, ldc, ep, nullptr);
	WskPlan pl = {};
	const bool aligned = stride_a % 4 == 0 && stride_b % 4 == 0 && stride_c % 4 == 0 && stride_pre % 4 == 0;
	t_plan_batch = batch;
	bla_status st = gemm_impl(stream, transa, transb, m, n, k, A, lda, B, ldb, C, ldc, ep, g_force_config < 0 && g_force_split <= 0 && aligned ? &pl : nullptr);
	t_plan_batch = 1;
	if (st) return st;
	if (pl.valid) {
		hipStream_t s = pick_stream(stream);
		const dim3 grid((unsigned)(pl.a.tiles_m * pl.a.tiles_n), (unsigned)batch);
#define BLA_B(T, AK, BK_) hipLaunchKernelGGL((gemm_f32_wsk_batched_kernel<T, AK, BK_>), grid, dim3(256), 0, s, pl.a, stride_a, stride_b, stride_c, stride_pre)
#define BLA_BT(AK, BK_) do { if (pl.a.wsk_tile == 16) BLA_B(16, AK, BK_); else BLA_B(32, AK, BK_); } while (0)
		if (pl.akc && !pl.bkc) BLA_BT(true, false);
		else if (pl.akc && pl.bkc) BLA_BT(true, true);
		else if (!pl.akc && !pl.bkc) BLA_BT(false, false);
		else BLA_BT(false, true);
#undef BLA_BT
#undef BLA_B
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return hip_fail(e, "gemm_f32_wsk_batched_kernel launch");
		snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_wsk_batched_%c%c_%dx%d_x%d", transa ? 't' : 'n', transb ? 't' : 'n', pl.a.tiles_m * pl.a.tiles_n, pl.a.wsk_tile, batch);
		return BLA_OK;
	}
	// (the first set has been issued by the planning call above)
	for (int i = 1; i < batch; i++) {
		bla_gemm_epilogue e2;
		if (ep) { e2 = *ep; if (e2.pre_act) e2.pre_act += (size_t)i * stride_pre; }
		st = gemm_impl(stream, transa, transb, m, n, k, A + (size_t)i * stride_a, lda, B + (size_t)i * stride_b, ldb, C + (size_t)i * stride_c, ldc, ep ? &e2 : nullptr, nullptr);
		if (st) return st;
	}
	return BLA_OK;
}

/* Up to three INDEPENDENT products issued together (none reads what another writes).  Three latency-bound products whose operands are all
 * K-contiguous (transa = 0, transb = 1) share ONE launch; two go through bla_gemm_pair_f32; anything else is issued one by one. */
bla_status bla_gemm_group_f32(void* stream, const bla_gemm_desc* d, int count) {
	BLA_REQUIRE(d && count >= 1 && count <= 3, BLA_ERR_INVALID, "1 to 3 descriptors");
	if (count == 1) return gemm_impl(stream, d[0].transa, d[0].transb, d[0].m, d[0].n, d[0].k, d[0].A, d[0].lda, d[0].B, d[0].ldb, d[0].C, d[0].ldc, d[0].ep, nullptr);
	if (count == 2) return bla_gemm_pair_f32(stream, &d[0], &d[1]);
	WskPlan pl[3] = {};
	const bool try_group = g_force_config < 0 && g_force_split <= 0;
	bla_status st;
	int planned = 0;
	for (int i = 0; i < 3; i++) {
		// a product that does not resolve to the un-split vectorised wave-split-K kernel launches right here (pl[i].valid stays false)
		st = gemm_impl(stream, d[i].transa, d[i].transb, d[i].m, d[i].n, d[i].k, d[i].A, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].ldc, d[i].ep,
		               try_group && !d[i].transa && d[i].transb ? &pl[i] : nullptr);
		if (st) return st;
		planned += pl[i].valid && !pl[i].a.softmax_grad ? 1 : 0;
	}
	hipStream_t s = pick_stream(stream);
	hipError_t e = hipSuccess;
	if (planned == 3) {
		const int t1 = pl[0].a.tiles_m * pl[0].a.tiles_n, t2 = t1 + pl[1].a.tiles_m * pl[1].a.tiles_n, t3 = t2 + pl[2].a.tiles_m * pl[2].a.tiles_n;
		hipLaunchKernelGGL(gemm_f32_wsk_triple_nt_kernel, dim3((unsigned)t3), dim3(256), 0, s, pl[0].a, pl[1].a, pl[2].a, t1, t2);
		e = hipGetLastError();
		snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_wsk_triple_nt_%dx%d+%dx%d+%dx%d", t1, pl[0].a.wsk_tile, t2 - t1, pl[1].a.wsk_tile, t3 - t2, pl[2].a.wsk_tile);
	} else {
		for (int i = 0; i < 3 && e == hipSuccess; i++)
			if (pl[i].valid) e = launch_wsk(pl[i].a, pl[i].akc, pl[i].bkc, true, true, dim3((unsigned)(pl[i].a.tiles_m * pl[i].a.tiles_n), 1), s);
	}
	if (e != hipSuccess) return hip_fail(e, "gemm_f32_wsk group launch");
	return BLA_OK;
}

}  // extern "C"
