// bla_gemm_kernel.h -- the direct-to-LDS fp32 MFMA GEMM kernel (gemm_f32_glds_kernel) with its argument block, shared by the two translation
// units that instantiate it: bla_gemm.hip (dense products) and bla_gather.hip (implicit-GEMM convolution: the gathered-operand variants).
// Split so that the two sets of instantiations compile side by side.
#pragma once
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
	//   mode 7 (forward / data gradient, 3x3, stride 1, image rows of 16 or 32 pixels, straight from the image): "im2col into LDS".  A tile is 128
	//           consecutive pixels = 128 / W whole rows of ONE image.  The contraction runs k = (g * 9 + t) * 16 + c: channel group g (16 channels),
	//           tap t, channel c of the group -- so the 9 slabs of a group all read the SAME 16 channel planes, and what sits in LDS is not the
	//           im2col slab [16][128] but the image window itself: per channel the 128 / W + 2 rows the tile's taps touch (rows outside the image
	//           from a zero word), 16 planes per group, two groups (double buffer).  The window of group g + 1 is fetched by 16-byte DMAs (ALIGNED:
	//           whole image rows) during the first slabs of group g -- nine slabs ahead of its use, one ninth of the im2col's bytes plus the halo
	//           rows -- and an MFMA's B fragment is read at plane c, row r + p, column x + q - 1; a lane at a row's end gets its out-of-row tap as a
	//           zero by a select.  No padded copy, no gather table, nothing per slab but one address add.  A [M][(g, t, c)] is re-ordered by the host.
	const float* g_img; const float* g_zero;
	const int2* g_ktab; const int2* g_ntab;   // {element offset, y | x << 16} per tap / per output pixel
	int g_mode, g_H, g_W, g_HWo, g_img_stride;
	int g_wo;                                 // mode 4 on the half-slab pipeline: width of the output map (g_W is the padded copy's row pitch there)
	// mode 3 on the half-slab pipeline, one pass over K: the adds the U-Net puts behind a convolution, applied where the tile is stored
	// (out = product + g_bias[image * g_bias_stride + row]; g_out2 = out + g_add, same layout as out; each optional)
	const float* g_bias; int g_bias_stride; const float* g_add; float* g_out2;
	int rc_global;   // host-side only: pick the instantiation that fetches row-contiguous operands with global_load_lds
	int wsk_tile;    // wave-split-K kernels: 32 (32x32 tiles, MFMA 32x32x2) or 16 (16x16 tiles, MFMA 16x16x4)
	// mode 3, several products over the SAME gathered image in one launch (the parity classes of a stride-2 data gradient, bla_conv.hip): blockIdx.y =
	// product; each brings its own kernels, contraction length, tap table and output plane; M, N, the image and the pixel table are shared
	// (the table lives in DEVICE memory: arrays inside this by-value block, indexed by blockIdx.y, made hipcc copy the whole block to scratch)
	int g_ncls;
	const GatherClass* g_cls;
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

// XCD-aware tile id: hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so give
// XCD x the contiguous chunk [x*q, (x+1)*q) of the (grouped) tile order.  Bijective for any count.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
	int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
	return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
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
// mode 7: 16-byte chunk q of a channel group's image window ([16 planes][rows][4 zeros + GW]) -> element offset in the image batch (channel group 0),
// -1 when it is a row's leading zero chunk, when its image row does not exist, or when it lies in the tail that rounds a group up to whole DMA instructions
// (the chunk is then fetched from the zero words).  The zero chunk in front of every row is both the column left of that row and the column right of the row
// before it: the taps dx = -1 / +1 of a row's first / last pixel read a zero from LDS, and the fragments need no masking (one v_cndmask per MFMA before).
template <int PL, int GW>
__device__ __forceinline__ int window_chunk_offset(int q, int i0, int img_h, int img_hw, int img_base) {
	constexpr int CPR = GW / 4 + 1;
	const int plane = q / (PL / 4), r = q - plane * (PL / 4), wrow = r / CPR, c4 = r - wrow * CPR;
	const int row = i0 - 1 + wrow;
	return (plane < 16 && c4 > 0 && (unsigned)row < (unsigned)img_h) ? img_base + plane * img_hw + row * GW + (c4 - 1) * 4 : -1;
}

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
          int WK = 1, int GW = 0>
__device__ __forceinline__ void gemm_f32_glds_body(GemmArgs p, const int blk_x, const int blk_y, const int blk_z, const int grid_x, const int grid_z) {
	// (blk_* / grid_*: this workgroup's place in ITS product's grid -- blockIdx / gridDim for a launch of one product, a slice of the linear grid when two
	// products share a launch, gather_pair_kernel below)
	static_assert(GATHER != 7 || (HS && (GW == 16 || GW == 32) && WM == 2 && WN == 2 && BN == 128), "mode 7: half-slab pipeline, image rows of 16 or 32 pixels");
	static_assert(GATHER == 0 || (AKC && BKC == (GATHER == 4) && NBUF == 2 && !PERSIST && BM == 128 && (BN == 128 || (BN == 256 && GATHER == 3 && HS)) && BK == 16 && WM * WN == 4),
	              "gather variants: A K-contiguous; modes 1-3 gather B as a [16][128] image (mode 3 on the half-slab pipeline: [16][256] too), mode 4 gathers A and takes a K-contiguous B");
	// mode 3: one wave-instruction of the B image (1 KiB = 256 floats) covers G3_RPI k-rows of BN / 4 sixteen-byte chunks each
	constexpr int G3_CPR = BN / 4, G3_RPI = 64 / (G3_CPR < 64 ? G3_CPR : 64);
	constexpr int NW = WM * WN * WK;
	constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
	constexpr bool G7 = GATHER == 7;
	constexpr int GWS = GW ? GW : 32;   // (a valid divisor in the instantiations that have no image width)
	constexpr int A_SZ = BM * BK, B_SZ = G7 ? 0 : BN * BK, KK = BK / 8, KKW = KK / WK;   // KKW: k-parts of a slab this wave multiplies
	// mode 7: the image window in LDS behind the two A slab buffers -- [2 groups][16 planes][W7_RH rows][4 zeros + GW] floats (a group rounded up to whole
	// 1-KiB DMA instructions, the tail zeros too), 16 bytes of slack before and after
	constexpr int W7_RH = G7 ? 128 / GWS + 2 : 1, W7_PITCH = GWS + 4, W7_PL = W7_RH * W7_PITCH, W7_GRP = (16 * W7_PL + 255) / 256 * 256, W7_NI = W7_GRP / 256, W7_PW = (W7_NI + 3) / 4;
	static_assert(!G7 || (W7_GRP > 16 * W7_PL && W7_PW <= 9), "the float behind a group's last row is a zero of its own tail; the next group's window arrives within the nine slabs of a group");
	static_assert(WK == 1 || (KK % WK == 0 && NBUF == 2 && !PERSIST && GATHER == 0 && !HS && !(TM == 4 && TN == 4)), "waves along K: the plain two-buffer pipeline only");
	typedef KcImage<BM, BK> AI;
	typedef KcImage<BN, BK> BI;
	extern __shared__ __attribute__((aligned(16))) float lds[];  // [NBUF][A_SZ + B_SZ], all LDS in this one array

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, h = lane >> 5;
	const int wk = wave / (WM * WN), wsp = wave % (WM * WN);   // group along K, position in the tile
	const int wm0 = (wsp / WN) * (BM / WM), wn0 = (wsp % WN) * (BN / WN);

	if (GATHER == 3 && HS && p.g_ncls > 1) {
		const GatherClass c = p.g_cls[blk_y];
		p.A = c.A; p.C = c.C; p.g_ktab = c.ktab; p.K = c.K; p.lda = c.K;
	}
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
	// Weight gradient (mode 4: few tiles, K cut over many workgroups): every tile of one K-split reads the same del_y columns, and the hardware
	// deals consecutive workgroup ids round-robin to the 8 XCDs -- with (tile, split) = (blockIdx.x, blockIdx.z) the tiles of a split land on
	// different XCDs and each L2 fetches that split's del_y for itself (349 MB of fills for 68 MB of operands at 128->128 @32x32 x64).  So XCD x takes
	// the splits s = x (mod 8) and walks them tile by tile: the tiles of a split are neighbours in ONE XCD's dispatch order and share its L2.
	int vblock = blk_x, zsplit = blk_z;
	if (GATHER == 4 && (grid_z & 7) == 0) {
		const int lin = blk_z * grid_x + blk_x, i = lin >> 3;
		zsplit = (i / grid_x) * 8 + (lin & 7); vblock = i % grid_x;
	}
	tile_origin(vblock, m0, n0);
	const int k_begin = zsplit * p.k_per_split;
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
		if ((GATHER >= 1 && GATHER <= 3) || G7) {
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
	// mode 4, half-slab pipeline: a slab is 16 consecutive output pixels starting at a multiple of 16, and the map is 4, 8 or a multiple of 16 pixels wide
	// (the host checks), so where a lane's chunk of four pixels sits RELATIVE to the slab's first pixel in the padded copy never changes: the lane's byte
	// offset (tap row + that place) is fixed for the whole K loop and goes into the buffer load's VGPR offset, the slab's first pixel (image, row, column:
	// wave-uniform) into its SGPR offset -- no table load, no select and no 64-bit address per DMA instruction (before: four scalar loads per slab, three
	// v_cndmask and a 64-bit add per instruction; the counters showed 0.7 VALU instructions per MFMA beside the MFMAs themselves).
	int g4_voff[A_NI], g4_pix0 = 0, g4_col = 0, g4_soff = 0, g4_step = 0, g4_step_end = 0;
	if (GATHER == 4 && HS) {
		const int wo = p.g_wo, wh = p.g_W;
#pragma unroll
		for (int i = 0; i < A_NI; i++) g4_voff[i] = (g4_tap[i] + (g4_chunk[i] / wo) * wh + g4_chunk[i] % wo) * 4;
		g4_col = g_r % wo; g4_pix0 = (g_r / wo) * wh + g4_col;
		g4_step = wo >= 16 ? 16 : (16 / wo) * wh;                  // to the next slab inside a row / over whole rows
		g4_step_end = wo >= 16 ? wh - wo + 16 : g4_step;            // from a row's last slab to the next row's first
		g4_soff = (g_img * p.g_img_stride + g4_pix0) * 4;
	}
	int g3_base = 0;              // mode 3: padded-image offset of this lane's four columns
	if (GATHER == 3) {
		int n = min(n0 + (lane % G3_CPR) * 4, p.N - 4);
		int b = n / p.g_HWo, r = n - b * p.g_HWo;
		g3_base = p.g_ntab[r].x + b * p.g_img_stride;
	}
	// mode 7: this wave's window DMA instructions jw = wave + 4 i of a group -- chunk q = 64 jw + lane of the group's 16 planes: plane q / (PL/4), window
	// row, 16-byte column -- as an element offset into the image batch (group 0; a group adds 16 planes) and whether that image row exists; the
	// fragment reads' lane offset; whether the lane's pixel column is a row's first / last
	// (the chunk's offset is worked out when its instruction is issued -- three times per nine slabs: kept in three registers and picked by the run-time
	// tap, hipcc turns the pick into a scratch array, and with it the whole argument block: 720 bytes of scratch per lane, seen in the ISA)
	int w7_i0 = 0, w7_base = 0;
	unsigned w7_lane = 0;
	if (G7) {
		const int b = n0 / p.g_HWo, i0 = (n0 - b * p.g_HWo) / GWS;    // the tile: rows i0 .. i0 + 128 / GW - 1 of image b
		w7_i0 = i0; w7_base = b * p.g_img_stride;
		w7_lane = (unsigned)((4 * h * W7_PL + ((wn0 + l31) / GWS) * W7_PITCH + 4 + (wn0 + l31) % GWS) * 4);
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
	constexpr bool A_BUF = (BUF_OK && (AKC || !RCG)) || GATHER == 3 || G7, B_BUF = (BUF_OK && (BKC || !RCG)) || GATHER == 4;   // padded-copy conv modes: their dense operand too
#if defined(__HIP_DEVICE_COMPILE__)
	// raw descriptors, no bounds (rows / columns past the matrix are fetched from clamped offsets); lane offsets in bytes
	__amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0x7fffffff, 0x00020000);
	__amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, 0x7fffffff, 0x00020000);
	__amdgpu_buffer_rsrc_t rsrc_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATHER == 4 ? p.g_img : p.A), 0, 0x7fffffff, 0x00020000);
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
	                          (GATHER == 0 || ((GATHER == 3 || GATHER == 4 || G7) && HS)) && !PERSIST && NBUF == 2;
	constexpr int NDMA = G7 ? A_NI + 1 : A_NI + B_NI;   // DMA instructions per wave per slab (8 at BK = 16, 16 at BK = 32; mode 7: A and at most one window instruction)
	size_t g_adv_a = 0, g_adv_b = 0;   // global-form operands of the half-slab pipeline: scalar advance added to the per-lane pointers
	int w7_tap = 0, w7_g = 0;                                    // mode 7: tap and channel group of the slab being computed
	const int w7_groups = G7 ? (p.g_img_stride / p.g_HWo) / 16 : 0, w7_gstep = 16 * p.g_HWo;
	// (plain values, not members of p: a select between p.g_img + x and p.g_zero hipcc rewrites as a load through a selected ADDRESS of the member,
	// which puts the whole argument block into scratch -- 720 bytes per lane, seen in the ISA)
	const float* const w7_img = p.g_img;
	const float* const w7_zero = p.g_zero;
#if defined(__HIP_DEVICE_COMPILE__)
	// Gather tables of the half-slab pipeline: the entries a slab's DMA instructions need are wave-uniform (tap rows of mode 3, the four
	// pixel chunks of mode 4), so they are SCALAR loads, issued when the cursor moves -- a whole slab before the DMA that uses them.  (A per-lane
	// table load in front of each DMA waits on vmcnt, i.e. for every LDS-DMA issued before it: 256->256 @16x16 ran 11 % slower that way.)
	int hs_tap[B_NI][2];
	auto hs_prefetch = [&]() {
		if (GATHER == 3) {
#pragma unroll
			for (int i = 0; i < B_NI; i++) {
				const int row = __builtin_amdgcn_readfirstlane(g_k + (wave * B_NI + i) * G3_RPI);
				hs_tap[i][0] = p.g_ktab[row].x; hs_tap[i][1] = G3_RPI == 2 ? p.g_ktab[row + 1].x : 0;
			}
		}
	};
	if (HALFSLAB && GATHER == 3) hs_prefetch();

	auto dma_one = [&](int buf, int d) {   // d-th DMA instruction of a slab; the offsets / gather cursors advance in dma_advance()
		float* base = lds + buf * (A_SZ + B_SZ);
		if (d < A_NI) {
			const int i = d;
			if (GATHER == 4) {   // gathered operand: 16-byte chunk of four pixels of this lane's tap row (padded image copy): fixed lane offset + the slab's scalar offset
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_img, (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, g4_voff[i], g4_soff, 0, 0);
			} else if (A_BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, voff_a[i], soff_a, 0, 0);
			else __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ga[i] + g_adv_a), (lds_ptr_t)(base + (wave * A_NI + i) * 256), 16, 0, 0);
		} else if (G7) {
			// the one window slot of a slab: instruction `w7_tap` of the NEXT group's window, in the first W7_PW slabs of a group (its buffer held the
			// group before this one, which nobody reads any more) -- nine slabs ahead of its first use
			if (w7_tap < W7_PW && wave + 4 * w7_tap < W7_NI && w7_g + 1 < w7_groups) {
				const int i = w7_tap;
				const int off = window_chunk_offset<W7_PL, GWS>((wave + 4 * i) * 64 + lane, w7_i0, p.g_H, p.g_HWo, w7_base);
				const float* src = off >= 0 ? w7_img + (off + (w7_g + 1) * w7_gstep) : w7_zero;
				float* dst = lds + 2 * A_SZ + 4 + ((w7_g + 1) & 1) * W7_GRP + (wave + 4 * i) * 256;
				__builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
			}
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
		if (G7) { soff_a += sa; return; }
		if (GATHER == 4) {   // B = del_y [image][N][HWo]: the pixel cursor wraps into the next image (HWo % 16 == 0: a slab never straddles two)
			const int r1 = g_r + (really ? BK : 0);
			const bool wrap = r1 >= p.g_HWo;
			soff_b += (really ? BK * 4 : 0) + (wrap ? (p.N - 1) * p.g_HWo * 4 : 0);
			g_r = wrap ? 0 : r1; g_img += wrap ? 1 : 0;
			const int c1 = g4_col + 16;
			const bool row_end = c1 >= p.g_wo;
			g4_pix0 = wrap ? 0 : g4_pix0 + (really ? (row_end ? g4_step_end : g4_step) : 0);
			g4_col = (wrap || row_end) ? 0 : (really ? c1 : g4_col);
			g4_soff = __builtin_amdgcn_readfirstlane((g_img * p.g_img_stride + g4_pix0) * 4);
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
			float* dst = p.splits > 1 ? p.slab + (size_t)zsplit * p.M * p.N : p.C;
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
		if constexpr (G7) {   // whole tiles inside one image; block `in` owns the 32 consecutive pixels wn0 + 32 in + l31
			const int b = n0 / p.g_HWo;
			const size_t img_off = (size_t)b * p.M * p.g_HWo + (n0 - b * p.g_HWo) + wn0 + l31;
			const float* bias = p.g_bias ? p.g_bias + (size_t)b * p.g_bias_stride : nullptr;
#pragma unroll
			for (int im = 0; im < TM; im++)
#pragma unroll
				for (int r = 0; r < 16; r++) {
					const int row = m0 + wm0 + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
					const size_t o = img_off + (size_t)row * p.g_HWo;
					const float bv = bias ? bias[row] : 0.f;
#pragma unroll
					for (int in = 0; in < TN; in++) {
						const float v = acc[im][in][r] + bv;
						p.C[o + in * 32] = v;
						if (p.g_out2) p.g_out2[o + in * 32] = v + p.g_add[o + in * 32];
					}
				}
			return;
		}
		if constexpr (GATHER == 3 && HALFSLAB) {   // whole tiles; a lane owns TN consecutive columns (the row-contiguous operand's interleaved blocks)
			const int col = n0 + wn0 + TN * l31;              // TN consecutive pixels of one image (HWo % 4 == 0)
			const int b = col / p.g_HWo, rr = col - b * p.g_HWo;
			const size_t img_off = (size_t)b * p.M * p.g_HWo + rr;
			float* cbase = (p.splits > 1 ? p.slab + (size_t)zsplit * p.M * p.N : p.C) + img_off;   // slabs are C-shaped
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
						if (p.splits > 1) p.slab[((size_t)zsplit * p.M + row) * p.N + col] = acc[im][in][r];
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
			                              // (mode 7 uses sb too: element [j][block] is one dword of the image window)
		};
		Frag P, Q;
		// Whole-slab form (FS) for the forward / data-gradient convolution kernels (64 x 64 per wave: 16 MFMAs per k-part): a wave that stops at every
		// k-part boundary for its fragments (land) AND at every slab for the barrier pays that fixed cost per 16 MFMAs -- a quarter of what the 256 x 256
		// kernel's 64-MFMA phases pay it for.  With 64 accumulator registers there is room for TWO whole-slab fragment sets: slab t + 1's fragments are
		// all read under slab t's 32 MFMAs, and a slab has ONE stop (land + vmcnt(0) + barrier, together) instead of three.
		constexpr bool FS = (GATHER == 3 || G7) && TM == 2 && TN == 2 && KK == 2;
		Frag F0[KK], F1[KK];
		constexpr int UA = AKC ? TM : 4, UB = G7 ? 4 * TN : BKC ? TN : 4, NU = UA + UB;   // fragment-read units per k-part
		// mode 7: byte address of (plane 0 of the group, tap) of the slab whose fragments are being read, plus this lane's part; [0] = slab t, [1] = slab t + 1
		unsigned w7_ad[2] = {0, 0}, w7_cur = 0;   // w7_cur: the A buffer offset of slab t (read_unit is told a slab by its A buffer)
		auto w7_addr = [&](int g, int tap) -> unsigned {   // tap (p, q): window row r + p, column x + q - 1
			const int pq = tap / 3;
			return lds0 + (unsigned)((2 * A_SZ + 4 + (g & 1) * W7_GRP + pq * W7_PITCH + (tap - 3 * pq) - 1) * 4) + w7_lane;
		};
		constexpr int NM = 4 * TM * TN;                                       // MFMAs per k-part
		auto opa = [&](const Frag& f, int im, int j) -> float { return AKC ? f.ka[im][j] : TM == 3 ? f.sa[j][im] : f.ra[j][im]; };
		auto opb = [&](const Frag& f, int in, int j) -> float { return G7 ? f.sb[j][in] : BKC ? f.kb[in][j] : TN == 3 ? f.sb[j][in] : f.rb[j][in]; };
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
			} else if constexpr (G7) {
				// unit x = (j, block): channel kk * 8 + 4 h + j of the group (the 4 h planes are in the lane's address), pixels block * 32 + l31 of the wave's 64;
				// `buf` is not a slab buffer here but which of the two slabs in flight (0: t, BUF_BYTES: t + 1)
				const int x = u - UA, j = x / TN, b = x % TN;
				const unsigned ad = buf == w7_cur ? w7_ad[0] : w7_ad[1];
				asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.sb[j][b]) : "v"(ad), "n"(((kk * 8 + j) * W7_PL + b * (32 / GWS) * W7_PITCH) * 4));
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
				if constexpr (G7) asm volatile("" : "+v"(f.sb[x / TN][x % TN]));
				else if constexpr (BKC) asm volatile("" : "+v"(f.kb[x]));
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
			for (int d = 0; d < (G7 ? A_NI : NDMA); d++) dma_one(buf, d);
			const bool adv = fetched + 1 < nkt;
			dma_advance(adv); fetched += adv ? 1 : 0;
		};
		auto slab = [&](int t) {
			const unsigned cur = (t & 1) * BUF_BYTES, nxt = ((t + 1) & 1) * BUF_BYTES;
			if constexpr (G7) {   // where slab t and slab t + 1 read the window
				const int tap1 = w7_tap == 8 ? 0 : w7_tap + 1, g1 = w7_tap == 8 ? w7_g + 1 : w7_g;
				w7_cur = cur; w7_ad[0] = w7_addr(w7_g, w7_tap); w7_ad[1] = w7_addr(g1, tap1);
			}
			// phases 0 .. KK-2: k-part q from one set while k-part q+1 of this slab is read into the other, the read units spread evenly
#pragma unroll
			for (int q = 0; q + 1 < KK; q++) {
				Frag& use = (q & 1) ? Q : P;
				Frag& fill = (q & 1) ? P : Q;
				land(use);
				// dense: the read units spread evenly over the phase.  Convolution modes (16 MFMAs per phase): one unit per MFMA from the start and the
				// rest of the MFMAs behind them -- spread evenly, the last unit is issued some 100 cycles before land() waits for it, less than an LDS round trip
				constexpr int FRONT = (GATHER != 0 && NU <= NM) ? 1 : 0;
#pragma unroll
				for (int u = 0; u < NU; u++) {
					read_unit(cur, q + 1, u, fill);
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int m = FRONT ? u : u * NM / NU; m < (FRONT ? (u + 1 < NU ? u + 1 : NM) : (u + 1) * NM / NU); m++) mf1(use, m);
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			// last phase (k-part KK-1 from Q): slab t+1 has landed for everyone, and everyone is done reading slab t
			land(Q);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			// The gather-table entries for the DMAs below were loaded a slab ago and have landed with everything else (vmcnt(0) above) -- but hipcc does not
			// know that wait, and puts its own s_waitcnt vmcnt(0) in front of their first use: BETWEEN this phase's DMA instructions, i.e. the wave would stop
			// until the DMAs it has just issued are back (the ISA showed it after the two A instructions of every slab).  Using the entries HERE makes hipcc put
			// that wait here, where it is free.
#if defined(__HIP_DEVICE_COMPILE__)
			if constexpr (GATHER == 3) {
#pragma unroll
				for (int i = 0; i < B_NI; i++) { asm volatile("" : "+v"(hs_tap[i][0])); if (G3_RPI == 2) asm volatile("" : "+v"(hs_tap[i][1])); }
			}
#endif
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
			if constexpr (G7) { const bool wrap = w7_tap == 8; w7_g += wrap ? 1 : 0; w7_tap = wrap ? 0 : w7_tap + 1; }   // on to slab t + 1
		};
		auto mf_fs = [&](Frag (&f)[KK], int idx) {   // idx-th MFMA of a slab: k-part major
			mf1(f[idx / NM], idx % NM);
		};
		auto slab_fs = [&](int t, Frag (&use)[KK], Frag (&fill)[KK]) {
			const unsigned nxt = ((t + 1) & 1) * BUF_BYTES;
			if constexpr (G7) {
				const int tap1 = w7_tap == 8 ? 0 : w7_tap + 1, g1 = w7_tap == 8 ? w7_g + 1 : w7_g;
				w7_cur = (t & 1) * BUF_BYTES; w7_ad[0] = w7_addr(w7_g, w7_tap); w7_ad[1] = w7_addr(g1, tap1);
			}
			// the one stop of the slab: this slab's fragments are in registers, slab t + 1 has landed for every wave, nobody needs slab t's LDS any more
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int kk = 0; kk < KK; kk++) land(use[kk]);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
#if defined(__HIP_DEVICE_COMPILE__)
			if constexpr (GATHER == 3) {   // (the table entries' wait, where it is free: see slab())
#pragma unroll
				for (int i = 0; i < B_NI; i++) { asm volatile("" : "+v"(hs_tap[i][0])); if (G3_RPI == 2) asm volatile("" : "+v"(hs_tap[i][1])); }
			}
#endif
			static_assert(!FS || KK * NU + NDMA <= KK * NM, "one memory instruction per MFMA");
			int idx = 0;
#pragma unroll
			for (int kk = 0; kk < KK; kk++)
#pragma unroll
				for (int u = 0; u < NU; u++) {   // slab t + 1 -> the other set (every LDS read before the first DMA)
					read_unit(nxt, kk, u, fill[kk]);
					__builtin_amdgcn_sched_barrier(0);
					mf_fs(use, idx++);
					__builtin_amdgcn_sched_barrier(0);
				}
#pragma unroll
			for (int d = 0; d < NDMA; d++) {     // slab t + 2 -> this slab's buffer
				dma_one(t & 1, d);
				__builtin_amdgcn_sched_barrier(0);
				mf_fs(use, idx++);
				__builtin_amdgcn_sched_barrier(0);
			}
			{ const bool adv = fetched + 1 < nkt; dma_advance(adv); fetched += adv ? 1 : 0; }
#pragma unroll
			for (int m = KK * NU + NDMA; m < KK * NM; m++) mf_fs(use, m);
			__builtin_amdgcn_sched_barrier(0);
			if constexpr (G7) { const bool wrap = w7_tap == 8; w7_g += wrap ? 1 : 0; w7_tap = wrap ? 0 : w7_tap + 1; }
		};
		if (nkt > 0) {
			if constexpr (G7) {   // the window of channel group 0, whole; group 1 follows during the first slabs
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
				for (int i = 0; i < W7_PW; i++)
					if (wave + 4 * i < W7_NI) {
						const int off = window_chunk_offset<W7_PL, GWS>((wave + 4 * i) * 64 + lane, w7_i0, p.g_H, p.g_HWo, w7_base);
						const float* src = off >= 0 ? w7_img + off : w7_zero;
						__builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds + 2 * A_SZ + 4 + (wave + 4 * i) * 256), 16, 0, 0);
					}
#endif
				w7_cur = 0; w7_ad[0] = w7_addr(0, 0);
			}
			fetch_slab(0);
			fetch_slab(1);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			if constexpr (FS) {
#pragma unroll
				for (int kk = 0; kk < KK; kk++)
#pragma unroll
					for (int u = 0; u < NU; u++) read_unit(0, kk, u, F0[kk]);
				int t = 0;
				for (; t + 1 < nkt; t += 2) { slab_fs(t, F0, F1); slab_fs(t + 1, F1, F0); }
				if (t < nkt) slab_fs(t, F0, F1);      // an odd number of slabs
			} else {
#pragma unroll
				for (int u = 0; u < NU; u++) read_unit(0, 0, u, P);
				for (int t = 0; t < nkt; t++) slab(t);
			}
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
			tile_origin(vb_next < total ? vb_next : blk_x, tm0, tn0);
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
		if (nkt > 0 && blk_x < total) {
			dma_next(0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
#pragma unroll
			for (int kk = 0; kk < KK; kk++) frags(lds, lds + A_SZ, kk, fa0[kk], fb0[kk]);
			dma_next(1);
			for (int vb = blk_x; vb < total; vb += grid_x) {
				tile_origin(vb, m0, n0);
				plan_next(vb + grid_x);
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

template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC, int MINW = 1, int NBUF = 2, bool PERSIST = false, int GATHER = 0, bool RCG = false, bool HS = false,
          int WK = 1, int GW = 0>
__global__ void __launch_bounds__(WM * WN * WK * 64, MINW) gemm_f32_glds_kernel(GemmArgs p) {
	gemm_f32_glds_body<BM, BN, BK, WM, WN, AKC, BKC, MINW, NBUF, PERSIST, GATHER, RCG, HS, WK, GW>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.z);
}

// Both gradients of one batched convolution in ONE launch: workgroups [0, blocks_w) run the weight gradient (gather mode 4; (tile, split) = (i % wx, i / wx)),
// the rest the data gradient (DG = gather mode 7 with rows of GW pixels, or mode 3; (tile, split) = (j % dx, j / dx)).  Neither reads what the other
// writes; dealt out in this order the data gradient's workgroups move in as the weight gradient's finish, so one product's prologue, drain and tail are
// covered by the other's body instead of standing alone on an idle chip (two contexts running them side by side measured 375 -> 331 us at 128 -> 128
// @32x32 x64; forking to a second stream per convolution cost as much in cross-stream waits as it gained).  Registers / LDS: the larger of the two.
template <int DG, int GW>
__global__ void __launch_bounds__(256, 1) gather_pair_kernel(GemmArgs w, GemmArgs d, int blocks_w, int wx, int wz, int dx, int dz) {
	const int b = (int)blockIdx.x;
	if (b < blocks_w) gemm_f32_glds_body<128, 128, 16, 2, 2, true, true, 1, 2, false, 4, false, true, 1, 0>(w, b % wx, 0, b / wx, wx, wz);
	else gemm_f32_glds_body<128, 128, 16, 2, 2, true, false, 1, 2, false, DG, false, true, 1, GW>(d, (b - blocks_w) % dx, 0, (b - blocks_w) / dx, dx, dz);
}

// (bla_gemm.hip) fold of split-K slabs: fixed order, epilogue applied
hipError_t launch_splitk_reduce(const GemmArgs& r, hipStream_t s);

}  // namespace bla
