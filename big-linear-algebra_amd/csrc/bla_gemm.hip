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

#include "bla_gemm_kernel.h"

namespace bla {

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
__global__ void __launch_bounds__(256) gemm_f32_wsk_pair_kernel(GemmArgs p, GemmArgs q, int tiles_p, const DoneHook* done) {
	__shared__ __attribute__((aligned(16))) WskShared sh;
	if ((int)blockIdx.x < tiles_p) {
		if (p.wsk_tile == 16) wsk_body<16, AKC1, BKC1, true, true>(p, (int)blockIdx.x, 0, sh);
		else wsk_body<32, AKC1, BKC1, true, true>(p, (int)blockIdx.x, 0, sh);
	} else {
		if (q.wsk_tile == 16) wsk_body<16, AKC2, BKC2, true, true>(q, (int)blockIdx.x - tiles_p, 0, sh);
		else wsk_body<32, AKC2, BKC2, true, true>(q, (int)blockIdx.x - tiles_p, 0, sh);
	}
	if (done) {   // the last gradient kernel of a data-parallel step: the workgroup that finishes last tells the peers (DoneHook, bla_internal.h)
		__shared__ int s_last;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are acknowledged
		__syncthreads();
		if (threadIdx.x == 0) {
			__threadfence_system();                          // ... and written back: the peers read over xGMI from memory
			s_last = __hip_atomic_fetch_add(done->arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
		}
		__syncthreads();
		if (s_last) {
			if (threadIdx.x == 0) *done->arrive = 0;
			const unsigned epoch = *done->epoch + 1;      // (the previous exchange finished before this launch: stream order)
			if ((int)threadIdx.x < done->world && (int)threadIdx.x != done->rank)
				__hip_atomic_store(done->peer_flags[threadIdx.x] + done->rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
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
__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(GemmArgs p) {
	size_t total = (size_t)p.M * p.N;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		float s = 0.f;
		for (int z = 0; z < p.splits; z++) s += p.slab[(size_t)z * total + i];
		epilogue_store(p, (int)(i / p.N), (int)(i % p.N), s);
	}
}

hipError_t launch_splitk_reduce(const GemmArgs& r, hipStream_t s) {
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
			// (128-row products with a long N -- the staged convolution's im2col product, 128 x 65536 x 1152 -- run 8-10 % faster on the half-slab 128 x 128 tile
			// than on config 3 and another 1-2 % on 128 x 256: 168.6 / 154.9 / 153.3 us, tools/gemm_rect_probe.py; hence 14 a hair above 3 and 15 in the list)
			const int cand[9] = {3, 7, 4, 11, 13, 14, kCfgHs192, 18, 15};
			const double eff[9] = {1.0, 0.95, 0.87, 1.03, 1.03, 1.005, 1.03, 0.93, 1.02};
			// Half-slab pipeline (configs 11, 13, 14, 17): one workgroup per CU by design (no x 0.88), whole tiles, plain epilogue, 16-byte aligned C
			// only.  The tile is picked so that the tile count is a whole number of rounds over the CUs: 4096^2 = 256 tiles of 256x256, 3072^2 = 256
			// of 192x192 (147.6 TFLOP/s against 128 on 64x64 tiles), 2048^2 = 256 of 128x128 (134.6 against 125.7 on 128x64).
			const bool big_ok = k >= 32 && ldc % 4 == 0 && (uintptr_t)C % 16 == 0 && !a.bias_row && !a.bias_col &&
			                    !a.pre_act && a.act == BLA_ACT_NONE && !a.relu_mask && a.beta == 0.f && !a.row_sum_a;
			double best = 0;
			for (int i = 0; i < 9; i++) {
				const Config& cc = kConfigs[cand[i]];
				const bool hs_tile = (i >= 3 && i < 7) || i == 8;
				if (hs_tile && !(big_ok && m % cc.bm == 0 && n % cc.bn == 0)) continue;
				long t = (long)((m + cc.bm - 1) / cc.bm) * ((n + cc.bn - 1) / cc.bn);
				// 64x64 tiles with two wave groups along K (config 18): the small problems where a CU holds one or two tiles (1024^3: 19.8 against 20.6 us,
				// 1280^3: 41.9 against 45.3, 1536^3: 71 against 77); needs whole 32-deep slabs
				if (i == 7 && !(k % 32 == 0 && t <= 3L * cus)) continue;
				long rounds = (t + cus - 1) / cus;
				double cost = (double)rounds * cc.bm * cc.bn / eff[i];
				if (t < 2L * cus && (i < 3 || i == 7)) cost /= 0.88;
				if (t < cus && i < 3) cost *= 1.25;   // ... and will have its K cut over workgroups: slabs and a fold launch (1280^3 on 128x64 tiles: 51 us)
				if (hs_tile && t < cus) cost *= 2;   // a partly filled chip: leave it to the smaller tiles / split-K
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
bla_status bla_gemm_pair_f32(void* stream, const bla_gemm_desc* p, const bla_gemm_desc* q) { return gemm_pair_with_hook(stream, p, q, nullptr, nullptr); }

}  // extern "C"

bla_status bla::gemm_pair_with_hook(void* stream, const bla_gemm_desc* p, const bla_gemm_desc* q, const DoneHook* d_hook, bool* posted) {
	if (posted) *posted = false;
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
#define BLA_PAIR_CASE(I) case I: hipLaunchKernelGGL((gemm_f32_wsk_pair_kernel<((I) & 8) != 0, ((I) & 4) != 0, ((I) & 2) != 0, ((I) & 1) != 0>), grid, dim3(256), 0, s, pp.a, pq.a, tp, d_hook); break
		switch (combo) {
			BLA_PAIR_CASE(0); BLA_PAIR_CASE(1); BLA_PAIR_CASE(2); BLA_PAIR_CASE(3); BLA_PAIR_CASE(4); BLA_PAIR_CASE(5); BLA_PAIR_CASE(6); BLA_PAIR_CASE(7);
			BLA_PAIR_CASE(8); BLA_PAIR_CASE(9); BLA_PAIR_CASE(10); BLA_PAIR_CASE(11); BLA_PAIR_CASE(12); BLA_PAIR_CASE(13); BLA_PAIR_CASE(14); BLA_PAIR_CASE(15);
		}
#undef BLA_PAIR_CASE
		e = hipGetLastError();
		if (posted && e == hipSuccess) *posted = d_hook != nullptr;
		snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_wsk_pair_%c%c+%c%c_%dx%d+%dx%d", pp.akc ? 'n' : 't', pp.bkc ? 't' : 'n', pq.akc ? 'n' : 't',
		         pq.bkc ? 't' : 'n', tp, pp.a.wsk_tile, tq, pq.a.wsk_tile);
	} else {
		if (pp.valid) e = launch_wsk(pp.a, pp.akc, pp.bkc, true, true, dim3((unsigned)tp, 1), s);
		if (e == hipSuccess && pq.valid) e = launch_wsk(pq.a, pq.akc, pq.bkc, true, true, dim3((unsigned)tq, 1), s);
	}
	if (e != hipSuccess) return hip_fail(e, "gemm_f32_wsk pair launch");
	return BLA_OK;
}

extern "C" {

/* C_i = op(A_i) op(B_i) for i < batch, operand i at base + i * stride (stride 0 = shared).  Latency-bound shapes (what a self-attention block over
 * a batch of images is made of) run as ONE launch of the wave-split-K kernel; anything else is issued set by set.  ep (alpha / beta, bias_row,
 * bias_col, act; pre_act at stride_pre) is shared by the sets. */
bla_status bla_gemm_batched_f32(void* stream, int transa, int transb, int m, int n, int k, const float* A, int lda, long stride_a, const float* B, int ldb,
                                long stride_b, float* C, int ldc, long stride_c, int batch, const bla_gemm_epilogue* ep, long stride_pre) {
	BLA_REQUIRE(batch >= 1, BLA_ERR_INVALID, "batch %d", batch);
	BLA_REQUIRE(!ep || (!ep->relu_mask && !ep->row_sum_a && !ep->softmax_grad), BLA_ERR_INVALID, "batched products take alpha / beta, the biases, act and pre_act only");
	if (batch == 1) return gemm_impl(stream, transa, transb, m, n, k, A, lda, B, ldb, C, ldc, ep, nullptr);
	if (g_force_config < 0 && g_force_split <= 0 && gemm_thin_applies(m, n, k, batch) && (!ep || (!ep->bias_col && ep->act == BLA_ACT_NONE))) {
		// short contraction, large output (the attention block's S x S products): HBM-bound, one wave per output block (bla_gemm_thin.hip)
		BLA_REQUIRE(A && B && C, BLA_ERR_INVALID, "null operand");
		const ThinPart part = {A, B, stride_a, stride_b, lda, ldb};
		bla_status st = gemm_thin_parts(stream, transa, transb, m, n, k, &part, 1, C, ldc, stride_c, batch, ep ? ep->alpha : 1.f, ep ? ep->beta : 0.f,
		                                ep ? ep->bias_row : nullptr, ep ? ep->pre_act : nullptr, ep ? ep->ld_pre : 0, stride_pre);
		if (st == BLA_OK) snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_thin_%c%c_x%d", transa ? 't' : 'n', transb ? 't' : 'n', batch);
		return st;
	}
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
