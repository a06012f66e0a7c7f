// bla_gather.hip -- implicit-GEMM convolution on the direct-to-LDS GEMM pipeline: the host side of the gathered-operand variants of
// gemm_f32_glds_kernel (bla_gemm_kernel.h; gather modes in GemmArgs) -- tile / split planning and the launches.  Called from bla_conv.hip.
#include "bla_gemm_kernel.h"

namespace bla {

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
// mode 4 on the half-slab kernel: a slab's 16 output pixels must sit at the same places relative to its first pixel whatever the slab (bla_gemm_kernel.h)
static bool gather4_fixed_offsets(int wo) { return wo == 4 || wo == 8 || (wo >= 16 && wo % 16 == 0); }
static int gather_k_per_split(int mode, int batch, int M, int N, int HWo) {
	const long K = (long)batch * HWo;
	if (mode != 2 && mode != 4) return (int)K;
	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
	// two workgroups per CU on either form.  On the side lane (the U-Net's weight gradients beside the main chain's kernels) ONE: the half-slab weight-gradient
	// kernel takes 108 registers, so one of its workgroups fits a CU BESIDE two of the main chain's gather kernels (188 / 172 registers each) and fills the matrix
	// pipe's bubbles instead of taking one of their two places -- batch-64 backward 9.83 -> 9.33 ms, batch 128 18.7 -> 16.9 (BLA_LANE_SLOTS=2: the old split)
	static const long lane_slots = [] { const char* e = getenv("BLA_LANE_SLOTS"); return e && *e ? atol(e) : 1L; }();
	const long slots = (ctx().side_lane && mode == 4 ? lane_slots : 2L) * cus;
	long splits = slots / tiles;
	const long slabs = K / 16;
	if (splits > slabs / 8) splits = slabs / 8;      // at least 8 slabs per split
	if (splits >= 32) splits &= ~7L;                 // a multiple of 8 where that costs little: the kernel then keeps the tiles of a split on one XCD (shared L2 fetch of del_y)
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
// one pass over K: where the tile is stored; taps cut over workgroups (small maps): in the fold of the slabs (gather_fold_epilogue_kernel)
bool gather3_fuses_epilogue(int M, int N, int K) { (void)K; return gather_hs(3, M, N); }

// The fold of a mode-3 product whose taps were cut over workgroups, with the adds the U-Net puts behind the convolution: slabs [split][image][M][HWo]
// summed in split order, + bias[image * stride + row], second output = that + add.  16 bytes per thread.
__global__ void __launch_bounds__(256) gather_fold_epilogue_kernel(const float4* __restrict__ slab, float4* __restrict__ out, int splits, size_t total4, const float* __restrict__ bias,
                                                                    int bias_stride, const float4* __restrict__ add, float4* __restrict__ out2, int M, int hwo4) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
		float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
		for (int z = 0; z < splits; z++) { const float4 v = slab[(size_t)z * total4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
		if (bias) {
			const size_t row = i / hwo4;
			const int image = (int)(row / M), ch = (int)(row - (size_t)image * M);
			const float b = bias[(size_t)image * bias_stride + ch];
			s.x += b; s.y += b; s.z += b; s.w += b;
		}
		out[i] = s;
		if (out2) { const float4 a = add[i]; out2[i] = make_float4(s.x + a.x, s.y + a.y, s.z + a.z, s.w + a.w); }
	}
}
// LDS of the image-window kernel (mode 7): two A slabs, two channel groups of the window -- [16 planes][128 / W + 2 rows][4 zeros + W], a group rounded up to
// whole 1-KiB DMA instructions -- and 16 bytes of slack at either end (the kernel's W7_* constants)
static size_t window_lds_bytes(int W) {
	const size_t plane = (size_t)(128 / W + 2) * (W + 4), group = (16 * plane + 255) / 256 * 256;
	return (2 * 128 * 16 + 2 * group + 8) * sizeof(float);
}
int gather_gemm_splits(int mode, int batch, int M, int N, int HWo) {
	if (mode != 2 && mode != 4) return 1;
	const long K = (long)batch * HWo;
	const int kps = gather_k_per_split(mode, batch, M, N, HWo);
	return (int)((K + kps - 1) / kps);
}

// Several forward-shaped products over ONE padded image in one launch (mode 3, half-slab pipeline, whole tiles): the four parity classes of a stride-2
// data gradient.  blockIdx.y = class.  The classes differ in contraction length (4F, 2F, 2F, F for 3x3 kernels); all 4 x tiles workgroups are resident
// at once (two per CU), so the ORDER decides which classes share a CU: longest with shortest, the two middle ones together (cls[] as given: the
// caller sorts).  Needs about two workgroups per CU in total; otherwise the caller runs the classes one by one with their taps cut over workgroups.
bool gather_classes_fit(int ncls, int M, int N) {
	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	return ncls >= 2 && ncls <= 4 && gather_hs(3, M, N) && (long)ncls * (M / 128) * (N / 128) >= 2L * cus - cus / 2;
}
bla_status gather_gemm_classes(hipStream_t s, int batch, int M, int N, const GatherClass* cls, const GatherClass* d_cls, int ncls, int ldc, const float* img,
                               const int2* ntab, int H, int W, int HWo, int img_stride) {
	BLA_REQUIRE(gather_classes_fit(ncls, M, N) && N % 4 == 0 && HWo % 4 == 0 && (long)batch * img_stride < (1L << 29), BLA_ERR_INVALID, "class launch: M=%d N=%d classes=%d", M, N, ncls);
	GemmArgs a = {};
	a.M = M; a.N = N; a.ldc = ldc; a.alpha = 1.f; a.act = BLA_ACT_NONE;
	a.g_img = img; a.g_zero = zero_word(); a.g_ntab = ntab; a.g_mode = 3; a.g_H = H; a.g_W = W; a.g_HWo = HWo; a.g_img_stride = img_stride;
	a.tiles_m = M / 128; a.tiles_n = N / 128; a.splits = 1; a.k_per_split = 1 << 30;
	a.g_ncls = ncls; a.g_cls = d_cls;       // d_cls: the same entries in device memory (the caller's launch wrote them)
	for (int i = 0; i < ncls; i++)
		BLA_REQUIRE(cls[i].K > 0 && cls[i].K % 16 == 0 && (uintptr_t)cls[i].A % 16 == 0, BLA_ERR_INVALID, "class %d: K = %d", i, cls[i].K);
	a.A = cls[0].A; a.C = cls[0].C; a.g_ktab = cls[0].ktab; a.K = cls[0].K; a.lda = a.K;
	const dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)ncls, 1), block(256);
	hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 3, false, true>), grid, block, 2 * (128 + 128) * 16 * sizeof(float), s, a);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

bla_status gather_gemm(hipStream_t s, int mode, int batch, int M, int N, int K, const float* A, int lda, float* C, int ldc, const float* img,
                       const int2* ktab, const int2* ntab, int H, int W, int HWo, int img_stride, const GatherEpilogue* ep, int wo) {
	BLA_REQUIRE((mode >= 1 && mode <= 4) || mode == 7, BLA_ERR_INVALID, "gather mode %d", mode);
	if (mode == 7) {   // the image window in LDS (3x3, stride 1): whole 128-pixel tiles inside one image, one pass over K, A = kernels re-ordered [M][(group, tap, channel)]
		const int ch = HWo > 0 ? img_stride / HWo : 0;
		BLA_REQUIRE(M % 128 == 0 && N % 128 == 0 && HWo == H * W && HWo % 128 == 0 && (W == 16 || W == 32) && ch % 16 == 0 && ch > 0 && K == 9 * ch && lda == K &&
		            (uintptr_t)A % 16 == 0 && (long)batch * img_stride < (1L << 29) && (long)N * M < (1L << 31), BLA_ERR_INVALID,
		            "mode 7 shape (M=%d N=%d K=%d H=%d W=%d C=%d)", M, N, K, H, W, ch);
		GemmArgs a = {};
		a.A = A; a.lda = lda; a.C = C; a.M = M; a.N = N; a.K = K; a.ldc = ldc;
		a.alpha = 1.f; a.act = BLA_ACT_NONE;
		a.g_img = img; a.g_zero = zero_word(); a.g_mode = 7; a.g_H = H; a.g_W = W; a.g_HWo = HWo; a.g_img_stride = img_stride;
		a.tiles_m = M / 128; a.tiles_n = N / 128; a.k_per_split = K; a.splits = 1;
		if (ep) { a.g_bias = ep->bias; a.g_bias_stride = ep->bias_stride; a.g_add = ep->add; a.g_out2 = ep->out2; }
		const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(256);
		const size_t lds_bytes = window_lds_bytes(W);
		if (W == 32) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 7, false, true, 1, 32>), grid, block, lds_bytes, s, a);
		else hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 7, false, true, 1, 16>), grid, block, lds_bytes, s, a);
		BLA_HIP(hipGetLastError());
		return BLA_OK;
	}
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
	const bool hs = gather_hs(mode, M, N) && (mode != 4 || gather4_fixed_offsets(wo));
	a.g_wo = wo;
	const bool with_ep = ep && (ep->bias || ep->out2);
	if (with_ep) {
		BLA_REQUIRE(mode == 3 && hs, BLA_ERR_INVALID, "the fused convolution epilogue needs the half-slab forward kernel (gather3_fuses_epilogue)");
		if (splits == 1) { a.g_bias = ep->bias; a.g_bias_stride = ep->bias_stride; a.g_add = ep->add; a.g_out2 = ep->out2; }
	}
	if (hs && mode == 3) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 3, false, true>), grid, block, lds_bytes, s, a);
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
		if (with_ep) {
			const size_t total4 = (size_t)M * N / 4;
			BLA_REQUIRE((uintptr_t)C % 16 == 0 && (!ep->out2 || ((uintptr_t)ep->out2 % 16 == 0 && (uintptr_t)ep->add % 16 == 0)), BLA_ERR_INVALID, "unaligned convolution output");
			const size_t blocks = (total4 + 255) / 256;
			hipLaunchKernelGGL(gather_fold_epilogue_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, (const float4*)a.slab, (float4*)C, splits, total4,
			                   ep->bias, ep->bias_stride, (const float4*)ep->add, (float4*)ep->out2, M, HWo / 4);
			BLA_HIP(hipGetLastError());
			return BLA_OK;
		}
		BLA_HIP(launch_splitk_reduce(r, s));
	}
	return BLA_OK;
}

// ---- both gradients of one convolution in one launch (gather_pair_kernel) -----------------------------------------------------------------------------
namespace { struct GatherPlan { GemmArgs a; dim3 grid; size_t lds; int mode, gw, hwo; size_t slab_floats; GatherEpilogue ep; }; }
static bla_status gather_plan(int mode, int batch, int M, int N, int K, const float* A, int lda, float* C, int ldc, const float* img, const int2* ktab, const int2* ntab, int H, int W,
                              int HWo, int img_stride, const GatherEpilogue* ep, GatherPlan* out);
// gather_plan = the planning half of gather_gemm for the forms the pair takes: the weight gradient on the half-slab mode 4, the data gradient on mode 7
// or on the half-slab mode 3 (whole 128 x 128 tiles).  Everything but the slab address: the caller lays both products' slabs out in one workspace.
bool gather_pair_fits(int mode, int M, int N) { return mode == 7 || gather_hs(mode, M, N); }
static bla_status gather_plan(int mode, int batch, int M, int N, int K, const float* A, int lda, float* C, int ldc, const float* img, const int2* ktab, const int2* ntab, int H, int W,
                              int HWo, int img_stride, const GatherEpilogue* ep, GatherPlan* out) {
	BLA_REQUIRE(out && (mode == 3 || mode == 4 || mode == 7) && gather_pair_fits(mode, M, N), BLA_ERR_INVALID, "gather_plan: mode %d M=%d N=%d", mode, M, N);
	GemmArgs a = {};
	a.alpha = 1.f; a.beta = 0.f; a.act = BLA_ACT_NONE;
	a.C = C; a.M = M; a.N = N; a.K = K; a.ldc = ldc;
	a.g_img = img; a.g_zero = zero_word(); a.g_mode = mode; a.g_H = H; a.g_W = W; a.g_HWo = HWo; a.g_img_stride = img_stride;
	a.tiles_m = M / 128; a.tiles_n = N / 128;
	out->mode = mode; out->gw = 0;
	if (mode == 7) {
		const int ch = HWo > 0 ? img_stride / HWo : 0;
		BLA_REQUIRE(M % 128 == 0 && N % 128 == 0 && HWo == H * W && HWo % 128 == 0 && (W == 16 || W == 32) && ch % 16 == 0 && ch > 0 && K == 9 * ch && lda == K &&
		            (uintptr_t)A % 16 == 0 && (long)batch * img_stride < (1L << 29) && (long)N * M < (1L << 31), BLA_ERR_INVALID, "mode 7 shape (M=%d N=%d K=%d H=%d W=%d C=%d)", M, N, K, H, W, ch);
		a.A = A; a.lda = lda; a.k_per_split = K; a.splits = 1;
		if (ep) { a.g_bias = ep->bias; a.g_bias_stride = ep->bias_stride; a.g_add = ep->add; a.g_out2 = ep->out2; }
		out->lds = window_lds_bytes(W);
		out->grid = dim3((unsigned)(a.tiles_m * a.tiles_n), 1, 1);
		out->gw = W;
	} else {
		BLA_REQUIRE(mode != 3 || (N % 4 == 0 && HWo % 4 == 0), BLA_ERR_INVALID, "mode 3 needs pixel counts that are multiples of 4");
		BLA_REQUIRE(K > 0 && K % 16 == 0 && lda % 4 == 0 && (uintptr_t)A % 16 == 0 && (mode != 4 || HWo % 16 == 0) && (long)batch * img_stride < (1L << 29), BLA_ERR_INVALID,
		            "gathered product needs K %% 16 == 0 and a 16-byte aligned dense operand (M=%d N=%d K=%d lda=%d)", M, N, K, lda);
		if (mode == 4) { a.A = nullptr; a.lda = 0; a.B = A; a.ldb = lda; } else { a.A = A; a.lda = lda; }
		a.g_ktab = ktab; a.g_ntab = ntab;
		const int splits = mode == 3 ? gather3_splits(M, N, K) : gather_gemm_splits(mode, batch, M, N, HWo);
		a.k_per_split = mode == 4 ? gather_k_per_split(mode, batch, M, N, HWo) : (K / 16 + splits - 1) / splits * 16;
		a.splits = splits;
		if (ep && (ep->bias || ep->out2) && splits == 1) { a.g_bias = ep->bias; a.g_bias_stride = ep->bias_stride; a.g_add = ep->add; a.g_out2 = ep->out2; }
		out->lds = 2 * (128 + 128) * 16 * sizeof(float);
		out->grid = dim3((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)splits);
	}
	out->a = a;
	out->slab_floats = a.splits > 1 ? (size_t)a.splits * M * N : 0;
	out->ep = ep ? *ep : GatherEpilogue{};
	out->hwo = HWo;
	return BLA_OK;
}
// w: a mode-4 plan, d: a mode-3 or mode-7 plan; slabs (a.slab) set by the caller where slab_floats > 0.  One launch, then the folds.
static bla_status gather_pair(hipStream_t s, const GatherPlan& w, const GatherPlan& d) {
	BLA_REQUIRE(w.mode == 4 && (d.mode == 3 || d.mode == 7) && (w.slab_floats == 0 || w.a.slab) && (d.slab_floats == 0 || d.a.slab), BLA_ERR_INVALID, "gather_pair: bad plans");
	const int wx = (int)w.grid.x, wz = (int)w.grid.z, dx = (int)d.grid.x, dz = (int)d.grid.z, blocks_w = wx * wz;
	const dim3 grid((unsigned)(blocks_w + dx * dz)), block(256);
	const size_t lds = w.lds > d.lds ? w.lds : d.lds;
	if (d.mode == 7 && d.gw == 32) hipLaunchKernelGGL((gather_pair_kernel<7, 32>), grid, block, lds, s, w.a, d.a, blocks_w, wx, wz, dx, dz);
	else if (d.mode == 7) hipLaunchKernelGGL((gather_pair_kernel<7, 16>), grid, block, lds, s, w.a, d.a, blocks_w, wx, wz, dx, dz);
	else hipLaunchKernelGGL((gather_pair_kernel<3, 0>), grid, block, lds, s, w.a, d.a, blocks_w, wx, wz, dx, dz);
	BLA_HIP(hipGetLastError());
	if (w.a.splits > 1) {
		GemmArgs r = w.a;
		r.M = w.a.N; r.N = w.a.M;      // the slabs hold the transposed tile: [split][N][M] -> C [N][M]
		BLA_HIP(launch_splitk_reduce(r, s));
	}
	if (d.a.splits > 1) {
		const bool with_ep = d.ep.bias || d.ep.out2;
		if (with_ep) {
			const size_t total4 = (size_t)d.a.M * d.a.N / 4, blocks = (total4 + 255) / 256;
			hipLaunchKernelGGL(gather_fold_epilogue_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, (const float4*)d.a.slab, (float4*)d.a.C, d.a.splits, total4,
			                   d.ep.bias, d.ep.bias_stride, (const float4*)d.ep.add, (float4*)d.ep.out2, d.a.M, d.hwo / 4);
			BLA_HIP(hipGetLastError());
		} else {
			GemmArgs r = d.a;
			r.ldc = r.N;
			BLA_HIP(launch_splitk_reduce(r, s));
		}
	}
	return BLA_OK;
}

static bla_status plan_of(const GatherProduct& g, int batch, GatherPlan* out) {
	BLA_REQUIRE(g.mode != 4 || gather4_fixed_offsets(g.wo), BLA_ERR_INVALID, "gather pair: the weight gradient's map is %d pixels wide (4, 8 or a multiple of 16)", g.wo);
	bla_status st = gather_plan(g.mode, batch, g.M, g.N, g.K, g.A, g.lda, g.C, g.ldc, g.img, g.ktab, g.ntab, g.H, g.W, g.HWo, g.img_stride, &g.ep, out);
	if (!st) out->a.g_wo = g.wo;
	return st;
}
size_t gather_product_slab_floats(const GatherProduct& g, int batch) {
	if (g.mode == 7) return 0;
	const int splits = g.mode == 3 ? gather3_splits(g.M, g.N, g.K) : gather_gemm_splits(g.mode, batch, g.M, g.N, g.HWo);
	return splits > 1 ? (size_t)splits * g.M * g.N : 0;
}
bla_status gather_pair_products(hipStream_t s, int batch, const GatherProduct& w, float* w_slab, const GatherProduct& d, float* d_slab) {
	GatherPlan pw, pd;
	bla_status st = plan_of(w, batch, &pw);
	if (st) return st;
	st = plan_of(d, batch, &pd);
	if (st) return st;
	pw.a.slab = pw.slab_floats ? w_slab : nullptr;
	pd.a.slab = pd.slab_floats ? d_slab : nullptr;
	return gather_pair(s, pw, pd);
}

}  // namespace bla
