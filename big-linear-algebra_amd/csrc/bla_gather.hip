// bla_gather.hip -- implicit-GEMM convolution on the direct-to-LDS GEMM pipeline: the host side of the gathered-operand variants of
// gemm_f32_glds_kernel (bla_gemm_kernel.h; modes 1 - 6 in GemmArgs) -- tile / split planning and the launches.  Called from bla_conv.hip.
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
	return use_hs && M % 128 == 0 && N % 128 == 0 && (mode == 3 || mode == 5 || ((mode == 4 || mode == 6) && hs4));
}
bool gather_whole_tiles(int mode, int M, int N) { return gather_hs(mode, M, N); }
static int gather_k_per_split(int mode, int batch, int M, int N, int HWo) {
	const long K = (long)batch * HWo;
	if (mode != 2 && mode != 4 && mode != 6) return (int)K;
	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
	const long slots = 2L * cus;   // two workgroups per CU on either form
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
bool gather3_fuses_epilogue(int M, int N, int K) { return gather_hs(3, M, N) && gather3_splits(M, N, K) == 1; }
int gather_gemm_splits(int mode, int batch, int M, int N, int HWo) {
	if (mode != 2 && mode != 4 && mode != 6) return 1;
	const long K = (long)batch * HWo;
	const int kps = gather_k_per_split(mode, batch, M, N, HWo);
	return (int)((K + kps - 1) / kps);
}

bla_status gather_gemm(hipStream_t s, int mode, int batch, int M, int N, int K, const float* A, int lda, float* C, int ldc, const float* img,
                       const int2* ktab, const int2* ntab, int H, int W, int HWo, int img_stride, const GatherEpilogue* ep) {
	BLA_REQUIRE(mode >= 1 && mode <= 6, BLA_ERR_INVALID, "gather mode %d", mode);
	const bool fwd16 = mode == 3 || mode == 5, wg16 = mode == 4 || mode == 6;   // the 16-byte forms: gathered B (forward / data gradient), gathered A (weight gradient)
	BLA_REQUIRE(!fwd16 || (N % 4 == 0 && HWo % 4 == 0 && N >= 4), BLA_ERR_INVALID, "mode %d needs pixel counts that are multiples of 4", mode);
	BLA_REQUIRE(!wg16 || M % 4 == 0, BLA_ERR_INVALID, "mode %d needs a tap count that is a multiple of 4", mode);
	BLA_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0 && lda % 4 == 0 && (uintptr_t)A % 16 == 0 && ((mode != 2 && !wg16) || HWo % 16 == 0), BLA_ERR_INVALID,
	            "gathered product needs K %% 16 == 0 and a 16-byte aligned dense operand (M=%d N=%d K=%d lda=%d)", M, N, K, lda);
	BLA_REQUIRE((long)batch * img_stride < (mode >= 3 ? (1L << 29) : (1L << 31)) && (long)N < (1L << 31), BLA_ERR_INVALID, "batch too large for 32-bit gather offsets");
	const bool hs = gather_hs(mode, M, N);
	BLA_REQUIRE((mode != 5 && mode != 6) || (hs && W % 4 == 0 && HWo == H * W && H < 65536 && W < 32768), BLA_ERR_INVALID,
	            "the unpadded modes need whole 128-wide tiles on the half-slab pipeline, stride 1 and rows of a multiple of four pixels (M=%d N=%d W=%d)", M, N, W);
	GemmArgs a = {};
	a.C = C; a.M = M; a.N = N; a.K = K; a.ldc = ldc;
	if (wg16) { a.A = nullptr; a.lda = 0; a.B = A; a.ldb = lda; }      // the dense operand (del_y) is the K-contiguous B
	else { a.A = A; a.lda = lda; a.B = nullptr; a.ldb = 0; }
	a.alpha = 1.f; a.beta = 0.f; a.act = BLA_ACT_NONE;
	a.g_img = img; a.g_zero = zero_word(); a.g_ktab = ktab; a.g_ntab = ntab; a.g_mode = mode; a.g_H = H; a.g_W = W; a.g_HWo = HWo; a.g_img_stride = img_stride;
	a.tiles_m = (M + 127) / 128; a.tiles_n = (N + 127) / 128;
	// forward / data gradient on the half-slab pipeline: 128 x 256 tiles (waves 2 x 2, each 64 x 128: half the LDS reads and DMA instructions per MFMA of the
	// 128 x 128 tile) once they give every CU a workgroup; BLA_CONV_BN=128 keeps the 128 x 128 tile, =256 forces the wide one
	static const int force_bn = [] { const char* e = getenv("BLA_CONV_BN"); return e ? atoi(e) : 0; }();
	const int cus = ctx().num_cus > 0 ? ctx().num_cus : 256;
	const int splits = fwd16 ? gather3_splits(M, N, K) : gather_gemm_splits(mode, batch, M, N, HWo);
	const bool bn256 = fwd16 && hs && N % 256 == 0 && force_bn != 128 && ((long)(M / 128) * (N / 256) >= cus || force_bn == 256) && splits == 1;
	if (bn256) a.tiles_n = N / 256;
	a.k_per_split = (mode == 2 || wg16) ? gather_k_per_split(mode, batch, M, N, HWo) : fwd16 ? (K / 16 + splits - 1) / splits * 16 : K;
	a.splits = splits; a.slab = nullptr;
	if (splits > 1) {
		void* ws;
		bla_status st = ensure_workspace((size_t)splits * M * N * sizeof(float), &ws);
		if (st) return st;
		a.slab = (float*)ws;
	}
	dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)splits), block(256);
	const size_t lds_bytes = 2 * (128 + 128) * 16 * sizeof(float), lds_wide = 2 * (128 + 256) * 16 * sizeof(float);
	// whole tiles: the half-slab pipeline (fragment sets per k-half, every LDS read and DMA dealt out between MFMAs) -- BLA_CONV_HS=0 keeps the older form
	if (ep && (ep->bias || ep->out2)) {
		BLA_REQUIRE(fwd16 && hs && splits == 1, BLA_ERR_INVALID, "the fused convolution epilogue needs the half-slab forward kernel in one pass over K (gather3_fuses_epilogue)");
		a.g_bias = ep->bias; a.g_bias_stride = ep->bias_stride; a.g_add = ep->add; a.g_out2 = ep->out2;
	}
	if (mode == 5 && bn256) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 256, 16, 2, 2, true, false, 1, 2, false, 5, false, true>), grid, block, lds_wide, s, a);
	else if (mode == 5) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 5, false, true>), grid, block, lds_bytes, s, a);
	else if (mode == 6) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, true, 1, 2, false, 6, false, true>), grid, block, lds_bytes, s, a);
	else if (bn256) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 256, 16, 2, 2, true, false, 1, 2, false, 3, false, true>), grid, block, lds_wide, s, a);
	else if (hs && mode == 3) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 3, false, true>), grid, block, lds_bytes, s, a);
	else if (hs) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, true, 1, 2, false, 4, false, true>), grid, block, lds_bytes, s, a);
	else if (mode == 1) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 1>), grid, block, lds_bytes, s, a);
	else if (mode == 2) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 2>), grid, block, lds_bytes, s, a);
	else if (mode == 3) hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, false, 1, 2, false, 3>), grid, block, lds_bytes, s, a);
	else hipLaunchKernelGGL((gemm_f32_glds_kernel<128, 128, 16, 2, 2, true, true, 1, 2, false, 4>), grid, block, lds_bytes, s, a);
	BLA_HIP(hipGetLastError());
	if (splits > 1) {
		GemmArgs r = a;
		if (wg16) { r.M = N; r.N = M; }   // the slabs hold the transposed tile: [split][N][M] -> C [N][M]
		if (fwd16) r.ldc = r.N;            // C-shaped slabs ([image][M][HWo]): a flat sum
		BLA_HIP(launch_splitk_reduce(r, s));
	}
	return BLA_OK;
}

}  // namespace bla
