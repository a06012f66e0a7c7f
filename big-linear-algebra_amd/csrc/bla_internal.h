// bla_internal.h -- shared by the HIP translation units of libbla_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include "../../include/bla.h"

namespace bla {

struct Context {
	bool ready = false;
	int users = 0;                  // threads on which this context is current (bla_context_set_current); guarded by the runtime's mutex
	int device = -1;
	hipStream_t stream = nullptr;
	void* workspace = nullptr;      // grow-only scratch (split-K slabs, reductions)
	size_t workspace_bytes = 0;
	void* workspace2 = nullptr;     // a second grow-only scratch for operands that must survive a kernel which uses the first (dilated gradients)
	size_t workspace2_bytes = 0;
	// The side lane: a second stream with its own scratch and a fork / join event pair.  The batched U-Net issues its weight gradients there (side_lane_fork:
	// the lane waits for what the caller's stream has issued so far; nothing on the caller's stream waits for the lane until side_lane_join) so that they run
	// beside the data gradients, norms and glue of the main chain.  While side_lane is set, ensure_workspace hands out the lane's scratch.
	hipStream_t side_stream = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	void* workspace3 = nullptr;
	size_t workspace3_bytes = 0;
	bool side_lane = false;
	unsigned* tile_counters = nullptr;   // 16384 arrival counters for in-launch split-K, zero between launches
	int num_cus = 0;
	char arch[64] = {0};
};

// The calling thread's current context: the one bla_context_set_current installed, else the process default (bla_init).
Context& ctx();
void set_error(const char* fmt, ...);
bla_status hip_fail(hipError_t e, const char* what);
bla_status require_ready();
// Scratch of at least `bytes` (device); valid until the next ensure_workspace call that grows it.
bla_status ensure_workspace(size_t bytes, void** out);
bla_status ensure_workspace2(size_t bytes, void** out);
// side lane (see Context): fork = the lane's stream (returned) waits for everything issued on `main` so far and becomes the target of ensure_workspace until
// side_lane_done(); join = `main` waits for everything issued on the lane
bla_status side_lane_fork(hipStream_t main, hipStream_t* lane);
void side_lane_done();
bla_status side_lane_join(hipStream_t main);
// out[i] = sum_{j<len} m[i*stride + j], i < count (bla_elementwise.hip)
bla_status window_sum(void* stream, const float* m, int count, int len, int stride, float* out);
inline hipStream_t pick_stream(void* s) { return s ? (hipStream_t)s : ctx().stream; }
// a device word that is always 0.0f (source address of the zero padding in gathered operands)
inline const float* zero_word() { return reinterpret_cast<const float*>(ctx().tile_counters + 16384); }
// Implicit-GEMM convolution over a batch of images on the direct-to-LDS 128x128x16 kernel, B operand gathered (bla_gemm.hip):
//   mode 1: C [batch][M][HWo] = A [M][K] . G,   n = (image, output pixel), k = tap;   ktab = taps, ntab = pixels
//   mode 2: C [M][N]          = sum_images A_b [M][HWo] . G_b^T,   n = tap, k = (image, pixel);   ktab = pixels, ntab = taps
//   mode 3: mode 1 on a zero-padded image copy (stride 1): no bounds checks, 16-byte DMA;  tables hold padded offsets
//   mode 4: mode 2 transposed on a padded copy (stride 1): M = taps, N = rows of A (= del_y [image][N][HWo]), C [N][M] = the weight gradient
// tables: {element offset, y | x << 16}.  Needs K % 16 == 0 (mode 2: HWo % 16 == 0), A 16-byte aligned with lda % 4 == 0.
int gather3_splits(int M, int N, int K);                               // ... of a mode-3 product (forward / data gradient on few tiles)
int gather_gemm_splits(int mode, int batch, int M, int N, int HWo);   // K splits (= slabs of M*N floats in the workspace) gather_gemm will use
struct GatherEpilogue { const float* bias; int bias_stride; const float* add; float* out2; };   // mode 3: out = product + bias[image * stride + row]; out2 = out + add
bool gather3_fuses_epilogue(int M, int N, int K);                      // true: gather_gemm(mode 3, ...) takes a GatherEpilogue
bla_status gather_gemm(hipStream_t s, int mode, int batch, int M, int N, int K, const float* A, int lda, float* C, int ldc, const float* img,
                       const int2* ktab, const int2* ntab, int H, int W, int HWo, int img_stride, const GatherEpilogue* ep = nullptr, int wo = 0);
// (wo: mode 4 -- the output map's width; 4, 8 or a multiple of 16 puts the weight gradient on the half-slab kernel with fixed lane offsets, anything else (or 0) on the older form)
// mode 3, up to four products over the same padded image in one launch (bla_gather.hip): each class its kernels A [M][K] (lda = K), tap table and output
struct GatherClass { const float* A; int K; const int2* ktab; float* C; };
bool gather_classes_fit(int ncls, int M, int N);
bla_status gather_gemm_classes(hipStream_t s, int batch, int M, int N, const GatherClass* cls, const GatherClass* d_cls, int ncls, int ldc, const float* img,
                               const int2* ntab, int H, int W, int HWo, int img_stride);

// Both gradients of one batched convolution in ONE launch (gather_pair_kernel, bla_gemm_kernel.h): w = the weight gradient as gather_gemm(mode 4, ...) takes it,
// d = the data gradient as gather_gemm(mode 7 or 3, ...) takes it (whole 128 x 128 tiles: gather_pair_fits); slabs: gather_product_slab_floats floats each (0: none)
struct GatherProduct { int mode, M, N, K; const float* A; int lda; float* C; int ldc; const float* img; const int2* ktab; const int2* ntab; int H, W, HWo, img_stride; GatherEpilogue ep; int wo; };   // wo: as gather_gemm's
bool gather_pair_fits(int mode, int M, int N);
size_t gather_product_slab_floats(const GatherProduct& g, int batch);
bla_status gather_pair_products(hipStream_t s, int batch, const GatherProduct& w, float* w_slab, const GatherProduct& d, float* d_slab);

// bla_conv.hip: implicit-GEMM convolution with the adds the U-Net puts behind it: out = conv + ep_bias[image * ep_bias_stride + channel];
// ep_out2 = out + ep_add, both optional.  One image: folded into the store; a batch: one pass behind the product.
// x_padded (batched callers): the zero-padded copy of d_x in conv_padded_layout(h, w, k, stride), written by the producer of d_x (the norm kernel in front):
// the padded-copy forward / the weight gradient then make none of their own.  Ignored on the paths that use no padded copy.
bla_status conv2d_forward_epilogue(void* stream, const float* d_x, const float* d_kern, float* d_out, int h, int w, int k, int c_in, int f_n, int stride,
                                   const float* ep_bias, const float* ep_add, float* ep_out2, int batch = 1, int ep_bias_stride = 0, const float* x_padded = nullptr,
                                   const float* prepared = nullptr);
bla_status conv2d_backward_batched(void* stream, const float* d_del_y, const float* d_x, const float* d_kern, float* d_del_kern, float* d_del_x, float* d_scratch, int batch,
                                   int h, int w, int k, int c_in, int f_n, int stride, const float* x_padded, const float* prepared = nullptr,
                                   const float* dy_padded = nullptr);   // dy_padded: the same for d_del_y (odd k, stride 1: the padded-copy data gradient reads it)
// Kernel matrices in the form a batched convolution's product reads them, prepared by the caller for MANY convolutions in one launch (the U-Net: once per
// pass instead of a 5-us launch in front of every product): mode 1 = window order of the forward kernels (gather mode 7), 2 = flipped + window order (its
// data gradient), 3 = flipped / transposed [C][F][k][k] (data gradient on the padded copy).  conv_kernel_prep_mode says which one a convolution's forward /
// data-gradient product takes (0: none); conv2d_forward_epilogue / conv2d_backward_batched take the prepared matrix as `prepared`.
struct KernelPrepJob { const float* src; float* dst; int f_n, c_n, k, mode; };
int conv_kernel_prep_mode(int batch, int h, int w, int k, int c_in, int f_n, int stride, bool data_gradient);
bla_status conv_prepare_kernels(void* stream, const KernelPrepJob* d_jobs, int njobs, size_t max_elements);
struct PadLayout { int w, wh, plane, pt, pl; };      // plane = 0: none
PadLayout conv_padded_layout(int h, int w, int k, int stride);
struct PadOut { float* dst; PadLayout L; };          // a producer's second output: dst[plane_index * L.plane + (y + L.pt) * L.wh + x + L.pl], halo zeroed once by the owner

// bla_conv_thin.hip: direct convolutions for at most four channels on one side (stride 1, k 1 or 3): forward with the optional epilogue, weight gradient
bool thin_conv_applies(int k, int c_in, int f_n, int stride);
bla_status thin_conv_forward(hipStream_t s, const float* x, const float* kern, float* out, int batch, int h, int w, int k, int c_in, int f_n, int pt, int pl,
                             const float* ep_bias, int ep_bias_stride, const float* ep_add, float* ep_out2);
bla_status thin_conv_wgrad(hipStream_t s, const float* del_y, const float* x, float* del_kern, int batch, int h, int w, int k, int c_in, int f_n, int pt, int pl);

// bla_gemm_thin.hip: batched products with a short contraction (k <= 64, a multiple of 8; m, n multiples of 32) and a large output, one wave per
// 32 x (32..128) block straight from global memory; up to three (A, B) pairs accumulate before the one store.  Strides in floats per batch index.
struct ThinPart { const float* A; const float* B; long sa, sb; int lda, ldb; };
bool gemm_thin_applies(int m, int n, int k, int batch);
bla_status gemm_thin_parts(void* stream, int transa, int transb, int m, int n, int k, const ThinPart* parts, int nparts, float* C, int ldc, long stride_c, int batch,
                           float alpha, float beta, const float* bias_row, float* pre, int ld_pre, long stride_pre);

// group norm + ReLU + dropout in one pass (relu = max(norm, 0), dropped = drop ? 0 : relu), model/cifar_unet.c:1056-1058
bla_status group_norm_relu_dropout(void* stream, const float* d_in, float* d_relu, const unsigned char* d_drop, float* d_dropped, float* d_stdevs, float* d_means,
                                   int channels, int group_size, int hw, const PadOut* pad = nullptr);   // d_drop may be NULL with pad (then the padded copy holds the ReLU output)
// group_norm_ddx with the ReLU gate on its input and the residual gradient added to its output (either may be NULL), model/cifar_unet.c:1204-1205,1219
bla_status group_norm_ddx_gated(void* stream, const float* d_source, float* d_dest, const float* d_data, const float* d_means, const float* d_stdevs,
                                int channels, int group_size, int hw, const float* d_relu_gate, const float* d_addend, const PadOut* pad = nullptr);

// bla_unet.hip: the batched ResNet block with the time-embedding projection hoisted out (bla_unet_model.hip forms all blocks' projections / time gradients in
// one launch each): RESNET_TDENSE_READY = ws->tdense is already filled; RESNET_DEFER_TIME_GRADS = only the per-image channel sums go to d_dtb
enum { RESNET_TDENSE_READY = 1, RESNET_DEFER_TIME_GRADS = 2, RESNET_WGRAD_SIDE = 4 };   // WGRAD_SIDE: the block's weight gradients go to the context's side lane
                                                                                         // (the caller keeps del_out and pads->g_out_b intact until it has joined the lane)
// padded copies of a block's two convolution inputs (relu1: B*cin planes, dp: B*cout planes; conv_padded_layout(h, w, k, 1); halo zeroed once by the owner):
// the forward pass's norm kernels fill them and say so (have1 / have2), both convolutions and both weight gradients then gather from them
struct ResnetPads { float* pad1; float* pad2; bool have1, have2;
                    const float *k1_fwd, *k2_fwd, *k1_bwd, *k2_bwd;
                    float* dy_pad;
                    float* g_res;
                    float* g_out_b; };   // RESNET_WGRAD_SIDE: this block's own buffer for the gradient in front of the first convolution (the side lane reads it later)   // scratch of B*cin*hw floats: the 1x1 residual convolution's data gradient, added inside the last norm gradient instead of by a pass of its own   // scratch of the block's resolution (B*cout planes, halo zeroed once): the padded gradient the first convolution's data gradient reads   // the two convolutions' prepared kernel matrices (KernelPrepJob outputs) or NULL
bla_status resnet_forward_single(void* stream, const float* d_x, const float* d_temb, const bla_resnet_params* p, const unsigned char* d_drop, const bla_resnet_ws* ws,
                                 float* d_result, int h, int w, int cin, int cout, int k, int tdim, int group_size, int flags);
bla_status resnet_forward_batched(void* stream, int batch, const float* d_x, const float* d_temb, const bla_resnet_params* p, const unsigned char* d_drop,
                                  const bla_resnet_ws* ws, float* d_result, int h, int w, int cin, int cout, int k, int tdim, int group_size, int flags, ResnetPads* pads);
bla_status resnet_backward_batched(void* stream, int batch, const float* d_del_out, const float* d_x, const float* d_temb, const bla_resnet_params* p,
                                   const bla_resnet_ws* ws, const bla_resnet_grads* g, const bla_resnet_scratch* sc, float* d_dtb, float* d_del_x, int h, int w,
                                   int cin, int cout, int k, int tdim, int group_size, int flags, const ResnetPads* pads);

// "My gradients are ready", posted by the LAST gradient kernel of a data-parallel step instead of by the exchange launch behind it (every
// step paid a launch boundary before any peer could start reading).  The block lives in device memory (built by bla_dp_connect); a launch that is
// handed it counts its workgroups in, and the last one to finish stores epoch + 1 into every peer's flag word for this rank -- each workgroup after
// its own s_waitcnt vmcnt(0) + barrier + system-scope release fence, the last arriver with an acquire on the count and system-scope release stores
// (the two-shot exchange's publish protocol).
struct DoneHook {
	unsigned* arrive;          // arrival counter, zero between launches
	const unsigned* epoch;     // the exchange object's state[0]: epoch of its last finished exchange
	int world, rank;
	unsigned* peer_flags[16];  // flag array A in rank p's memory (own slot unused)
};
// bla_gemm.hip: bla_gemm_pair_f32 with the hook handed to the pair launch; *posted says whether a launch took it (a pair that falls back to two
// launches does not)
bla_status gemm_pair_with_hook(void* stream, const bla_gemm_desc* p, const bla_gemm_desc* q, const DoneHook* d_hook, bool* posted);
// bla_dp.hip: the device-resident hook of a connected exchange object (NULL for a single rank); the exchange of `parity` with the flag push left
// to whoever took the hook
const DoneHook* dp_done_hook(bla_dp* dp);
bla_status dp_allreduce(bla_dp* dp, void* stream, int parity, float* d_out, float* d_target, float alpha, bool flags_posted);

// bla_dp.hip: identity of an exchange object that survives address reuse (a destroyed and re-created object never has the same id)
unsigned long long dp_identity(const bla_dp* dp);
// gradient-bucket parity of the next data-parallel step through this exchange object / count one step (several trainers -- the full-batch
// one and the one for an epoch's last, shorter batch -- may share an exchange object: the parity must alternate over ALL their steps)
int dp_parity(const bla_dp* dp);
void dp_advance(bla_dp* dp);

// The callers of this library are the reference's C programs: they seed libc rand() once (srand(42), model/mnist_nn.c:513) and draw from it
// between library calls (MNIST sampler, U-Net dropout).  HIP runtime initialisation and RCCL communicator set-up draw from / reseed rand()
// themselves (measured: tests/c/rand_stream.c), which would shift the program's stream against the reference's CPU run.  glibc's rand() is
// random() on the current state array: park the caller's state while those calls run and put it back afterwards.
// Process-wide and nestable: one host thread per rank may be inside the library at once (bla.h), and a blocking call such as
// ncclCommInitRank must not hold a lock the other ranks' threads need.  The FIRST guard to be entered parks the caller's state (initstate
// onto a scratch array in static storage), the LAST one to leave puts it back; only the depth counter and the swap are under a mutex, never
// the guarded call.  While any guard is open, rand() draws (the runtime's, RCCL's) go to the scratch state.
void rand_guard_enter();
void rand_guard_leave();
struct RandStreamGuard {
	RandStreamGuard() { rand_guard_enter(); }
	~RandStreamGuard() { rand_guard_leave(); }
	RandStreamGuard(const RandStreamGuard&) = delete;
	RandStreamGuard& operator=(const RandStreamGuard&) = delete;
};

#define BLA_HIP(call)                                               \
	do {                                                            \
		hipError_t _e = (call);                                     \
		if (_e != hipSuccess) return ::bla::hip_fail(_e, #call);    \
	} while (0)

#define BLA_REQUIRE(cond, status, ...)                              \
	do {                                                            \
		if (!(cond)) { ::bla::set_error(__VA_ARGS__); return status; } \
	} while (0)

}  // namespace bla
