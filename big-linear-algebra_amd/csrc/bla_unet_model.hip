// bla_unet_model.hip -- the reference's U-Net (model/cifar_unet.c) assembled from the device-resident blocks of bla_unet.hip:
// forward() (:1099-1166) and backward() (:1351-1436) for one image -- or for a batch of images at once (bla_unet_create_batched: every
// activation carries a leading image index, every image its own time embedding, the gradients are summed over the images: what a mini-batch
// of the reference's one-image-at-a-time loop accumulates) -- parameters and gradients in two flat buckets.
//
//   18 ResNet blocks (8 down, 2 mid, 8 up), 5 self-attention blocks (two at resolution 2 on the way down, one in the middle, two at
//   resolution 2 on the way up), 3 stride-2 down-convolutions, 3 nearest-neighbour up-samplings (each followed by a channel-changing
//   convolution only where the two resolutions' widths differ, :1131,1141,1151), 4 skip concatenations, output group-norm + ReLU + conv.
//
// The reference's own model is work in progress: its backward pass walks out of bounds, and several call sites hand the
// wrong buffers around.  This composition is the INTENDED network, block for block what the reference's functions compute when they are
// wired as their comments and the forward pass say:
//   * the second attention block of the third up-sampling stage has its own parameters and reads the second ResNet block's result
//     (as written :1150 calls the first block's parameters again and :1151 then reads an output nobody wrote);
//   * the data gradient of the three stride-2 convolutions is the stride-2 adjoint (as written :1412,1420,1430 pass stride 1 and _col2im
//     indexes out of bounds);
//   * a skip connection's gradient is ADDED to the gradient that arrives along the main path (as written :1424-1427 adds it first and
//     then lets _backward_attention overwrite the sum);
//   * convolution gradients land in the gradient bucket (as written :1203,1216 hand conv_ddx the parameters as the sink).
// Every block is pinned on its own against the reference's functions (tests/test_resnet.py, test_attention.py, test_conv_gpu.py,
// test_unet_glue.py); the composition is pinned against the same composition of the oracle's blocks (tests/test_unet_model.py).
#include "bla_internal.h"
#include <cstring>
#include <string>
#include <vector>

using namespace bla;

namespace {
struct Tensor { size_t off, count; std::string name; };
struct Res {
	int cin, cout, h, w;
	size_t conv1, conv2, tw, tb, res;   // offsets in the buckets (res = SIZE_MAX when cin == cout)
	bla_resnet_ws ws;
	float* result;
	ResnetPads pads = {nullptr, nullptr, false, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // batched: zero-haloed padded copies of the two convolution inputs, filled by the norm kernels
	float* dtb = nullptr;                 // batched: per-image channel sums of the time-projection gradient [B][cout], kept until the pass's last launch
	size_t drop_off;                      // offset of this block's dropout decisions in the caller's mask
};
// one ResNet block's time-embedding projection (model/cifar_unet.c:1051-1052) and its gradients (:1191-1199); all 18 run as one launch each way
struct TimeJob { const float* w; const float* bias; float* tdense; const float* dtb; float* g_tw; float* g_tb; int cout; };
struct Att {
	int c, h, w;
	size_t wq, wk, wv, wo, bias;
	bla_attention_ws fwd;
	float* out;
};
struct Conv {
	int cin, cout, h, w, stride, k;       // h, w: INPUT size
	size_t kern;
	float* out;
	bool present;
	const float *prep_fwd = nullptr, *prep_bwd = nullptr;   // batched: the kernel matrix as the forward / data-gradient product reads it (KernelPrepJob outputs)
};
constexpr size_t kNone = (size_t)-1;
}  // namespace

struct bla_unet {
	bla_unet_config cfg;
	int batch = 1;
	int H[4], W[4];
	std::vector<Tensor> tensors;
	size_t count = 0, drop_count = 0;
	float *params = nullptr, *grads = nullptr;
	Res res[18];
	Att att[5];
	Conv down[3], up[3], outc;
	float *cat[4] = {nullptr, nullptr, nullptr, nullptr};      // skip concatenations, stage order up_1 .. up_4
	float *nn[3] = {nullptr, nullptr, nullptr};                // nearest-neighbour outputs
	float *out_relu = nullptr, *out_mu = nullptr, *out_sd = nullptr;
	unsigned char* zero_drop = nullptr;
	// backward
	float *t1 = nullptr, *t2 = nullptr, *t3 = nullptr, *gskip[4] = {nullptr, nullptr, nullptr, nullptr};
	TimeJob* time_jobs = nullptr;                // device, one per ResNet block (batch > 1)
	float* g_res = nullptr;                                      // batched: the residual 1x1 convolutions' data gradient (largest block input)
	// batched, weight gradients on the context's side lane (BLA_UNET_SIDE=0: off): every gradient buffer of a backward pass is written ONCE (a pool instead of
	// three rotating buffers), every block has its own g_out_b -- what the lane reads stays intact until the pass's one join
	bool side = false;
	std::vector<float*> gpool;
	float* dy_pad[4] = {nullptr, nullptr, nullptr, nullptr};   // per resolution: padded gradient scratch of the blocks' first convolutions (halo zeroed once)
	KernelPrepJob *prep_fwd = nullptr, *prep_bwd = nullptr;   // device: every convolution's kernel matrix re-ordered / flipped in ONE launch per pass
	int n_prep_fwd = 0, n_prep_bwd = 0;
	size_t prep_max = 0;
	float *dtb = nullptr, *partials = nullptr;   // batched blocks: per-image time-bias sums [B][Cout], per-image attention weight gradients [B][C*d]
	bla_resnet_scratch sc = {};
	bla_attention_ws agrad = {};
	const float* last_x = nullptr; const float* last_temb = nullptr; const unsigned char* last_drop = nullptr;
	std::vector<void*> owned;
};

namespace {
bla_status dalloc(bla_unet* m, float** p, size_t floats) {
	void* q = nullptr;
	BLA_HIP(hipMalloc(&q, (floats ? floats : 1) * sizeof(float)));
	m->owned.push_back(q);
	*p = (float*)q;
	return BLA_OK;
}
size_t add_tensor(bla_unet* m, const std::string& name, size_t count) {
	size_t off = m->count;
	m->tensors.push_back(Tensor{off, count, name});
	m->count += (count + 3) / 4 * 4;    // every tensor 16-byte aligned inside the bucket
	return off;
}
void plan_res(bla_unet* m, Res& r, const std::string& name, int cin, int cout, int h, int w) {
	const int k = m->cfg.kernel, t = m->cfg.time_dim;
	r.cin = cin; r.cout = cout; r.h = h; r.w = w;
	r.conv1 = add_tensor(m, name + ".conv_1_kernels", (size_t)cout * cin * k * k);
	r.conv2 = add_tensor(m, name + ".conv_2_kernels", (size_t)cout * cout * k * k);
	r.tw = add_tensor(m, name + ".time_weights", (size_t)t * cout);
	r.tb = add_tensor(m, name + ".time_biases", (size_t)cout);
	r.res = cin != cout ? add_tensor(m, name + ".residual_conv_kernels", (size_t)cout * cin) : kNone;
	r.drop_off = m->drop_count;
	m->drop_count += (size_t)cout * h * w;
}
void plan_att(bla_unet* m, Att& a, const std::string& name, int c, int h, int w) {
	const int d = m->cfg.key_dim;
	a.c = c; a.h = h; a.w = w;
	a.wq = add_tensor(m, name + ".Q_proj", (size_t)c * d);
	a.wk = add_tensor(m, name + ".K_proj", (size_t)c * d);
	a.wv = add_tensor(m, name + ".V_proj", (size_t)c * d);
	a.wo = add_tensor(m, name + ".weights", (size_t)d * c);
	a.bias = add_tensor(m, name + ".biases", (size_t)c);
}
void plan_conv(bla_unet* m, Conv& c, const std::string& name, int cin, int cout, int h, int w, int stride, bool present = true) {
	c.cin = cin; c.cout = cout; c.h = h; c.w = w; c.stride = stride; c.k = m->cfg.kernel; c.present = present; c.out = nullptr;
	c.kern = present ? add_tensor(m, name, (size_t)cout * cin * c.k * c.k) : kNone;
}
// The second ReLU's output before dropout: the backward pass gates on the dropped form alone (dp = drop ? 0 : relu2 is zero wherever relu2 is), so a batched
// model whose norms run on the 16-byte kernels (rows of whole float4, a batch that folds into the channel count) does not store it
bool keep_relu2(const bla_unet* m, const Res& r) {
	const int gs = m->cfg.group_size;
	return !(m->batch > 1 && (r.h * r.w) % 4 == 0 && (r.cout % gs == 0 || r.cout < gs));
}
bla_status alloc_res(bla_unet* m, Res& r) {
	const size_t hw = (size_t)r.h * r.w * m->batch;   // (every buffer of the block: B times its single-image size)
	const int gs = m->cfg.group_size, g1 = (r.cin + gs - 1) / gs * m->batch, g2 = (r.cout + gs - 1) / gs * m->batch;
	bla_status st;
	if ((st = dalloc(m, &r.ws.mu1, g1)) || (st = dalloc(m, &r.ws.sd1, g1)) || (st = dalloc(m, &r.ws.relu1, r.cin * hw)) || (st = dalloc(m, &r.ws.c1, r.cout * hw)) ||
	    (st = dalloc(m, &r.ws.tdense, (size_t)r.cout * m->batch)) || (st = dalloc(m, &r.ws.mu2, g2)) || (st = dalloc(m, &r.ws.sd2, g2)) ||
	    (keep_relu2(m, r) ? (st = dalloc(m, &r.ws.relu2, r.cout * hw)) : (r.ws.relu2 = nullptr, BLA_OK)) ||
	    (st = dalloc(m, &r.ws.dp, r.cout * hw)) || (st = dalloc(m, &r.ws.c2, r.cout * hw)) || (st = dalloc(m, &r.result, r.cout * hw)))
		return st;
	r.ws.res = nullptr;
	if (r.cin != r.cout && (st = dalloc(m, &r.ws.res, r.cout * hw))) return st;
	if (m->batch == 1) return BLA_OK;
	if ((st = dalloc(m, &r.dtb, (size_t)r.cout * m->batch))) return st;
	if (r.cin != r.cout && r.cin > 4) {      // (one scratch for all blocks: they run one after the other)
		size_t widest = 0;
		for (int i = 0; i < 18; i++) if (m->res[i].cin != m->res[i].cout) widest = std::max(widest, (size_t)m->batch * m->res[i].cin * m->res[i].h * m->res[i].w);
		if (!m->g_res && (st = dalloc(m, &m->g_res, widest))) return st;
		r.pads.g_res = m->g_res;
	}
	// padded copies (conv_padded_layout): only where a tiled convolution will read them -- not for the 3-channel input of the first block (direct kernels)
	static const bool pads_on = [] { const char* e = getenv("BLA_UNET_PADS"); return !(e && e[0] == '0'); }();
	const PadLayout L = conv_padded_layout(r.h, r.w, m->cfg.kernel, 1);
	if (pads_on && L.plane > 0) {
		const size_t p1 = (size_t)m->batch * r.cin * L.plane, p2 = (size_t)m->batch * r.cout * L.plane;
		if (r.cin > 4) { if ((st = dalloc(m, &r.pads.pad1, p1))) return st; BLA_HIP(hipMemsetAsync(r.pads.pad1, 0, p1 * sizeof(float), ctx().stream)); }
		if ((st = dalloc(m, &r.pads.pad2, p2))) return st;
		BLA_HIP(hipMemsetAsync(r.pads.pad2, 0, p2 * sizeof(float), ctx().stream));
		// one scratch per resolution for the padded gradient in front of conv_1's data gradient (all blocks of a resolution share layout and halo)
		int res_i = 0;
		while (res_i < 3 && m->H[res_i] != r.h) res_i++;
		const size_t widest = (size_t)m->batch * std::max(std::max(m->cfg.dims[0], m->cfg.dims[1]), std::max(m->cfg.dims[2], m->cfg.dims[3])) * L.plane;
		if (!m->dy_pad[res_i]) {
			if ((st = dalloc(m, &m->dy_pad[res_i], widest))) return st;
			BLA_HIP(hipMemsetAsync(m->dy_pad[res_i], 0, widest * sizeof(float), ctx().stream));
		}
		r.pads.dy_pad = m->dy_pad[res_i];
	}
	return BLA_OK;
}
bla_status alloc_att_ws(bla_unet* m, bla_attention_ws& ws, size_t s1, size_t d) {
	bla_status st;
	const size_t s = s1, ss = s1 * s1 * m->batch;
	d *= m->batch;
	if ((st = dalloc(m, &ws.q, s * d)) || (st = dalloc(m, &ws.k, s * d)) || (st = dalloc(m, &ws.v, s * d)) || (st = dalloc(m, &ws.attention, s * d)) ||
	    (st = dalloc(m, &ws.scores_raw, ss)) || (st = dalloc(m, &ws.weights, ss)))
		return st;
	return BLA_OK;
}

// tdense[b][c] = sum_t temb[b][t] W[t][c] + bias[c] for every block at once: grid (block, 8 images, 64 channels); a workgroup = 64 channels x 4 quarters of
// the t range, the eight embedding rows in LDS, the quarters folded through LDS in quarter order (t ascending inside a quarter)
__global__ void __launch_bounds__(256) time_dense_all_kernel(const TimeJob* __restrict__ jobs, const float* __restrict__ temb, int batch, int tdim) {
	extern __shared__ float te[];             // [8][tdim], then [4][8][64] partial sums
	float* part = te + 8 * tdim;
	const TimeJob j = jobs[blockIdx.x];
	const int c0 = blockIdx.z * 64;
	if (c0 >= j.cout) return;                 // (whole workgroups: before any barrier)
	const int b0 = blockIdx.y * 8, nb = min(8, batch - b0);
	for (int e = threadIdx.x; e < 8 * tdim; e += 256) te[e] = e < nb * tdim ? temb[(size_t)b0 * tdim + e] : 0.f;
	__syncthreads();
	const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6, c = c0 + lane;
	const int per = (tdim + 3) / 4, t0 = quarter * per, t1 = min(tdim, t0 + per);
	float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	if (c < j.cout)
#pragma unroll 8
		for (int t = t0; t < t1; t++) {      // (eight weight loads in flight: one by one this loop is 128 dependent round trips, 53 us)
			const float wv = j.w[(size_t)t * j.cout + c];
#pragma unroll
			for (int i = 0; i < 8; i++) acc[i] = fmaf(te[i * tdim + t], wv, acc[i]);
		}
#pragma unroll
	for (int i = 0; i < 8; i++) part[(quarter * 8 + i) * 64 + lane] = acc[i];
	__syncthreads();
	for (int e = threadIdx.x; e < 8 * 64; e += 256) {
		const int i = e >> 6, l = e & 63;
		if (i < nb && c0 + l < j.cout)
			j.tdense[(size_t)(b0 + i) * j.cout + c0 + l] = ((part[(0 * 8 + i) * 64 + l] + part[(1 * 8 + i) * 64 + l]) + (part[(2 * 8 + i) * 64 + l] + part[(3 * 8 + i) * 64 + l])) + j.bias[c0 + l];
	}
}
// g_tw[t][c] = sum_b temb[b][t] dtb[b][c], g_tb[c] = sum_b dtb[b][c] (images in order): grid (block, tdim / 8), thread = channel
__global__ void __launch_bounds__(256) time_grads_all_kernel(const TimeJob* __restrict__ jobs, const float* __restrict__ temb, int batch, int tdim) {
	const TimeJob j = jobs[blockIdx.x];
	const int t0 = blockIdx.y * 8, nt = min(8, tdim - t0);
	for (int c = threadIdx.x; c < j.cout; c += 256) {
		float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sb = 0.f;
#pragma unroll 8
		for (int b = 0; b < batch; b++) {
			const float d = j.dtb[(size_t)b * j.cout + c];
			sb += d;
#pragma unroll
			for (int i = 0; i < 8; i++) acc[i] = fmaf(i < nt ? temb[(size_t)b * tdim + t0 + i] : 0.f, d, acc[i]);
		}
		for (int i = 0; i < nt; i++) j.g_tw[(size_t)(t0 + i) * j.cout + c] = acc[i];
		if (blockIdx.y == 0) j.g_tb[c] = sb;
	}
}
// del_Y = 2 (prediction - noise), model/cifar_unet.c:1353-1364
__global__ void __launch_bounds__(256) unet_loss_grad_kernel(const float* __restrict__ out, const float* __restrict__ noise, float* __restrict__ g, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) g[i] = 2.f * (out[i] - noise[i]);
}
// _concat_skip, model/cifar_unet.c:1088-1097, per image: cat[b] = (a[b], skip[b]), both halves n floats
__global__ void __launch_bounds__(256) unet_concat_kernel(const float* __restrict__ a, const float* __restrict__ skip, float* __restrict__ cat, size_t n, size_t total) {
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
		const size_t b = e / (2 * n), r = e - b * 2 * n;
		cat[e] = r < n ? a[b * n + r] : skip[b * n + r - n];
	}
}
// the same 16 bytes at a time (n a multiple of 4)
__global__ void __launch_bounds__(256) unet_concat4_kernel(const float4* __restrict__ a, const float4* __restrict__ skip, float4* __restrict__ cat, size_t n4, size_t total4) {
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
		const size_t b = e / (2 * n4), r = e - b * 2 * n4;
		cat[e] = r < n4 ? a[b * n4 + r] : skip[b * n4 + r - n4];
	}
}
__global__ void __launch_bounds__(256) unet_split4_kernel(const float4* __restrict__ g_cat, float4* __restrict__ g_main, float4* __restrict__ g_skip, size_t n4, size_t total4) {
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
		const size_t b = e / (2 * n4), r = e - b * 2 * n4;
		const float4 v = g_cat[e];
		if (r < n4) g_main[b * n4 + r] = v; else g_skip[b * n4 + r - n4] = v;
	}
}
// _split_concat, :1339-1349, per image: the first half is the main path's gradient, the second the skip connection's
__global__ void __launch_bounds__(256) unet_split_kernel(const float* __restrict__ g_cat, float* __restrict__ g_main, float* __restrict__ g_skip, size_t n, size_t total) {
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
		const size_t b = e / (2 * n), r = e - b * 2 * n;
		const float v = g_cat[e];
		if (r < n) g_main[b * n + r] = v; else g_skip[b * n + r - n] = v;
	}
}
unsigned grid_of(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }
}  // namespace

extern "C" {

bla_status bla_unet_create(bla_unet** out, const bla_unet_config* cfg) { return bla_unet_create_batched(out, cfg, 1); }

bla_status bla_unet_create_batched(bla_unet** out, const bla_unet_config* cfg, int batch) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(out && cfg, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(batch >= 1 && batch <= 4096, BLA_ERR_INVALID, "batch %d", batch);
	BLA_REQUIRE(cfg->image_h > 0 && cfg->image_w > 0 && cfg->in_channels > 0 && cfg->time_dim > 0 && cfg->kernel > 0 && cfg->group_size > 0 && cfg->key_dim > 0 &&
	            cfg->dims[0] > 0 && cfg->dims[1] > 0 && cfg->dims[2] > 0 && cfg->dims[3] > 0, BLA_ERR_INVALID, "bad U-Net configuration");
	BLA_REQUIRE(cfg->image_h % 8 == 0 && cfg->image_w % 8 == 0, BLA_ERR_INVALID, "image size must be a multiple of 8 (three halvings, each undone by a x2 up-sampling)");
	bla_unet* m = new bla_unet();
	m->cfg = *cfg;
	m->batch = batch;
	const size_t B = (size_t)batch;
	const int* D = cfg->dims;
	m->H[0] = cfg->image_h; m->W[0] = cfg->image_w;
	for (int i = 1; i < 4; i++) { m->H[i] = (m->H[i - 1] + 1) / 2; m->W[i] = (m->W[i - 1] + 1) / 2; }   // RESOLUTION_n_HEIGHT, :39-46
	const int* H = m->H; const int* W = m->W;
	// parameter bucket, in the order of first use by forward()
	plan_res(m, m->res[0], "down_1_resnet_1", cfg->in_channels, D[0], H[0], W[0]);
	plan_res(m, m->res[1], "down_1_resnet_2", D[0], D[0], H[0], W[0]);
	plan_conv(m, m->down[0], "down_1_conv_kernels", D[0], D[1], H[0], W[0], 2);
	plan_res(m, m->res[2], "down_2_resnet_1", D[1], D[1], H[1], W[1]);
	plan_att(m, m->att[0], "down_2_self_attention_1", D[1], H[1], W[1]);
	plan_res(m, m->res[3], "down_2_resnet_2", D[1], D[1], H[1], W[1]);
	plan_att(m, m->att[1], "down_2_self_attention_2", D[1], H[1], W[1]);
	plan_conv(m, m->down[1], "down_2_conv_kernels", D[1], D[2], H[1], W[1], 2);
	plan_res(m, m->res[4], "down_3_resnet_1", D[2], D[2], H[2], W[2]);
	plan_res(m, m->res[5], "down_3_resnet_2", D[2], D[2], H[2], W[2]);
	plan_conv(m, m->down[2], "down_3_conv_kernels", D[2], D[3], H[2], W[2], 2);
	plan_res(m, m->res[6], "down_4_resnet_1", D[3], D[3], H[3], W[3]);
	plan_res(m, m->res[7], "down_4_resnet_2", D[3], D[3], H[3], W[3]);
	plan_res(m, m->res[8], "mid_resnet_1", D[3], D[3], H[3], W[3]);
	plan_att(m, m->att[2], "mid_self_attention", D[3], H[3], W[3]);
	plan_res(m, m->res[9], "mid_resnet_2", D[3], D[3], H[3], W[3]);
	plan_res(m, m->res[10], "up_1_resnet_1", 2 * D[3], D[3], H[3], W[3]);
	plan_res(m, m->res[11], "up_1_resnet_2", D[3], D[3], H[3], W[3]);
	plan_conv(m, m->up[0], "up_1_conv_kernels", D[3], D[2], H[2], W[2], 1, D[3] != D[2]);
	plan_res(m, m->res[12], "up_2_resnet_1", 2 * D[2], D[2], H[2], W[2]);
	plan_res(m, m->res[13], "up_2_resnet_2", D[2], D[2], H[2], W[2]);
	plan_conv(m, m->up[1], "up_2_conv_kernels", D[2], D[1], H[1], W[1], 1, D[2] != D[1]);
	plan_res(m, m->res[14], "up_3_resnet_1", 2 * D[1], D[1], H[1], W[1]);
	plan_att(m, m->att[3], "up_3_self_attention_1", D[1], H[1], W[1]);
	plan_res(m, m->res[15], "up_3_resnet_2", D[1], D[1], H[1], W[1]);
	plan_att(m, m->att[4], "up_3_self_attention_2", D[1], H[1], W[1]);
	plan_conv(m, m->up[2], "up_3_conv_kernels", D[1], D[0], H[0], W[0], 1, D[1] != D[0]);
	plan_res(m, m->res[16], "up_4_resnet_1", 2 * D[0], D[0], H[0], W[0]);
	plan_res(m, m->res[17], "up_4_resnet_2", D[0], D[0], H[0], W[0]);
	plan_conv(m, m->outc, "output_conv_kernels", D[0], cfg->in_channels, H[0], W[0], 1);

	auto fail = [&](bla_status s) { (void)bla_unet_destroy(m); return s; };
	if ((st = dalloc(m, &m->params, m->count)) || (st = dalloc(m, &m->grads, m->count))) return fail(st);
	BLA_HIP(hipMemsetAsync(m->params, 0, m->count * sizeof(float), ctx().stream));
	BLA_HIP(hipMemsetAsync(m->grads, 0, m->count * sizeof(float), ctx().stream));
	size_t max_act = 0, max_s = 0, max_flip = 0, max_cout_hw = 0, max_cin_hw = 0, max_cd = 1, max_cout = 1;
	for (Res& r : m->res) {
		if ((st = alloc_res(m, r))) return fail(st);
		const size_t hw = (size_t)r.h * r.w * B;
		max_act = std::max(max_act, (size_t)std::max(r.cin, r.cout) * hw);
		max_cout_hw = std::max(max_cout_hw, r.cout * hw); max_cin_hw = std::max(max_cin_hw, r.cin * hw); max_cout = std::max(max_cout, (size_t)r.cout);
		max_flip = std::max(max_flip, (size_t)r.cout * std::max(r.cin, r.cout) * cfg->kernel * cfg->kernel);
	}
	for (Att& a : m->att) {
		const size_t s = (size_t)a.h * a.w;
		if ((st = alloc_att_ws(m, a.fwd, s, cfg->key_dim)) || (st = dalloc(m, &a.out, B * a.c * s))) return fail(st);
		max_s = std::max(max_s, s);
		max_cd = std::max(max_cd, (size_t)a.c * cfg->key_dim);
	}
	for (int i = 0; i < 3; i++) {
		Conv& c = m->down[i];
		if ((st = dalloc(m, &c.out, B * c.cout * H[i + 1] * W[i + 1]))) return fail(st);
		max_flip = std::max(max_flip, (size_t)c.cout * c.cin * c.k * c.k);
		Conv& u = m->up[i];
		if ((st = dalloc(m, &m->nn[i], B * u.cin * u.h * u.w))) return fail(st);
		if (u.present) {
			if ((st = dalloc(m, &u.out, B * u.cout * u.h * u.w))) return fail(st);
			max_flip = std::max(max_flip, (size_t)u.cout * u.cin * u.k * u.k);
		}
		max_act = std::max(max_act, B * u.cin * u.h * u.w);
	}
	const int stage_dim[4] = {D[3], D[2], D[1], D[0]}, stage_res[4] = {3, 2, 1, 0};
	for (int i = 0; i < 4; i++) {
		const size_t n = B * stage_dim[i] * H[stage_res[i]] * W[stage_res[i]];
		if ((st = dalloc(m, &m->cat[i], 2 * n)) || (st = dalloc(m, &m->gskip[i], n))) return fail(st);
		max_act = std::max(max_act, 2 * n);
	}
	const size_t hw0 = (size_t)H[0] * W[0] * B;
	const int g0 = (D[0] + cfg->group_size - 1) / cfg->group_size * batch;
	if ((st = dalloc(m, &m->outc.out, cfg->in_channels * hw0)) || (st = dalloc(m, &m->out_relu, D[0] * hw0)) || (st = dalloc(m, &m->out_mu, g0)) ||
	    (st = dalloc(m, &m->out_sd, g0)))
		return fail(st);
	max_flip = std::max(max_flip, (size_t)cfg->in_channels * D[0] * cfg->kernel * cfg->kernel);
	static const bool side_on = [] { const char* e = getenv("BLA_UNET_SIDE"); return !(e && e[0] == '0'); }();
	m->side = side_on && batch > 1 && 48 * max_act * sizeof(float) <= ((size_t)16 << 30);   // (the write-once pool: at most 16 GiB of it -- beyond that, one stream and three rotating buffers)
	if (m->side) {
		for (int i = 0; i < 48; i++) { float* q; if ((st = dalloc(m, &q, max_act))) return fail(st); m->gpool.push_back(q); }
		for (Res& r : m->res) if ((st = dalloc(m, &r.pads.g_out_b, (size_t)r.cout * r.h * r.w * B))) return fail(st);
	}
	if ((st = dalloc(m, &m->t1, max_act)) || (st = dalloc(m, &m->t2, max_act)) || (st = dalloc(m, &m->t3, max_act)) || (st = dalloc(m, &m->sc.g_out_a, max_cout_hw)) ||
	    (st = dalloc(m, &m->sc.g_out_b, max_cout_hw)) || (st = dalloc(m, &m->sc.g_in, max_cin_hw)) || (st = dalloc(m, &m->sc.flip, max_flip)) ||
	    (st = alloc_att_ws(m, m->agrad, max_s, cfg->key_dim)) || (st = dalloc(m, &m->dtb, B * max_cout)) || (st = dalloc(m, &m->partials, B * max_cd)))
		return fail(st);
	void* z = nullptr;
	const size_t zero_bytes = m->drop_count ? B * m->drop_count : 1;
	BLA_HIP(hipMalloc(&z, zero_bytes));
	m->owned.push_back(z);
	m->zero_drop = (unsigned char*)z;
	BLA_HIP(hipMemsetAsync(z, 0, zero_bytes, ctx().stream));
	if (cfg->time_dim <= 1024) {   // the 18 time-embedding projections (and, for a batch, their gradients) as one launch each way (eight embedding rows in LDS): the jobs' addresses are fixed from here on
		TimeJob jobs[18];
		for (int i = 0; i < 18; i++) {
			const Res& r = m->res[i];
			jobs[i] = TimeJob{m->params + r.tw, m->params + r.tb, r.ws.tdense, r.dtb, m->grads + r.tw, m->grads + r.tb, r.cout};
		}
		void* tj = nullptr;
		BLA_HIP(hipMalloc(&tj, sizeof jobs));
		m->owned.push_back(tj);
		m->time_jobs = (TimeJob*)tj;
		BLA_HIP(hipMemcpy(tj, jobs, sizeof jobs, hipMemcpyHostToDevice));
	}
	static const bool prep_on = [] { const char* e = getenv("BLA_UNET_PREP"); return !(e && e[0] == '0'); }();
	if (batch > 1 && prep_on) {
		// One launch at the head of forward() and one at the head of backward() put every convolution's kernels into the form its product reads (window order,
		// flipped): 49 five-microsecond launches per pass become 2.  The parameters do not change inside a pass, so the prepared copies are valid for it.
		std::vector<KernelPrepJob> jf, jb;
		auto add = [&](size_t kern_off, int h, int w, int k, int cin, int cout, int stride, bool want_dgrad, const float** pf, const float** pb) -> bla_status {
			const size_t n = (size_t)cout * cin * k * k;
			const int mf = conv_kernel_prep_mode(batch, h, w, k, cin, cout, stride, false), mb = want_dgrad ? conv_kernel_prep_mode(batch, h, w, k, cin, cout, stride, true) : 0;
			for (int which = 0; which < 2; which++) {
				const int mode = which ? mb : mf;
				if (!mode) continue;
				float* dst;
				bla_status s2 = dalloc(m, &dst, n);
				if (s2) return s2;
				(which ? jb : jf).push_back(KernelPrepJob{m->params + kern_off, dst, cout, cin, k, mode});
				*(which ? pb : pf) = dst;
				m->prep_max = std::max(m->prep_max, n);
			}
			return BLA_OK;
		};
		for (int i = 0; i < 18; i++) {
			Res& r = m->res[i];
			if ((st = add(r.conv1, r.h, r.w, cfg->kernel, r.cin, r.cout, 1, i != 0, &r.pads.k1_fwd, &r.pads.k1_bwd)) ||     // (block 0 forms no gradient of the image)
			    (st = add(r.conv2, r.h, r.w, cfg->kernel, r.cout, r.cout, 1, true, &r.pads.k2_fwd, &r.pads.k2_bwd)))
				return fail(st);
		}
		for (int i = 0; i < 3; i++)
			if (m->up[i].present && (st = add(m->up[i].kern, m->up[i].h, m->up[i].w, m->up[i].k, m->up[i].cin, m->up[i].cout, 1, true, &m->up[i].prep_fwd, &m->up[i].prep_bwd))) return fail(st);
		auto upload = [&](const std::vector<KernelPrepJob>& v, KernelPrepJob** d, int* n) -> bla_status {
			*n = (int)v.size();
			if (v.empty()) return BLA_OK;
			void* q = nullptr;
			BLA_HIP(hipMalloc(&q, v.size() * sizeof(KernelPrepJob)));
			m->owned.push_back(q);
			BLA_HIP(hipMemcpy(q, v.data(), v.size() * sizeof(KernelPrepJob), hipMemcpyHostToDevice));
			*d = (KernelPrepJob*)q;
			return BLA_OK;
		};
		if ((st = upload(jf, &m->prep_fwd, &m->n_prep_fwd)) || (st = upload(jb, &m->prep_bwd, &m->n_prep_bwd))) return fail(st);
	}
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	*out = m;
	return BLA_OK;
}

bla_status bla_unet_destroy(bla_unet* m) {
	if (!m) return BLA_OK;
	(void)hipDeviceSynchronize();
	for (void* p : m->owned) (void)hipFree(p);
	delete m;
	return BLA_OK;
}

size_t bla_unet_param_count(const bla_unet* m) { return m ? m->count : 0; }
float* bla_unet_params(bla_unet* m) { return m ? m->params : nullptr; }
float* bla_unet_grads(bla_unet* m) { return m ? m->grads : nullptr; }
float* bla_unet_output(bla_unet* m) { return m ? m->outc.out : nullptr; }
size_t bla_unet_dropout_count(const bla_unet* m) { return m ? m->drop_count * m->batch : 0; }
int bla_unet_batch(const bla_unet* m) { return m ? m->batch : 0; }
int bla_unet_tensor_count(const bla_unet* m) { return m ? (int)m->tensors.size() : 0; }

bla_status bla_unet_tensor_info(const bla_unet* m, int index, size_t* offset, size_t* count, char* name, int name_len) {
	BLA_REQUIRE(m && index >= 0 && index < (int)m->tensors.size(), BLA_ERR_INVALID, "tensor index out of range");
	const Tensor& t = m->tensors[index];
	if (offset) *offset = t.off;
	if (count) *count = t.count;
	if (name && name_len > 0) { strncpy(name, t.name.c_str(), name_len - 1); name[name_len - 1] = '\0'; }
	return BLA_OK;
}

/* forward(), model/cifar_unet.c:1099-1166; a batched model takes [B][C][H][W] images and [B][time_dim] embeddings */
bla_status bla_unet_forward_f32(bla_unet* m, void* stream, const float* d_x, const float* d_time_embedding, const unsigned char* d_drop) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(m && d_x && d_time_embedding, BLA_ERR_INVALID, "null argument");
	const bla_unet_config& c = m->cfg;
	const int* D = c.dims; const int* H = m->H; const int* W = m->W;
	const float* P = m->params;
	const int B = m->batch;
	hipStream_t s = pick_stream(stream);
	m->last_x = d_x; m->last_temb = d_time_embedding; m->last_drop = d_drop;
	auto res = [&](int i, const float* in) -> bla_status {
		Res& r = m->res[i];
		bla_resnet_params p = {P + r.conv1, P + r.conv2, P + r.tw, P + r.tb, r.res != kNone ? P + r.res : nullptr};
		// the dropout decisions: block by block in forward order, inside a block image by image
		return resnet_forward_batched(stream, B, in, d_time_embedding, &p, (d_drop ? d_drop : m->zero_drop) + r.drop_off * B, &r.ws, r.result, r.h, r.w, r.cin,
		                              r.cout, c.kernel, c.time_dim, c.group_size, m->time_jobs ? RESNET_TDENSE_READY : 0, B > 1 ? &r.pads : nullptr);
	};
	if (m->time_jobs) {   // every block's time-embedding projection depends on the embedding alone: one launch for the 18 of them
		int max_cout = 0;
		for (int i = 0; i < 18; i++) max_cout = std::max(max_cout, m->res[i].cout);
		hipLaunchKernelGGL(time_dense_all_kernel, dim3(18, (unsigned)((B + 7) / 8), (unsigned)((max_cout + 63) / 64)), dim3(256),
		                   ((size_t)8 * c.time_dim + 4 * 8 * 64) * sizeof(float), s, m->time_jobs, d_time_embedding, B, c.time_dim);
		BLA_HIP(hipGetLastError());
	}
	auto att = [&](int i, const float* in) -> bla_status {
		Att& a = m->att[i];
		return bla_attention_forward_batched_f32(stream, B, in, P + a.wq, P + a.wk, P + a.wv, P + a.wo, P + a.bias, &a.fwd, a.out, a.c, a.h * a.w, c.key_dim);
	};
	auto conv = [&](Conv& k, const float* in) -> bla_status {
		return conv2d_forward_epilogue(stream, in, P + k.kern, k.out, k.h, k.w, k.k, k.cin, k.cout, k.stride, nullptr, nullptr, nullptr, B, 0, nullptr, k.prep_fwd);
	};
	if (m->n_prep_fwd) { st = conv_prepare_kernels(stream, m->prep_fwd, m->n_prep_fwd, m->prep_max); if (st) return st; }
	auto concat = [&](int stage, const float* a, const float* skip, size_t n) -> bla_status {   // _concat_skip, :1088-1097 (n: floats per image and half)
		if (n % 4 == 0 && ((uintptr_t)a | (uintptr_t)skip | (uintptr_t)m->cat[stage]) % 16 == 0)
			hipLaunchKernelGGL(unet_concat4_kernel, dim3(grid_of(n / 2 * B)), dim3(256), 0, s, (const float4*)a, (const float4*)skip, (float4*)m->cat[stage], n / 4, n / 2 * B);
		else hipLaunchKernelGGL(unet_concat_kernel, dim3(grid_of(2 * n * B)), dim3(256), 0, s, a, skip, m->cat[stage], n, 2 * n * B);
		BLA_HIP(hipGetLastError());
		return BLA_OK;
	};
	auto upsample = [&](int i, const float* in, const float** next) -> bla_status {   // _nearest_neighbours (+ the optional convolution), :1125-1131
		Conv& u = m->up[i];
		bla_status s2 = bla_nearest_neighbours_f32(stream, in, m->nn[i], B * u.cin, H[3 - i], W[3 - i], u.h, u.w, 2);
		*next = m->nn[i];
		if (!s2 && u.present) { s2 = conv(u, m->nn[i]); *next = u.out; }
		return s2;
	};
#define TRY(x) do { st = (x); if (st) return st; } while (0)
	// down
	TRY(res(0, d_x)); TRY(res(1, m->res[0].result));
	TRY(conv(m->down[0], m->res[1].result));
	TRY(res(2, m->down[0].out)); TRY(att(0, m->res[2].result)); TRY(res(3, m->att[0].out)); TRY(att(1, m->res[3].result));
	TRY(conv(m->down[1], m->att[1].out));
	TRY(res(4, m->down[1].out)); TRY(res(5, m->res[4].result));
	TRY(conv(m->down[2], m->res[5].result));
	TRY(res(6, m->down[2].out)); TRY(res(7, m->res[6].result));
	// mid
	TRY(res(8, m->res[7].result)); TRY(att(2, m->res[8].result)); TRY(res(9, m->att[2].out));
	// up
	const float* next;
	TRY(concat(0, m->res[9].result, m->res[7].result, (size_t)D[3] * H[3] * W[3]));
	TRY(res(10, m->cat[0])); TRY(res(11, m->res[10].result));
	TRY(upsample(0, m->res[11].result, &next));
	TRY(concat(1, next, m->res[5].result, (size_t)D[2] * H[2] * W[2]));
	TRY(res(12, m->cat[1])); TRY(res(13, m->res[12].result));
	TRY(upsample(1, m->res[13].result, &next));
	TRY(concat(2, next, m->res[3].result, (size_t)D[1] * H[1] * W[1]));
	TRY(res(14, m->cat[2])); TRY(att(3, m->res[14].result)); TRY(res(15, m->att[3].out)); TRY(att(4, m->res[15].result));
	TRY(upsample(2, m->att[4].out, &next));
	TRY(concat(3, next, m->res[1].result, (size_t)D[0] * H[0] * W[0]));
	TRY(res(16, m->cat[3])); TRY(res(17, m->res[16].result));
	// output, :1163-1165
	TRY(bla_group_norm_relu_batched_f32(stream, B, m->res[17].result, m->out_relu, m->out_sd, m->out_mu, D[0], c.group_size, H[0] * W[0]));
	TRY(conv(m->outc, m->out_relu));
	return BLA_OK;
}

/* backward(), model/cifar_unet.c:1351-1436 (intended wiring, see the head of this file): fills the gradient bucket for the images, time
 * embeddings and dropout decisions of the last forward pass (summed over the images of a batch). */
bla_status bla_unet_backward_f32(bla_unet* m, void* stream, const float* d_noise) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(m && d_noise, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(m->last_x, BLA_ERR_INVALID, "bla_unet_forward_f32 has not run");
	const bla_unet_config& c = m->cfg;
	const int* D = c.dims; const int* H = m->H; const int* W = m->W;
	const float* P = m->params; float* G = m->grads;
	const float* temb = m->last_temb;
	const int B = m->batch;
	hipStream_t s = pick_stream(stream);
	// ResNet block i: gradient `g` of its result -> `out` (gradient of its input `x`)
	auto res = [&](int i, const float* g, const float* x, float* out) -> bla_status {
		Res& r = m->res[i];
		bla_resnet_params p = {P + r.conv1, P + r.conv2, P + r.tw, P + r.tb, r.res != kNone ? P + r.res : nullptr};
		bla_resnet_grads gr = {G + r.conv1, G + r.conv2, G + r.tw, G + r.tb, r.res != kNone ? G + r.res : nullptr};
		return resnet_backward_batched(stream, B, g, x, temb, &p, &r.ws, &gr, &m->sc, B > 1 ? r.dtb : m->dtb, out, r.h, r.w, r.cin, r.cout, c.kernel, c.time_dim,
		                               c.group_size, (B > 1 && m->time_jobs ? RESNET_DEFER_TIME_GRADS : 0) | (m->side ? RESNET_WGRAD_SIDE : 0), B > 1 ? &r.pads : nullptr);
	};
	auto att = [&](int i, const float* g, const float* x, float* out) -> bla_status {
		Att& a = m->att[i];
		return bla_attention_backward_batched_f32(stream, B, g, x, P + a.wq, P + a.wk, P + a.wv, P + a.wo, &a.fwd, &m->agrad, m->partials, G + a.wq, G + a.wk, G + a.wv,
		                                          G + a.wo, out, a.c, a.h * a.w, c.key_dim, 0);
	};
	auto conv = [&](Conv& k, const float* g, const float* x, float* out) -> bla_status {
		if (m->side) {      // weight gradient on the side lane (g and x stay intact until the join below), data gradient here
			hipStream_t lane;
			bla_status s2 = side_lane_fork(s, &lane);
			if (s2) return s2;
			s2 = conv2d_backward_batched(lane, g, x, P + k.kern, G + k.kern, nullptr, m->sc.flip, B, k.h, k.w, k.k, k.cin, k.cout, k.stride, nullptr, nullptr);
			side_lane_done();
			if (s2) return s2;
			return conv2d_backward_batched(stream, g, x, P + k.kern, nullptr, out, m->sc.flip, B, k.h, k.w, k.k, k.cin, k.cout, k.stride, nullptr, k.prep_bwd);
		}
		return conv2d_backward_batched(stream, g, x, P + k.kern, G + k.kern, out, m->sc.flip, B, k.h, k.w, k.k, k.cin, k.cout, k.stride, nullptr, k.prep_bwd);
	};
	if (m->n_prep_bwd) { st = conv_prepare_kernels(stream, m->prep_bwd, m->n_prep_bwd, m->prep_max); if (st) return st; }
	// up-sampling stage i backwards: gradient of (the optional convolution's output | the resized map) -> gradient of the map before resizing;
	// g, tmp and out are three different buffers
	auto upsample = [&](int i, const float* g, float* tmp, float* out) -> bla_status {
		Conv& u = m->up[i];
		const float* gn = g;
		if (u.present) { bla_status s2 = conv(u, g, m->nn[i], tmp); if (s2) return s2; gn = tmp; }
		return bla_nearest_neighbours_ddx_f32(stream, gn, out, B * u.cin, u.h, u.w, H[3 - i], W[3 - i], 2);
	};
	// _split_concat, :1339-1349: the first half of a concatenation's gradient goes on along the main path, the second is the skip connection's
	auto split = [&](int stage, const float* g_cat, float* g_main, size_t n) -> bla_status {
		if (n % 4 == 0 && ((uintptr_t)g_cat | (uintptr_t)g_main | (uintptr_t)m->gskip[stage]) % 16 == 0)
			hipLaunchKernelGGL(unet_split4_kernel, dim3(grid_of(n / 2 * B)), dim3(256), 0, s, (const float4*)g_cat, (float4*)g_main, (float4*)m->gskip[stage], n / 4, n / 2 * B);
		else hipLaunchKernelGGL(unet_split_kernel, dim3(grid_of(2 * n * B)), dim3(256), 0, s, g_cat, g_main, m->gskip[stage], n, 2 * n * B);
		BLA_HIP(hipGetLastError());
		return BLA_OK;
	};
	// three rotating gradient buffers: a block never writes the buffer it reads.  With the weight gradients on the side lane every WRITE takes a fresh buffer
	// from the pool instead (F(x) below), so what the lane still reads -- the gradient a block or convolution came in with -- is never overwritten in this pass
	float *a = m->t1, *b = m->t2, *cbuf = m->t3;
	size_t gp = 0;
	auto F = [&](float*& p) -> float* { if (m->side) p = m->gpool[gp++ % m->gpool.size()]; return p; };
	const size_t hw0 = (size_t)H[0] * W[0];
	const size_t n0 = (size_t)D[0] * hw0, n1 = (size_t)D[1] * H[1] * W[1], n2 = (size_t)D[2] * H[2] * W[2], n3 = (size_t)D[3] * H[3] * W[3];
	const int nout = (int)(c.in_channels * hw0 * B);
	hipLaunchKernelGGL(unet_loss_grad_kernel, dim3(grid_of((size_t)nout)), dim3(256), 0, s, m->outc.out, d_noise, F(a), nout);   // :1353-1364
	BLA_HIP(hipGetLastError());
	// output processing, :1367-1369: convolution, ReLU gate, group norm
	TRY(conv(m->outc, a, m->out_relu, F(b)));
	TRY(bla_group_norm_ddx_gated_batched_f32(stream, B, b, F(a), m->res[17].result, m->out_mu, m->out_sd, D[0], c.group_size, (int)hw0, m->out_relu, nullptr));
	// fourth up-sampling stage, :1372-1374
	TRY(res(17, a, m->res[16].result, F(b))); TRY(res(16, b, m->cat[3], F(a)));
	TRY(split(3, a, F(b), n0));
	// third, :1377-1383 (resize and the optional convolution, then attention 2, ResNet 2, attention 1, ResNet 1)
	TRY(upsample(2, b, F(a), F(cbuf)));
	TRY(att(4, cbuf, m->res[15].result, F(a))); TRY(res(15, a, m->att[3].out, F(b))); TRY(att(3, b, m->res[14].result, F(a))); TRY(res(14, a, m->cat[2], F(b)));
	TRY(split(2, b, F(a), n1));
	// second, :1386-1390
	TRY(upsample(1, a, F(b), F(cbuf)));
	TRY(res(13, cbuf, m->res[12].result, F(a))); TRY(res(12, a, m->cat[1], F(b)));
	TRY(split(1, b, F(a), n2));
	// first, :1393-1397
	TRY(upsample(0, a, F(b), F(cbuf)));
	TRY(res(11, cbuf, m->res[10].result, F(a))); TRY(res(10, a, m->cat[0], F(b)));
	TRY(split(0, b, F(a), n3));
	// middle, :1400-1402
	TRY(res(9, a, m->att[2].out, F(b))); TRY(att(2, b, m->res[8].result, F(a))); TRY(res(8, a, m->res[7].result, F(b)));
	// fourth down-sampling stage, :1405-1409: the skip's gradient joins the main path's
	TRY(bla_add_f32(stream, b, m->gskip[0], n3 * B));
	TRY(res(7, b, m->res[6].result, F(a))); TRY(res(6, a, m->down[2].out, F(b)));
	// third, :1412-1417
	TRY(conv(m->down[2], b, m->res[5].result, F(a)));
	TRY(bla_add_f32(stream, a, m->gskip[1], n2 * B));
	TRY(res(5, a, m->res[4].result, F(b))); TRY(res(4, b, m->down[1].out, F(a)));
	// second, :1420-1427
	TRY(conv(m->down[1], a, m->att[1].out, F(b)));
	TRY(att(1, b, m->res[3].result, F(a)));
	TRY(bla_add_f32(stream, a, m->gskip[2], n1 * B));
	TRY(res(3, a, m->att[0].out, F(b))); TRY(att(0, b, m->res[2].result, F(a))); TRY(res(2, a, m->down[0].out, F(b)));
	// first, :1430-1435
	TRY(conv(m->down[0], b, m->res[1].result, F(a)));
	TRY(bla_add_f32(stream, a, m->gskip[3], n0 * B));
	TRY(res(1, a, m->res[0].result, F(b)));
	TRY(res(0, b, m->last_x, nullptr));   // nothing consumes the gradient of the image: the first block forms its weight gradients only
	if (B > 1 && m->time_jobs) {   // the 18 blocks' time-weight / time-bias gradients from the channel sums each block left behind (:1191-1199)
		hipLaunchKernelGGL(time_grads_all_kernel, dim3(18, (unsigned)((c.time_dim + 7) / 8)), dim3(256), 0, s, m->time_jobs, temb, B, c.time_dim);
		BLA_HIP(hipGetLastError());
	}
	if (m->side) TRY(side_lane_join(s));      // the weight gradients are in when this stream moves on
	return BLA_OK;
}
#undef TRY

}  // extern "C"
