// bla_layer.hip -- lib/layer.h on the device, batched: the reference's generic dense-layer chain (feed_forward,
// lib/layer.c:6-20; back_propagate_errors with its recursion toward the input, :48-107) for B samples (columns) at once, parameters resident.
// The reference works on one column vector per call with host function pointers for the activation; here the activation is one of a small
// set of device functions and every product is n x p . p x B on the MFMA GEMM.  With B = 1 the arithmetic is layer.c's, step by step:
//   forward   raw = W a_prev + b ; nodes = act(raw)                                                     (:10-17)
//   backward  g_L = 2 (a_L - y)                                                                          (:87-90)
//             h_l = act'(raw_l) (.) g_l ;  delta_l = h_l * (-learn_rate)                                 (:62-65, 92-95)
//             g_{l-1} = W_l^T h_l      with the weights as they were BEFORE this call                    (:53-59: the recursion runs before any update)
//             W_l += delta_l a_{l-1}^T ;  b_l += delta_l   (summed over the B columns)                   (:67-75, 97-105)
#include "bla_internal.h"
#include <vector>

using namespace bla;

struct bla_layer_net {
	int layers, batch;                 // layers counts the input layer
	std::vector<int> n, act;
	std::vector<float> act_p;
	std::vector<size_t> w_off, b_off;  // per layer l >= 1
	size_t count;
	float* params;
	std::vector<float*> raw, nodes;    // [n_l][B], l >= 1 (nodes[0] = the caller's input)
	float *g, *g2, *h, *delta;         // scratch [max n][B]
	const float* input;
	std::vector<void*> owned;
};

namespace {
__device__ __forceinline__ float act_f(int act, float p, float x) {
	switch (act) {
		case BLA_LAYER_ACT_SCALE: return x * p;
		case BLA_LAYER_ACT_RELU: return x < 0.f ? 0.f : x;
		case BLA_LAYER_ACT_LEAKY: return x < 0.f ? x * p : x;
		default: return x;
	}
}
__device__ __forceinline__ float act_ddx_f(int act, float p, float x) {
	switch (act) {
		case BLA_LAYER_ACT_SCALE: return p;
		case BLA_LAYER_ACT_RELU: return x > 0.f ? 1.f : 0.f;
		case BLA_LAYER_ACT_LEAKY: return x > 0.f ? 1.f : p;
		default: return 1.f;
	}
}
__global__ void __launch_bounds__(256) layer_act_kernel(const float* __restrict__ raw, float* __restrict__ nodes, int act, float p, size_t n) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) nodes[i] = act_f(act, p, raw[i]);
}
// g == nullptr: the output layer, g = 2 (nodes - expect)   (lib/layer.c:87-90)
__global__ void __launch_bounds__(256) layer_delta_kernel(const float* __restrict__ raw, const float* __restrict__ g, const float* __restrict__ nodes,
                                                          const float* __restrict__ expect, float* __restrict__ h, float* __restrict__ delta, int act, float p,
                                                          float neg_lr, size_t n) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		const float gi = g ? g[i] : 2.f * (nodes[i] - expect[i]);
		const float hi = act_ddx_f(act, p, raw[i]) * gi;
		h[i] = hi;
		delta[i] = hi * neg_lr;
	}
}
unsigned blocks_for(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }
bla_status lalloc(bla_layer_net* m, float** p, size_t floats) {
	void* q = nullptr;
	BLA_HIP(hipMalloc(&q, (floats ? floats : 1) * sizeof(float)));
	m->owned.push_back(q);
	*p = (float*)q;
	return BLA_OK;
}
}  // namespace

extern "C" {

bla_status bla_layer_net_create(bla_layer_net** out, const int* sizes, int num_layers, int batch, const int* acts, const float* act_params) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(out && sizes && acts && num_layers >= 2 && batch > 0, BLA_ERR_INVALID, "need at least an input and one computing layer");
	for (int l = 0; l < num_layers; l++) BLA_REQUIRE(sizes[l] > 0, BLA_ERR_INVALID, "layer %d has %d nodes", l, sizes[l]);
	bla_layer_net* m = new bla_layer_net();
	m->layers = num_layers; m->batch = batch; m->input = nullptr;
	m->n.assign(sizes, sizes + num_layers);
	m->act.assign(num_layers, BLA_LAYER_ACT_IDENTITY); m->act_p.assign(num_layers, 0.f);
	m->w_off.assign(num_layers, 0); m->b_off.assign(num_layers, 0);
	m->raw.assign(num_layers, nullptr); m->nodes.assign(num_layers, nullptr);
	size_t o = 0; int widest = 0;
	for (int l = 1; l < num_layers; l++) {
		m->act[l] = acts[l - 1]; m->act_p[l] = act_params ? act_params[l - 1] : 0.f;
		m->w_off[l] = o; o += (size_t)sizes[l] * sizes[l - 1];
		m->b_off[l] = o; o += (size_t)sizes[l];
		o = (o + 3) / 4 * 4;
		widest = sizes[l] > widest ? sizes[l] : widest;
	}
	m->count = o;
	auto fail = [&](bla_status s) { (void)bla_layer_net_destroy(m); return s; };
	if ((st = lalloc(m, &m->params, o))) return fail(st);
	for (int l = 1; l < num_layers; l++)
		if ((st = lalloc(m, &m->raw[l], (size_t)sizes[l] * batch)) || (st = lalloc(m, &m->nodes[l], (size_t)sizes[l] * batch))) return fail(st);
	const size_t sc = (size_t)(widest > sizes[0] ? widest : sizes[0]) * batch;
	if ((st = lalloc(m, &m->g, sc)) || (st = lalloc(m, &m->g2, sc)) || (st = lalloc(m, &m->h, sc)) || (st = lalloc(m, &m->delta, sc))) return fail(st);
	BLA_HIP(hipMemsetAsync(m->params, 0, o * sizeof(float), ctx().stream));
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	*out = m;
	return BLA_OK;
}

bla_status bla_layer_net_destroy(bla_layer_net* m) {
	if (!m) return BLA_OK;
	(void)hipDeviceSynchronize();
	for (void* p : m->owned) (void)hipFree(p);
	delete m;
	return BLA_OK;
}

size_t bla_layer_net_param_count(const bla_layer_net* m) { return m ? m->count : 0; }
float* bla_layer_net_params(bla_layer_net* m) { return m ? m->params : nullptr; }
float* bla_layer_net_weights(bla_layer_net* m, int layer) { return m && layer >= 1 && layer < m->layers ? m->params + m->w_off[layer] : nullptr; }
float* bla_layer_net_biases(bla_layer_net* m, int layer) { return m && layer >= 1 && layer < m->layers ? m->params + m->b_off[layer] : nullptr; }
float* bla_layer_net_nodes(bla_layer_net* m, int layer) { return m && layer >= 1 && layer < m->layers ? m->nodes[layer] : nullptr; }
float* bla_layer_net_raw_nodes(bla_layer_net* m, int layer) { return m && layer >= 1 && layer < m->layers ? m->raw[layer] : nullptr; }

/* feed_forward on every computing layer, lib/layer.c:6-20; d_x: [n0][B] */
bla_status bla_layer_net_forward_f32(bla_layer_net* m, void* stream, const float* d_x) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(m && d_x, BLA_ERR_INVALID, "null argument");
	hipStream_t s = pick_stream(stream);
	const int B = m->batch;
	m->input = d_x;
	const float* prev = d_x;
	for (int l = 1; l < m->layers; l++) {
		bla_gemm_epilogue ep = {};
		ep.alpha = 1.f; ep.bias_row = m->params + m->b_off[l];                                       // matrix_multiply + matrix_add (:10-11)
		st = bla_gemm_f32(s, 0, 0, m->n[l], B, m->n[l - 1], m->params + m->w_off[l], m->n[l - 1], prev, B, m->raw[l], B, &ep);
		if (st) return st;
		const size_t n = (size_t)m->n[l] * B;
		hipLaunchKernelGGL(layer_act_kernel, dim3(blocks_for(n)), dim3(256), 0, s, m->raw[l], m->nodes[l], m->act[l], m->act_p[l], n);   // :16-17
		BLA_HIP(hipGetLastError());
		prev = m->nodes[l];
	}
	return BLA_OK;
}

/* back_propagate_errors (lib/layer.c:80-107) and its recursion (:48-78) on the activations of the last forward pass; d_expect: [n_L][B] */
bla_status bla_layer_net_backward_f32(bla_layer_net* m, void* stream, const float* d_expect, float learn_rate) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(m && d_expect, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(m->input, BLA_ERR_INVALID, "bla_layer_net_forward_f32 has not run");
	hipStream_t s = pick_stream(stream);
	const int B = m->batch, L = m->layers - 1;
	float *g = nullptr, *gn = m->g;     // g: cost gradient w.r.t. this layer's activation (nullptr = the output layer computes it itself)
	for (int l = L; l >= 1; l--) {
		const size_t n = (size_t)m->n[l] * B;
		hipLaunchKernelGGL(layer_delta_kernel, dim3(blocks_for(n)), dim3(256), 0, s, m->raw[l], g, m->nodes[l], d_expect, m->h, m->delta, m->act[l], m->act_p[l],
		                   -learn_rate, n);
		BLA_HIP(hipGetLastError());
		float* W = m->params + m->w_off[l];
		if (l > 1) {   // the gradient for the layer below, from the weights as they still are (:53-59)
			st = bla_gemm_f32(s, 1, 0, m->n[l - 1], B, m->n[l], W, m->n[l - 1], m->h, B, gn, B, nullptr);
			if (st) return st;
		}
		// W += delta . a_prev^T, b += rowsum(delta)   (:67-69 / :97-99 the rank-1 product, :72-73 / :102-103 the update)
		bla_gemm_epilogue ep = {};
		ep.alpha = 1.f; ep.beta = 1.f; ep.row_sum_a = m->params + m->b_off[l]; ep.row_sum_alpha = 1.f; ep.row_sum_beta = 1.f;
		const float* a_prev = l > 1 ? m->nodes[l - 1] : m->input;
		st = bla_gemm_f32(s, 0, 1, m->n[l], m->n[l - 1], B, m->delta, B, a_prev, B, W, m->n[l - 1], &ep);
		if (st == BLA_ERR_INVALID) {   // a product too large for the latency-bound kernel (the accumulated row sum lives there): the bias step as its own pass
			ep.row_sum_a = nullptr;
			st = bla_gemm_f32(s, 0, 1, m->n[l], m->n[l - 1], B, m->delta, B, a_prev, B, W, m->n[l - 1], &ep);
			if (st) return st;
			st = bla_col_sum_f32(s, m->delta, m->n[l], B, m->h, BLA_COLSUM_INTENDED);   // (h of this layer has been consumed above)
			if (st) return st;
			st = bla_add_f32(s, m->params + m->b_off[l], m->h, (size_t)m->n[l]);
		}
		if (st) return st;
		g = gn; gn = gn == m->g ? m->g2 : m->g;
	}
	return BLA_OK;
}

}  // extern "C"
