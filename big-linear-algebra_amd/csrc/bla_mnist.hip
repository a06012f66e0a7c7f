// bla_mnist.hip -- device-resident trainer for the reference's MNIST-NN (model/mnist_nn.c): the hot loop
// :218-315 (scale, 3 x {GEMM, bias, activation}, loss gradient, 5 transposed GEMMs, bias gradients, SGD
// update) with parameters, activations and gradients resident in HBM and no host round trip per batch.
//
// Why this exists beside the drop-in matrix.h: the reference program mutates ->data on the host between
// library calls (mnist_nn.c:38-51,204-217), so the host-coherent API can never keep tensors on the device;
// samples/s is therefore measured on this trainer, which reproduces the same arithmetic step by step
// (tests/test_mnist_gpu.py pins it to the golden vectors the reference produced).
//
// Data layout: samples are COLUMNS (input [n0][B], one-hot labels [n3][B], model/mnist_nn.c:199,209).
// Parameters live in ONE flat fp32 bucket in the order W1,b1,W2,b2,W3,b3 (row-major, the CSV order of
// data/mnist_nn/*.csv); gradients in a second bucket with the same layout -- the unit a data-parallel
// all-reduce exchanges (SUM over ranks: every weight gradient is a plain sum over batch columns, no 1/B,
// model/mnist_nn.c:260-293), after which every rank applies the identical update.
//
// Fusions relative to the reference's call sequence (all are re-orderings of independent elementwise work):
//   matrix_multiply + matrix_add_tile_columns + clone_matrix + relu  -> one GEMM with bias/ReLU epilogue that
//        stores Z (pre-activation) and A                                        (:221-229)
//   clone + scale(-1) + add + scale(1/n0) around softmax              -> softmax kernel with gradient tail (:233-268)
//   transpose + multiply + transpose                                  -> NT / TN GEMM                     (:267-292)
//   clone + relu_ddx + multiply_elementwise                           -> GEMM epilogue mask on Z          (:276-278,287-289)
//   6 x {clip (no-op), scale(lr), add}                                -> one axpy over the flat bucket     (:296-315)
#include "bla_internal.h"
#include <vector>

struct bla_mnist_nn {
	int n[4];
	int batch;
	size_t off[6];        // offsets of W1,b1,W2,b2,W3,b3 in the buckets
	size_t count;         // total parameters
	float* params; float* grads; bool own_buckets;
	float* x;             // scaled input [n0][B]
	float* x_raw; float* y;   // optional resident input/label buffers (graph replay)
	float *z1, *a1, *z2, *a2, *z3, *a3, *dz3, *dz2, *dz1;
	std::vector<void*> owned;
	hipGraph_t graph; hipGraphExec_t graph_exec; bool graph_ready; int graph_colsum; float graph_lr;
	// data-parallel step: one graph per gradient-bucket parity (the exchange object double-buffers the bucket)
	hipGraph_t dp_graph[2]; hipGraphExec_t dp_exec[2]; bool dp_ready[2]; unsigned long long dp_bound_id; float dp_lr;
	// loss / accuracy bookkeeping (model/mnist_nn.c:237-257): per-column device accumulators, fed by the output layer's launch when enabled
	double* m_loss; unsigned* m_correct; bool metrics_on;
};

using namespace bla;


static bla_status dev_alloc(bla_mnist_nn* nn, float** p, size_t floats) {
	void* q = nullptr;
	BLA_HIP(hipMalloc(&q, (floats ? floats : 1) * sizeof(float)));
	nn->owned.push_back(q);
	*p = (float*)q;
	return BLA_OK;
}

extern "C" {

bla_status bla_mnist_nn_create(bla_mnist_nn** out, const int* sizes, int batch) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(out && sizes, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(batch > 0 && sizes[0] > 0 && sizes[1] > 0 && sizes[2] > 0 && sizes[3] > 0, BLA_ERR_INVALID, "bad layer sizes / batch");
	bla_mnist_nn* nn = new bla_mnist_nn();
	for (int i = 0; i < 4; i++) nn->n[i] = sizes[i];
	nn->batch = batch;
	size_t o = 0;
	for (int l = 0; l < 3; l++) {
		nn->off[2 * l] = o; o += (size_t)sizes[l + 1] * sizes[l];
		nn->off[2 * l + 1] = o; o += (size_t)sizes[l + 1];
	}
	// keep every W 16-byte aligned inside the bucket when the sizes allow it (784*256, 256, ... all are multiples of 4
	// for the reference architecture; odd architectures fall back to the scalar-load GEMM variant automatically)
	nn->count = o;
	nn->own_buckets = true;
	nn->graph_ready = false;
	nn->dp_ready[0] = nn->dp_ready[1] = false; nn->dp_bound_id = 0; nn->dp_lr = 0.f;
	nn->m_loss = nullptr; nn->m_correct = nullptr; nn->metrics_on = false;
	const size_t B = batch;
	st = dev_alloc(nn, &nn->params, o); if (st) return st;
	st = dev_alloc(nn, &nn->grads, o); if (st) return st;
	st = dev_alloc(nn, &nn->x, sizes[0] * B); if (st) return st;
	st = dev_alloc(nn, &nn->x_raw, sizes[0] * B); if (st) return st;
	st = dev_alloc(nn, &nn->y, sizes[3] * B); if (st) return st;
	st = dev_alloc(nn, &nn->z1, sizes[1] * B); if (st) return st;
	st = dev_alloc(nn, &nn->a1, sizes[1] * B); if (st) return st;
	st = dev_alloc(nn, &nn->z2, sizes[2] * B); if (st) return st;
	st = dev_alloc(nn, &nn->a2, sizes[2] * B); if (st) return st;
	st = dev_alloc(nn, &nn->z3, sizes[3] * B); if (st) return st;
	st = dev_alloc(nn, &nn->a3, sizes[3] * B); if (st) return st;
	st = dev_alloc(nn, &nn->dz3, sizes[3] * B); if (st) return st;
	st = dev_alloc(nn, &nn->dz2, sizes[2] * B); if (st) return st;
	st = dev_alloc(nn, &nn->dz1, sizes[1] * B); if (st) return st;
	// on the context's stream and waited for: that stream is non-blocking, a NULL-stream memset would not order against the first step
	BLA_HIP(hipMemsetAsync(nn->params, 0, o * sizeof(float), ctx().stream));
	BLA_HIP(hipMemsetAsync(nn->grads, 0, o * sizeof(float), ctx().stream));
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	void* ws;   // split-K slabs: grow the shared workspace now so that no step (or graph capture) ever reallocates
	st = ensure_workspace((size_t)64 << 20, &ws); if (st) return st;
	*out = nn;
	return BLA_OK;
}

bla_status bla_mnist_nn_destroy(bla_mnist_nn* nn) {
	if (!nn) return BLA_OK;
	if (nn->graph_ready) { (void)hipGraphExecDestroy(nn->graph_exec); (void)hipGraphDestroy(nn->graph); }
	for (int i = 0; i < 2; i++)
		if (nn->dp_ready[i]) { (void)hipGraphExecDestroy(nn->dp_exec[i]); (void)hipGraphDestroy(nn->dp_graph[i]); }
	for (void* p : nn->owned) (void)hipFree(p);
	delete nn;
	return BLA_OK;
}

size_t bla_mnist_nn_param_count(const bla_mnist_nn* nn) { return nn ? nn->count : 0; }
float* bla_mnist_nn_params(bla_mnist_nn* nn) { return nn ? nn->params : nullptr; }
float* bla_mnist_nn_grads(bla_mnist_nn* nn) { return nn ? nn->grads : nullptr; }
float* bla_mnist_nn_input(bla_mnist_nn* nn) { return nn ? nn->x_raw : nullptr; }
float* bla_mnist_nn_labels(bla_mnist_nn* nn) { return nn ? nn->y : nullptr; }

bla_status bla_mnist_nn_use_buckets(bla_mnist_nn* nn, float* d_params, float* d_grads) {
	BLA_REQUIRE(nn && d_params && d_grads, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(!nn->graph_ready, BLA_ERR_INVALID, "buckets cannot change after a graph was captured");
	for (int i = 0; i < 2; i++)   // recorded data-parallel steps hold the old parameter bucket: drop them, the next dp_step records again
		if (nn->dp_ready[i]) { (void)hipGraphExecDestroy(nn->dp_exec[i]); (void)hipGraphDestroy(nn->dp_graph[i]); nn->dp_ready[i] = false; }
	BLA_HIP(hipDeviceSynchronize());
	BLA_HIP(hipMemcpyAsync(d_params, nn->params, nn->count * sizeof(float), hipMemcpyDeviceToDevice, ctx().stream));
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	nn->params = d_params; nn->grads = d_grads; nn->own_buckets = false;
	return BLA_OK;
}

bla_status bla_mnist_nn_set_params(bla_mnist_nn* nn, const float* h_flat) {
	BLA_REQUIRE(nn && h_flat, BLA_ERR_INVALID, "null argument");
	// On the context's own (non-blocking) stream and waited for: a NULL-stream hipMemcpy from pageable memory returns once the bytes are
	// staged, its DMA is only ordered against blocking streams -- a step launched right behind it on the context stream read stale weights
	// (seen from the C trainer, whose host side is fast enough to get there first).
	BLA_HIP(hipDeviceSynchronize());
	BLA_HIP(hipMemcpyAsync(nn->params, h_flat, nn->count * sizeof(float), hipMemcpyHostToDevice, ctx().stream));
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	return BLA_OK;
}

bla_status bla_mnist_nn_get_params(bla_mnist_nn* nn, float* h_flat) {
	BLA_REQUIRE(nn && h_flat, BLA_ERR_INVALID, "null argument");
	BLA_HIP(hipDeviceSynchronize());
	BLA_HIP(hipMemcpyAsync(h_flat, nn->params, nn->count * sizeof(float), hipMemcpyDeviceToHost, ctx().stream));
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	return BLA_OK;
}

/* which: 0..8 = z1,a1,z2,a2,z3,a3,dz3,dz2,dz1 (device pointers, [rows][B]) -- for parity tests */
bla_status bla_mnist_nn_activation(bla_mnist_nn* nn, int which, float** d_ptr, int* rows) {
	BLA_REQUIRE(nn && d_ptr && rows && which >= 0 && which < 9, BLA_ERR_INVALID, "bad argument");
	float* p[9] = {nn->z1, nn->a1, nn->z2, nn->a2, nn->z3, nn->a3, nn->dz3, nn->dz2, nn->dz1};
	int r[9] = {nn->n[1], nn->n[1], nn->n[2], nn->n[2], nn->n[3], nn->n[3], nn->n[3], nn->n[2], nn->n[1]};
	*d_ptr = p[which]; *rows = r[which];
	return BLA_OK;
}

/* Forward + backward for one batch: fills the gradient bucket (un-scaled sums over the batch columns).
 * d_x_raw: [n0][B] pixels 0..255 (scaled by 1/255.0F here, model/mnist_nn.c:218); d_y: one-hot [n3][B].
 * NULL for either means "use the trainer's resident input / label buffer". */
static bla_status forward_backward_into(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, int colsum_mode, float* grads,
                                        const DoneHook* hook = nullptr, bool* posted = nullptr);

bla_status bla_mnist_nn_forward_backward(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, int colsum_mode) {
	BLA_REQUIRE(nn, BLA_ERR_INVALID, "null trainer");
	return forward_backward_into(nn, stream, d_x_raw, d_y, colsum_mode, nn->grads);
}

/* Z1 = W1 (x / 255) + b1, A1 = relu; Z2, A2 likewise; Z3 = W3 A2 + b3, A3 = softmax per column, dZ3 = (A3 - Y) / n0; with the
 * loss / accuracy accumulators fed from the output layer's launch when enabled (model/mnist_nn.c:218-268). */
static bla_status forward_pass(bla_mnist_nn* nn, hipStream_t s, const float* d_x_raw, const float* d_y, bool with_backward) {
	const int n0 = nn->n[0], n1 = nn->n[1], n2 = nn->n[2], n3 = nn->n[3], B = nn->batch;
	float *W1 = nn->params + nn->off[0], *b1 = nn->params + nn->off[1], *W2 = nn->params + nn->off[2], *b2 = nn->params + nn->off[3];
	float *W3 = nn->params + nn->off[4], *b3 = nn->params + nn->off[5];
	const float xs = 1 / 255.0F;
	bla_status st;
	bla_gemm_epilogue ep = {};
	ep.alpha = xs; ep.bias_row = b1; ep.pre_act = nn->z1; ep.ld_pre = B; ep.act = BLA_ACT_RELU;
	st = bla_gemm_f32(s, 0, 0, n1, B, n0, W1, n0, d_x_raw, B, nn->a1, B, &ep); if (st) return st;         // :221-224
	ep.alpha = 1.f; ep.bias_row = b2; ep.pre_act = nn->z2;
	st = bla_gemm_f32(s, 0, 0, n2, B, n1, W2, n1, nn->a1, B, nn->a2, B, &ep); if (st) return st;          // :226-229
	// output layer: Z3 = W3 A2 + b3, A3 = softmax per column, dZ3 = (A3 - Y) * (1/n0)   (:231-234,260-268;
	// scale = 1 / (double) LAYER_INPUT_SIZE).  Fused into the product's tail when the layer fits one tile row.
	ep.bias_row = b3; ep.pre_act = nn->z3; ep.act = BLA_ACT_NONE;
	const float gscale = (float)(1 / (double)n0);
	if (n3 <= 32) {
		ep.softmax_y = d_y; ep.softmax_scale = gscale; ep.softmax_grad = nn->dz3;
		if (nn->metrics_on) { ep.softmax_loss_acc = nn->m_loss; ep.softmax_correct_acc = nn->m_correct; }
		st = bla_gemm_f32(s, 0, 0, n3, B, n2, W3, n2, nn->a2, B, nn->a3, B, &ep); if (st) return st;
	} else {
		st = bla_gemm_f32(s, 0, 0, n3, B, n2, W3, n2, nn->a2, B, nn->a3, B, &ep); if (st) return st;
		st = bla_softmax_cols_grad_f32(s, nn->a3, n3, B, d_y, gscale, nn->dz3); if (st) return st;
	}
	return BLA_OK;
}

// hook: the data-parallel step's "gradients ready" post -- handed to the LAST gradient launch (the {dW2 | dW1} pair), whose last workgroup tells the
// peers; *posted says whether that happened (otherwise the exchange launch pushes the flags itself, as before)
static bla_status forward_backward_into(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, int colsum_mode, float* grads, const DoneHook* hook,
                                        bool* posted) {
	if (posted) *posted = false;
	bla_status st = require_ready();
	if (st) return st;
	if (!d_x_raw) d_x_raw = nn->x_raw;
	if (!d_y) d_y = nn->y;
	const int n0 = nn->n[0], n1 = nn->n[1], n2 = nn->n[2], n3 = nn->n[3], B = nn->batch;
	if (colsum_mode == BLA_COLSUM_AS_WRITTEN && (n1 > B || n2 > B || n3 > B)) {
		set_error("matrix_col_sum as written is out of bounds for layer sizes %d/%d/%d at batch %d (rows > cols, SURVEY Q2)", n1, n2, n3, B);
		return BLA_ERR_UNDEFINED;
	}
	float *W2 = nn->params + nn->off[2], *W3 = nn->params + nn->off[4];
	float *dW1 = grads + nn->off[0], *db1 = grads + nn->off[1], *dW2 = grads + nn->off[2], *db2 = grads + nn->off[3];
	float *dW3 = grads + nn->off[4], *db3 = grads + nn->off[5];
	hipStream_t s = pick_stream(stream);

	// The input scale of :218 (x *= 1/255.0F, a float expression) is folded into the two products that read X:
	// W1.(s x) = s (W1.x) and dZ1.(s x)^T = s (dZ1.x^T) -- no pass over the 784 x B input at all.
	const float xs = 1 / 255.0F;
	const bool fuse_db = colsum_mode == BLA_COLSUM_INTENDED;   // true row sums ride along the dW products

	st = forward_pass(nn, s, d_x_raw, d_y, true); if (st) return st;
	// (Measured and not kept: running the layer-3 / layer-2 weight-gradient products on a side stream -- parallel branches of
	// the captured graph -- to take two launches off the critical path.  The cross-queue dependencies cost more than the
	// launches: 69.6 us per step instead of 56.)
	// Backward in three launches: dZ2 alone; then the pairs {dW3 | dZ1} and {dW2 | dW1} -- independent products share a launch (the
	// reference runs all five one after the other, :267-293), and putting the small dW2 beside the large dW1 hides it completely.
	bla_gemm_epilogue em = {};
	em.alpha = 1.f; em.relu_mask = nn->z2; em.ld_mask = B;
	st = bla_gemm_f32(s, 1, 0, n2, B, n3, W3, n2, nn->dz3, B, nn->dz2, B, &em); if (st) return st;   // dZ2 = (W3^T dZ3) (.) relu'(Z2), :273-278
	bla_gemm_epilogue eg = {};
	eg.alpha = 1.f; eg.row_sum_a = fuse_db ? db3 : nullptr;
	bla_gemm_epilogue em1 = {};
	em1.alpha = 1.f; em1.relu_mask = nn->z1; em1.ld_mask = B;
	{
		bla_gemm_desc dw = {0, 1, n3, n2, B, nn->dz3, B, nn->a2, B, dW3, n2, &eg};       // dW3 = dZ3 . A2^T, :267-270; db3 :271
		bla_gemm_desc dz = {1, 0, n1, B, n2, W2, n1, nn->dz2, B, nn->dz1, B, &em1};      // dZ1, :284-289
		st = bla_gemm_pair_f32(s, &dw, &dz); if (st) return st;
	}
	if (!fuse_db) { st = bla_col_sum_f32(s, nn->dz3, n3, B, db3, colsum_mode); if (st) return st; }
	bla_gemm_epilogue eg2 = {}, eg1 = {};
	eg2.alpha = 1.f; eg2.row_sum_a = fuse_db ? db2 : nullptr;
	eg1.alpha = xs; eg1.row_sum_a = fuse_db ? db1 : nullptr;
	{
		bla_gemm_desc d2 = {0, 1, n2, n1, B, nn->dz2, B, nn->a1, B, dW2, n1, &eg2};      // dW2, :279-282
		bla_gemm_desc d1 = {0, 1, n1, n0, B, nn->dz1, B, d_x_raw, B, dW1, n0, &eg1};     // dW1 (the 1/255 of :218 folded into alpha), :290-293
		st = gemm_pair_with_hook(s, &d2, &d1, fuse_db ? hook : nullptr, posted); if (st) return st;   // (with separate bias sums behind it, it is not the last launch)
	}
	if (!fuse_db) {
		st = bla_col_sum_f32(s, nn->dz2, n2, B, db2, colsum_mode); if (st) return st;
		st = bla_col_sum_f32(s, nn->dz1, n1, B, db1, colsum_mode); if (st) return st;
	}
	return BLA_OK;
}

/* params += lr * grads over the whole bucket: the six {clip (no-op at INFINITY), scale, add} triples of :296-315.
 * The reference's learn rate is the float -0.02 (:186). */
bla_status bla_mnist_nn_apply(bla_mnist_nn* nn, void* stream, float lr) {
	BLA_REQUIRE(nn, BLA_ERR_INVALID, "null trainer");
	return bla_axpy_f32(stream, nn->params, nn->grads, lr, nn->count);
}

static bool can_fuse_update(const bla_mnist_nn* nn, int colsum_mode);
static bla_status fused_update_step(bla_mnist_nn* nn, hipStream_t s, float lr, const float* d_x_raw, const float* d_y);

bla_status bla_mnist_nn_train_step(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, float lr, int colsum_mode) {
	bla_status st = bla_mnist_nn_forward_backward(nn, stream, d_x_raw, d_y, colsum_mode);
	if (st) return st;
	return bla_mnist_nn_apply(nn, stream, lr);
}

/* Forward + backward + update with the update folded into the weight-gradient products: W_l += lr * (dZ_l . A_{l-1}^T) through the
 * products' alpha / beta (C = W_l), b_l += lr * rowsum(dZ_l) through the scaled row sum -- no gradient bucket, no axpy launch, and the
 * order changes so that no product updates a matrix another product of the same launch still reads:
 *     forward x 3;  dZ2 (reads W3);  {W3, b3 update | dZ1 (reads W2)};  {W2, b2 update | W1, b1 update}
 * 6 launches instead of 7.  Rounding differs from "scale the gradient, then add" (:296-315) by one fused multiply-add per weight.
 * Only for layer shapes whose every product runs on the wave-split-K kernel (the scaled row sum lives there) and true row sums. */
static bool can_fuse_update(const bla_mnist_nn* nn, int colsum_mode) {
	auto t32 = [](int m, int n) { return (long)((m + 31) / 32) * ((n + 31) / 32); };
	const int n0 = nn->n[0], n1 = nn->n[1], n2 = nn->n[2], n3 = nn->n[3], B = nn->batch;
	return colsum_mode == BLA_COLSUM_INTENDED && B <= 1280 && n0 <= 1280 && n1 <= 1280 && n2 <= 1280 && n3 <= 32 && t32(n1, B) <= 512 && t32(n2, B) <= 512 &&
	       t32(n3, n2) <= 512 && t32(n2, n1) <= 512 && t32(n1, n0) <= 512;
}

static bla_status fused_update_step(bla_mnist_nn* nn, hipStream_t s, float lr, const float* d_x_raw, const float* d_y) {
	const int n0 = nn->n[0], n1 = nn->n[1], n2 = nn->n[2], n3 = nn->n[3], B = nn->batch;
	float *W1 = nn->params + nn->off[0], *b1 = nn->params + nn->off[1], *W2 = nn->params + nn->off[2], *b2 = nn->params + nn->off[3];
	float *W3 = nn->params + nn->off[4], *b3 = nn->params + nn->off[5];
	const float* x = d_x_raw ? d_x_raw : nn->x_raw; const float* y = d_y ? d_y : nn->y;
	const float xs = 1 / 255.0F;
	bla_status st = forward_pass(nn, s, x, y, true);
	if (st) return st;
	bla_gemm_epilogue em2 = {};
	em2.alpha = 1.f; em2.relu_mask = nn->z2; em2.ld_mask = B;
	st = bla_gemm_f32(s, 1, 0, n2, B, n3, W3, n2, nn->dz3, B, nn->dz2, B, &em2); if (st) return st;   // dZ2: last reader of W3
	bla_gemm_epilogue u3 = {}, em1 = {};
	u3.alpha = lr; u3.beta = 1.f; u3.row_sum_a = b3; u3.row_sum_alpha = lr; u3.row_sum_beta = 1.f;
	em1.alpha = 1.f; em1.relu_mask = nn->z1; em1.ld_mask = B;
	{
		bla_gemm_desc up = {0, 1, n3, n2, B, nn->dz3, B, nn->a2, B, W3, n2, &u3};                             // W3 += lr dZ3 A2^T, b3 += lr rowsum(dZ3)
		bla_gemm_desc dz = {1, 0, n1, B, n2, W2, n1, nn->dz2, B, nn->dz1, B, &em1};                           // dZ1: last reader of W2
		st = bla_gemm_pair_f32(s, &up, &dz); if (st) return st;
	}
	bla_gemm_epilogue u2 = {}, u1 = {};
	u2.alpha = lr; u2.beta = 1.f; u2.row_sum_a = b2; u2.row_sum_alpha = lr; u2.row_sum_beta = 1.f;
	u1.alpha = lr * xs; u1.beta = 1.f; u1.row_sum_a = b1; u1.row_sum_alpha = lr; u1.row_sum_beta = 1.f;
	{
		bla_gemm_desc p2 = {0, 1, n2, n1, B, nn->dz2, B, nn->a1, B, W2, n1, &u2};
		bla_gemm_desc p1 = {0, 1, n1, n0, B, nn->dz1, B, x, B, W1, n0, &u1};
		st = bla_gemm_pair_f32(s, &p2, &p1); if (st) return st;
	}
	return BLA_OK;
}

/* model/mnist_nn.c:221-234 alone (and the whole of run(), :447-463). */
bla_status bla_mnist_nn_forward(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn, BLA_ERR_INVALID, "null trainer");
	return forward_pass(nn, pick_stream(stream), d_x_raw ? d_x_raw : nn->x_raw, d_y ? d_y : nn->y, false);
}

static void drop_graphs(bla_mnist_nn* nn) {
	if (nn->graph_ready) { (void)hipGraphExecDestroy(nn->graph_exec); (void)hipGraphDestroy(nn->graph); nn->graph_ready = false; }
	for (int i = 0; i < 2; i++)
		if (nn->dp_ready[i]) { (void)hipGraphExecDestroy(nn->dp_exec[i]); (void)hipGraphDestroy(nn->dp_graph[i]); nn->dp_ready[i] = false; }
}

/* Loss / accuracy bookkeeping of model/mnist_nn.c:237-257 (and run(), :476-490) inside the output layer's launch: one double and one
 * counter per batch column, each owned by one thread of that launch.  Recorded graphs bake the pointers in, so toggling drops them. */
bla_status bla_mnist_nn_metrics_enable(bla_mnist_nn* nn, int on) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn, BLA_ERR_INVALID, "null trainer");
	BLA_REQUIRE(!on || nn->n[3] <= 32, BLA_ERR_INVALID, "device-side loss / accuracy needs an output layer of at most 32 classes (fused softmax tail)");
	if ((on != 0) == nn->metrics_on) return BLA_OK;
	BLA_HIP(hipDeviceSynchronize());
	drop_graphs(nn);
	if (on && !nn->m_loss) {
		float* p = nullptr;
		st = dev_alloc(nn, &p, 2 * (size_t)nn->batch); if (st) return st;     // doubles
		nn->m_loss = (double*)p;
		st = dev_alloc(nn, &p, (size_t)nn->batch); if (st) return st;
		nn->m_correct = (unsigned*)p;
	}
	if (on) {
		BLA_HIP(hipMemsetAsync(nn->m_loss, 0, nn->batch * sizeof(double), ctx().stream));
		BLA_HIP(hipMemsetAsync(nn->m_correct, 0, nn->batch * sizeof(unsigned), ctx().stream));
		BLA_HIP(hipStreamSynchronize(ctx().stream));
	}
	nn->metrics_on = on != 0;
	return BLA_OK;
}

bla_status bla_mnist_nn_metrics_read(bla_mnist_nn* nn, double* loss_sum, long long* num_correct, int reset) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn && loss_sum && num_correct, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(nn->metrics_on, BLA_ERR_INVALID, "bla_mnist_nn_metrics_enable has not run");
	BLA_HIP(hipDeviceSynchronize());
	std::vector<double> l(nn->batch); std::vector<unsigned> c(nn->batch);
	BLA_HIP(hipMemcpyAsync(l.data(), nn->m_loss, nn->batch * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
	BLA_HIP(hipMemcpyAsync(c.data(), nn->m_correct, nn->batch * sizeof(unsigned), hipMemcpyDeviceToHost, ctx().stream));
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	double ls = 0; long long cs = 0;
	for (int i = 0; i < nn->batch; i++) { ls += l[i]; cs += c[i]; }   // column order
	*loss_sum = ls; *num_correct = cs;
	if (reset) {
		BLA_HIP(hipMemsetAsync(nn->m_loss, 0, nn->batch * sizeof(double), ctx().stream));
		BLA_HIP(hipMemsetAsync(nn->m_correct, 0, nn->batch * sizeof(unsigned), ctx().stream));
		BLA_HIP(hipStreamSynchronize(ctx().stream));
	}
	return BLA_OK;
}

/* The batch construction loop of model/mnist_nn.c:204-217 on the device: column k of the input is example idx[k] of the resident,
 * feature-major dataset (lib/mnist_csv2.c's layout), column k of the label matrix its one-hot label. */
namespace {
__global__ void __launch_bounds__(256) mnist_gather_kernel(const float* __restrict__ X, const float* __restrict__ labels, int num_examples,
                                                           const int* __restrict__ idx, float* __restrict__ x_raw, float* __restrict__ y, int n0, int n3, int B) {
	const int total = (n0 + n3) * B;
	for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		const int row = e / B, k = e - row * B;
		const int ex = idx[k];
		if (row < n0) x_raw[e] = X[(size_t)row * num_examples + ex];                    // input_data[k + B * p] = ex.X[p * num_examples], :208-210
		else y[e - n0 * B] = (int)labels[ex] == row - n0 ? 1.f : 0.f;                    // expectations[k + expectation * B] = 1, :212-216
	}
}
}  // namespace

bla_status bla_mnist_nn_gather_batch(bla_mnist_nn* nn, void* stream, const float* d_X, const float* d_labels, int num_examples, const int* d_indices) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn && d_X && d_labels && d_indices && num_examples > 0, BLA_ERR_INVALID, "null / empty argument");
	const int total = (nn->n[0] + nn->n[3]) * nn->batch;
	hipLaunchKernelGGL(mnist_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, pick_stream(stream), d_X, d_labels, num_examples, d_indices,
	                   nn->x_raw, nn->y, nn->n[0], nn->n[3], nn->batch);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

/* The fused-update step issued directly on the stream (no graph): six launches from one host call.  Measured faster than replaying the same
 * six launches as a graph (41.7 vs 45.4 us per step) as long as the host keeps up; the gradient bucket is not written.  Falls back to
 * bla_mnist_nn_train_step where the fused form does not apply. */
bla_status bla_mnist_nn_fused_step(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, float lr, int colsum_mode) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn, BLA_ERR_INVALID, "null trainer");
	if (!can_fuse_update(nn, colsum_mode)) return bla_mnist_nn_train_step(nn, stream, d_x_raw, d_y, lr, colsum_mode);
	return fused_update_step(nn, pick_stream(stream), lr, d_x_raw, d_y);
}

/* Capture one whole step (resident input/label buffers -> updated parameters) into a hipGraph and replay it:
 * the step is ~25 launches of a few microseconds each, i.e. launch-bound when issued one by one.
 * with_update = 0 captures forward+backward only (data-parallel: the all-reduce sits between the two halves). */
bla_status bla_mnist_nn_graph_step(bla_mnist_nn* nn, void* stream, float lr, int colsum_mode, int with_update) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn, BLA_ERR_INVALID, "null trainer");
	hipStream_t s = pick_stream(stream);
	if (!nn->graph_ready || nn->graph_colsum != colsum_mode * 2 + with_update || nn->graph_lr != lr) {
		if (nn->graph_ready) { (void)hipGraphExecDestroy(nn->graph_exec); (void)hipGraphDestroy(nn->graph); nn->graph_ready = false; }
		BLA_HIP(hipStreamSynchronize(s));
		BLA_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
		if (with_update && can_fuse_update(nn, colsum_mode)) st = fused_update_step(nn, s, lr, nullptr, nullptr);
		else {
			st = bla_mnist_nn_forward_backward(nn, s, nullptr, nullptr, colsum_mode);
			if (!st && with_update) st = bla_mnist_nn_apply(nn, s, lr);
		}
		hipError_t e = hipStreamEndCapture(s, &nn->graph);
		if (st) { if (e == hipSuccess) (void)hipGraphDestroy(nn->graph); return st; }
		if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
		BLA_HIP(hipGraphInstantiate(&nn->graph_exec, nn->graph, nullptr, nullptr, 0));
		nn->graph_ready = true; nn->graph_colsum = colsum_mode * 2 + with_update; nn->graph_lr = lr;
	}
	BLA_HIP(hipGraphLaunch(nn->graph_exec, s));
	return BLA_OK;
}

/* One data-parallel step as ONE graph launch: forward + backward on this rank's batch columns into the exchange object's
 * gradient bucket of the current parity, then the fused all-reduce + update (bla_dp_allreduce_f32 with target = params,
 * alpha = lr).  Collective: every rank calls it the same number of times.  Needs true row sums for the bias gradients
 * (BLA_COLSUM_INTENDED): matrix_col_sum as written is not separable over columns. */
bla_status bla_mnist_nn_dp_step(bla_mnist_nn* nn, bla_dp* dp, void* stream, float lr, int colsum_mode) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn && dp, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(colsum_mode == BLA_COLSUM_INTENDED, BLA_ERR_INVALID, "a sharded batch needs BLA_COLSUM_INTENDED (true row sums)");
	BLA_REQUIRE(bla_dp_count(dp) == nn->count, BLA_ERR_SHAPE, "exchange bucket holds %zu floats, the trainer has %zu parameters", bla_dp_count(dp), nn->count);
	hipStream_t s = pick_stream(stream);
	// (the exchange object's id, not its address: a destroyed and re-created object may come back at the same address, and the recorded
	// graphs would then hold the freed buckets and closed peer mappings)
	if (nn->dp_bound_id != dp_identity(dp) || nn->dp_lr != lr) {
		for (int i = 0; i < 2; i++)
			if (nn->dp_ready[i]) { (void)hipGraphExecDestroy(nn->dp_exec[i]); (void)hipGraphDestroy(nn->dp_graph[i]); nn->dp_ready[i] = false; }
		nn->dp_bound_id = dp_identity(dp); nn->dp_lr = lr;
	}
	const int par = dp_parity(dp);
	if (!nn->dp_ready[par]) {
		BLA_HIP(hipStreamSynchronize(s));
		BLA_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
		bool posted = false;
		st = forward_backward_into(nn, s, nullptr, nullptr, colsum_mode, bla_dp_bucket(dp, par), dp_done_hook(dp), &posted);
		if (!st) st = dp_allreduce(dp, s, par, nullptr, nn->params, lr, posted);
		hipError_t e = hipStreamEndCapture(s, &nn->dp_graph[par]);
		if (st) { if (e == hipSuccess) (void)hipGraphDestroy(nn->dp_graph[par]); return st; }
		if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
		BLA_HIP(hipGraphInstantiate(&nn->dp_exec[par], nn->dp_graph[par], nullptr, nullptr, 0));
		nn->dp_ready[par] = true;
	}
	BLA_HIP(hipGraphLaunch(nn->dp_exec[par], s));
	dp_advance(dp);
	return BLA_OK;
}

/* The same data-parallel step issued directly on the stream (seven launches from one host call) instead of replayed as a graph; the two
 * forms may be mixed freely (they share the bucket parity). */
bla_status bla_mnist_nn_dp_step_direct(bla_mnist_nn* nn, bla_dp* dp, void* stream, float lr, int colsum_mode) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn && dp, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(colsum_mode == BLA_COLSUM_INTENDED, BLA_ERR_INVALID, "a sharded batch needs BLA_COLSUM_INTENDED (true row sums)");
	BLA_REQUIRE(bla_dp_count(dp) == nn->count, BLA_ERR_SHAPE, "exchange bucket holds %zu floats, the trainer has %zu parameters", bla_dp_count(dp), nn->count);
	hipStream_t s = pick_stream(stream);
	const int par = dp_parity(dp);
	bool posted = false;
	st = forward_backward_into(nn, s, nullptr, nullptr, colsum_mode, bla_dp_bucket(dp, par), dp_done_hook(dp), &posted);
	if (st) return st;
	st = dp_allreduce(dp, s, par, nullptr, nn->params, lr, posted);
	if (st) return st;
	dp_advance(dp);
	return BLA_OK;
}

}  // extern "C"
