// bla_rccl.hip -- the library form of the data-parallel exchange (BASELINE north_star: "RCCL all-reduce over xGMI on the
// weight/bias gradients after each backward"; SURVEY 8(b) bla_dp_{init,allreduce}, 8(e) "ncclAllReduce as the required baseline").
//
// One communicator per rank (one process per GPU, or one context per GPU inside one process), created from a 128-byte unique id
// that rank 0 makes and the host program hands to the other ranks over any channel it has.  The exchange is ONE in-place
// ncclAllReduce(ncclFloat, ncclSum) of the flat gradient bucket (235,146 floats for the reference's 784-256-128-10 network:
// dW1, db1, dW2, db2, dW3, db3) between backward (model/mnist_nn.c:260-293) and the update (:296-315); the gradient is a sum
// over batch columns, so the SUM needs no rescale.  The hand-written peer-read kernel of bla_dp.hip does the same job in one
// launch fused with the update; both are callable from C and bench.py reports which one ran.
#include "bla_internal.h"
#include <rccl/rccl.h>
#include <cstring>

using namespace bla;

struct bla_rccl {
	ncclComm_t comm;
	int rank, world, device;
};

static bla_status nccl_fail(ncclResult_t r, const char* what) {
	set_error("RCCL error %d (%s) in %s", (int)r, ncclGetErrorString(r), what);
	return BLA_ERR_HIP;
}
#define BLA_NCCL(call)                                        \
	do {                                                      \
		ncclResult_t _r = (call);                             \
		if (_r != ncclSuccess) return nccl_fail(_r, #call);   \
	} while (0)

extern "C" {

bla_status bla_dp_rccl_unique_id(void* id128) {
	BLA_REQUIRE(id128, BLA_ERR_INVALID, "null argument");
	RandStreamGuard keep_callers_rand_stream;   // RCCL's bootstrap draws from rand()
	static_assert(sizeof(ncclUniqueId) == BLA_RCCL_ID_BYTES, "unique id size");
	ncclUniqueId id;
	BLA_NCCL(ncclGetUniqueId(&id));
	memcpy(id128, &id, sizeof id);
	return BLA_OK;
}

bla_status bla_dp_rccl_init(bla_rccl** out, const void* id128, int rank, int world) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(out && id128, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(world >= 1 && rank >= 0 && rank < world, BLA_ERR_INVALID, "rank %d / world %d", rank, world);
	RandStreamGuard keep_callers_rand_stream;
	ncclUniqueId id;
	memcpy(&id, id128, sizeof id);
	BLA_HIP(hipSetDevice(ctx().device));
	ncclComm_t comm;
	BLA_NCCL(ncclCommInitRank(&comm, world, id, rank));
	bla_rccl* c = new bla_rccl{comm, rank, world, ctx().device};
	*out = c;
	return BLA_OK;
}

bla_status bla_dp_rccl_destroy(bla_rccl* c) {
	if (!c) return BLA_OK;
	RandStreamGuard keep_callers_rand_stream;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	ncclResult_t r = ncclCommDestroy(c->comm);
	delete c;
	if (ctx().ready) (void)hipSetDevice(ctx().device);
	if (r != ncclSuccess) return nccl_fail(r, "ncclCommDestroy");
	return BLA_OK;
}

/* d_buf[i] = SUM over ranks of d_buf[i], in place, asynchronous on `stream`. */
bla_status bla_dp_rccl_allreduce_f32(bla_rccl* c, void* stream, float* d_buf, size_t count) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(c && d_buf, BLA_ERR_INVALID, "null argument");
	RandStreamGuard keep_callers_rand_stream;
	BLA_NCCL(ncclAllReduce(d_buf, d_buf, count, ncclFloat, ncclSum, c->comm, pick_stream(stream)));
	return BLA_OK;
}

/* One data-parallel step over the library collective: forward + backward on this rank's batch columns into the trainer's gradient
 * bucket, ncclAllReduce(SUM) of the bucket, params += lr * sum.  All on `stream`, asynchronous. */
bla_status bla_mnist_nn_dp_step_rccl(bla_mnist_nn* nn, bla_rccl* c, void* stream, float lr, int colsum_mode) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(nn && c, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(colsum_mode == BLA_COLSUM_INTENDED, BLA_ERR_INVALID, "a sharded batch needs BLA_COLSUM_INTENDED (true row sums)");
	st = bla_mnist_nn_forward_backward(nn, stream, nullptr, nullptr, colsum_mode);
	if (st) return st;
	st = bla_dp_rccl_allreduce_f32(c, stream, bla_mnist_nn_grads(nn), bla_mnist_nn_param_count(nn));
	if (st) return st;
	return bla_mnist_nn_apply(nn, stream, lr);
}

}  // extern "C"
