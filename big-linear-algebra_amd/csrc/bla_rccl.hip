// bla_rccl.hip -- the library form of the data-parallel exchange (BASELINE north_star: "RCCL all-reduce over xGMI on the
// weight/bias gradients after each backward"; SURVEY 8(b) bla_dp_{init,allreduce}, 8(e) "ncclAllReduce as the required baseline").
//
// One communicator per rank (one process per GPU, or one context per GPU inside one process), created from a 128-byte unique id
// that rank 0 makes and the host program hands to the other ranks over any channel it has.  The exchange is ONE in-place
// ncclAllReduce(ncclFloat, ncclSum) of the flat gradient bucket (235,146 floats for the reference's 784-256-128-10 network:
// dW1, db1, dW2, db2, dW3, db3) between backward (model/mnist_nn.c:260-293) and the update (:296-315); the gradient is a sum
// over batch columns, so the SUM needs no rescale.  The hand-written peer-read kernel of bla_dp.hip does the same job in one
// launch fused with the update; both are callable from C and bench.py reports which one ran.
//
// librccl is NOT a link dependency: it is opened on first use (dlopen of the soname: the copy torch.distributed already loaded
// when there is one -- one collective library per process -- else the one on this library's RUNPATH or under /opt/rocm/lib), so a single-GPU
// build or box without RCCL loads and runs; the bla_dp_rccl_* entry points then return BLA_ERR_NO_DEVICE with the reason in bla_last_error().
#include "bla_internal.h"
#include <rccl/rccl.h>   // types and prototypes only
#include <dlfcn.h>
#include <cstring>
#include <mutex>

using namespace bla;

struct bla_rccl {
	ncclComm_t comm;
	int rank, world, device;
};

namespace {
struct RcclApi {
	void* handle = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	bool tried = false;
};
RcclApi g_api;
std::mutex g_api_mu;

const RcclApi* rccl_api() {
	std::lock_guard<std::mutex> lk(g_api_mu);
	if (g_api.tried) return g_api.handle ? &g_api : nullptr;
	g_api.tried = true;
	RandStreamGuard keep_callers_rand_stream;
	void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process (torch.distributed's "nccl" backend)
	if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);     // this library's RUNPATH (torch/lib), LD_LIBRARY_PATH, ld.so.cache
	if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!h) return nullptr;
#define BLA_SYM(name) g_api.name = (decltype(g_api.name))dlsym(h, "nccl" #name); if (!g_api.name) { dlclose(h); return nullptr; }
	BLA_SYM(GetUniqueId) BLA_SYM(CommInitRank) BLA_SYM(CommInitAll) BLA_SYM(CommDestroy) BLA_SYM(AllReduce) BLA_SYM(GetErrorString) BLA_SYM(GroupStart) BLA_SYM(GroupEnd)
#undef BLA_SYM
	g_api.handle = h;
	return &g_api;
}

bla_status nccl_fail(const RcclApi* api, ncclResult_t r, const char* what) {
	set_error("RCCL error %d (%s) in %s", (int)r, api->GetErrorString(r), what);
	return BLA_ERR_HIP;
}
}  // namespace

#define BLA_RCCL_API(api)                                                                                                          \
	const RcclApi* api = rccl_api();                                                                                               \
	BLA_REQUIRE(api, BLA_ERR_NO_DEVICE, "librccl.so.1 could not be opened (%s): the RCCL leg is unavailable, use the peer-read exchange (bla_dp_*)", dlerror())
#define BLA_NCCL(api, call)                                        \
	do {                                                           \
		ncclResult_t _r = (api)->call;                             \
		if (_r != ncclSuccess) return nccl_fail(api, _r, #call);   \
	} while (0)

extern "C" {

int bla_dp_rccl_available(void) { return rccl_api() ? 1 : 0; }

bla_status bla_dp_rccl_unique_id(void* id128) {
	BLA_REQUIRE(id128, BLA_ERR_INVALID, "null argument");
	BLA_RCCL_API(api);
	RandStreamGuard keep_callers_rand_stream;   // RCCL's bootstrap draws from rand()
	static_assert(sizeof(ncclUniqueId) == BLA_RCCL_ID_BYTES, "unique id size");
	ncclUniqueId id;
	BLA_NCCL(api, GetUniqueId(&id));
	memcpy(id128, &id, sizeof id);
	return BLA_OK;
}

/* ncclCommInitRank: COLLECTIVE and BLOCKING -- it returns when all `world` ranks have called it.  One process per rank, or one host thread
 * per rank; a single host thread that drives several ranks creates them with bla_dp_rccl_init_all instead (calling this for rank 0 from the
 * only thread would wait forever for ranks that thread has not created yet). */
bla_status bla_dp_rccl_init(bla_rccl** out, const void* id128, int rank, int world) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(out && id128, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(world >= 1 && rank >= 0 && rank < world, BLA_ERR_INVALID, "rank %d / world %d", rank, world);
	BLA_RCCL_API(api);
	RandStreamGuard keep_callers_rand_stream;
	ncclUniqueId id;
	memcpy(&id, id128, sizeof id);
	BLA_HIP(hipSetDevice(ctx().device));
	ncclComm_t comm;
	BLA_NCCL(api, CommInitRank(&comm, world, id, rank));
	bla_rccl* c = new bla_rccl{comm, rank, world, ctx().device};
	*out = c;
	return BLA_OK;
}

/* All ranks of ONE process from one host thread (ncclCommInitAll): out[r] = the communicator of rank r on devices[r] (distinct devices;
 * RCCL refuses duplicates).  The ranks' collectives must then be issued between bla_dp_rccl_group_begin / _end. */
bla_status bla_dp_rccl_init_all(bla_rccl** out, const int* devices, int world) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(out && devices && world >= 1 && world <= 64, BLA_ERR_INVALID, "bad argument (world %d)", world);
	BLA_RCCL_API(api);
	RandStreamGuard keep_callers_rand_stream;
	ncclComm_t comms[64];
	BLA_NCCL(api, CommInitAll(comms, world, devices));
	for (int r = 0; r < world; r++) out[r] = new bla_rccl{comms[r], r, world, devices[r]};
	if (ctx().ready) (void)hipSetDevice(ctx().device);
	return BLA_OK;
}

/* ncclGroupStart / ncclGroupEnd: one host thread issuing the collectives of several local ranks brackets them (without the bracket the
 * first rank's collective blocks the thread before the second rank's is issued). */
bla_status bla_dp_rccl_group_begin(void) {
	BLA_RCCL_API(api);
	BLA_NCCL(api, GroupStart());
	return BLA_OK;
}
bla_status bla_dp_rccl_group_end(void) {
	BLA_RCCL_API(api);
	RandStreamGuard keep_callers_rand_stream;
	BLA_NCCL(api, GroupEnd());
	return BLA_OK;
}

bla_status bla_dp_rccl_destroy(bla_rccl* c) {
	if (!c) return BLA_OK;
	const RcclApi* api = rccl_api();
	if (!api) { delete c; return BLA_OK; }
	RandStreamGuard keep_callers_rand_stream;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	ncclResult_t r = api->CommDestroy(c->comm);
	delete c;
	if (ctx().ready) (void)hipSetDevice(ctx().device);
	if (r != ncclSuccess) return nccl_fail(api, r, "ncclCommDestroy");
	return BLA_OK;
}

/* d_buf[i] = SUM over ranks of d_buf[i], in place, asynchronous on `stream`. */
bla_status bla_dp_rccl_allreduce_f32(bla_rccl* c, void* stream, float* d_buf, size_t count) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(c && d_buf, BLA_ERR_INVALID, "null argument");
	BLA_RCCL_API(api);
	RandStreamGuard keep_callers_rand_stream;
	BLA_NCCL(api, AllReduce(d_buf, d_buf, count, ncclFloat, ncclSum, c->comm, pick_stream(stream)));
	return BLA_OK;
}

/* One data-parallel step over the library collective: forward + backward on this rank's batch columns into the trainer's gradient
 * bucket, ncclAllReduce(SUM) of the bucket, params += lr * sum.  All on `stream`, asynchronous.  (Several local ranks driven by one host
 * thread: bla_mnist_nn_forward_backward per rank, then the ranks' bla_dp_rccl_allreduce_f32 inside one group, then bla_mnist_nn_apply per rank.) */
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
