// bla_runtime.hip -- device selection, stream, workspace, memory and event helpers
// behind the C-ABI of include/bla.h.  No CPU fallback lives here or anywhere in
// this library: if HIP cannot give us a gfx950 device every compute entry point
// returns BLA_ERR_NO_DEVICE.
#include "bla_internal.h"
#include <cstring>
#include <mutex>

namespace bla {

static Context g_ctx;                          // the process default (bla_init)
static thread_local Context* t_ctx = nullptr;  // bla_context_set_current
static thread_local char g_err[512] = "no error";
static std::mutex g_mu;

Context& ctx() { return t_ctx ? *t_ctx : g_ctx; }

void set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

bla_status hip_fail(hipError_t e, const char* what) {
	set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
	return BLA_ERR_HIP;
}

bla_status require_ready() {
	if (!ctx().ready) {
		set_error("bla runtime not initialised: call bla_init(device) first (no CPU fallback exists)");
		return BLA_ERR_NO_DEVICE;
	}
	return BLA_OK;
}

// libc rand() state of the caller, parked while HIP / RCCL calls that draw from rand() run (bla_internal.h)
static std::mutex g_rand_mu;
static int g_rand_depth = 0;
static char g_rand_scratch[128];
static char* g_rand_saved = nullptr;
void rand_guard_enter() {
	std::lock_guard<std::mutex> lk(g_rand_mu);
	if (g_rand_depth++ == 0) g_rand_saved = initstate(1u, g_rand_scratch, sizeof g_rand_scratch);
}
void rand_guard_leave() {
	std::lock_guard<std::mutex> lk(g_rand_mu);
	if (--g_rand_depth == 0 && g_rand_saved) { (void)setstate(g_rand_saved); g_rand_saved = nullptr; }
}

bla_status side_lane_fork(hipStream_t main, hipStream_t* lane) {
	Context& c = ctx();
	if (!c.side_stream) {
		// the lane's stream runs at the LOWEST priority: the caller's critical path is on the other stream, the lane's workgroups take what that leaves
		// (batch-64 U-Net backward 9.87-9.91 -> 9.78-9.81 ms; BLA_LANE_PRIORITY=default|high for the other two settings)
		const char* pr = getenv("BLA_LANE_PRIORITY");
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		if (pr && pr[0] == 'd') BLA_HIP(hipStreamCreateWithFlags(&c.side_stream, hipStreamNonBlocking));
		else BLA_HIP(hipStreamCreateWithPriority(&c.side_stream, hipStreamNonBlocking, pr && pr[0] == 'h' ? greatest : least));
		BLA_HIP(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
		BLA_HIP(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
	}
	BLA_HIP(hipEventRecord(c.ev_fork, main));
	BLA_HIP(hipStreamWaitEvent(c.side_stream, c.ev_fork, 0));
	c.side_lane = true;
	*lane = c.side_stream;
	return BLA_OK;
}
void side_lane_done() { ctx().side_lane = false; }
bla_status side_lane_join(hipStream_t main) {
	Context& c = ctx();
	if (!c.side_stream) return BLA_OK;
	BLA_HIP(hipEventRecord(c.ev_join, c.side_stream));
	BLA_HIP(hipStreamWaitEvent(main, c.ev_join, 0));
	return BLA_OK;
}

bla_status ensure_workspace(size_t bytes, void** out) {
	Context& c = ctx();
	if (c.side_lane) {     // issued for the side lane: its own scratch
		if (bytes > c.workspace3_bytes) {
			if (c.workspace3) { BLA_HIP(hipDeviceSynchronize()); BLA_HIP(hipFree(c.workspace3)); c.workspace3 = nullptr; c.workspace3_bytes = 0; }
			const size_t want = bytes < (size_t)(64u << 20) ? (size_t)(64u << 20) : bytes;
			BLA_HIP(hipMalloc(&c.workspace3, want));
			c.workspace3_bytes = want;
		}
		*out = c.workspace3;
		return BLA_OK;
	}
	if (bytes > c.workspace_bytes) {
		// Grow-only.  Callers serialise on one stream, so freeing after a sync is safe.
		if (c.workspace) {
			BLA_HIP(hipDeviceSynchronize());
			BLA_HIP(hipFree(c.workspace));
			c.workspace = nullptr;
			c.workspace_bytes = 0;
		}
		size_t want = bytes < (size_t)(64u << 20) ? (size_t)(64u << 20) : bytes;
		BLA_HIP(hipMalloc(&c.workspace, want));
		c.workspace_bytes = want;
	}
	*out = c.workspace;
	return BLA_OK;
}

bla_status ensure_workspace2(size_t bytes, void** out) {
	Context& c = ctx();
	if (bytes > c.workspace2_bytes) {
		if (c.workspace2) {
			BLA_HIP(hipDeviceSynchronize());
			BLA_HIP(hipFree(c.workspace2));
			c.workspace2 = nullptr;
			c.workspace2_bytes = 0;
		}
		BLA_HIP(hipMalloc(&c.workspace2, bytes));
		c.workspace2_bytes = bytes;
	}
	*out = c.workspace2;
	return BLA_OK;
}

}  // namespace bla

using namespace bla;

namespace bla {
typedef float mfma_acc_t __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(1024) mfma_rate_kernel(int iters, float* sink) {
	mfma_acc_t c0, c1, c2, c3;   // four independent chains (64 AGPRs: fits 4 waves per SIMD)
#pragma unroll
	for (int r = 0; r < 16; r++) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
	float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
	for (int it = 0; it < iters; it++) {   // written out as asm so that the chains stay exactly as they are; 8 MFMAs per trip
		asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\tv_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n\t"
		             "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\tv_mfma_f32_32x32x2_f32 %3, %4, %5, %3\n\t"
		             "v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\tv_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n\t"
		             "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\tv_mfma_f32_32x32x2_f32 %3, %4, %5, %3"
		             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
	}
	asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results are read by VALU below: software-managed hazard
	float t = c0[0] + c1[0] + c2[0] + c3[0];
	if (t == 123.456f) sink[0] = t;   // keep the accumulators live
}
typedef float mfma_acc4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(1024) mfma_rate_16x16_kernel(int iters, float* sink) {   // the other fp32 shape: v_mfma_f32_16x16x4_f32, 8 passes
	mfma_acc4_t c0, c1, c2, c3, c4, c5, c6, c7;
#pragma unroll
	for (int r = 0; r < 4; r++) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; c4[r] = 0.f; c5[r] = 0.f; c6[r] = 0.f; c7[r] = 0.f; }
	float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
	for (int it = 0; it < iters; it++) {   // 16 MFMAs of 2048 FLOP each per trip = the FLOPs of 8 MFMAs 32x32x2
		asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n\tv_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n\t"
		             "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n\tv_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n\tv_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n\t"
		             "v_mfma_f32_16x16x4_f32 %6, %8, %9, %6\n\tv_mfma_f32_16x16x4_f32 %7, %8, %9, %7\n\t"
		             "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n\tv_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n\t"
		             "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n\tv_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n\tv_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n\t"
		             "v_mfma_f32_16x16x4_f32 %6, %8, %9, %6\n\tv_mfma_f32_16x16x4_f32 %7, %8, %9, %7"
		             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7) : "v"(a), "v"(b));
	}
	asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
	float t = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];
	if (t == 123.456f) sink[0] = t;
}
}  // namespace bla

extern "C" {

int bla_device_count(void) {
	RandStreamGuard keep_callers_rand_stream;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// release whatever a context holds, ready or not (a context whose set-up failed half way holds a stream but is not ready)
static void close_context(Context& c) {
	if (!c.stream && !c.workspace && !c.workspace2 && !c.tile_counters && !c.side_stream && !c.workspace3) { c = Context(); return; }
	RandStreamGuard keep_callers_rand_stream;
	if (c.device >= 0) { (void)hipSetDevice(c.device); (void)hipDeviceSynchronize(); }
	if (c.side_stream) (void)hipStreamDestroy(c.side_stream);
	if (c.ev_fork) (void)hipEventDestroy(c.ev_fork);
	if (c.ev_join) (void)hipEventDestroy(c.ev_join);
	if (c.workspace3) (void)hipFree(c.workspace3);
	if (c.stream) (void)hipStreamDestroy(c.stream);
	if (c.workspace) (void)hipFree(c.workspace);
	if (c.workspace2) (void)hipFree(c.workspace2);
	if (c.tile_counters) (void)hipFree(c.tile_counters);
	c = Context();
}

// create stream / counters of one context on `device` (the caller holds g_mu).  On failure nothing stays allocated and the calling thread is
// back on the device it was on.
static bla_status open_context(Context& c, int device) {
	RandStreamGuard keep_callers_rand_stream;   // the first HIP calls of a process initialise the runtime, which draws from rand()
	int n = bla_device_count();
	if (n <= 0) {
		set_error("no HIP device visible (hipGetDeviceCount = %d); this library has no CPU path", n);
		return BLA_ERR_NO_DEVICE;
	}
	BLA_REQUIRE(device >= 0 && device < n, BLA_ERR_INVALID, "device %d out of range [0,%d)", device, n);
	int before = -1;
	(void)hipGetDevice(&before);
	auto body = [&]() -> bla_status {
		BLA_HIP(hipSetDevice(device));
		c.device = device;
		hipDeviceProp_t prop;
		BLA_HIP(hipGetDeviceProperties(&prop, device));
		if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
			set_error("device %d is %s; libbla_hip.so carries gfx950 code objects only", device, prop.gcnArchName);
			return BLA_ERR_NO_DEVICE;
		}
		BLA_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
		BLA_HIP(hipMalloc((void**)&c.tile_counters, (16384 + 64) * sizeof(unsigned)));   // + 64 words that stay zero (zero_word())
		// zeroed on the context's own stream and waited for: the stream is non-blocking, a NULL-stream memset would not order against it
		BLA_HIP(hipMemsetAsync(c.tile_counters, 0, (16384 + 64) * sizeof(unsigned), c.stream));
		BLA_HIP(hipStreamSynchronize(c.stream));
		c.num_cus = prop.multiProcessorCount;
		strncpy(c.arch, prop.gcnArchName, sizeof(c.arch) - 1);
		c.ready = true;
		return BLA_OK;
	};
	bla_status st = body();
	if (st) {
		close_context(c);
		if (before >= 0) (void)hipSetDevice(before);
	}
	return st;
}

bla_status bla_init(int device) {
	std::lock_guard<std::mutex> lk(g_mu);
	if (g_ctx.ready && g_ctx.device == device) {
		if (!t_ctx) BLA_HIP(hipSetDevice(device));
		return BLA_OK;
	}
	close_context(g_ctx);   // switching device: drop the old stream / workspace
	return open_context(g_ctx, device);
}

bla_status bla_shutdown(void) {
	std::lock_guard<std::mutex> lk(g_mu);
	close_context(g_ctx);
	return BLA_OK;
}

struct bla_context { Context c; };

/* Further contexts beside the default one: a context = {device, stream, split-K workspace, arrival counters}.  A host that drives
 * several GPUs (or several replicas on one GPU) from ONE process creates one context per rank and makes it current on the calling
 * thread before it issues that rank's bla_* calls. */
bla_status bla_context_create(bla_context** out, int device) {
	BLA_REQUIRE(out, BLA_ERR_INVALID, "null out pointer");
	std::lock_guard<std::mutex> lk(g_mu);
	bla_context* c = new bla_context();
	bla_status st = open_context(c->c, device);
	if (st) delete c;
	else *out = c;
	Context& cur = ctx();   // open_context moved this thread to `device`: go back to where the caller was
	if (cur.ready) (void)hipSetDevice(cur.device);
	return st;
}

bla_status bla_context_set_current(bla_context* c) {
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (t_ctx) t_ctx->users--;
		t_ctx = c ? &c->c : nullptr;
		if (t_ctx) t_ctx->users++;
	}
	Context& cur = ctx();
	if (cur.ready) BLA_HIP(hipSetDevice(cur.device));
	return BLA_OK;
}

/* Refused (BLA_ERR_INVALID) while the context is current on ANOTHER thread: that thread's next bla_* call would use freed memory. */
bla_status bla_context_destroy(bla_context* c) {
	if (!c) return BLA_OK;
	std::lock_guard<std::mutex> lk(g_mu);
	const int mine = t_ctx == &c->c ? 1 : 0;
	BLA_REQUIRE(c->c.users - mine == 0, BLA_ERR_INVALID, "the context is still current on %d other thread(s): bla_context_set_current(NULL) there first", c->c.users - mine);
	if (mine) { t_ctx = nullptr; c->c.users--; }
	close_context(c->c);
	delete c;
	Context& cur = ctx();
	if (cur.ready) (void)hipSetDevice(cur.device);
	return BLA_OK;
}

/* The calling program's libc rand() stream around HIP / RCCL calls it makes ITSELF between library calls (the library's own entry points do
 * this internally): enter parks the stream, leave puts it back; nestable, process-wide, any thread. */
void bla_rand_guard_enter(void) { rand_guard_enter(); }
void bla_rand_guard_leave(void) { rand_guard_leave(); }

int bla_is_initialized(void) { return ctx().ready ? 1 : 0; }
const char* bla_last_error(void) { return g_err; }
const char* bla_version(void) { return "bla-hip 0.1 (gfx950)"; }

const char* bla_status_string(bla_status s) {
	switch (s) {
		case BLA_OK: return "ok";
		case BLA_ERR_INVALID: return "invalid argument";
		case BLA_ERR_SHAPE: return "shape mismatch";
		case BLA_ERR_NO_DEVICE: return "no device / not initialised";
		case BLA_ERR_HIP: return "HIP runtime error";
		case BLA_ERR_UNDEFINED: return "undefined in the reference";
		case BLA_ERR_TIMEOUT: return "a rank never arrived at the exchange";
		default: return "unknown status";
	}
}

bla_status bla_device_name(char* buf, int buflen) {
	bla_status s = require_ready();
	if (s) return s;
	BLA_REQUIRE(buf && buflen > 0, BLA_ERR_INVALID, "null buffer");
	snprintf(buf, buflen, "%s (%d CUs)", ctx().arch, ctx().num_cus);
	return BLA_OK;
}

bla_status bla_malloc(void** p, size_t bytes) {
	bla_status s = require_ready();
	if (s) return s;
	BLA_REQUIRE(p, BLA_ERR_INVALID, "null out pointer");
	*p = nullptr;
	if (bytes == 0) return BLA_OK;
	BLA_HIP(hipMalloc(p, bytes));
	return BLA_OK;
}

bla_status bla_free(void* p) {
	if (!p) return BLA_OK;
	bla_status s = require_ready();
	if (s) return s;
	BLA_HIP(hipFree(p));
	return BLA_OK;
}

bla_status bla_memcpy_h2d(void* d, const void* h, size_t bytes, void* stream) {
	bla_status s = require_ready();
	if (s) return s;
	if (bytes == 0) return BLA_OK;
	BLA_REQUIRE(d && h, BLA_ERR_INVALID, "null pointer in memcpy_h2d");
	BLA_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, pick_stream(stream)));
	return BLA_OK;
}

bla_status bla_memcpy_d2h(void* h, const void* d, size_t bytes, void* stream) {
	bla_status s = require_ready();
	if (s) return s;
	if (bytes == 0) return BLA_OK;
	BLA_REQUIRE(d && h, BLA_ERR_INVALID, "null pointer in memcpy_d2h");
	BLA_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, pick_stream(stream)));
	return BLA_OK;
}

bla_status bla_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
	bla_status s = require_ready();
	if (s) return s;
	if (bytes == 0) return BLA_OK;
	BLA_REQUIRE(dst && src, BLA_ERR_INVALID, "null pointer in memcpy_d2d");
	BLA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, pick_stream(stream)));
	return BLA_OK;
}

bla_status bla_memset(void* d, int byte, size_t bytes, void* stream) {
	bla_status s = require_ready();
	if (s) return s;
	if (bytes == 0) return BLA_OK;
	BLA_REQUIRE(d, BLA_ERR_INVALID, "null pointer in memset");
	BLA_HIP(hipMemsetAsync(d, byte, bytes, pick_stream(stream)));
	return BLA_OK;
}

bla_status bla_stream_sync(void* stream) {
	bla_status s = require_ready();
	if (s) return s;
	BLA_HIP(hipStreamSynchronize(pick_stream(stream)));
	return BLA_OK;
}

void* bla_default_stream(void) { return (void*)ctx().stream; }

bla_status bla_event_create(void** ev) {
	bla_status s = require_ready();
	if (s) return s;
	BLA_REQUIRE(ev, BLA_ERR_INVALID, "null out pointer");
	hipEvent_t e;
	BLA_HIP(hipEventCreate(&e));
	*ev = (void*)e;
	return BLA_OK;
}

bla_status bla_event_destroy(void* ev) {
	if (!ev) return BLA_OK;
	BLA_HIP(hipEventDestroy((hipEvent_t)ev));
	return BLA_OK;
}

bla_status bla_event_record(void* ev, void* stream) {
	bla_status s = require_ready();
	if (s) return s;
	BLA_REQUIRE(ev, BLA_ERR_INVALID, "null event");
	BLA_HIP(hipEventRecord((hipEvent_t)ev, pick_stream(stream)));
	return BLA_OK;
}

bla_status bla_event_elapsed_ms(void* a, void* b, float* ms) {
	bla_status s = require_ready();
	if (s) return s;
	BLA_REQUIRE(a && b && ms, BLA_ERR_INVALID, "null argument");
	BLA_HIP(hipEventSynchronize((hipEvent_t)b));
	BLA_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
	return BLA_OK;
}

struct BlaGraph { hipGraph_t graph; hipGraphExec_t exec; };

bla_status bla_graph_begin(void* stream) {
	bla_status st = require_ready();
	if (st) return st;
	hipStream_t s = pick_stream(stream);
	BLA_HIP(hipStreamSynchronize(s));
	BLA_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
	return BLA_OK;
}

bla_status bla_graph_end(void* stream, void** graph) {
	BLA_REQUIRE(graph, BLA_ERR_INVALID, "null argument");
	hipStream_t s = pick_stream(stream);
	hipGraph_t g = nullptr;
	BLA_HIP(hipStreamEndCapture(s, &g));
	hipGraphExec_t ex = nullptr;
	hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
	if (e != hipSuccess) { (void)hipGraphDestroy(g); return hip_fail(e, "hipGraphInstantiate"); }
	BlaGraph* bg = new BlaGraph{g, ex};
	*graph = bg;
	return BLA_OK;
}

bla_status bla_graph_launch(void* graph, void* stream) {
	BLA_REQUIRE(graph, BLA_ERR_INVALID, "null graph");
	BLA_HIP(hipGraphLaunch(((BlaGraph*)graph)->exec, pick_stream(stream)));
	return BLA_OK;
}

bla_status bla_graph_destroy(void* graph) {
	if (!graph) return BLA_OK;
	BlaGraph* bg = (BlaGraph*)graph;
	(void)hipGraphExecDestroy(bg->exec);
	(void)hipGraphDestroy(bg->graph);
	delete bg;
	return BLA_OK;
}

/* Diagnostics: a workgroup of `waves` wavefronts issues `iters` x 8 independent v_mfma_f32_32x32x2_f32 with no memory traffic at all:
 * the rate this reaches is the practical ceiling of the fp32 MFMA pipe (clock and issue gaps included) that a GEMM can approach. */
bla_status bla_diag_mfma_rate(void* stream, int blocks, int waves, int iters, float* d_sink) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(blocks > 0 && waves != 0 && waves <= 16 && waves >= -16 && iters > 0 && d_sink, BLA_ERR_INVALID, "bad argument");
	if (waves < 0) hipLaunchKernelGGL(bla::mfma_rate_16x16_kernel, dim3(blocks), dim3(-waves * 64), 0, pick_stream(stream), iters, d_sink);   // negative: the 16x16x4 shape
	else hipLaunchKernelGGL(bla::mfma_rate_kernel, dim3(blocks), dim3(waves * 64), 0, pick_stream(stream), iters, d_sink);
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

}  // extern "C"
