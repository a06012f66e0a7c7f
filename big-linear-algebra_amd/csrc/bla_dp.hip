// bla_dp.hip -- the one exchange step of the data-parallel MNIST-NN path: SUM all-reduce of the flat
// gradient bucket, fused with the SGD update, as ONE kernel that reads the peers' buckets directly over xGMI.
//
// Why not a library ring: the bucket is 0.94 MB and a step is ~50 us of kernels, so the exchange is latency-bound.
// A ring all-reduce pays 2(R-1) = 14 dependent xGMI hops; here every GPU pulls the R-1 peer buckets concurrently
// over its R-1 point-to-point links (one hop, ~0.94 MB per link) and adds them in rank order, so
//   * the sum is bit-identical on every rank (same order everywhere) -- parameters never drift apart,
//   * the update `params += lr * sum` rides in the same pass (no second kernel, no extra read of the bucket),
//   * the kernel is an ordinary graph node: a whole step (forward, backward, exchange, update) is one hipGraphLaunch.
//
// Synchronisation is one monotonic flag per (reader, writer) pair, pushed by the writer into the reader's memory,
// and two gradient buckets used alternately:
//   writer r, epoch e : [kernels fill bucket[e&1]]  ->  exchange kernel: flag[p][r] = e for every p  (system-scope release)
//   reader p, epoch e : spin until flag[p][r] >= e for every r (system-scope acquire), then read bucket_r[e&1]
// Rank r overwrites bucket[(e+1)&1] only after its own epoch-e exchange finished, which needed every peer's
// epoch-e flag, which each peer pushed after completing its epoch-(e-1) exchange -- the last reader of that bucket.
// So no second barrier is needed.
//
// From four ranks up the exchange runs in two shots inside the same kernel (each link then carries 2/R of a bucket instead of a
// whole one -- 4x less at R = 8): rank r sums slice r of every peer's bucket (rank order) into its own `reduced` buffer, the
// workgroup that finishes that phase last pushes a second flag, and every rank then pulls the R-1 reduced slices of its peers.
// All ranks end up with the owner's sums, so the result is bit-identical to the one-shot form.  The second phase waits on
// workgroups of the same launch on other GPUs, so that launch is kept small enough (128 workgroups) to be fully resident.
//
// Buckets and reduced slices are fine-grained device memory, the flag words a separate uncached allocation; both are exported /
// opened with hipIpc*MemHandle between processes and addressed directly between ranks of one process.  A wait that exceeds the time-out raises the
// status word instead of spinning forever.
#include "bla_internal.h"
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <random>
#include <unistd.h>

using namespace bla;

namespace {
constexpr int kMaxWorld = 16;
constexpr long long kTicksPerMs = 100000LL;        // wall_clock64 runs at 100 MHz; the wait for a peer gives up after BLA_DP_TIMEOUT_MS (default 4000)

struct DpKernelArgs {
	const float* src[kMaxWorld];          // bucket of rank r for this parity (own: local pointer)
	unsigned* peer_flags[kMaxWorld];      // flag array living in rank p's memory
	unsigned* flags;                      // own flag array: flags[r] = last epoch rank r published to me
	unsigned* state;                      // local: [0] epoch of the last finished exchange, [1] arrival counter, [2] status
	float* out;                           // optional: out[i] = sum
	float* target;                        // optional: target[i] += alpha * sum
	float alpha;
	unsigned n4;                          // float4 groups, the last one possibly partial
	unsigned count;                       // floats in out / target (the source buckets are padded allocations)
	int world, rank;
	// two-shot form
	const float* red[kMaxWorld];          // reduced slice of rank r for this parity (own: local pointer, written in phase A)
	float* red_self;
	unsigned* peer_flags_b[kMaxWorld];    // second flag array ("my reduced slice is ready") in rank p's memory
	unsigned* flags_b;
	unsigned per4;                        // float4 groups per slice: slice r = [r * per4, min((r + 1) * per4, n4))
	long long timeout_ticks;
	int flags_posted;                     // the last gradient kernel of the step has already told the peers (DoneHook): no push here
};

__device__ __forceinline__ float4 load16_nt(const float* base, size_t i4) {
	typedef float vf4 __attribute__((ext_vector_type(4)));
	vf4 x = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(base) + i4);
	return make_float4(x.x, x.y, x.z, x.w);
}

// out / target hold `count` floats (possibly a foreign, unpadded buffer): the last float4 group may be partial
__device__ __forceinline__ void deliver(const DpKernelArgs& a, unsigned i, float4 s) {
	if (4 * i + 4 <= a.count) {
		if (a.out) reinterpret_cast<float4*>(a.out)[i] = s;
		if (a.target) {
			float4 t = reinterpret_cast<float4*>(a.target)[i];
			t.x += a.alpha * s.x; t.y += a.alpha * s.y; t.z += a.alpha * s.z; t.w += a.alpha * s.w;
			reinterpret_cast<float4*>(a.target)[i] = t;
		}
	} else {
		const float sv[4] = {s.x, s.y, s.z, s.w};
		for (unsigned j = 0; 4 * i + j < a.count; j++) {
			if (a.out) a.out[4 * i + j] = sv[j];
			if (a.target) a.target[4 * i + j] += a.alpha * sv[j];
		}
	}
}

// wait until flags[r] has reached `epoch` for every r < world (threads r < world poll); returns false on time-out
__device__ __forceinline__ bool wait_flags(const unsigned* flags, int world, int rank, unsigned epoch, int* s_fail, long long timeout_ticks) {
	if ((int)threadIdx.x < world && (int)threadIdx.x != rank) {   // (a rank's own data is ordered by its stream: no flag to itself)
		const long long t0 = wall_clock64();
		while ((int)(__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
			__builtin_amdgcn_s_sleep(4);
			if (wall_clock64() - t0 > timeout_ticks) { *s_fail = 1; break; }
		}
	}
	__syncthreads();
	return *s_fail == 0;
}

__global__ void __launch_bounds__(256) dp_allreduce_twoshot_kernel(DpKernelArgs a) {
	__shared__ int s_fail;
	__shared__ int s_last;
	const unsigned epoch = a.state[0] + 1;
	if (threadIdx.x == 0) { s_fail = 0; s_last = 0; }
	__syncthreads();
	if (!a.flags_posted && blockIdx.x == 0 && (int)threadIdx.x < a.world && (int)threadIdx.x != a.rank)   // (the system-scope release is the fence)
		__hip_atomic_store(a.peer_flags[threadIdx.x] + a.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	bool ok = wait_flags(a.flags, a.world, a.rank, epoch, &s_fail, a.timeout_ticks);
	// phase A: my slice of every bucket, summed in rank order -> reduced buffer (for the peers) and my own out / target
	const unsigned lo = a.rank * a.per4, hi = min(lo + a.per4, a.n4);
	if (ok) {
		for (unsigned i = lo + blockIdx.x * 256 + threadIdx.x; i < hi; i += gridDim.x * 256) {
			float4 v[kMaxWorld];
#pragma unroll
			for (int r = 0; r < kMaxWorld; r++)
				if (r < a.world) v[r] = load16_nt(a.src[r], i);
			float4 s = v[0];
#pragma unroll
			for (int r = 1; r < kMaxWorld; r++)
				if (r < a.world) { s.x += v[r].x; s.y += v[r].y; s.z += v[r].z; s.w += v[r].w; }
			reinterpret_cast<float4*>(a.red_self)[i - lo] = s;
			deliver(a, i, s);
		}
	}
	// The workgroup that leaves phase A last publishes "my reduced slice is ready" to every peer.  Ordering:
	//   writer workgroup w : stores to red_self -> every wave s_waitcnt vmcnt(0) (its stores are acknowledged by w's L2; the barrier
	//                        alone does not wait for them, and lane 0's fence only covers its own wave's queue) -> barrier ->
	//                        lane 0: system-scope RELEASE fence (writes w's XCD-private L2 back: the peers read over xGMI from memory,
	//                        they do not snoop any L2 -- so every writing workgroup needs its own, the last arriver's would only
	//                        clean the last arriver's L2) -> arrival RMW
	//   last arriver       : arrival RMW with ACQUIRE: it reads the end of the RMW chain, which continues every writer's release
	//                        sequence, so all writers' fences synchronise with it -> system-scope RELEASE store of flag B
	//   peer               : system-scope ACQUIRE load of flag B -> barrier -> loads of red[]
	// A workgroup that wrote nothing has nothing to publish and skips the fence (it still takes part in the count).
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		if (lo + blockIdx.x * 256 < hi) __threadfence_system();
		s_last = __hip_atomic_fetch_add(a.state + 3, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
	}
	__syncthreads();
	if (s_last) {
		if (threadIdx.x == 0) a.state[3] = 0;
		if ((int)threadIdx.x < a.world && (int)threadIdx.x != a.rank)
			__hip_atomic_store(a.peer_flags_b[threadIdx.x] + a.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	// phase B: the peers' reduced slices
	ok = wait_flags(a.flags_b, a.world, a.rank, epoch, &s_fail, a.timeout_ticks) && ok;
	if (ok) {
		const unsigned rest = a.n4 - (hi - lo);   // groups owned by the peers
		for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < rest; j += gridDim.x * 256) {
			const unsigned i = j < lo ? j : j + (hi - lo);      // skip my own slice
			const unsigned owner = i / a.per4;
			deliver(a, i, load16_nt(a.red[owner], i - owner * a.per4));
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_fail) atomicOr(a.state + 2, 1u);
		// No fence: every workgroup consumed state[0] (its epoch) long before its own arrival, so the last arriver's store cannot be seen by
		// a read of this launch; the next launch is ordered by the stream.
		if (atomicAdd(a.state + 1, 1u) == gridDim.x - 1) {
			a.state[1] = 0;
			__hip_atomic_store(a.state, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

__global__ void __launch_bounds__(256) dp_allreduce_kernel(DpKernelArgs a) {
	__shared__ int s_fail;
	const unsigned epoch = a.state[0] + 1;   // bumped by the last workgroup of this launch, after every workgroup has read it
	if (threadIdx.x == 0) s_fail = 0;
	__syncthreads();
	// this rank's gradient kernels finished before this launch; the system-scope release makes their bytes visible to the peers
	if (!a.flags_posted && blockIdx.x == 0 && (int)threadIdx.x < a.world && (int)threadIdx.x != a.rank)
		__hip_atomic_store(a.peer_flags[threadIdx.x] + a.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	if ((int)threadIdx.x < a.world && (int)threadIdx.x != a.rank) {   // (a rank's own data is ordered by its stream: no flag to itself)
		const long long t0 = wall_clock64();
		while ((int)(__hip_atomic_load(a.flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
			__builtin_amdgcn_s_sleep(4);
			if (wall_clock64() - t0 > a.timeout_ticks) { s_fail = 1; break; }
		}
	}
	__syncthreads();
	const bool fail = s_fail != 0;
	if (!fail) {
		for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < a.n4; i += gridDim.x * 256) {
			// Plain 16-byte loads, all in flight together (one xGMI round trip per element group, not one per peer): the
			// system-scope acquire on the flags above (buffer_inv sc0 sc1, then the workgroup barrier) already discarded every
			// line this CU or this L2 could hold of the peers' fine-grained buckets.
			float4 v[kMaxWorld];
#pragma unroll
			for (int r = 0; r < kMaxWorld; r++)
				if (r < a.world) {
					typedef float vf4 __attribute__((ext_vector_type(4)));
					vf4 x = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(a.src[r]) + i);
					v[r] = make_float4(x.x, x.y, x.z, x.w);
				}
			float4 s = v[0];
#pragma unroll
			for (int r = 1; r < kMaxWorld; r++)
				if (r < a.world) { s.x += v[r].x; s.y += v[r].y; s.z += v[r].z; s.w += v[r].w; }   // rank order: identical on every rank
			if (4 * i + 4 <= a.count) {
				if (a.out) reinterpret_cast<float4*>(a.out)[i] = s;
				if (a.target) {
					float4 t = reinterpret_cast<float4*>(a.target)[i];
					t.x += a.alpha * s.x; t.y += a.alpha * s.y; t.z += a.alpha * s.z; t.w += a.alpha * s.w;
					reinterpret_cast<float4*>(a.target)[i] = t;
				}
			} else {   // ragged tail of a foreign (unpadded) destination
				const float sv[4] = {s.x, s.y, s.z, s.w};
				for (unsigned j = 0; 4 * i + j < a.count; j++) {
					if (a.out) a.out[4 * i + j] = sv[j];
					if (a.target) a.target[4 * i + j] += a.alpha * sv[j];
				}
			}
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		if (fail) atomicOr(a.state + 2, 1u);
		if (atomicAdd(a.state + 1, 1u) == gridDim.x - 1) {   // last workgroup: every workgroup has read state[0] (no fence needed, see above)
			a.state[1] = 0;
			__hip_atomic_store(a.state, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}
}  // namespace

struct bla_dp {
	unsigned long long id;       // never reused (an address can be)
	int rank, world, device;
	size_t count, padded;        // floats; padded to a multiple of 1024
	char* base; size_t bytes;    // own fine-grained allocation: [bucket 0][bucket 1][reduced 0][reduced 1]
	char* flags;                 // own flag words, a separate UNCACHED allocation: [flags A: 2048 B][flags B: 2048 B].  The peers write
	                             // them over xGMI straight into this GPU's memory while this GPU polls them: they must never sit in its L2
	bool flags_uncached;
	size_t red_off;
	unsigned per4;               // float4 groups per slice of the two-shot form
	bool twoshot;
	unsigned* state;             // local (ordinary device memory)
	void* peer[kMaxWorld];       // peer data allocations (own slot = base): IPC mappings, or plain pointers for ranks of this process
	void* peer_flags[kMaxWorld];
	bool peer_ipc[kMaxWorld];    // opened through hipIpcOpenMemHandle (to be closed)
	bool connected;
	unsigned resident_blocks;    // workgroups of the two-shot kernel this device holds at once (occupancy x CUs), measured at create
	long long timeout_ticks;
	DoneHook* hook;              // device copy of the "gradients ready" post for the last gradient kernel (built by connect; NULL for one rank)
	unsigned long long steps;    // data-parallel steps issued through this object: the trainers take their bucket parity from here
};

// What one rank tells the others (BLA_DP_HANDLE_BYTES opaque bytes): IPC handles for other processes, plain pointers for ranks that
// live in the same process (hipIpcOpenMemHandle refuses a handle of the opening process itself).
struct DpHandle {
	hipIpcMemHandle_t data, flags;
	unsigned long long pid, nonce;   // same process <=> both equal
	unsigned long long base, flags_base;
	int device;
	unsigned magic;
};
static_assert(sizeof(DpHandle) <= BLA_DP_HANDLE_BYTES, "handle blob too small");
static unsigned long long process_nonce() {
	static const unsigned long long n = ((unsigned long long)std::random_device{}() << 32) ^ std::random_device{}() ^ (unsigned long long)(uintptr_t)&kMaxWorld;
	return n;
}
static std::atomic<unsigned long long> g_next_dp_id{1};
namespace bla {
unsigned long long dp_identity(const bla_dp* dp) { return dp ? dp->id : 0; }
int dp_parity(const bla_dp* dp) { return (int)(dp->steps & 1); }
void dp_advance(bla_dp* dp) { dp->steps++; }
}

extern "C" {

bla_status bla_dp_create(bla_dp** out, int rank, int world, size_t count) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(out && count > 0 && count < ((size_t)1 << 31), BLA_ERR_INVALID, "null / empty / oversized argument");
	BLA_REQUIRE(world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world, BLA_ERR_INVALID, "rank %d / world %d (max %d)", rank, world, kMaxWorld);
	bla_dp* dp = new bla_dp();
	dp->id = g_next_dp_id.fetch_add(1);
	dp->rank = rank; dp->world = world; dp->count = count; dp->device = ctx().device;
	dp->padded = (count + 1023) / 1024 * 1024;
	const unsigned n4 = (unsigned)((count + 3) / 4);
	dp->per4 = (n4 + world - 1) / world;
	dp->red_off = 2 * dp->padded * sizeof(float);
	dp->bytes = dp->red_off + 2 * (size_t)dp->per4 * 16;
	// two shots pay a second flag round to move 2/R of a bucket per link instead of a whole one: worth it from four ranks up
	const char* algo = getenv("BLA_DP_ALGO");
	dp->twoshot = algo ? strcmp(algo, "twoshot") == 0 : world >= 4;
	void *p = nullptr, *f = nullptr, *s = nullptr;
	hipError_t e = hipExtMallocWithFlags(&p, dp->bytes, hipDeviceMallocFinegrained);
	if (e != hipSuccess) { delete dp; return hip_fail(e, "hipExtMallocWithFlags(fine-grained exchange buffer)"); }
	dp->flags_uncached = hipExtMallocWithFlags(&f, 4096, hipDeviceMallocUncached) == hipSuccess;
	if (!dp->flags_uncached) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&f, 4096, hipDeviceMallocFinegrained); }
	// (on the context's own non-blocking stream, then a device-wide wait: NULL-stream work is not ordered against that stream)
	if (e == hipSuccess) e = hipMemsetAsync(p, 0, dp->bytes, ctx().stream);
	if (e == hipSuccess) e = hipMemsetAsync(f, 0, 4096, ctx().stream);
	if (e == hipSuccess) e = hipMalloc(&s, 64);
	if (e == hipSuccess) e = hipMemsetAsync(s, 0, 64, ctx().stream);
	if (e == hipSuccess) e = hipDeviceSynchronize();
	if (e != hipSuccess) { (void)hipFree(p); if (f) (void)hipFree(f); if (s) (void)hipFree(s); delete dp; return hip_fail(e, "exchange state allocation"); }
	dp->base = (char*)p; dp->flags = (char*)f; dp->state = (unsigned*)s;
	for (int r = 0; r < kMaxWorld; r++) { dp->peer[r] = nullptr; dp->peer_flags[r] = nullptr; dp->peer_ipc[r] = false; }
	dp->peer[rank] = dp->base; dp->peer_flags[rank] = dp->flags;
	dp->connected = world == 1;
	dp->hook = nullptr;
	dp->steps = 0;
	// The two-shot kernel's second phase waits for workgroups of the same launch (here and on the peers): its grid must be resident at once.  What
	// "at once" allows is asked of the runtime (occupancy of that kernel x CUs), not assumed.
	int occ = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dp_allreduce_twoshot_kernel, 256, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 1; }
	dp->resident_blocks = (unsigned)occ * (unsigned)(ctx().num_cus > 0 ? ctx().num_cus : 256);
	const char* tmo = getenv("BLA_DP_TIMEOUT_MS");
	const long long ms = tmo && atoll(tmo) > 0 ? atoll(tmo) : 4000;
	dp->timeout_ticks = ms * kTicksPerMs;
	*out = dp;
	return BLA_OK;
}

bla_status bla_dp_destroy(bla_dp* dp) {
	if (!dp) return BLA_OK;
	(void)hipSetDevice(dp->device);
	(void)hipDeviceSynchronize();
	for (int r = 0; r < dp->world; r++)
		if (r != dp->rank && dp->peer_ipc[r]) {
			if (dp->peer[r]) (void)hipIpcCloseMemHandle(dp->peer[r]);
			if (dp->peer_flags[r]) (void)hipIpcCloseMemHandle(dp->peer_flags[r]);
		}
	(void)hipFree(dp->base);
	(void)hipFree(dp->flags);
	(void)hipFree(dp->state);
	if (dp->hook) (void)hipFree(dp->hook);
	delete dp;
	if (ctx().ready) (void)hipSetDevice(ctx().device);
	return BLA_OK;
}

/* BLA_DP_HANDLE_BYTES opaque bytes that the other ranks pass to bla_dp_connect. */
bla_status bla_dp_export(bla_dp* dp, void* handle) {
	BLA_REQUIRE(dp && handle, BLA_ERR_INVALID, "null argument");
	DpHandle h;
	memset(&h, 0, sizeof h);
	BLA_HIP(hipIpcGetMemHandle(&h.data, dp->base));
	BLA_HIP(hipIpcGetMemHandle(&h.flags, dp->flags));
	h.pid = (unsigned long long)getpid(); h.nonce = process_nonce();
	h.base = (unsigned long long)(uintptr_t)dp->base; h.flags_base = (unsigned long long)(uintptr_t)dp->flags;
	h.device = dp->device; h.magic = 0xB1A0D902u;
	memset(handle, 0, BLA_DP_HANDLE_BYTES);
	memcpy(handle, &h, sizeof h);
	return BLA_OK;
}

/* handles: world x BLA_DP_HANDLE_BYTES, slot r = what rank r exported (the own slot is ignored).  Every rank must have created its
 * exchange object before any rank connects (the caller's handle exchange is that barrier).  Ranks of another process are mapped
 * through IPC; ranks of this process are addressed directly (peer access is enabled when they sit on another device). */
bla_status bla_dp_connect(bla_dp* dp, const void* handles) {
	BLA_REQUIRE(dp && handles, BLA_ERR_INVALID, "null argument");
	BLA_REQUIRE(!dp->connected || dp->world == 1, BLA_ERR_INVALID, "already connected");
	BLA_HIP(hipSetDevice(dp->device));
	for (int r = 0; r < dp->world; r++) {
		if (r == dp->rank) continue;
		DpHandle h;
		memcpy(&h, (const char*)handles + (size_t)r * BLA_DP_HANDLE_BYTES, sizeof h);
		BLA_REQUIRE(h.magic == 0xB1A0D902u, BLA_ERR_INVALID, "slot %d does not hold a bla_dp_export blob", r);
		if (h.pid == (unsigned long long)getpid() && h.nonce == process_nonce()) {
			if (h.device != dp->device) {
				hipError_t e = hipDeviceEnablePeerAccess(h.device, 0);
				if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
				else if (e != hipSuccess) { set_error("hipDeviceEnablePeerAccess(device %d -> %d): %s", dp->device, h.device, hipGetErrorString(e)); return BLA_ERR_HIP; }
			}
			dp->peer[r] = (void*)(uintptr_t)h.base; dp->peer_flags[r] = (void*)(uintptr_t)h.flags_base;
			continue;
		}
		void *p = nullptr, *f = nullptr;
		hipError_t e = hipIpcOpenMemHandle(&p, h.data, hipIpcMemLazyEnablePeerAccess);
		if (e == hipSuccess) e = hipIpcOpenMemHandle(&f, h.flags, hipIpcMemLazyEnablePeerAccess);
		if (e != hipSuccess) {
			if (p) (void)hipIpcCloseMemHandle(p);
			set_error("hipIpcOpenMemHandle(rank %d's exchange buffer): %s", r, hipGetErrorString(e));
			return BLA_ERR_HIP;
		}
		dp->peer[r] = p; dp->peer_flags[r] = f; dp->peer_ipc[r] = true;
	}
	// the "gradients ready" post for the last gradient kernel of a step: arrival counter = state[4], epoch = state[0], the peers' flag arrays A
	if (dp->world > 1 && !dp->hook) {
		DoneHook h;
		memset(&h, 0, sizeof h);
		h.arrive = dp->state + 4; h.epoch = dp->state; h.world = dp->world; h.rank = dp->rank;
		for (int r = 0; r < dp->world; r++) h.peer_flags[r] = (unsigned*)dp->peer_flags[r];
		void* d = nullptr;
		BLA_HIP(hipMalloc(&d, sizeof h));
		hipError_t e = hipMemcpyAsync(d, &h, sizeof h, hipMemcpyHostToDevice, ctx().stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
		if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "upload of the gradients-ready hook"); }
		dp->hook = (DoneHook*)d;
	}
	dp->connected = true;
	if (ctx().ready) (void)hipSetDevice(ctx().device);
	return BLA_OK;
}

float* bla_dp_bucket(bla_dp* dp, int parity) { return dp ? (float*)(dp->base + (size_t)(parity & 1) * dp->padded * sizeof(float)) : nullptr; }
size_t bla_dp_count(const bla_dp* dp) { return dp ? dp->count : 0; }

/* sum_i = SUM over ranks r (ascending) of bucket_r[parity][i];  d_out[i] = sum_i (if d_out);  d_target[i] += alpha * sum_i
 * (if d_target).  Collective: every rank enqueues the same sequence of calls with the same parity; parities alternate.
 * Asynchronous on `stream`, capturable into a hipGraph (the epoch lives in device memory). */
bla_status bla_dp_allreduce_f32(bla_dp* dp, void* stream, int parity, float* d_out, float* d_target, float alpha) {
	return bla::dp_allreduce(dp, stream, parity, d_out, d_target, alpha, false);
}

}  // extern "C"

const DoneHook* bla::dp_done_hook(bla_dp* dp) {
	static const bool off = [] { const char* e = getenv("BLA_DP_POST"); return e && e[0] == '0'; }();   // BLA_DP_POST=0: the exchange launch pushes the flags itself
	return dp && !off ? dp->hook : nullptr;
}

bla_status bla::dp_allreduce(bla_dp* dp, void* stream, int parity, float* d_out, float* d_target, float alpha, bool flags_posted) {
	bla_status st = require_ready();
	if (st) return st;
	BLA_REQUIRE(dp, BLA_ERR_INVALID, "null exchange");
	BLA_REQUIRE(dp->connected, BLA_ERR_INVALID, "bla_dp_connect has not run");
	BLA_REQUIRE(((uintptr_t)d_out | (uintptr_t)d_target) % 16 == 0, BLA_ERR_INVALID, "out / target must be 16-byte aligned");
	DpKernelArgs a = {};
	for (int r = 0; r < dp->world; r++) {
		a.src[r] = (const float*)((char*)dp->peer[r] + (size_t)(parity & 1) * dp->padded * sizeof(float));
		a.peer_flags[r] = (unsigned*)dp->peer_flags[r];
	}
	a.flags = (unsigned*)dp->flags;
	a.state = dp->state;
	a.out = d_out; a.target = d_target; a.alpha = alpha;
	a.n4 = (unsigned)((dp->count + 3) / 4); a.count = (unsigned)dp->count;
	a.world = dp->world; a.rank = dp->rank;
	a.timeout_ticks = dp->timeout_ticks;
	a.flags_posted = flags_posted ? 1 : 0;
	unsigned cus = (unsigned)(ctx().num_cus > 0 ? ctx().num_cus : 256);
	// Every workgroup of an exchange launch spins until the peers' launches have shown up, so all launches of a group must be resident
	// at once.  One rank per GPU: always true.  Several ranks REHEARSING on one GPU (tests, BLA_BENCH_SHARE_GPU) share its workgroup
	// slots with each other and with the gradient kernels the spinning ranks wait for: BLA_DP_MAX_BLOCKS caps the grid there.
	static const unsigned max_blocks = [] { const char* e = getenv("BLA_DP_MAX_BLOCKS"); int v = e ? atoi(e) : 0; return v > 0 ? (unsigned)v : 0u; }();
	if (max_blocks && cus > max_blocks) cus = max_blocks;
	if (dp->twoshot && dp->world > 1) {
		for (int r = 0; r < dp->world; r++) {
			a.red[r] = (const float*)((char*)dp->peer[r] + dp->red_off + (size_t)(parity & 1) * dp->per4 * 16);
			a.peer_flags_b[r] = (unsigned*)((char*)dp->peer_flags[r] + 2048);
		}
		a.red_self = (float*)(dp->base + dp->red_off + (size_t)(parity & 1) * dp->per4 * 16);
		a.flags_b = (unsigned*)(dp->flags + 2048);
		a.per4 = dp->per4;
		// the second phase waits for workgroups of this same launch (here and on the peers): keep the grid fully resident
		unsigned blocks = (a.n4 - a.per4 + 255) / 256;
		// a quarter of what the device holds at once, at most 128 (more workgroups than that do not make a 0.94 MB exchange faster): the grid stays resident
		// beside whatever else the stream's neighbours run.  BLA_DP_MAX_BLOCKS overrides (several ranks rehearsing on ONE device share its slots).
		const unsigned quarter = dp->resident_blocks / 4 ? dp->resident_blocks / 4 : 1;
		const unsigned cap = max_blocks ? cus : (quarter < 128 ? quarter : 128);
		if (blocks > cap) blocks = cap;
		if (blocks < 1) blocks = 1;
		hipLaunchKernelGGL(dp_allreduce_twoshot_kernel, dim3(blocks), dim3(256), 0, pick_stream(stream), a);
	} else {
		unsigned blocks = (a.n4 + 255) / 256;
		if (blocks > cus) blocks = cus;
		if (blocks < 1) blocks = 1;
		hipLaunchKernelGGL(dp_allreduce_kernel, dim3(blocks), dim3(256), 0, pick_stream(stream), a);
	}
	BLA_HIP(hipGetLastError());
	return BLA_OK;
}

extern "C" {

/* 0 = healthy; 1 = a wait for a peer's flag timed out in some earlier exchange (the sums of that exchange were skipped).
 * Synchronises the device. */
bla_status bla_dp_status(bla_dp* dp, int* status) {
	BLA_REQUIRE(dp && status, BLA_ERR_INVALID, "null argument");
	unsigned s[3];
	BLA_HIP(hipDeviceSynchronize());
	BLA_HIP(hipMemcpyAsync(s, dp->state, sizeof s, hipMemcpyDeviceToHost, ctx().stream));
	BLA_HIP(hipStreamSynchronize(ctx().stream));
	*status = (int)s[2];
	return BLA_OK;
}

/* BLA_ERR_TIMEOUT when a wait for a peer's flag has timed out in some earlier exchange (nothing was delivered by that exchange: out / target keep
 * what they held), else BLA_OK.  Synchronises the device. */
bla_status bla_dp_check(bla_dp* dp) {
	int st = 0;
	bla_status r = bla_dp_status(dp, &st);
	if (r) return r;
	BLA_REQUIRE(st == 0, BLA_ERR_TIMEOUT, "a rank never arrived at the gradient exchange: the wait for its flag timed out (BLA_DP_TIMEOUT_MS) and the sums were not delivered");
	return BLA_OK;
}
int bla_dp_resident_blocks(const bla_dp* dp) { return dp ? (int)dp->resident_blocks : 0; }

}  // extern "C"
