/* The drop-in contract includes libc rand(): the reference's programs seed it once (srand(42), model/mnist_nn.c:513) and draw from it
 * between library calls (the MNIST sampler, lib/mnist_csv2.c:36-62; the U-Net's dropout, model/cifar_unet.c:1032-1042).  The GPU runtime
 * behind this library must not disturb that stream.  Prints one line per stage: "ok" or the first mismatch. */
#include "bla.h"
#include <stdio.h>
#include <stdlib.h>

static int expect[4];
static int bad = 0;
/* each stage: srand(42), the library operation, then the first draws must be the seeded ones */
static void check(const char* name) {
	for (int i = 0; i < 4; i++) {
		int v = rand();
		if (v != expect[i]) { printf("%s: draw %d is %d, the seeded stream has %d\n", name, i, v, expect[i]); bad = 1; srand(42); return; }
	}
	printf("%s: ok\n", name);
	srand(42);
}
#define stage(name, draws) check(name)

int main(void) {
	srand(42);
	for (int i = 0; i < 4; i++) expect[i] = rand();
	srand(42);
	stage("before init", 4);
	if (bla_init(0) != BLA_OK) { printf("no device: %s\n", bla_last_error()); return 2; }
	stage("after bla_init", 4);
	void *a, *b, *c;
	float h[64 * 64];
	for (int i = 0; i < 64 * 64; i++) h[i] = (float)(i % 7);
	if (bla_malloc(&a, sizeof h) || bla_malloc(&b, sizeof h) || bla_malloc(&c, sizeof h)) return 3;
	stage("after bla_malloc", 4);
	bla_memcpy_h2d(a, h, sizeof h, NULL); bla_memcpy_h2d(b, h, sizeof h, NULL); bla_stream_sync(NULL);
	stage("after copies", 4);
	bla_gemm_f32(NULL, 0, 0, 64, 64, 64, a, 64, b, 64, c, 64, NULL); bla_stream_sync(NULL);
	stage("after the first launch", 4);
	bla_context* ctx;
	if (bla_context_create(&ctx, 0) == BLA_OK) { bla_context_set_current(ctx); bla_context_set_current(NULL); bla_context_destroy(ctx); }
	stage("after a second context", 4);
	bla_dp* dp;
	if (bla_dp_create(&dp, 0, 1, 1000) == BLA_OK) bla_dp_destroy(dp);
	stage("after an exchange object", 4);
	bla_rccl* comm; char id[BLA_RCCL_ID_BYTES];
	if (bla_dp_rccl_unique_id(id) == BLA_OK && bla_dp_rccl_init(&comm, id, 0, 1) == BLA_OK) {
		bla_dp_rccl_allreduce_f32(comm, NULL, c, 64); bla_stream_sync(NULL); bla_dp_rccl_destroy(comm);
	}
	stage("after an RCCL communicator", 4);
	return bad;
}
