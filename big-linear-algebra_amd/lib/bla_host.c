/* bla_host.c -- see bla_host.h */
#include "bla_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_started = 0;
static float* g_buf[BLA_HOST_SLOTS];
static size_t g_cap[BLA_HOST_SLOTS];

void bla_host_fail(const char* what, bla_status st) {
	fflush(stdout);
	fprintf(stderr, "big-linear-algebra (MI355X backend): %s failed: %s -- %s\n", what, bla_status_string(st), bla_last_error());
	exit(1);
}

void bla_host_init(void) {
	if (g_started) return;
	const char* dev = getenv("BLA_DEVICE");
	bla_status st = bla_init(dev ? atoi(dev) : 0);
	if (st != BLA_OK) bla_host_fail("bla_init", st);   /* no device -> no result: there is no CPU fallback */
	g_started = 1;
}

int bla_host_strict(void) {
	const char* s = getenv("BLA_STRICT_REFERENCE");
	return s && s[0] == '1';
}

float* bla_host_buf(int slot, size_t floats) {
	bla_host_init();
	if (floats == 0) floats = 1;
	if (floats > g_cap[slot]) {
		if (g_buf[slot]) BLA_TRY(bla_free(g_buf[slot]));
		size_t cap = floats + floats / 4;
		void* p = NULL;
		BLA_TRY(bla_malloc(&p, cap * sizeof(float)));
		g_buf[slot] = (float*)p;
		g_cap[slot] = cap;
	}
	return g_buf[slot];
}

float* bla_host_up(int slot, const float* h, size_t floats) {
	float* d = bla_host_buf(slot, floats);
	if (floats) BLA_TRY(bla_memcpy_h2d(d, h, floats * sizeof(float), NULL));
	return d;
}

void bla_host_down(float* h, const float* d, size_t floats) {
	if (floats) BLA_TRY(bla_memcpy_d2h(h, d, floats * sizeof(float), NULL));
	BLA_TRY(bla_stream_sync(NULL));
}
