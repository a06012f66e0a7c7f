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

static float* g_pack[2];
static size_t g_pack_cap[2];

float* bla_host_pack_block(int which, size_t floats) {
	if (floats > g_pack_cap[which]) {
		free(g_pack[which]);
		g_pack_cap[which] = floats + floats / 4;
		g_pack[which] = malloc(g_pack_cap[which] * sizeof(float));
		if (!g_pack[which]) { fprintf(stderr, "big-linear-algebra (MI355X backend): out of host memory for a %zu-float staging block\n", floats); exit(1); }
	}
	return g_pack[which];
}

/* channel arrays (conv.h / norm.h): `count` equally sized Matrix planes of matrix_float_t <-> one contiguous device buffer */
#define PLANE_WORDS(n) ((size_t)(n) * (sizeof(matrix_float_t) / sizeof(float)))
float* bla_host_up_planes(int slot, Matrix* ch, int count) {
	const size_t per = (size_t)ch[0].rows * ch[0].cols;
	int contiguous = 1;
	for (int c = 1; c < count && contiguous; c++) contiguous = ch[c].data == ch[0].data + c * per;
	if (contiguous) return bla_host_up(slot, (const float*)ch[0].data, PLANE_WORDS(per * count));
	matrix_float_t* block = (matrix_float_t*)bla_host_pack_block(0, PLANE_WORDS(per * count));
	for (int c = 0; c < count; c++) memcpy(block + c * per, ch[c].data, per * sizeof(matrix_float_t));
	float* d = bla_host_up(slot, (const float*)block, PLANE_WORDS(per * count));
	BLA_TRY(bla_stream_sync(NULL));       /* the block is reused by the next operand */
	return d;
}

void bla_host_down_planes(Matrix* ch, int count, const float* d) {
	const size_t per = (size_t)ch[0].rows * ch[0].cols;
	matrix_float_t* block = (matrix_float_t*)bla_host_pack_block(1, PLANE_WORDS(per * count));
	bla_host_down((float*)block, d, PLANE_WORDS(per * count));
	for (int c = 0; c < count; c++) memcpy(ch[c].data, block + c * per, per * sizeof(matrix_float_t));
}
