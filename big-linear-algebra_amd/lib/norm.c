/* norm.c -- host side of the drop-in lib/norm.h (reference lib/norm.c:5-93): channels are gathered into one
 * [C][H*W] device buffer, one workgroup per group does the mean / variance / normalise passes. */
#include "norm.h"
#include "bla_host.h"

/* one copy per channel array (bla_host_up_planes packs the per-channel Matrix planes first) */
#define gather bla_host_up_planes
#define scatter bla_host_down_planes

void group_norm(Matrix* in, Matrix* out, matrix_float_t* stdevs, matrix_float_t* means, int channels, int group_size) {
	int hw = in[0].rows * in[0].cols, groups = (channels + group_size - 1) / group_size;
	float* din = gather(0, in, channels);
	float* dout = bla_host_buf(1, (size_t)hw * channels);
	float* dsd = bla_host_buf(2, groups);
	float* dmu = bla_host_buf(3, groups);
	BLA_TRY(bla_group_norm_f32(NULL, din, dout, dsd, dmu, channels, group_size, hw));
	bla_host_down(stdevs, dsd, groups);
	bla_host_down(means, dmu, groups);
	scatter(out, channels, dout);
}

void group_norm_ddx(Matrix* source, Matrix* dest, Matrix* data, matrix_float_t* means, matrix_float_t* stdevs, int channels, int group_size) {
	int hw = source[0].rows * source[0].cols, groups = (channels + group_size - 1) / group_size;
	float* dsrc = gather(0, source, channels);
	float* ddata = gather(1, data, channels);
	float* ddest = bla_host_buf(2, (size_t)hw * channels);
	float* dmu = bla_host_up(3, means, groups);
	float* dsd = bla_host_up(4, stdevs, groups);
	BLA_TRY(bla_group_norm_ddx_f32(NULL, dsrc, ddest, ddata, dmu, dsd, channels, group_size, hw));
	scatter(dest, channels, ddest);
}
