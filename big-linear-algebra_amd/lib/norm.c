/* norm.c -- host side of the drop-in lib/norm.h (reference lib/norm.c:5-93): channels are gathered into one
 * [C][H*W] device buffer, one workgroup per group does the mean / variance / normalise passes. */
#include "norm.h"
#include "bla_dev.h"

/* one copy per channel array (bla_host_up_planes packs the per-channel Matrix planes first) */
#define gather dev_up_planes
#define scatter dev_down_planes

void group_norm(Matrix* in, Matrix* out, matrix_float_t* stdevs, matrix_float_t* means, int channels, int group_size) {
	int hw = in[0].rows * in[0].cols, groups = (channels + group_size - 1) / group_size;
	bla_elem_t* din = gather(0, in, channels);
	bla_elem_t* dout = dev_buf(1, (size_t)hw * channels);
	bla_elem_t* dsd = dev_buf(2, groups);
	bla_elem_t* dmu = dev_buf(3, groups);
	BLA_TRY(DEV(group_norm)(NULL, din, dout, dsd, dmu, channels, group_size, hw));
	dev_down(stdevs, dsd, groups);
	dev_down(means, dmu, groups);
	scatter(out, channels, dout);
}

void group_norm_ddx(Matrix* source, Matrix* dest, Matrix* data, matrix_float_t* means, matrix_float_t* stdevs, int channels, int group_size) {
	int hw = source[0].rows * source[0].cols, groups = (channels + group_size - 1) / group_size;
	bla_elem_t* dsrc = gather(0, source, channels);
	bla_elem_t* ddata = gather(1, data, channels);
	bla_elem_t* ddest = dev_buf(2, (size_t)hw * channels);
	bla_elem_t* dmu = dev_up(3, means, groups);
	bla_elem_t* dsd = dev_up(4, stdevs, groups);
	BLA_TRY(DEV(group_norm_ddx)(NULL, dsrc, ddest, ddata, dmu, dsd, channels, group_size, hw));
	scatter(dest, channels, ddest);
}
