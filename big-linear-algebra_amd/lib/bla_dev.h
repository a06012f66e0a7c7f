/* bla_dev.h -- private: the element type on the device = matrix_float_t (float by default, double under -DBLA_FP64: the reference's own type,
 * lib/matrix.h:4), the matching C-ABI entry of include/bla.h, and typed views of the staging helpers of bla_host.c (which count in floats). */
#ifndef BLA_DEV_H
#define BLA_DEV_H
#include "bla_host.h"
#ifdef BLA_FP64
typedef double bla_elem_t;
#define DEV(name) bla_##name##_f64
#else
typedef float bla_elem_t;
#define DEV(name) bla_##name##_f32
#endif
#define WORDS(n) ((size_t)(n) * (sizeof(matrix_float_t) / sizeof(float)))
static inline bla_elem_t* dev_up(int slot, const matrix_float_t* h, size_t n) { return (bla_elem_t*)bla_host_up(slot, (const float*)h, WORDS(n)); }
static inline bla_elem_t* dev_buf(int slot, size_t n) { return (bla_elem_t*)bla_host_buf(slot, WORDS(n)); }
static inline void dev_down(matrix_float_t* h, const bla_elem_t* d, size_t n) { bla_host_down((float*)h, (const float*)d, WORDS(n)); }
static inline bla_elem_t* dev_up_planes(int slot, Matrix* planes, int count) { return (bla_elem_t*)bla_host_up_planes(slot, planes, count); }
static inline void dev_down_planes(Matrix* planes, int count, const bla_elem_t* d) { bla_host_down_planes(planes, count, (const float*)d); }
#endif
