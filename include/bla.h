/*
 * bla.h -- C-ABI of the MI355X (gfx950) backend for the dense linear-algebra hot
 * path of damians13/big-linear-algebra.
 *
 * Plain C: pointers, ints and floats only -- no HIP, torch or C++ types.  The
 * host side (big-linear-algebra_amd/lib/ *.c, compiled by gcc, API-identical to
 * the reference's lib/matrix.h, lib/conv.h, lib/norm.h, lib/util.h, lib/layer.h)
 * and the Python tests/bench bind exactly these symbols.
 *
 * Conventions
 *   - every matrix is dense row-major fp32, element (r,c) at p[r*ld + c]
 *     (the reference's Matrix layout, lib/matrix.h:6-11, with ld = cols);
 *   - pointers named d_* / "device" are device pointers (bla_malloc or any HIP
 *     allocation, e.g. a torch tensor's data_ptr());
 *   - `stream` is a hipStream_t passed as void*; NULL = the library's own stream;
 *   - every entry point returns a bla_status (0 = BLA_OK); bla_last_error()
 *     gives the text.  Nothing here ever falls back to a CPU implementation:
 *     without a usable device every compute call fails with BLA_ERR_NO_DEVICE.
 *   - launches are asynchronous on `stream`; call bla_stream_sync to wait.
 *
 * Each compute entry point cites the reference function it replaces.
 */
#ifndef BLA_H
#define BLA_H

#include <stddef.h>

#if defined(BLA_BUILDING)
#define BLA_API __attribute__((visibility("default")))
#else
#define BLA_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef int bla_status;
enum {
	BLA_OK = 0,
	BLA_ERR_INVALID = 1,   /* bad argument (null pointer, negative size, ld too small) */
	BLA_ERR_SHAPE = 2,     /* operand shapes do not conform */
	BLA_ERR_NO_DEVICE = 3, /* no usable gfx950 device / runtime not initialised */
	BLA_ERR_HIP = 4,       /* a HIP runtime call failed; see bla_last_error() */
	BLA_ERR_UNDEFINED = 5  /* the reference itself is undefined here (e.g. col2im with stride != 1) */
};

/* ---- runtime -------------------------------------------------------------- */
BLA_API bla_status bla_init(int device);            /* select device, create stream + workspace; idempotent */
BLA_API bla_status bla_shutdown(void);
BLA_API int bla_is_initialized(void);
BLA_API int bla_device_count(void);          /* 0 when no device / no driver */
BLA_API const char* bla_last_error(void);
BLA_API const char* bla_status_string(bla_status s);
BLA_API const char* bla_version(void);
BLA_API bla_status bla_device_name(char* buf, int buflen);   /* gcnArchName of the active device */

BLA_API bla_status bla_malloc(void** d_ptr, size_t bytes);
BLA_API bla_status bla_free(void* d_ptr);
BLA_API bla_status bla_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
BLA_API bla_status bla_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
BLA_API bla_status bla_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
BLA_API bla_status bla_memset(void* d_dst, int byte, size_t bytes, void* stream);
BLA_API bla_status bla_stream_sync(void* stream);
BLA_API void* bla_default_stream(void);

/* Wall-clock free device timers (HIP events on `stream`) for bench.py's roofline leg. */
BLA_API bla_status bla_event_create(void** ev);
BLA_API bla_status bla_event_destroy(void* ev);
BLA_API bla_status bla_event_record(void* ev, void* stream);
BLA_API bla_status bla_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* syncs on ev_stop */

/* ---- GEMM: replaces matrix_multiply_inplace / matrix_multiply (lib/matrix.c:35-57)
 * and every matrix_transpose + multiply + transpose-back sandwich around it
 * (model/mnist_nn.c:267-292, lib/conv.c:221-227) via transa/transb.
 *
 *   C[m x n] = epilogue( alpha * op(A)[m x k] . op(B)[k x n] )
 *   op(A) = A (m x k, lda >= k) if !transa, else A^T with A stored k x m (lda >= m)
 *   op(B) = B (k x n, ldb >= n) if !transb, else B^T with B stored n x k (ldb >= k)
 *
 * Arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32), one fp32 rounding per product,
 * fp32 accumulation; summation order over k differs from the reference's
 * k-ascending scalar chain (documented tolerance: DESIGN.md).
 *
 * Epilogue, applied in this order to v = alpha*acc (all optional, NULL/0 = off):
 *   v += bias_row[r]          bias per output ROW  (matrix_add_tile_columns with an m x 1 b, lib/matrix.c:189-195)
 *   v += bias_col[c]          bias per output COLUMN (matrix_add_tile_rows, lib/matrix.c:199-205)
 *   pre_act[r*ld_pre + c] = v   (keeps Z next to A = act(Z), model/mnist_nn.c:221-224)
 *   act == BLA_ACT_RELU: v = v < 0 ? 0 : v        (lib/util.c:7-13)
 *   relu_mask: v *= (relu_mask[r*ld_mask + c] > 0 ? 1 : 0)   (relu_ddx + hadamard, model/mnist_nn.c:276-278)
 *   beta != 0: v += beta * C[r*ldc + c]
 */
enum { BLA_ACT_NONE = 0, BLA_ACT_RELU = 1 };

typedef struct bla_gemm_epilogue {
	float alpha;               /* 0 is NOT treated specially; use 1.0f for a plain product */
	float beta;
	const float* bias_row;     /* device, length m, or NULL */
	const float* bias_col;     /* device, length n, or NULL */
	float* pre_act;            /* device m x n (ld_pre), or NULL */
	int ld_pre;
	int act;                   /* BLA_ACT_* */
	const float* relu_mask;    /* device m x n (ld_mask), or NULL */
	int ld_mask;
} bla_gemm_epilogue;

BLA_API bla_status bla_gemm_f32(void* stream, int transa, int transb, int m, int n, int k,
                        const float* d_a, int lda, const float* d_b, int ldb,
                        float* d_c, int ldc, const bla_gemm_epilogue* ep /* NULL = plain product */);

/* Tuning/diagnostics: force a tile configuration (-1 = automatic) and split-K factor (0 = automatic). */
BLA_API bla_status bla_gemm_set_config(int config, int split_k);
/* Name of the kernel variant the last bla_gemm_f32 call launched (for profiles/). */
BLA_API const char* bla_gemm_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* BLA_H */
