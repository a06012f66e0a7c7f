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
	BLA_ERR_UNDEFINED = 5, /* the reference itself is undefined here (e.g. col2im with stride != 1) */
	BLA_ERR_TIMEOUT = 6    /* a rank of the gradient exchange never arrived (bla_dp_check) */
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

/* Further contexts beside the default one bla_init creates: a context = {device, stream, split-K workspace, arrival counters}.
 * A host that drives several GPUs -- or several replicas on one GPU -- from ONE process ("single-process multi-device,
 * one host thread or one stream per device") creates one context per rank and makes it current on the calling thread before it issues
 * that rank's bla_* calls; NULL = back to the default context.  Objects (trainers, exchange objects, device memory) belong to the
 * device of the context that was current when they were created. */
typedef struct bla_context bla_context;
BLA_API bla_status bla_context_create(bla_context** out, int device);
BLA_API bla_status bla_context_set_current(bla_context* c);
BLA_API bla_status bla_context_destroy(bla_context* c);   /* BLA_ERR_INVALID while the context is current on another thread (set NULL there first, also before that thread ends) */

/* libc rand() belongs to the calling program (the reference's programs seed it once and draw from it between library calls: sampler
 * lib/mnist_csv2.c:36-62, dropout model/cifar_unet.c:1032-1042), but HIP runtime start-up and RCCL set-up draw from it too.  Every library
 * entry that reaches them parks the caller's stream by itself; a host program that makes such calls ITSELF brackets them with this pair.
 * Process-wide and nestable: the first enter (any thread) parks the stream, the last leave puts it back.  While a guard is open other threads
 * must not draw from rand() and expect the program's stream. */
BLA_API void bla_rand_guard_enter(void);
BLA_API void bla_rand_guard_leave(void);

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

/* Launch-bound sequences (one image through a U-Net block is a dozen launches of a few microseconds): record any sequence of
 * bla_* calls issued on `stream` between begin and end into a hipGraph and replay it with one launch.  Run the sequence once
 * eagerly first -- scratch buffers and gather tables are created on first use, and allocating is not allowed while recording.
 * The recorded calls are not executed during recording; pointers and sizes are frozen into the graph. */
/* diagnostics: `blocks` workgroups of `waves` wavefronts each issue iters x 8 independent fp32 MFMAs (32x32x2) and nothing else --
 * the practical ceiling of the matrix pipe (tools/mfma_peak.py) */
BLA_API bla_status bla_diag_mfma_rate(void* stream, int blocks, int waves, int iters, float* d_sink);
BLA_API bla_status bla_graph_begin(void* stream);
BLA_API bla_status bla_graph_end(void* stream, void** graph);
BLA_API bla_status bla_graph_launch(void* graph, void* stream);
BLA_API bla_status bla_graph_destroy(void* graph);

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
 * A NULL epilogue means alpha = 1 and nothing else.  When passing a struct, zero-initialise it and set alpha.
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
	/* by-products fused into the latency-bound kernels (run as separate passes elsewhere):
	 *   row_sum_a[r] = sum_k A[r][k] (needs !transa): with A = dZ this is the bias gradient "sum over the batch
	 *   columns", the documented intent of matrix_col_sum (model/mnist_nn.c:271,282,293);
	 *   softmax_y/softmax_grad (m <= 32, no other post-op than bias_row/pre_act): C = column softmax of the result,
	 *   softmax_grad = (C - softmax_y) * softmax_scale, both with leading dimension ldc (model/mnist_nn.c:234,260-268). */
	float* row_sum_a;
	const float* softmax_y;
	float softmax_scale;
	float* softmax_grad;
	/* row_sum_a[r] = row_sum_beta * row_sum_a[r] + row_sum_alpha * sum_k A[r][k]; both 0 (a zero-initialised struct) = plain store of the
	 * sum.  With alpha = learn rate, beta = 1 the bias update b += lr * db rides along the weight-gradient product (latency-bound kernels only). */
	float row_sum_alpha, row_sum_beta;
	/* with the fused softmax tail: per-COLUMN accumulators (length n) of the reference's loss / accuracy bookkeeping, model/mnist_nn.c:237-257:
	 *   softmax_loss_acc[c]    += -sum_r y[r][c] * log(p[r][c] + 1e-15)            (double, LOSS_EPSILON of :15)
	 *   softmax_correct_acc[c] += y[pred][c] == 1, pred = first row with the largest probability (> 0)
	 * both NULL = off.  Summed over the columns they are the batch totals the reference adds into its epoch averages. */
	double* softmax_loss_acc;
	unsigned* softmax_correct_acc;
} bla_gemm_epilogue;

BLA_API bla_status bla_gemm_f32(void* stream, int transa, int transb, int m, int n, int k,
                        const float* d_a, int lda, const float* d_b, int ldb,
                        float* d_c, int ldc, const bla_gemm_epilogue* ep /* NULL = plain product */);
/* Two independent products (neither reads what the other writes) issued together: when both are latency-bound shapes they
 * share one launch and overlap -- dW_l = dZ_l.A^T beside dZ_{l-1} = W_l^T.dZ_l in model/mnist_nn.c:267-289, which the
 * reference runs one after the other.  Otherwise identical to two bla_gemm_f32 calls. */
typedef struct bla_gemm_desc {
	int transa, transb, m, n, k;
	const float* A; int lda;
	const float* B; int ldb;
	float* C; int ldc;
	const bla_gemm_epilogue* ep;   /* may be NULL */
} bla_gemm_desc;
BLA_API bla_status bla_gemm_pair_f32(void* stream, const bla_gemm_desc* p, const bla_gemm_desc* q);
/* Up to three independent products; three latency-bound ones with K-contiguous operands on both sides (transa = 0, transb = 1: the three
 * weight gradients of one MNIST-NN step, model/mnist_nn.c:267-292) share ONE launch.  Otherwise identical to separate calls. */
BLA_API bla_status bla_gemm_group_f32(void* stream, const bla_gemm_desc* descs, int count);

/* Tuning/diagnostics: force a tile configuration (-1 = automatic) and split-K factor (0 = automatic).  Configurations
 * (csrc/bla_gemm.hip): 0-2 register-staged tiles (any shape); 3/4/5/7 direct-to-LDS 128x128x16, 64x64x16, 128x128x32, 128x64x16
 * (16-byte aligned operands, k a multiple of the slab depth); 6 wave-split-K 32x32 for latency-bound shapes (16: its 16x16-tile form, which the automatic
 * choice takes when the 32x32 tiling would leave CUs idle); 8/9 256x128-class
 * three-buffer tiles; 10 persistent 128x128; 11/12/13 the one-workgroup-per-CU half-slab kernels 256x256x16, 256x256x32, 128x512x16
 * and 14 / 15 / 17 their 128x128, 128x256 and 192x192 forms (whole tiles, plain alpha epilogue only; the automatic choice takes the one whose tile count
 * is a whole number of rounds over the CUs); 18: 64x64 tiles on 32-deep slabs with two wave groups along K (1024^3-class products, one or two tiles
 * per CU).  A forced configuration that cannot take the call fails with
 * BLA_ERR_INVALID; the automatic choice never does. */
BLA_API bla_status bla_gemm_set_config(int config, int split_k);
/* Name of the kernel variant the last bla_gemm_f32 call launched (for profiles/). */
BLA_API const char* bla_gemm_last_kernel(void);

/* ---- elementwise / broadcast / transpose / reductions: lib/matrix.c:59-205, lib/util.c:7-55 -------------
 * All operate in place on device memory exactly like their reference counterparts operate on host
 * memory; n = rows*cols.  HBM-bound; sums accumulate in fp64 and are rounded to fp32 once. */
BLA_API bla_status bla_scale_f32(void* stream, float* d_m, size_t n, float f);                 /* matrix_scale, lib/matrix.c:59-63 */
BLA_API bla_status bla_add_f32(void* stream, float* d_a, const float* d_b, size_t n);          /* matrix_add: a += b, lib/matrix.c:65-69 */
BLA_API bla_status bla_hadamard_f32(void* stream, float* d_a, const float* d_b, size_t n);     /* matrix_multiply_elementwise, lib/matrix.c:95-103 (shape check is the caller's) */
BLA_API bla_status bla_axpy_f32(void* stream, float* d_y, const float* d_x, float alpha, size_t n); /* y += alpha*x: matrix_scale+matrix_add pair, model/mnist_nn.c:303-315 */
BLA_API bla_status bla_relu_f32(void* stream, float* d, size_t n);                             /* relu, lib/util.c:7-13 */
BLA_API bla_status bla_relu_ddx_f32(void* stream, float* d, size_t n);                         /* relu_ddx, model/mnist_nn.c:47-51 */
BLA_API bla_status bla_add_tile_columns_f32(void* stream, float* d_a, int a_rows, int a_cols, const float* d_b, int b_cols); /* lib/matrix.c:189-195 */
BLA_API bla_status bla_add_tile_rows_f32(void* stream, float* d_a, int a_rows, int a_cols, const float* d_b);                /* lib/matrix.c:199-205 */
BLA_API bla_status bla_transpose_f32(void* stream, const float* d_in, float* d_out, int rows, int cols);  /* out (cols x rows) = in^T; matrix_transpose, lib/matrix.c:105-118 */
BLA_API bla_status bla_row_sum_f32(void* stream, const float* d_m, int rows, int cols, float* d_out);     /* 1 x cols, matrix_row_sum, lib/matrix.c:123-133 */
/* matrix_col_sum, lib/matrix.c:138-148.  AS_WRITTEN reproduces out[i] = sum_{j<cols} flat[i*rows+j] and returns
 * BLA_ERR_UNDEFINED where the reference reads out of bounds (rows > cols); INTENDED gives true row sums. */
enum { BLA_COLSUM_AS_WRITTEN = 0, BLA_COLSUM_INTENDED = 1 };
BLA_API bla_status bla_col_sum_f32(void* stream, const float* d_m, int rows, int cols, float* d_out, int mode);
BLA_API bla_status bla_frobenius_f32(void* stream, const float* d_m, size_t n, float* d_out);  /* d_out[0] = sqrt(sum x^2), lib/matrix.c:150-158 */
BLA_API bla_status bla_max_f32(void* stream, const float* d_m, size_t n, float* d_out);        /* d_out[0] = max (-inf when empty), lib/matrix.c:160-168 */
BLA_API bla_status bla_zscore_f32(void* stream, float* d_m, size_t n);                         /* matrix_z_score_normalize, lib/matrix.c:170-185 */
BLA_API bla_status bla_softmax_cols_f32(void* stream, float* d, int rows, int cols);           /* softmax per column, lib/util.c:15-34 */
BLA_API bla_status bla_softmax_rows_f32(void* stream, float* d, int rows, int cols);           /* softmax_row_wise, lib/util.c:36-55 */
/* softmax per column, then d_grad = (softmax - y) * scale in the same pass (model/mnist_nn.c:234,263-268) */
BLA_API bla_status bla_softmax_cols_grad_f32(void* stream, float* d, int rows, int cols, const float* d_y, float scale, float* d_grad);

/* ---- the matrix.h path in the reference's own element type (lib/matrix.h:4: double) for the -DBLA_FP64 build of the host layer
 * (SURVEY 7.0(1)): same meaning as the _f32 entries above, GEMM on v_mfma_f64_16x16x4_f64.  This is the <= 1e-12 comparison mode against the
 * reference's CPU results (only the order of additions differs), not a performance path. */
BLA_API bla_status bla_gemm_f64(void* stream, int transa, int transb, int m, int n, int k, const double* d_a, int lda, const double* d_b, int ldb,
                                double* d_c, int ldc, double alpha, double beta);                             /* lib/matrix.c:35-57 */
BLA_API bla_status bla_scale_f64(void* stream, double* d_m, size_t n, double f);                             /* :59-63 */
BLA_API bla_status bla_add_f64(void* stream, double* d_a, const double* d_b, size_t n);                      /* :65-69 */
BLA_API bla_status bla_hadamard_f64(void* stream, double* d_a, const double* d_b, size_t n);                 /* :95-103 */
BLA_API bla_status bla_add_tile_columns_f64(void* stream, double* d_a, int a_rows, int a_cols, const double* d_b, int b_cols);   /* :189-195 */
BLA_API bla_status bla_add_tile_rows_f64(void* stream, double* d_a, int a_rows, int a_cols, const double* d_b);                  /* :199-205 */
BLA_API bla_status bla_transpose_f64(void* stream, const double* d_in, double* d_out, int rows, int cols);   /* :105-118 */
BLA_API bla_status bla_row_sum_f64(void* stream, const double* d_m, int rows, int cols, double* d_out);      /* :123-133 */
BLA_API bla_status bla_col_sum_f64(void* stream, const double* d_m, int rows, int cols, double* d_out, int mode);   /* :138-148 */
BLA_API bla_status bla_frobenius_f64(void* stream, const double* d_m, size_t n, double* d_out);              /* :150-158 */
BLA_API bla_status bla_max_f64(void* stream, const double* d_m, size_t n, double* d_out);                    /* :160-168 */
BLA_API bla_status bla_zscore_f64(void* stream, double* d_m, size_t n);                                      /* :170-185 (sigma through sqrtf, as there) */
/* conv.h / norm.h / util.h in the same element type (csrc/bla_f64_conv.hip): the index maps of lib/conv.c, conv() / conv_ddx() with their two
 * products on the f64 GEMM, group norm and its gradient, relu, the two softmaxes -- argument meaning as the _f32 entries of the same name. */
BLA_API bla_status bla_im2col_f64(void* stream, const double* d_x, double* d_out, int h, int w, int k, int c_in, int stride);                 /* lib/conv.c:8-77 */
BLA_API bla_status bla_col2im_f64(void* stream, const double* d_cols, double* d_out, int h, int w, int k, int c_n, int stride);              /* :80-135 */
BLA_API bla_status bla_kernels_to_matrix_f64(void* stream, const double* d_kern, double* d_mat, int f_n, int c_n, int k);                     /* :138-153 */
BLA_API bla_status bla_matrix_to_kernels_f64(void* stream, const double* d_mat, double* d_kern, int f_n, int c_n, int k);                     /* :156-171 */
BLA_API bla_status bla_reshape_channels_matrix_f64(void* stream, double* d_channels, const double* d_matrix, int c_n, int hw);                /* :174-187, as written */
BLA_API bla_status bla_reshape_matrix_channels_f64(void* stream, double* d_matrix, const double* d_channels, int c_n, int hw);                /* :190-203, as written */
BLA_API bla_status bla_conv_forward_f64(void* stream, const double* d_x, const double* d_kern, double* d_im2col, double* d_kmat, double* d_product, double* d_output,
                                        int h, int w, int k, int c_in, int f_n, int stride);                                                  /* :205-212 */
BLA_API bla_status bla_conv_backward_f64(void* stream, const double* d_del_y, const double* d_im2col, const double* d_kmat, double* d_del_q, double* d_del_kmat,
                                         double* d_del_kern, double* d_del_col, double* d_del_x, int h, int w, int k, int c_in, int f_n, int stride);   /* :214-229 */
BLA_API bla_status bla_group_norm_f64(void* stream, const double* d_in, double* d_out, double* d_stdevs, double* d_means, int channels, int group_size, int hw);   /* lib/norm.c:5-50 */
BLA_API bla_status bla_group_norm_ddx_f64(void* stream, const double* d_source, double* d_dest, const double* d_data, const double* d_means, const double* d_stdevs,
                                          int channels, int group_size, int hw);                                                               /* :52-93 */
BLA_API bla_status bla_relu_f64(void* stream, double* d, size_t n);                                                                            /* lib/util.c:7-13 */
BLA_API bla_status bla_softmax_cols_f64(void* stream, double* d, int rows, int cols);                                                          /* :15-34 */
BLA_API bla_status bla_softmax_rows_f64(void* stream, double* d, int rows, int cols);                                                          /* :36-55 */

/* ---- convolution stages, lib/conv.c.  Images are contiguous [C][H][W]; kernels [F][C][k][k]; workspaces are
 * the reference's ConvData members (lib/conv.h:6-11): im2col [Ho*Wo][k*k*C], kernel_matrix [k*k*C][F],
 * product [Ho*Wo][F], output [F][Ho][Wo].  TF "SAME" padding, Ho = ceil((float)H/stride) (lib/conv.c:13-28,55-56). */
BLA_API bla_status bla_conv_out_hw(int h, int w, int stride, int* ho, int* wo);
BLA_API bla_status bla_im2col_f32(void* stream, const float* d_x, float* d_out, int h, int w, int k, int c_in, int stride);       /* _im2col, lib/conv.c:8-77 */
/* _col2im, lib/conv.c:80-135: the reference is defined for stride 1 only (it walks the image grid instead of the output grid).
 * Other strides give the INTENDED operation, the adjoint of _im2col (d_cols is [Ho*Wo][k*k*C], d_out [C][h][w]); with BLA_STRICT_REFERENCE=1
 * in the environment they return BLA_ERR_UNDEFINED instead. */
BLA_API bla_status bla_col2im_f32(void* stream, const float* d_cols, float* d_out, int h, int w, int k, int c_n, int stride);
BLA_API bla_status bla_kernels_to_matrix_f32(void* stream, const float* d_kern, float* d_mat, int f_n, int c_n, int k);          /* _reshape_kernels_matrix, lib/conv.c:138-153 */
BLA_API bla_status bla_matrix_to_kernels_f32(void* stream, const float* d_mat, float* d_kern, int f_n, int c_n, int k);          /* _reshape_matrix_kernels, lib/conv.c:156-171 */
/* The two channel reshapes keep the reference's names, argument order AND as-written direction:
 * reshape_channels_matrix(channels, matrix) writes channels <- matrix; reshape_matrix_channels(matrix, channels) writes matrix <- channels. */
BLA_API bla_status bla_reshape_channels_matrix_f32(void* stream, float* d_channels, const float* d_matrix, int c_n, int hw);     /* lib/conv.c:174-187 */
BLA_API bla_status bla_reshape_matrix_channels_f32(void* stream, float* d_matrix, const float* d_channels, int c_n, int hw);     /* lib/conv.c:190-203 */
/* conv(), lib/conv.c:205-212, intended composition (GEMM result reaches output; as written the last step overwrites product
 * from the stale output and never writes output -- the host layer offers that literal mode too). */
BLA_API bla_status bla_conv_forward_f32(void* stream, const float* d_x, const float* d_kern, float* d_im2col, float* d_kmat, float* d_product,
                                        float* d_output, int h, int w, int k, int c_in, int f_n, int stride);
/* conv_ddx(), lib/conv.c:214-229, intended composition.  h, w are the INPUT's size; del_y is [F][Ho][Wo].  Stride 1 is the reference's only
 * defined case; other strides use the adjoint _col2im above (BLA_STRICT_REFERENCE=1: BLA_ERR_UNDEFINED). */
BLA_API bla_status bla_conv_backward_f32(void* stream, const float* d_del_y, const float* d_im2col, const float* d_kmat, float* d_del_q,
                                         float* d_del_kmat, float* d_del_kern, float* d_del_col, float* d_del_x, int h, int w, int k, int c_in,
                                         int f_n, int stride);
/* Implicit-GEMM convolution for device-resident callers: the im2col matrix is gathered inside the MFMA kernel and
 * never written (no ConvData workspaces).  Values equal conv()'s `output` / conv_ddx()'s del_kernels and del_input
 * (lib/conv.c:205-229, intended composition).  Any stride.  The data gradient at stride != 1 (undefined in the reference) is
 * the adjoint of the forward map -- the stride-1 convolution of the zero-dilated del_y with the flipped kernels -- as the U-Net's three
 * down-convolutions need it (model/cifar_unet.c:1105,1111,1115; backward :1412,1420,1430); BLA_STRICT_REFERENCE=1 refuses it
 * (BLA_ERR_UNDEFINED).  d_scratch: F*C*k*k floats, needed only when d_del_x != NULL. */
BLA_API bla_status bla_conv2d_forward_f32(void* stream, const float* d_x, const float* d_kern, float* d_out, int h, int w, int k, int c_in, int f_n, int stride);
BLA_API bla_status bla_conv2d_backward_f32(void* stream, const float* d_del_y, const float* d_x, const float* d_kern, float* d_del_kern,
                                           float* d_del_x, float* d_scratch, int h, int w, int k, int c_in, int f_n, int stride);
/* `batch` images through the same kernels in one launch (the reference has no batch dimension: one conv() / conv_ddx() call
 * per image, model/cifar_unet.c:1105-1165).  x [B][C][H][W], out / del_y [B][F][Ho][Wo], del_x [B][C][H][W];
 * del_kern = SUM over the images of the per-image weight gradient, folded in image order. */
BLA_API bla_status bla_conv2d_forward_batched_f32(void* stream, const float* d_x, const float* d_kern, float* d_out, int batch, int h, int w, int k,
                                                  int c_in, int f_n, int stride);
BLA_API bla_status bla_conv2d_backward_batched_f32(void* stream, const float* d_del_y, const float* d_x, const float* d_kern, float* d_del_kern,
                                                   float* d_del_x, float* d_scratch, int batch, int h, int w, int k, int c_in, int f_n, int stride);
/* group_norm / group_norm_ddx, lib/norm.c:5-93, on [C][H*W]; quirk Q3 kept (epsilon == 0, "stdevs" holds the variance,
 * out = (x - mean) / variance).  Note the reference's argument orders (lib/norm.h:6-7). */
BLA_API bla_status bla_group_norm_f32(void* stream, const float* d_in, float* d_out, float* d_stdevs, float* d_means, int channels, int group_size, int hw);
BLA_API bla_status bla_group_norm_ddx_f32(void* stream, const float* d_source, float* d_dest, const float* d_data, const float* d_means,
                                          const float* d_stdevs, int channels, int group_size, int hw);

/* ---- U-Net glue ops around the conv path, model/cifar_unet.c; channel arrays are [C][H*W] ----
 * _add_time_embedding (:1024-1030) is bla_add_tile_columns_f32(x, C, H*W, t, 1); _concat_skip / _split_concat
 * (:1088-1097,1339-1349) are bla_memcpy_d2d on channel ranges; the time-bias gradient (:1191-1196) is
 * bla_col_sum_f32(..., BLA_COLSUM_INTENDED). */
BLA_API bla_status bla_relu_mask_f32(void* stream, float* d_dest, const float* d_source, const float* d_relu_result, size_t n);   /* multi_channel_relu_ddx, :241-253 */
/* _dropout (:1032-1042): the reference draws `(float) rand() / RAND_MAX < DROPOUT_RATE` per element in order; the host
 * makes those draws (same libc stream) and passes them as d_drop (non-zero = dropped). */
BLA_API bla_status bla_dropout_f32(void* stream, const float* d_x, float* d_y, const unsigned char* d_drop, size_t n);
BLA_API bla_status bla_dropout_mask_f32(void* stream, float* d_x, const float* d_dropout_result, size_t n);                         /* _dropout_mask, :1168-1178 */
BLA_API bla_status bla_nearest_neighbours_f32(void* stream, const float* d_in, float* d_out, int channels, int in_h, int in_w, int out_h, int out_w, int scale);   /* :1074-1086 */
BLA_API bla_status bla_nearest_neighbours_ddx_f32(void* stream, const float* d_source, float* d_dest, int channels, int src_h, int src_w, int dest_h, int dest_w, int scale);   /* :1229-1244 */
BLA_API bla_status bla_softmax_ddx_f32(void* stream, const float* d_softmax_output, const float* d_gradient, float* d_out, int rows, int dim);   /* _softmax_ddx, :1246-1259 */

/* Self-attention block of the U-Net (model/cifar_unet.c:999-1022 forward, :1261-1337 backward), device-resident, with the
 * channel reshapes in the direction their call sites need (as written they are swapped and the block reads stale
 * buffers).  x/out/del_y/del_x: [C][S], S = H*W; wq/wk/wv: [C][d]; w: [d][C]; bias: [C].
 * Workspaces (caller allocated): q,k,v,attention [S][d]; scores_raw (= attention_weights_raw, scaled scores) and
 * weights (= attention_weights, softmax) [S][S].  The backward's gradient workspace reuses the struct:
 * q,k,v,attention = del_Q,del_K,del_V,del_P; scores_raw = del_I; weights = del_S.
 * jacobian_from_raw != 0 reproduces :1307 literally (_softmax_ddx gets the RAW scores where the softmax output is meant). */
typedef struct bla_attention_ws { float *q, *k, *v, *scores_raw, *weights, *attention; } bla_attention_ws;
BLA_API bla_status bla_attention_forward_f32(void* stream, const float* d_x, const float* d_wq, const float* d_wk, const float* d_wv, const float* d_w,
                                             const float* d_bias, const bla_attention_ws* ws, float* d_out, int c, int s, int d);
BLA_API bla_status bla_attention_backward_f32(void* stream, const float* d_del_y, const float* d_x, const float* d_wq, const float* d_wk,
                                              const float* d_wv, const float* d_w, const bla_attention_ws* fwd, const bla_attention_ws* grad,
                                              float* d_del_wq, float* d_del_wk, float* d_del_wv, float* d_del_w, float* d_del_x, int c, int s, int d,
                                              int jacobian_from_raw);

/* ResNet block of the U-Net (model/cifar_unet.c:1044-1072 forward, :1180-1227 backward), device-resident, intended
 * composition (conv() delivers its result; gradients land in the gradient struct -- as written :1203,1216 hand conv_ddx
 * the parameter kernels as the sink).  x/del_x: [Cin][H*W]; result/del_out: [Cout][H*W]; temb: [T];
 * conv1 [Cout][Cin][k][k]; conv2 [Cout][Cout][k][k]; time_w [T][Cout]; time_b [Cout]; res [Cout][Cin][1][1] or NULL when
 * Cin == Cout; d_drop [Cout*H*W]: the host's rand() draws, non-zero = dropped.  Stride 1. */
typedef struct bla_resnet_params { const float *conv1, *conv2, *time_w, *time_b, *res; } bla_resnet_params;
typedef struct bla_resnet_grads { float *conv1, *conv2, *time_w, *time_b, *res; } bla_resnet_grads;
/* saved by the forward pass for the backward pass: mu/sd per group; relu1 [Cin][HW]; c1 (conv 1 + time embedding), relu2, dp
 * (dropout output), c2, res (residual conv output, may be NULL when Cin == Cout) [Cout][HW]; tdense [Cout] */
typedef struct bla_resnet_ws { float *mu1, *sd1, *relu1, *c1, *tdense, *mu2, *sd2, *relu2, *dp, *c2, *res; } bla_resnet_ws;
/* backward scratch: g_out_a, g_out_b [Cout][HW]; g_in [Cin][HW]; flip [Cout*max(Cin,Cout)*k*k] */
typedef struct bla_resnet_scratch { float *g_out_a, *g_out_b, *g_in, *flip; } bla_resnet_scratch;
BLA_API bla_status bla_group_norm_relu_f32(void* stream, const float* d_in, float* d_out, float* d_stdevs, float* d_means, int channels, int group_size, int hw);   /* group_norm then relu, fused */
BLA_API bla_status bla_sum_f32(void* stream, float* d_out, const float* d_a, const float* d_b, size_t n);   /* out = a + b, :1067-1071 */
BLA_API bla_status bla_resnet_forward_f32(void* stream, const float* d_x, const float* d_temb, const bla_resnet_params* p, const unsigned char* d_drop,
                                          const bla_resnet_ws* ws, float* d_result, int h, int w, int cin, int cout, int k, int tdim, int group_size);
BLA_API bla_status bla_resnet_backward_f32(void* stream, const float* d_del_out, const float* d_x, const float* d_temb, const bla_resnet_params* p,
                                           const bla_resnet_ws* ws, const bla_resnet_grads* g, const bla_resnet_scratch* sc, float* d_del_x, int h, int w,
                                           int cin, int cout, int k, int tdim, int group_size);

/* The same blocks for a batch of images (what a mini-batch of the reference's one-image-at-a-time training loop computes, gradients summed over the
 * images): x / out / del_* [B][C][H*W], temb [B][T] (every image its own time step), d_drop [B][Cout*H*W]; every workspace / scratch buffer is B
 * times its single-image size (tdense [B][Cout], mu / sd [B][groups]).  The convolutions run as batched implicit GEMMs, the norms over B*C
 * channels, each per-image product of the attention block as one launch over the batch.  batch = 1 is the single-image entry point.
 * d_partials: [B][C*d] scratch (per-image weight gradients before their sum); d_dtb: [B][Cout] scratch.  bla_resnet_backward_batched_f32 takes
 * d_del_x = NULL when nothing consumes the gradient of the block's input (a network's first block): only the weight gradients are formed. */
BLA_API bla_status bla_gemm_batched_f32(void* stream, int transa, int transb, int m, int n, int k, const float* A, int lda, long stride_a, const float* B, int ldb,
                                        long stride_b, float* C, int ldc, long stride_c, int batch, const bla_gemm_epilogue* ep, long stride_pre);
BLA_API bla_status bla_group_norm_relu_batched_f32(void* stream, int batch, const float* d_in, float* d_out, float* d_stdevs, float* d_means, int channels,
                                                   int group_size, int hw);
BLA_API bla_status bla_group_norm_ddx_gated_batched_f32(void* stream, int batch, const float* d_source, float* d_dest, const float* d_data, const float* d_means,
                                                        const float* d_stdevs, int channels, int group_size, int hw, const float* d_relu_gate,
                                                        const float* d_addend);
BLA_API bla_status bla_attention_forward_batched_f32(void* stream, int batch, const float* d_x, const float* d_wq, const float* d_wk, const float* d_wv,
                                                     const float* d_w, const float* d_bias, const bla_attention_ws* ws, float* d_out, int c, int s, int d);
BLA_API bla_status bla_attention_backward_batched_f32(void* stream, int batch, const float* d_del_y, const float* d_x, const float* d_wq, const float* d_wk,
                                                      const float* d_wv, const float* d_w, const bla_attention_ws* fwd, const bla_attention_ws* grad,
                                                      float* d_partials, float* d_del_wq, float* d_del_wk, float* d_del_wv, float* d_del_w, float* d_del_x,
                                                      int c, int s, int d, int jacobian_from_raw);
BLA_API bla_status bla_resnet_forward_batched_f32(void* stream, int batch, const float* d_x, const float* d_temb, const bla_resnet_params* p,
                                                  const unsigned char* d_drop, const bla_resnet_ws* ws, float* d_result, int h, int w, int cin, int cout, int k,
                                                  int tdim, int group_size);
BLA_API bla_status bla_resnet_backward_batched_f32(void* stream, int batch, const float* d_del_out, const float* d_x, const float* d_temb,
                                                   const bla_resnet_params* p, const bla_resnet_ws* ws, const bla_resnet_grads* g, const bla_resnet_scratch* sc,
                                                   float* d_dtb, float* d_del_x, int h, int w, int cin, int cout, int k, int tdim, int group_size);

/* ---- lib/layer.h on the device, batched: feed_forward (lib/layer.c:6-20) and back_propagate_errors with its recursion
 * (:48-107) for `batch` samples (columns) at once, parameters resident in one bucket (W_1, b_1, W_2, b_2, ...; W_l is n_l x n_{l-1}).  The
 * reference's activation callbacks become one of a few device functions; with batch = 1 the arithmetic is layer.c's step by step, with more
 * columns the weight / bias steps are summed over the columns (all gradients taken at the weights as they were before the call). */
enum { BLA_LAYER_ACT_IDENTITY = 0, BLA_LAYER_ACT_SCALE = 1 /* a = p x, a' = p (main.c:7-17 with p = 0.1) */, BLA_LAYER_ACT_RELU = 2,
       BLA_LAYER_ACT_LEAKY = 3 /* a = x < 0 ? p x : x */ };
typedef struct bla_layer_net bla_layer_net;
/* sizes[num_layers] incl. the input layer; acts / act_params [num_layers - 1] for the computing layers */
BLA_API bla_status bla_layer_net_create(bla_layer_net** out, const int* sizes, int num_layers, int batch, const int* acts, const float* act_params);
BLA_API bla_status bla_layer_net_destroy(bla_layer_net* m);
BLA_API size_t bla_layer_net_param_count(const bla_layer_net* m);
BLA_API float* bla_layer_net_params(bla_layer_net* m);
BLA_API float* bla_layer_net_weights(bla_layer_net* m, int layer);     /* device, layer >= 1 */
BLA_API float* bla_layer_net_biases(bla_layer_net* m, int layer);
BLA_API float* bla_layer_net_nodes(bla_layer_net* m, int layer);       /* [n_layer][batch] after a forward pass */
BLA_API float* bla_layer_net_raw_nodes(bla_layer_net* m, int layer);
/* d_x is read again by the backward pass (it is the first computing layer's a_prev, lib/layer.c:67): keep it valid until then */
BLA_API bla_status bla_layer_net_forward_f32(bla_layer_net* m, void* stream, const float* d_x /* [n_0][batch] */);
BLA_API bla_status bla_layer_net_backward_f32(bla_layer_net* m, void* stream, const float* d_expect /* [n_L][batch] */, float learn_rate);

/* ---- the U-Net of model/cifar_unet.c assembled from the blocks above: forward() (:1099-1166) and backward() (:1351-1436) for one image.
 * 18 ResNet blocks, 5 self-attention blocks, 3 stride-2 convolutions, 3 nearest-neighbour up-samplings (+ a convolution where the widths of the
 * two resolutions differ), 4 skip concatenations, output group norm + ReLU + convolution.  Parameters and gradients live in two flat buckets;
 * bla_unet_tensor_info enumerates the tensors (names follow the reference's struct members, e.g. "down_2_resnet_1.conv_1_kernels",
 * "mid_self_attention.Q_proj", "up_3_conv_kernels") in the order forward() first uses them.  The wiring is the INTENDED network: see
 * csrc/bla_unet_model.hip for the four places where the reference's work-in-progress call sites differ.
 * Reference constants (:26-37): image 32 x 32 x 3, dims {128, 256, 256, 256}, time_dim 512, kernel 3, group_size 32, key_dim 16. */
typedef struct bla_unet_config { int image_h, image_w, in_channels, dims[4], time_dim, kernel, group_size, key_dim; } bla_unet_config;
typedef struct bla_unet bla_unet;
BLA_API bla_status bla_unet_create(bla_unet** out, const bla_unet_config* cfg);
/* The same network for `batch` images per pass: d_x / the output / d_noise are [B][C][H][W], d_time_embedding [B][time_dim] (every image its own
 * time step), the gradients are summed over the images (what `batch` passes of the reference's one-image loop accumulate).  d_drop: the blocks
 * in forward order, inside a block image by image.  bla_unet_create = batch 1. */
BLA_API bla_status bla_unet_create_batched(bla_unet** out, const bla_unet_config* cfg, int batch);
BLA_API int bla_unet_batch(const bla_unet* m);
BLA_API bla_status bla_unet_destroy(bla_unet* m);
BLA_API size_t bla_unet_param_count(const bla_unet* m);         /* floats in each bucket (every tensor starts 16-byte aligned) */
BLA_API float* bla_unet_params(bla_unet* m);                    /* device */
BLA_API float* bla_unet_grads(bla_unet* m);                     /* device; written by bla_unet_backward_f32 */
BLA_API float* bla_unet_output(bla_unet* m);                    /* device, [in_channels][H][W]: the predicted noise */
BLA_API int bla_unet_tensor_count(const bla_unet* m);
BLA_API bla_status bla_unet_tensor_info(const bla_unet* m, int index, size_t* offset, size_t* count, char* name, int name_len);
/* _dropout (:1032-1042) draws one decision per element of every ResNet block's second ReLU, in forward order: d_drop holds
 * bla_unet_dropout_count() of them (non-zero = dropped; the host makes the draws from its own rand() stream), NULL = keep everything. */
BLA_API size_t bla_unet_dropout_count(const bla_unet* m);
BLA_API bla_status bla_unet_forward_f32(bla_unet* m, void* stream, const float* d_x /* [C][H][W] */, const float* d_time_embedding /* [time_dim] */,
                                        const unsigned char* d_drop);
/* del_Y = 2 (prediction - noise) (:1353-1364), then every block backwards; uses the activations of the last forward pass */
BLA_API bla_status bla_unet_backward_f32(bla_unet* m, void* stream, const float* d_noise /* [C][H][W] */);

/* ---- device-resident MNIST-NN trainer: the hot loop of model/mnist_nn.c:218-315 with everything in HBM -------
 * sizes = {n0, n1, n2, n3} (784, 256, 128, 10 in the reference, model/mnist_nn.c:25-28); samples are columns.
 * Parameters sit in one flat bucket ordered W1,b1,W2,b2,W3,b3 (each row-major), gradients in a second bucket of
 * the same layout (un-scaled sums over the batch columns).  Data parallelism = SUM-all-reduce the gradient
 * bucket between bla_mnist_nn_forward_backward and bla_mnist_nn_apply.
 * colsum_mode: BLA_COLSUM_AS_WRITTEN reproduces matrix_col_sum literally (needs n_i <= batch, else
 * BLA_ERR_UNDEFINED); BLA_COLSUM_INTENDED uses true row sums (required for sharded batches). */
typedef struct bla_mnist_nn bla_mnist_nn;
BLA_API bla_status bla_mnist_nn_create(bla_mnist_nn** out, const int* sizes /* [4] */, int batch);
BLA_API bla_status bla_mnist_nn_destroy(bla_mnist_nn* nn);
BLA_API size_t bla_mnist_nn_param_count(const bla_mnist_nn* nn);
BLA_API float* bla_mnist_nn_params(bla_mnist_nn* nn);     /* device pointer, param_count floats */
BLA_API float* bla_mnist_nn_grads(bla_mnist_nn* nn);      /* device pointer, param_count floats */
BLA_API float* bla_mnist_nn_input(bla_mnist_nn* nn);      /* resident raw-pixel buffer [n0][batch] (used when d_x_raw == NULL) */
BLA_API float* bla_mnist_nn_labels(bla_mnist_nn* nn);     /* resident one-hot buffer [n3][batch] (used when d_y == NULL) */
/* Adopt caller-owned buckets (e.g. tensors a collective library registered); current parameters are copied over. */
BLA_API bla_status bla_mnist_nn_use_buckets(bla_mnist_nn* nn, float* d_params, float* d_grads);
BLA_API bla_status bla_mnist_nn_set_params(bla_mnist_nn* nn, const float* h_flat);
BLA_API bla_status bla_mnist_nn_get_params(bla_mnist_nn* nn, float* h_flat);
BLA_API bla_status bla_mnist_nn_activation(bla_mnist_nn* nn, int which /* 0..8: z1,a1,z2,a2,z3,a3,dz3,dz2,dz1 */, float** d_ptr, int* rows);
BLA_API bla_status bla_mnist_nn_forward_backward(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, int colsum_mode);
BLA_API bla_status bla_mnist_nn_apply(bla_mnist_nn* nn, void* stream, float lr /* reference: (float)-0.02 */);
BLA_API bla_status bla_mnist_nn_train_step(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, float lr, int colsum_mode);
/* ---- the rest of the reference's training loop, device-resident (examples/mnist_nn_gpu.c drives these from C) ----
 * Batch construction, model/mnist_nn.c:204-217: with the whole dataset resident in HBM in the reference's feature-major layout
 * (lib/mnist_csv2.c: X[pixel * num_examples + example], labels y[example]) the host only sends the example indices its sampler drew;
 * x_raw[p][k] = X[p * num_examples + idx[k]], one-hot[label][k] = 1 land in the trainer's resident input / label buffers. */
BLA_API bla_status bla_mnist_nn_gather_batch(bla_mnist_nn* nn, void* stream, const float* d_X, const float* d_labels, int num_examples,
                                             const int* d_indices);
/* Forward pass only (model/mnist_nn.c:221-234; the whole of run(), :447-463): fills z1..a3 (and dz3, which run() ignores). */
BLA_API bla_status bla_mnist_nn_forward(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y);
/* Loss / accuracy bookkeeping (model/mnist_nn.c:237-257, run(): :476-490) inside the output layer's launch: once enabled every forward
 * pass adds its batch's cross-entropy (double) and its number of correct predictions to device-side accumulators -- no extra launch, no
 * host round trip per batch.  read: synchronises, returns the totals since the last reset (summed over the batch columns in order). */
BLA_API bla_status bla_mnist_nn_metrics_enable(bla_mnist_nn* nn, int on);
BLA_API bla_status bla_mnist_nn_metrics_read(bla_mnist_nn* nn, double* loss_sum, long long* num_correct, int reset);

/* One step with the update folded into the weight-gradient products (see graph_step below), issued directly on the stream: six launches
 * from one host call, no gradient bucket.  Falls back to bla_mnist_nn_train_step where the fused form does not apply. */
BLA_API bla_status bla_mnist_nn_fused_step(bla_mnist_nn* nn, void* stream, const float* d_x_raw, const float* d_y, float lr, int colsum_mode);
/* Same step from the resident buffers, captured once into a hipGraph and replayed (launch-bound otherwise).  with_update = 1 and
 * BLA_COLSUM_INTENDED on the reference's layer sizes takes the fused form: the update rides inside the weight-gradient products
 * (W += lr * dZ.A^T, b += lr * rowsum(dZ)), six launches, and the gradient bucket is NOT written; with_update = 0 always fills it. */
BLA_API bla_status bla_mnist_nn_graph_step(bla_mnist_nn* nn, void* stream, float lr, int colsum_mode, int with_update);

/* ---- data-parallel exchange (the one exchange step of the MNIST-NN path) ------------------------
 * The reference has no multi-device code; this is what a data-parallel driver of model/mnist_nn.c needs between
 * backward (:260-293) and the update (:296-315): SUM of the flat gradient bucket over the ranks (the reference's
 * gradient is a sum over batch columns, so no rescale).  One process per GPU; each rank owns two gradient buckets
 * (used alternately) in fine-grained memory that the peers map through IPC and read directly over xGMI; the sum is
 * taken in rank order on every rank (bit-identical results) and the update is fused.  See csrc/bla_dp.hip. */
typedef struct bla_dp bla_dp;
#define BLA_DP_HANDLE_BYTES 256
BLA_API bla_status bla_dp_create(bla_dp** out, int rank, int world, size_t count /* floats per bucket */);
BLA_API bla_status bla_dp_destroy(bla_dp* dp);
/* BLA_DP_HANDLE_BYTES opaque bytes for the other ranks (exchange them with any host-side channel: MPI, torch.distributed, a file,
 * or a plain array when all ranks live in one process) */
BLA_API bla_status bla_dp_export(bla_dp* dp, void* handle);
/* handles: world x BLA_DP_HANDLE_BYTES, slot r = rank r's export (own slot ignored); call once, after every rank has exported.
 * Ranks in other processes are mapped through IPC, ranks of the calling process (other contexts) are addressed directly. */
BLA_API bla_status bla_dp_connect(bla_dp* dp, const void* handles);
BLA_API float* bla_dp_bucket(bla_dp* dp, int parity);   /* device pointer of this rank's bucket 0 / 1 */
BLA_API size_t bla_dp_count(const bla_dp* dp);
/* sum_i = SUM_r bucket_r[parity][i] (r ascending); d_out[i] = sum_i if d_out; d_target[i] += alpha * sum_i if d_target.
 * Collective and asynchronous on `stream`; successive calls alternate the parity; capturable into a hipGraph. */
BLA_API bla_status bla_dp_allreduce_f32(bla_dp* dp, void* stream, int parity, float* d_out, float* d_target, float alpha);
/* *status = 0 healthy, 1 = some earlier exchange gave up waiting for a peer (4 s) and skipped its sums; synchronises */
BLA_API bla_status bla_dp_status(bla_dp* dp, int* status);
/* the same as a return code: BLA_ERR_TIMEOUT when some earlier exchange gave up waiting (BLA_DP_TIMEOUT_MS, default 4000) -- that exchange delivered
 * NOTHING (out / target keep what they held; never zeros in place of sums); the exchange object is then out of step with its peers: destroy it */
BLA_API bla_status bla_dp_check(bla_dp* dp);
/* workgroups of the two-shot exchange kernel this device holds at once (occupancy x CUs, asked of the runtime at create): its grid is capped at a
 * quarter of that (at most 128), so that the phase that waits for workgroups of the same launch is always fully resident */
BLA_API int bla_dp_resident_blocks(const bla_dp* dp);
/* forward + backward + exchange + update of one data-parallel step as one graph launch (BLA_COLSUM_INTENDED only) */
BLA_API bla_status bla_mnist_nn_dp_step(bla_mnist_nn* nn, bla_dp* dp, void* stream, float lr, int colsum_mode);
/* the same step issued directly on the stream (seven launches from one host call); may be mixed with the graph form */
BLA_API bla_status bla_mnist_nn_dp_step_direct(bla_mnist_nn* nn, bla_dp* dp, void* stream, float lr, int colsum_mode);

/* ---- the same exchange through the library collective: RCCL ncclAllReduce(ncclFloat, ncclSum) over xGMI (north_star) ----
 * librccl is opened on first use, not linked: bla_dp_rccl_available() says whether it could be (a single-GPU box needs none).
 * One process (or one host thread) per rank: rank 0 makes the 128-byte unique id, the host program hands it to the other ranks over any
 * channel, every rank calls bla_dp_rccl_init on its current context's device -- ncclCommInitRank is COLLECTIVE and BLOCKS until all ranks
 * have called it, so a single host thread must not call it rank after rank.
 * ONE host thread driving all ranks of a process: bla_dp_rccl_init_all (ncclCommInitAll, distinct devices), then per step every rank's
 * bla_dp_rccl_allreduce_f32 between bla_dp_rccl_group_begin / _end.
 * Not executed with world > 1 anywhere yet: the build pool has one-GPU boxes and RCCL refuses duplicate devices (tests: world 1). */
typedef struct bla_rccl bla_rccl;
#define BLA_RCCL_ID_BYTES 128
BLA_API int bla_dp_rccl_available(void);
BLA_API bla_status bla_dp_rccl_unique_id(void* id128);
BLA_API bla_status bla_dp_rccl_init(bla_rccl** out, const void* id128, int rank, int world);   /* ncclCommInitRank; collective, blocking */
BLA_API bla_status bla_dp_rccl_init_all(bla_rccl** out, const int* devices, int world);        /* ncclCommInitAll: out[world] from one thread */
BLA_API bla_status bla_dp_rccl_group_begin(void);                                               /* ncclGroupStart */
BLA_API bla_status bla_dp_rccl_group_end(void);                                                 /* ncclGroupEnd */
BLA_API bla_status bla_dp_rccl_destroy(bla_rccl* c);
BLA_API bla_status bla_dp_rccl_allreduce_f32(bla_rccl* c, void* stream, float* d_buf, size_t count);   /* in place, SUM, async on stream */
/* forward + backward into the trainer's gradient bucket, ncclAllReduce of the bucket, params += lr * sum (model/mnist_nn.c:218-315 sharded) */
BLA_API bla_status bla_mnist_nn_dp_step_rccl(bla_mnist_nn* nn, bla_rccl* c, void* stream, float lr, int colsum_mode);

#ifdef __cplusplus
}
#endif
#endif /* BLA_H */
