"""ctypes binding of csrc/libbla_hip.so (the C-ABI declared in include/bla.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "csrc", "libbla_hip.so")
_lib = None

ACT_NONE, ACT_RELU = 0, 1


class BlaError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"bla status {status}: {msg}")
        self.status = status


class Epilogue(C.Structure):
    """struct bla_gemm_epilogue (include/bla.h)."""
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("bias_row", C.c_void_p), ("bias_col", C.c_void_p),
                ("pre_act", C.c_void_p), ("ld_pre", C.c_int), ("act", C.c_int), ("relu_mask", C.c_void_p),
                ("ld_mask", C.c_int), ("row_sum_a", C.c_void_p), ("softmax_y", C.c_void_p), ("softmax_scale", C.c_float),
                ("softmax_grad", C.c_void_p), ("row_sum_alpha", C.c_float), ("row_sum_beta", C.c_float),
                ("softmax_loss_acc", C.c_void_p), ("softmax_correct_acc", C.c_void_p)]


class GemmDesc(C.Structure):
    """struct bla_gemm_desc (include/bla.h): one product of a bla_gemm_pair_f32 call."""
    _fields_ = [("transa", C.c_int), ("transb", C.c_int), ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int),
                ("ep", C.POINTER(Epilogue))]


class AttentionWs(C.Structure):
    """struct bla_attention_ws (include/bla.h): six device pointers."""
    _fields_ = [(n, C.c_void_p) for n in ("q", "k", "v", "scores_raw", "weights", "attention")]


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(n, C.c_void_p) for n in fields], "__doc__": f"struct {name} of include/bla.h (device pointers)"})


ResnetParams = _ptr_struct("bla_resnet_params", ("conv1", "conv2", "time_w", "time_b", "res"))
ResnetGrads = _ptr_struct("bla_resnet_grads", ("conv1", "conv2", "time_w", "time_b", "res"))
ResnetWs = _ptr_struct("bla_resnet_ws", ("mu1", "sd1", "relu1", "c1", "tdense", "mu2", "sd2", "relu2", "dp", "c2", "res"))
ResnetScratch = _ptr_struct("bla_resnet_scratch", ("g_out_a", "g_out_b", "g_in", "flip"))

_VP, _I, _F, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t
# name -> (restype, argtypes); every symbol include/bla.h declares must appear here (tests check both ways)
SIGNATURES = {
    "bla_init": (_I, [_I]), "bla_shutdown": (_I, []), "bla_is_initialized": (_I, []), "bla_device_count": (_I, []),
    "bla_last_error": (C.c_char_p, []), "bla_status_string": (C.c_char_p, [_I]), "bla_version": (C.c_char_p, []),
    "bla_device_name": (_I, [C.c_char_p, _I]),
    "bla_context_create": (_I, [C.POINTER(_VP), _I]), "bla_context_set_current": (_I, [_VP]), "bla_context_destroy": (_I, [_VP]),
    "bla_malloc": (_I, [C.POINTER(_VP), _SZ]), "bla_free": (_I, [_VP]),
    "bla_memcpy_h2d": (_I, [_VP, _VP, _SZ, _VP]), "bla_memcpy_d2h": (_I, [_VP, _VP, _SZ, _VP]),
    "bla_memcpy_d2d": (_I, [_VP, _VP, _SZ, _VP]), "bla_memset": (_I, [_VP, _I, _SZ, _VP]),
    "bla_stream_sync": (_I, [_VP]), "bla_default_stream": (_VP, []),
    "bla_event_create": (_I, [C.POINTER(_VP)]), "bla_event_destroy": (_I, [_VP]), "bla_event_record": (_I, [_VP, _VP]),
    "bla_event_elapsed_ms": (_I, [_VP, _VP, C.POINTER(_F)]),
    "bla_gemm_f32": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _I, _VP, _I, C.POINTER(Epilogue)]),
    "bla_gemm_set_config": (_I, [_I, _I]), "bla_gemm_last_kernel": (C.c_char_p, []),
    "bla_scale_f32": (_I, [_VP, _VP, _SZ, _F]), "bla_add_f32": (_I, [_VP, _VP, _VP, _SZ]),
    "bla_hadamard_f32": (_I, [_VP, _VP, _VP, _SZ]), "bla_axpy_f32": (_I, [_VP, _VP, _VP, _F, _SZ]),
    "bla_relu_f32": (_I, [_VP, _VP, _SZ]), "bla_relu_ddx_f32": (_I, [_VP, _VP, _SZ]),
    "bla_add_tile_columns_f32": (_I, [_VP, _VP, _I, _I, _VP, _I]), "bla_add_tile_rows_f32": (_I, [_VP, _VP, _I, _I, _VP]),
    "bla_transpose_f32": (_I, [_VP, _VP, _VP, _I, _I]), "bla_row_sum_f32": (_I, [_VP, _VP, _I, _I, _VP]),
    "bla_col_sum_f32": (_I, [_VP, _VP, _I, _I, _VP, _I]), "bla_frobenius_f32": (_I, [_VP, _VP, _SZ, _VP]),
    "bla_max_f32": (_I, [_VP, _VP, _SZ, _VP]), "bla_zscore_f32": (_I, [_VP, _VP, _SZ]),
    "bla_softmax_cols_f32": (_I, [_VP, _VP, _I, _I]), "bla_softmax_rows_f32": (_I, [_VP, _VP, _I, _I]),
    "bla_softmax_cols_grad_f32": (_I, [_VP, _VP, _I, _I, _VP, _F, _VP]),
    "bla_gemm_f64": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _I, _VP, _I, C.c_double, C.c_double]),
    "bla_scale_f64": (_I, [_VP, _VP, _SZ, C.c_double]), "bla_add_f64": (_I, [_VP, _VP, _VP, _SZ]), "bla_hadamard_f64": (_I, [_VP, _VP, _VP, _SZ]),
    "bla_add_tile_columns_f64": (_I, [_VP, _VP, _I, _I, _VP, _I]), "bla_add_tile_rows_f64": (_I, [_VP, _VP, _I, _I, _VP]),
    "bla_transpose_f64": (_I, [_VP, _VP, _VP, _I, _I]), "bla_row_sum_f64": (_I, [_VP, _VP, _I, _I, _VP]), "bla_col_sum_f64": (_I, [_VP, _VP, _I, _I, _VP, _I]),
    "bla_frobenius_f64": (_I, [_VP, _VP, _SZ, _VP]), "bla_max_f64": (_I, [_VP, _VP, _SZ, _VP]), "bla_zscore_f64": (_I, [_VP, _VP, _SZ]),
    "bla_im2col_f64": (_I, [_VP, _VP, _VP] + [_I] * 5), "bla_col2im_f64": (_I, [_VP, _VP, _VP] + [_I] * 5),
    "bla_kernels_to_matrix_f64": (_I, [_VP, _VP, _VP, _I, _I, _I]), "bla_matrix_to_kernels_f64": (_I, [_VP, _VP, _VP, _I, _I, _I]),
    "bla_reshape_channels_matrix_f64": (_I, [_VP, _VP, _VP, _I, _I]), "bla_reshape_matrix_channels_f64": (_I, [_VP, _VP, _VP, _I, _I]),
    "bla_conv_forward_f64": (_I, [_VP] * 7 + [_I] * 6), "bla_conv_backward_f64": (_I, [_VP] * 9 + [_I] * 6),
    "bla_group_norm_f64": (_I, [_VP] * 5 + [_I] * 3), "bla_group_norm_ddx_f64": (_I, [_VP] * 6 + [_I] * 3),
    "bla_relu_f64": (_I, [_VP, _VP, _SZ]), "bla_softmax_cols_f64": (_I, [_VP, _VP, _I, _I]), "bla_softmax_rows_f64": (_I, [_VP, _VP, _I, _I]),
    "bla_conv_out_hw": (_I, [_I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "bla_im2col_f32": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I]), "bla_col2im_f32": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I]),
    "bla_kernels_to_matrix_f32": (_I, [_VP, _VP, _VP, _I, _I, _I]), "bla_matrix_to_kernels_f32": (_I, [_VP, _VP, _VP, _I, _I, _I]),
    "bla_reshape_channels_matrix_f32": (_I, [_VP, _VP, _VP, _I, _I]), "bla_reshape_matrix_channels_f32": (_I, [_VP, _VP, _VP, _I, _I]),
    "bla_conv_forward_f32": (_I, [_VP] * 7 + [_I] * 6), "bla_conv_backward_f32": (_I, [_VP] * 9 + [_I] * 6),
    "bla_conv2d_forward_f32": (_I, [_VP] * 4 + [_I] * 6), "bla_conv2d_backward_f32": (_I, [_VP] * 7 + [_I] * 6),
    "bla_conv2d_forward_batched_f32": (_I, [_VP] * 4 + [_I] * 7), "bla_conv2d_backward_batched_f32": (_I, [_VP] * 7 + [_I] * 7),
    "bla_group_norm_f32": (_I, [_VP] * 5 + [_I] * 3), "bla_group_norm_ddx_f32": (_I, [_VP] * 6 + [_I] * 3),
    "bla_relu_mask_f32": (_I, [_VP, _VP, _VP, _VP, _SZ]), "bla_dropout_f32": (_I, [_VP, _VP, _VP, _VP, _SZ]),
    "bla_dropout_mask_f32": (_I, [_VP, _VP, _VP, _SZ]), "bla_nearest_neighbours_f32": (_I, [_VP, _VP, _VP] + [_I] * 6),
    "bla_nearest_neighbours_ddx_f32": (_I, [_VP, _VP, _VP] + [_I] * 6), "bla_softmax_ddx_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I]),
    "bla_attention_forward_f32": (_I, [_VP] * 9 + [_I] * 3), "bla_attention_backward_f32": (_I, [_VP] * 14 + [_I] * 4),
    "bla_group_norm_relu_f32": (_I, [_VP] * 5 + [_I] * 3), "bla_sum_f32": (_I, [_VP, _VP, _VP, _VP, _SZ]),
    "bla_resnet_forward_f32": (_I, [_VP] * 7 + [_I] * 7), "bla_resnet_backward_f32": (_I, [_VP] * 9 + [_I] * 7),
    "bla_layer_net_create": (_I, [C.POINTER(_VP), C.POINTER(_I), _I, _I, C.POINTER(_I), C.POINTER(_F)]), "bla_layer_net_destroy": (_I, [_VP]),
    "bla_layer_net_param_count": (_SZ, [_VP]), "bla_layer_net_params": (_VP, [_VP]), "bla_layer_net_weights": (_VP, [_VP, _I]),
    "bla_layer_net_biases": (_VP, [_VP, _I]), "bla_layer_net_nodes": (_VP, [_VP, _I]), "bla_layer_net_raw_nodes": (_VP, [_VP, _I]),
    "bla_layer_net_forward_f32": (_I, [_VP, _VP, _VP]), "bla_layer_net_backward_f32": (_I, [_VP, _VP, _VP, _F]),
    "bla_unet_create_batched": (_I, [C.POINTER(_VP), _VP, _I]), "bla_unet_batch": (_I, [_VP]),
    "bla_unet_create": (_I, [C.POINTER(_VP), _VP]), "bla_unet_destroy": (_I, [_VP]), "bla_unet_param_count": (_SZ, [_VP]),
    "bla_unet_params": (_VP, [_VP]), "bla_unet_grads": (_VP, [_VP]), "bla_unet_output": (_VP, [_VP]), "bla_unet_tensor_count": (_I, [_VP]),
    "bla_unet_tensor_info": (_I, [_VP, _I, C.POINTER(_SZ), C.POINTER(_SZ), C.c_char_p, _I]), "bla_unet_dropout_count": (_SZ, [_VP]),
    "bla_unet_forward_f32": (_I, [_VP, _VP, _VP, _VP, _VP]), "bla_unet_backward_f32": (_I, [_VP, _VP, _VP]),
    "bla_mnist_nn_create": (_I, [C.POINTER(_VP), C.POINTER(_I), _I]), "bla_mnist_nn_destroy": (_I, [_VP]),
    "bla_mnist_nn_param_count": (_SZ, [_VP]), "bla_mnist_nn_params": (_VP, [_VP]), "bla_mnist_nn_grads": (_VP, [_VP]),
    "bla_mnist_nn_input": (_VP, [_VP]), "bla_mnist_nn_labels": (_VP, [_VP]),
    "bla_mnist_nn_use_buckets": (_I, [_VP, _VP, _VP]), "bla_mnist_nn_set_params": (_I, [_VP, _VP]),
    "bla_mnist_nn_get_params": (_I, [_VP, _VP]), "bla_mnist_nn_activation": (_I, [_VP, _I, C.POINTER(_VP), C.POINTER(_I)]),
    "bla_mnist_nn_forward_backward": (_I, [_VP, _VP, _VP, _VP, _I]), "bla_mnist_nn_apply": (_I, [_VP, _VP, _F]),
    "bla_mnist_nn_train_step": (_I, [_VP, _VP, _VP, _VP, _F, _I]), "bla_mnist_nn_graph_step": (_I, [_VP, _VP, _F, _I, _I]),
    "bla_mnist_nn_fused_step": (_I, [_VP, _VP, _VP, _VP, _F, _I]),
    "bla_mnist_nn_gather_batch": (_I, [_VP, _VP, _VP, _VP, _I, _VP]), "bla_mnist_nn_forward": (_I, [_VP, _VP, _VP, _VP]),
    "bla_mnist_nn_metrics_enable": (_I, [_VP, _I]), "bla_mnist_nn_metrics_read": (_I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_longlong), _I]),
    "bla_gemm_batched_f32": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, C.c_long, _VP, _I, C.c_long, _VP, _I, C.c_long, _I, C.POINTER(Epilogue), C.c_long]),
    "bla_group_norm_relu_batched_f32": (_I, [_VP, _I] + [_VP] * 4 + [_I] * 3),
    "bla_group_norm_ddx_gated_batched_f32": (_I, [_VP, _I] + [_VP] * 5 + [_I] * 3 + [_VP] * 2),
    "bla_attention_forward_batched_f32": (_I, [_VP, _I] + [_VP] * 8 + [_I] * 3), "bla_attention_backward_batched_f32": (_I, [_VP, _I] + [_VP] * 14 + [_I] * 4),
    "bla_resnet_forward_batched_f32": (_I, [_VP, _I] + [_VP] * 6 + [_I] * 7), "bla_resnet_backward_batched_f32": (_I, [_VP, _I] + [_VP] * 9 + [_I] * 7),
    "bla_gemm_pair_f32": (_I, [_VP, _VP, _VP]), "bla_gemm_group_f32": (_I, [_VP, _VP, _I]),
    "bla_diag_mfma_rate": (_I, [_VP, _I, _I, _I, _VP]),
    "bla_graph_begin": (_I, [_VP]), "bla_graph_end": (_I, [_VP, C.POINTER(_VP)]), "bla_graph_launch": (_I, [_VP, _VP]),
    "bla_graph_destroy": (_I, [_VP]),
    "bla_dp_create": (_I, [C.POINTER(_VP), _I, _I, C.c_size_t]), "bla_dp_destroy": (_I, [_VP]),
    "bla_dp_export": (_I, [_VP, _VP]), "bla_dp_connect": (_I, [_VP, _VP]),
    "bla_dp_bucket": (_VP, [_VP, _I]), "bla_dp_count": (C.c_size_t, [_VP]),
    "bla_dp_allreduce_f32": (_I, [_VP, _VP, _I, _VP, _VP, _F]), "bla_dp_status": (_I, [_VP, C.POINTER(_I)]),
    "bla_dp_rccl_unique_id": (_I, [_VP]), "bla_dp_rccl_init": (_I, [C.POINTER(_VP), _VP, _I, _I]), "bla_dp_rccl_destroy": (_I, [_VP]),
    "bla_dp_rccl_allreduce_f32": (_I, [_VP, _VP, _VP, _SZ]), "bla_mnist_nn_dp_step_rccl": (_I, [_VP, _VP, _VP, _F, _I]),
    "bla_dp_check": (_I, [_VP]), "bla_dp_resident_blocks": (_I, [_VP]),
    "bla_dp_rccl_available": (_I, []), "bla_dp_rccl_init_all": (_I, [C.POINTER(_VP), C.POINTER(_I), _I]), "bla_dp_rccl_group_begin": (_I, []),
    "bla_dp_rccl_group_end": (_I, []), "bla_rand_guard_enter": (None, []), "bla_rand_guard_leave": (None, []),
    "bla_mnist_nn_dp_step": (_I, [_VP, _VP, _VP, _F, _I]), "bla_mnist_nn_dp_step_direct": (_I, [_VP, _VP, _VP, _F, _I]),
}


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (file name without version, SONAME
    libamdhip64.so.7); libbla_hip.so asks for libamdhip64.so.7.  If we are loaded before torch, the dynamic linker
    would satisfy that from /opt/rocm and a later `import torch` would bring a SECOND runtime into the process
    (torch then reports no device, and stream handles are not interchangeable).  Loading torch's copy first makes
    our NEEDED entry resolve to it by SONAME, whichever import order the application uses.  Pure C programs
    (the drop-in host layer) never see torch and simply use the system runtime."""
    from .build import torch_lib_dir
    tl = torch_lib_dir()
    if tl:
        try:
            C.CDLL(os.path.join(tl, "libamdhip64.so"), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The loaded C-ABI library.  Fails loudly when it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                               "this package has no CPU fallback")
        _preload_hip_runtime()
        L = C.CDLL(_SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise BlaError(status, lib().bla_last_error().decode())


def is_available():
    """True when the library is built and a HIP device is visible."""
    return os.path.exists(_SO) and lib().bla_device_count() > 0


def init(device=0):
    check(lib().bla_init(device))


def sync(stream=None):
    check(lib().bla_stream_sync(stream))


class DeviceArray:
    """A dense row-major fp32 matrix (or any shape) in HBM, owned via bla_malloc."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().bla_malloc(C.byref(p), max(self.nbytes, 4)))
        self.ptr = p.value

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                lib().bla_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    @property
    def ld(self):
        return self.shape[-1]

    def copy_from(self, a, stream=None):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.shape == self.shape, (a.shape, self.shape)
        check(lib().bla_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes, stream))
        sync(stream)  # pageable source must stay alive until the copy is done
        return self

    def numpy(self, stream=None):
        out = np.empty(self.shape, self.dtype)
        check(lib().bla_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, stream))
        sync(stream)
        return out

    def fill_bytes(self, byte, stream=None):
        check(lib().bla_memset(self.ptr, byte, self.nbytes, stream))
        return self


def to_device(a, dtype=np.float32):
    a = np.asarray(a)
    return DeviceArray(a.shape, dtype).copy_from(a)


def empty(shape, dtype=np.float32):
    return DeviceArray(shape, dtype)


def zeros(shape, dtype=np.float32):
    return DeviceArray(shape, dtype).fill_bytes(0)


def _ptr(x):
    if x is None:
        return None
    return x.ptr if isinstance(x, DeviceArray) else int(x)


def gemm_desc(a, b, c, transa=False, transb=False, epilogue=None):
    """Descriptor for gemm_pair: op(A) op(B) -> C on device arrays; `epilogue` is an Epilogue (kept alive by the caller)."""
    m = a.shape[1] if transa else a.shape[0]
    k = a.shape[0] if transa else a.shape[1]
    n = b.shape[0] if transb else b.shape[1]
    return GemmDesc(int(transa), int(transb), m, n, k, a.ptr, a.ld, b.ptr, b.ld, c.ptr, c.ld,
                    C.pointer(epilogue) if epilogue is not None else None)


def gemm_pair(p, q, stream=None):
    check(lib().bla_gemm_pair_f32(stream, C.byref(p), C.byref(q)))


def gemm(a, b, c, transa=False, transb=False, alpha=1.0, beta=0.0, bias_row=None, bias_col=None, pre_act=None,
         act=ACT_NONE, relu_mask=None, stream=None, m=None, n=None, k=None, lda=None, ldb=None, ldc=None,
         row_sum_a=None, softmax_y=None, softmax_scale=0.0, softmax_grad=None, row_sum_alpha=0.0, row_sum_beta=0.0):
    """C = epilogue(alpha * op(A) op(B)) on device arrays; shapes default to the arrays' own."""
    if m is None:
        m = a.shape[1] if transa else a.shape[0]
    if k is None:
        k = a.shape[0] if transa else a.shape[1]
    if n is None:
        n = b.shape[0] if transb else b.shape[1]
    kb = b.shape[1] if transb else b.shape[0]
    if isinstance(b, DeviceArray) and kb != k and ldb is None:
        raise BlaError(2, f"inner dimensions differ: {k} vs {kb}")
    ep = Epilogue(alpha, beta, _ptr(bias_row), _ptr(bias_col), _ptr(pre_act), pre_act.ld if pre_act is not None else 0,
                  act, _ptr(relu_mask), relu_mask.ld if relu_mask is not None else 0, _ptr(row_sum_a), _ptr(softmax_y),
                  softmax_scale, _ptr(softmax_grad), row_sum_alpha, row_sum_beta)
    check(lib().bla_gemm_f32(stream, int(transa), int(transb), m, n, k, _ptr(a), lda or a.ld, _ptr(b), ldb or b.ld,
                             _ptr(c), ldc or c.ld, C.byref(ep)))
    return c
