"""Host-side mirror of the reference's MNIST-NN training loop (model/mnist_nn.c:164-394) over the
device-resident trainer of the C-ABI (bla_mnist_nn_*), plus the data-parallel wrapper.

Names follow the reference program: layer sizes 784 -> 256 -> 128 -> 10, samples are columns, one step =
forward, backward, SGD update with the float learn rate -0.02; the six parameter matrices load from / save
to the reference's CSV layout through the flat bucket (W1,b1,W2,b2,W3,b3).
"""
import ctypes as C

import numpy as np

from . import native

LAYER_SIZES = (784, 256, 128, 10)                 # model/mnist_nn.c:25-28
LEARN_RATE = float(np.float32(-0.02))             # float epoch_learn_rate = -SGD_LEARN_RATE_MULTIPLIER, :186
COLSUM_AS_WRITTEN, COLSUM_INTENDED = 0, 1
HANDLE_BYTES = 256                                # BLA_DP_HANDLE_BYTES
ACTIVATION_NAMES = ["z1", "a1", "z2", "a2", "z3", "a3", "dz3", "dz2", "dz1"]


def bucket_shapes(sizes=LAYER_SIZES):
    n0, n1, n2, n3 = sizes
    return [(n1, n0), (n1, 1), (n2, n1), (n2, 1), (n3, n2), (n3, 1)]


def flatten_params(params):
    return np.concatenate([np.ascontiguousarray(p, np.float32).ravel() for p in params])


def split_bucket(flat, sizes=LAYER_SIZES):
    out, off = [], 0
    for r, c in bucket_shapes(sizes):
        out.append(np.asarray(flat[off:off + r * c]).reshape(r, c)); off += r * c
    return out


class MnistNN:
    """One replica of the trainer on the current device."""

    def __init__(self, batch, sizes=LAYER_SIZES, colsum_mode=COLSUM_INTENDED):
        self.L = native.lib()
        self.sizes, self.batch, self.colsum_mode = tuple(sizes), int(batch), colsum_mode
        h = C.c_void_p()
        native.check(self.L.bla_mnist_nn_create(C.byref(h), (C.c_int * 4)(*self.sizes), self.batch))
        self.h = h
        self.count = int(self.L.bla_mnist_nn_param_count(h))

    def close(self):
        if getattr(self, "h", None):
            self.L.bla_mnist_nn_destroy(self.h)
            self.h = None

    __del__ = close

    # -- parameters --------------------------------------------------------------------------------------
    def set_params(self, params):
        flat = flatten_params(params) if isinstance(params, (list, tuple)) else np.ascontiguousarray(params, np.float32)
        assert flat.size == self.count
        native.check(self.L.bla_mnist_nn_set_params(self.h, flat.ctypes.data))

    def get_params(self):
        flat = np.empty(self.count, np.float32)
        native.check(self.L.bla_mnist_nn_get_params(self.h, flat.ctypes.data))
        return split_bucket(flat, self.sizes)

    def grads(self):
        flat = np.empty(self.count, np.float32)
        native.sync()
        native.check(self.L.bla_memcpy_d2h(flat.ctypes.data, self.L.bla_mnist_nn_grads(self.h), flat.nbytes, None))
        native.sync()
        return split_bucket(flat, self.sizes)

    def activation(self, name):
        p, rows = C.c_void_p(), C.c_int()
        native.check(self.L.bla_mnist_nn_activation(self.h, ACTIVATION_NAMES.index(name), C.byref(p), C.byref(rows)))
        out = np.empty((rows.value, self.batch), np.float32)
        native.sync()
        native.check(self.L.bla_memcpy_d2h(out.ctypes.data, p, out.nbytes, None))
        native.sync()
        return out

    def use_buckets(self, params_ptr, grads_ptr):
        native.check(self.L.bla_mnist_nn_use_buckets(self.h, params_ptr, grads_ptr))

    @property
    def params_ptr(self):
        return self.L.bla_mnist_nn_params(self.h)

    @property
    def grads_ptr(self):
        return self.L.bla_mnist_nn_grads(self.h)

    # -- data ----------------------------------------------------------------------------------------------
    def load_batch(self, x_raw, y, stream=None):
        """x_raw [n0][B] raw pixels 0..255, y one-hot [n3][B] -> the trainer's resident buffers."""
        x_raw = np.ascontiguousarray(x_raw, np.float32); y = np.ascontiguousarray(y, np.float32)
        assert x_raw.shape == (self.sizes[0], self.batch) and y.shape == (self.sizes[3], self.batch)
        native.check(self.L.bla_memcpy_h2d(self.L.bla_mnist_nn_input(self.h), x_raw.ctypes.data, x_raw.nbytes, stream))
        native.check(self.L.bla_memcpy_h2d(self.L.bla_mnist_nn_labels(self.h), y.ctypes.data, y.nbytes, stream))
        native.sync(stream)

    # -- the step ------------------------------------------------------------------------------------------
    def forward_backward(self, stream=None):
        native.check(self.L.bla_mnist_nn_forward_backward(self.h, stream, None, None, self.colsum_mode))

    def apply(self, lr=LEARN_RATE, stream=None):
        native.check(self.L.bla_mnist_nn_apply(self.h, stream, lr))

    def train_step(self, lr=LEARN_RATE, stream=None):
        native.check(self.L.bla_mnist_nn_train_step(self.h, stream, None, None, lr, self.colsum_mode))

    def graph_step(self, lr=LEARN_RATE, stream=None, with_update=True):
        native.check(self.L.bla_mnist_nn_graph_step(self.h, stream, lr, self.colsum_mode, int(with_update)))

    def fused_step(self, lr=LEARN_RATE, stream=None):
        """Forward, backward and update in six launches issued directly (update folded into the weight-gradient products; no gradient bucket)."""
        native.check(self.L.bla_mnist_nn_fused_step(self.h, stream, None, None, lr, self.colsum_mode))

    def dp_step(self, exchange, lr=LEARN_RATE, stream=None, graph=True):
        """One data-parallel step (forward, backward, direct-xGMI all-reduce fused with the update): one graph launch, or (graph=False)
        the same seven launches issued directly from one host call."""
        f = self.L.bla_mnist_nn_dp_step if graph else self.L.bla_mnist_nn_dp_step_direct
        native.check(f(self.h, exchange.h, stream, lr, self.colsum_mode))

    def dp_step_rccl(self, comm, lr=LEARN_RATE, stream=None):
        """One data-parallel step over the library collective: forward + backward, ncclAllReduce(SUM) of the gradient bucket, update."""
        native.check(self.L.bla_mnist_nn_dp_step_rccl(self.h, comm.h, stream, lr, self.colsum_mode))


class Context:
    """bla_context_*: one more {device, stream, workspace} beside the default one, for hosts that drive several ranks from one
    process.  make_current() installs it on the calling thread; every bla_* call issued after that belongs to this rank."""

    def __init__(self, device=0):
        self.L = native.lib()
        h = C.c_void_p()
        native.check(self.L.bla_context_create(C.byref(h), device))
        self.h = h

    def make_current(self):
        native.check(self.L.bla_context_set_current(self.h))
        return self

    @staticmethod
    def restore_default():
        native.check(native.lib().bla_context_set_current(None))

    def close(self):
        if self.h:
            self.L.bla_context_destroy(self.h)
            self.h = None


class RcclComm:
    """bla_dp_rccl_*: the library collective (ncclAllReduce SUM over xGMI) behind the C-ABI.  `broadcast_bytes(b) -> bytes` hands rank 0's
    128-byte unique id to every rank over any host channel."""

    def __init__(self, rank, world, broadcast_bytes=None):
        self.L = native.lib()
        ident = C.create_string_buffer(128)
        if rank == 0:
            native.check(self.L.bla_dp_rccl_unique_id(ident))
        raw = ident.raw
        if world > 1:
            raw = broadcast_bytes(raw if rank == 0 else None)
        h = C.c_void_p()
        native.check(self.L.bla_dp_rccl_init(C.byref(h), raw, rank, world))
        self.h, self.rank, self.world = h, rank, world

    def allreduce(self, ptr, count, stream=None):
        native.check(self.L.bla_dp_rccl_allreduce_f32(self.h, stream, ptr, count))

    def close(self):
        if self.h:
            self.L.bla_dp_rccl_destroy(self.h)
            self.h = None


class Exchange:
    """bla_dp_*: the gradient exchange of the data-parallel step.  `all_gather_bytes(b) -> [bytes per rank]` is any
    host-side channel (torch.distributed.all_gather_object, MPI, ...): it only carries the 256-byte handle blobs once."""

    def __init__(self, rank, world, count, all_gather_bytes=None):
        """With `all_gather_bytes` the peers are connected here; without it (and world > 1) call export() / connect() yourself,
        e.g. to let every rank agree on a fallback between the local steps."""
        self.L = native.lib()
        h = C.c_void_p()
        native.check(self.L.bla_dp_create(C.byref(h), rank, world, count))
        self.h, self.rank, self.world, self.count = h, rank, world, count
        if world > 1 and all_gather_bytes is not None:
            self.connect(all_gather_bytes(self.export()))

    def export(self):
        buf = C.create_string_buffer(HANDLE_BYTES)
        native.check(self.L.bla_dp_export(self.h, buf))
        return buf.raw

    def connect(self, handles):
        assert len(handles) == self.world and all(len(x) == HANDLE_BYTES for x in handles)
        native.check(self.L.bla_dp_connect(self.h, b"".join(handles)))

    def bucket(self, parity):
        return self.L.bla_dp_bucket(self.h, parity)

    def allreduce(self, parity, out=None, target=None, alpha=0.0, stream=None):
        native.check(self.L.bla_dp_allreduce_f32(self.h, stream, parity, out, target, alpha))

    def status(self):
        s = C.c_int()
        native.check(self.L.bla_dp_status(self.h, C.byref(s)))
        return s.value

    def close(self):
        if self.h:
            self.L.bla_dp_destroy(self.h)
            self.h = None


def shard_columns(n_cols, world, rank):
    """Columns [lo, hi) of a batch that rank `rank` of `world` owns (model/mnist_nn.c has samples as columns;
    every weight gradient is a plain sum over columns, :260-293, so shards add up exactly)."""
    assert n_cols % world == 0, "global batch must divide over the ranks"
    per = n_cols // world
    return rank * per, (rank + 1) * per


def data_parallel_step(local_forward_backward, grads_tensor, apply_update, dist=None):
    """One data-parallel SGD step: local forward/backward on this rank's column shard, SUM all-reduce of the flat
    gradient bucket (no rescale: the reference's gradient is a sum over the batch, not a mean), identical update on
    every rank.  `dist` is torch.distributed (RCCL on GPUs, gloo in CPU tests) or None for a single rank."""
    local_forward_backward()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(grads_tensor, op=dist.ReduceOp.SUM)
    apply_update()
