"""Deterministic, numpy-version-independent input generators shared by
oracle/gen_golden.py (which ran the reference on these inputs) and the tests
(which regenerate the same inputs instead of storing them)."""
import numpy as np


def splitmix(seed, n):
    """splitmix64 counter hash -> n uint64 (arithmetic wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(seed, shape, lo=-0.5, hi=0.5, dtype=np.float64):
    """uniform [lo,hi) from 24 random bits, rounded through fp32 so that the very
    same value is exactly representable in the fp32 build and the fp64 reference."""
    n = int(np.prod(shape))
    u = (splitmix(seed, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).astype(dtype).reshape(shape)


def randint(seed, shape, n):
    return (splitmix(seed, int(np.prod(shape))) % np.uint64(n)).astype(np.int64).reshape(shape)


def sample_idx(size, count=4096):
    """Indices of the strided sample a digest keeps of a big output."""
    if size <= count:
        return np.arange(size)
    return (np.arange(count, dtype=np.int64) * 2654435761 + 12345) % size


class Golden:
    """Accessor for one tests/golden/*.npz: full arrays or digests (sum, abs-sum, sample)."""

    def __init__(self, path):
        self.z = np.load(path)

    def __contains__(self, name):
        return name in self.z.files or (name + "__sum") in self.z.files

    def __getitem__(self, name):
        return self.z[name]

    def mean_abs(self, name):
        """Mean |value| of a recorded output (full or digest): the natural absolute scale for a tolerance."""
        if self.is_digest(name):
            return float(self.z[name + "__asum"]) / float(np.prod(self.z[name + "__shape"]))
        return float(np.abs(self.z[name]).mean())

    def is_digest(self, name):
        return (name + "__sum") in self.z.files

    def check(self, name, got, rtol=0.0, atol=0.0, exact=False):
        """Compare `got` with the recorded value.  exact=True demands bit equality
        (oracle vs reference); otherwise |got-ref| <= atol + rtol*|ref| elementwise on the
        stored part and a matching relative error on the digest sums."""
        got = np.asarray(got, np.float64)
        if not self.is_digest(name):
            ref = self.z[name]
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            if exact:
                assert np.array_equal(got, ref), f"{name}: not bit-identical, max|d|={np.abs(got-ref).max()}"
            else:
                err = np.abs(got - ref); lim = atol + rtol * np.abs(ref)
                assert (err <= lim).all(), f"{name}: max excess {(err-lim).max():.3e} (max err {err.max():.3e})"
            return
        shape = tuple(self.z[name + "__shape"])
        assert got.shape == shape, (name, got.shape, shape)
        flat = got.ravel()
        smp = flat[sample_idx(flat.size)]
        ref = self.z[name + "__sample"]
        if exact:
            assert np.array_equal(smp, ref), f"{name}: sample not bit-identical"
            assert flat.sum(dtype=np.float64) == float(self.z[name + "__sum"])
            assert np.abs(flat).sum(dtype=np.float64) == float(self.z[name + "__asum"])
        else:
            err = np.abs(smp - ref); lim = atol + rtol * np.abs(ref)
            assert (err <= lim).all(), f"{name}: sample max excess {(err-lim).max():.3e}"
            asum = float(self.z[name + "__asum"])
            budget = (atol * flat.size + rtol * asum) * 1.0 + 1e-300
            assert abs(flat.sum(dtype=np.float64) - float(self.z[name + "__sum"])) <= budget, name
            assert abs(np.abs(flat).sum(dtype=np.float64) - asum) <= budget, name
