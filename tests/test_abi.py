"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/bla.h declares (and nothing undeclared), every symbol is bound by the Python layer, and
-- with no GPU in this container -- the product refuses to compute instead of falling back."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "bla.h")).read()
    return sorted(set(re.findall(r"BLA_API[^;(]*?\b(bla_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound(pkg):
    decl = declared_symbols()
    assert len(decl) > 20
    so = os.path.join(ROOT, "big-linear-algebra_amd", "csrc", "libbla_hip.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = sorted(set(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("bla_")))
    assert exported == decl, (set(decl) ^ set(exported))
    assert sorted(pkg.native.SIGNATURES) == decl, set(decl) ^ set(pkg.native.SIGNATURES)
    L = pkg.lib()
    for name in decl:
        assert hasattr(L, name)


def test_no_torch_types_in_abi():
    txt = open(os.path.join(ROOT, "include", "bla.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)          # declarations only, comments may name them
    for banned in ("torch", "at::", "hipStream_t", "#include <hip"):
        assert banned not in txt


def test_fails_loudly_without_a_device(pkg):
    import torch
    if torch.cuda.is_available() or pkg.lib().bla_device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(pkg.BlaError) as e:
        pkg.init(0)
    assert e.value.status == 3 and "no CPU path" in str(e.value)
    # compute entry points refuse too (no silent CPU fallback)
    assert pkg.lib().bla_gemm_f32(None, 0, 0, 2, 2, 2, None, 2, None, 2, None, 2, None) == 3
    assert pkg.lib().bla_scale_f32(None, None, 4, 1.0) == 3
    assert b"no CPU fallback" in pkg.lib().bla_last_error() or b"not initialised" in pkg.lib().bla_last_error()


def test_product_never_imports_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/."""
    pk = os.path.join(ROOT, "big-linear-algebra_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                for pat in (r"import\s+oracle", r"from\s+oracle", r"liboracle", r"oracle/", r"ora(32|64)_", r"_ref/libref", r"import\s+ref\b"):
                    assert not re.search(pat, txt), (os.path.join(dp, f), pat)


def test_rand_guard_overlapping_threads_keep_the_callers_stream(pkg):
    """the guard that parks the calling program's libc rand() state around HIP / RCCL calls is process-wide and nestable -- two
    threads inside the library at once (one host thread per rank) must leave the program's srand(42) stream exactly where it was, whatever
    the order in which they enter and leave, and whatever is drawn from rand() while a guard is open."""
    import ctypes
    import threading
    libc = ctypes.CDLL("libc.so.6")
    L = pkg.lib()
    libc.srand(42); want = [libc.rand() for _ in range(12)]
    libc.srand(42); got = [libc.rand() for _ in range(4)]
    a_in, b_in, a_out = threading.Event(), threading.Event(), threading.Event()

    def rank_a():
        L.bla_rand_guard_enter(); a_in.set()
        b_in.wait(5)
        for _ in range(3):
            libc.rand()          # what the runtime / RCCL would draw while the stream is parked
        L.bla_rand_guard_leave(); a_out.set()      # A leaves first: B's guard is still open, the stream must stay parked

    def rank_b():
        a_in.wait(5)
        L.bla_rand_guard_enter(); b_in.set()
        a_out.wait(5)
        libc.rand()
        L.bla_rand_guard_leave()
    ta, tb = threading.Thread(target=rank_a), threading.Thread(target=rank_b)
    ta.start(); tb.start(); ta.join(10); tb.join(10)
    assert not ta.is_alive() and not tb.is_alive()
    got += [libc.rand() for _ in range(4)]
    L.bla_rand_guard_enter(); L.bla_rand_guard_enter(); libc.rand(); L.bla_rand_guard_leave(); libc.rand(); L.bla_rand_guard_leave()   # nested on one thread
    got += [libc.rand() for _ in range(4)]
    assert got == want
