"""In-tree native build of the package (no pip, no JIT cache):

    csrc/*.hip      --hipcc --offload-arch=gfx950-->  csrc/libbla_hip.so   (HIP kernels + C-ABI, include/bla.h)
    lib/*.c         --gcc-->                          lib/libbla_host.so   (drop-in matrix.h/conv.h/... host API)

The HIP runtime we link against is the one PyTorch-ROCm ships (torch/lib/libamdhip64.so, SONAME
libamdhip64.so) so that a process that also imports torch (bench.py: torch.distributed over RCCL)
holds exactly ONE HIP runtime and torch's stream handles are valid in our launches.
hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container.
"""
import glob
import importlib.util
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "lib")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIP_SO = os.path.join(CSRC, "libbla_hip.so")
HOST_SO = os.path.join(HOST, "libbla_host.so")
HOST_F64_SO = os.path.join(HOST, "libbla_host_f64.so")   # matrix.h in the reference's own element type (-DBLA_FP64)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def torch_lib_dir():
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return None
    d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError(f"native build failed: {cmd[0]} exited {r.returncode}")
    return r.stdout


def build_hip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        only = os.environ.get("BLA_BUILD_ONLY")   # experiments on one translation unit: "bla_gather.hip,bla_conv.hip" rebuilds just those (the others keep their objects)
        if only and os.path.basename(s) not in only.split(",") and os.path.exists(o):
            continue
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                   "-DBLA_BUILDING", "-I", INCLUDE, "-c", s, "-o", o] + os.environ.get("BLA_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DBLA_WSK_DIAG
            jobs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), cmd))
    for s, p, cmd in jobs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + out + "\n")
            raise RuntimeError(f"hipcc failed on {os.path.basename(s)}")
        if verbose and out.strip():
            print(out)
    if force or jobs or _newer(HIP_SO, objs):
        tl = torch_lib_dir()
        link = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", HIP_SO] + objs
        if tl:  # one HIP runtime per process: bind to the runtime torch will load (see module docstring)
            link = ["g++", "-shared", "-fPIC", "-o", HIP_SO] + objs + ["-L", tl, "-l:libamdhip64.so", f"-Wl,-rpath,{tl}",
                    "-Wl,--no-undefined"]
            link += ["-ldl"]   # bla_rccl.hip opens librccl.so.1 on first use (no link dependency: a box without RCCL still loads this library)
        else:
            link += ["-ldl"]
        _run(link)
    return HIP_SO


def build_host(force=False):
    # lib/mnist_csv.c (legacy streaming reader) and lib/mnist_csv2.c define the same names by the reference's design: like the reference's
    # build, a program links one of them; the shared library carries mnist_csv2 (what model/mnist_nn.c uses)
    srcs = sorted(f for f in glob.glob(os.path.join(HOST, "*.c")) if os.path.basename(f) != "mnist_csv.c")
    if not srcs:
        return None
    hdrs = glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if force or _newer(HOST_SO, srcs + hdrs + [HIP_SO]):
        _run(["gcc", "-std=c99", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INCLUDE, "-o", HOST_SO] + srcs +
             ["-L", CSRC, "-l:libbla_hip.so", f"-Wl,-rpath,{CSRC}", "-lm"])
    # -DBLA_FP64: matrix.h, conv.h, norm.h, util.h and the readers in the reference's own element type (layer.h's float* callbacks do not fit it)
    f64_srcs = [os.path.join(HOST, f) for f in ("matrix.c", "conv.c", "norm.c", "util.c", "bla_host.c", "csv.c", "mnist_csv2.c", "cifar10.c", "bmp.c")]
    if force or _newer(HOST_F64_SO, f64_srcs + hdrs + [HIP_SO]):
        _run(["gcc", "-std=c99", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-DBLA_FP64", "-I", INCLUDE, "-o", HOST_F64_SO] + f64_srcs +
             ["-L", CSRC, "-l:libbla_hip.so", f"-Wl,-rpath,{CSRC}", "-lm"])
    return HOST_SO


def build_native(force=False, verbose=False):
    hip = build_hip(force, verbose)
    host = build_host(force)
    return hip, host


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
