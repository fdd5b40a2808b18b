"""Build the native libraries (gfx950 only) with hipcc, in-tree:

    estdepth_amd/lib/libestd_hip.so         the HIP kernels behind the torch-free C ABI (include/estd_hip.h)
    estdepth_amd/lib/libestd_torch_ops.so   TORCH_LIBRARY(estdepth_hip) operator registration over that C ABI (csrc/torch_ops.cpp)

    python -m estdepth_amd.build [--force] [--verbose]

The libraries are git-ignored build products: a fresh clone has to run this once (hipcc cross-compiles without a GPU;
``__graft_entry__.build()`` does it), and they travel to the GPU box with the working tree.  Staleness is decided by
mtime of sources and headers.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj" + os.environ.get("ESTD_LIB_SUFFIX", ""))
LIB = os.path.join(OUT_DIR, "libestd_hip%s.so" % os.environ.get("ESTD_LIB_SUFFIX", ""))
# Kernels with a default caller.  The superseded A/B kernels -- depth-only Winograd conv3d, row-only Winograd conv2d, the two 3 x bf16
# operand-split kernels (none has a default caller since round 4; the two-axis Winograd kernels cover every instance and are faster),
# the operand-reuse rebuild of the two-axis 32 -> 32 instance (conv3d_wino2x.hip: at parity with the 8-wave kernel, superseded by conv3d_wino3.hip) --
# are built, exported (include/estd_hip.h: #ifdef ESTD_BUILD_AB), bound and tested only with ESTD_BUILD_AB=1 in the environment.
SOURCES = ["conv3d_mfma.hip", "conv3d_wino2.hip", "conv3d_wino3.hip", "conv3d_xout.hip", "conv3d_wino2_c16.hip", "conv2d_mfma.hip", "conv2d_wino2.hip",
           "plane_sweep.hip", "est_fusion.hip", "refine2d.hip", "conv1x1.hip", "conv2d_taps.hip"]
AB_SOURCES = ["conv3d_wino.hip", "conv3d_split_bf16.hip", "conv2d_wino.hip", "conv2d_split_bf16.hip", "conv3d_wino2x.hip"]
BUILD_AB = os.environ.get("ESTD_BUILD_AB", "0") == "1"
if BUILD_AB:
    SOURCES = SOURCES + AB_SOURCES
HEADERS = [os.path.join(ROOT, "include", "estd_hip.h"), os.path.join(CSRC, "estd_common.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + (["-DESTD_BUILD_AB=1"] if BUILD_AB else [])
MODE_STAMP = os.path.join(OUT_DIR, ".build_mode" + os.environ.get("ESTD_LIB_SUFFIX", ""))


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    mode = "ab" if BUILD_AB else "default"
    relink = (open(MODE_STAMP).read().strip() if os.path.exists(MODE_STAMP) else "default" if os.path.exists(LIB) else "") != mode
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + os.environ.get("ESTD_BUILD_DEFS", "").split() + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for log in ex.map(run, jobs):
                if verbose and log:
                    sys.stderr.write(log)
    if force or jobs or relink or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    build_torch_ops(force=force or relink)
    open(MODE_STAMP, "w").write(mode)       # only now: a failed operator-library rebuild after a mode switch is retried by the next build()
    return LIB


TORCH_OPS_SRC = os.path.join(CSRC, "torch_ops.cpp")
TORCH_OPS_LIB = os.path.join(OUT_DIR, "libestd_torch_ops%s.so" % os.environ.get("ESTD_LIB_SUFFIX", ""))


def build_torch_ops(force=False):
    """TORCH_LIBRARY wrappers: host-only C++ against libtorch (no device code), linked to libestd_hip.so via $ORIGIN."""
    if not (force or _stale(TORCH_OPS_LIB, [TORCH_OPS_SRC, HEADERS[0], LIB])):
        return TORCH_OPS_LIB
    import torch
    from torch.utils import cpp_extension
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-I" + os.path.join(ROOT, "include")] \
        + (["-DESTD_BUILD_AB=1"] if BUILD_AB else [])
    cmd += ["-I" + p for p in cpp_extension.include_paths("cuda")]
    cmd += [TORCH_OPS_SRC, "-o", TORCH_OPS_LIB, "-L" + OUT_DIR, "-l:" + os.path.basename(LIB), "-L" + tlib,
            "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
    return TORCH_OPS_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
