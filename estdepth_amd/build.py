"""Build libestd_hip.so (gfx950 only) with hipcc.  In-tree output: estdepth_amd/lib/libestd_hip.so

    python -m estdepth_amd.build [--force] [--verbose]
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
SOURCES = ["conv3d_mfma.hip", "conv3d_split_bf16.hip", "conv2d_mfma.hip", "conv2d_split_bf16.hip", "plane_sweep.hip", "est_fusion.hip"]
HEADERS = [os.path.join(ROOT, "include", "estd_hip.h"), os.path.join(CSRC, "estd_common.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


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
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
