"""Build recipe for the C oracle (gcc, OpenMP).  Output: oracle/_build/libestd_oracle.so"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libestd_oracle.so")
SRC = os.path.join(HERE, "estd_oracle.c")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force) and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    # -ffp-contract=off: every C expression rounds like the separate ATen ops it restates; fused multiply-adds appear only
    # where the reference has them (the GEMM kernels behind matmul/bmm), spelled fmaf() in the source
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fno-fast-math", "-ffp-contract=off",
           "-shared", "-fPIC", SRC, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
