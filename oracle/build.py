"""Build recipe for the C oracle (test infrastructure; see yn_oracle.c header).

    python oracle/build.py        ->  oracle/libyn_oracle.so

There is no `oracle/_ref`: the reference is pure Python (zero native
files), so there is nothing to compile from /root/reference; the oracle is
pinned by fixtures generated from the imported reference instead.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "yn_oracle.c")
OUT = os.path.join(HERE, "libyn_oracle.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-ftree-vectorize", "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
           "-fvisibility=hidden", "-shared", "-fPIC", "-std=c11", SRC, "-o", OUT, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
