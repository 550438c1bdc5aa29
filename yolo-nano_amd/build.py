"""Build recipe for libyolonano_hip.so (hipcc, gfx950 only, in-tree).

    python -m yolo_nano_amd.build      ->  yolo-nano_amd/libyolonano_hip.so

hipcc cross-compiles for gfx950 without a GPU.  kernels_post.hip is built with
-ffp-contract=off because its NMS arithmetic must match numpy's float32 operation
sequence bit for bit (models/yolo_nano.py:159-188).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libyolonano_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = [("kernels_conv.hip", []), ("kernels_post.hip", ["-ffp-contract=off"]), ("kernels_train.hip", []), ("kernels_bwd.hip", []), ("kernels_h16.hip", []), ("kernels_chain.hip", []), ("kernels_pipe.hip", []), ("kernels_stage.hip", []), ("yn_api.hip", [])]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result",
          "-Wno-pass-failed"]


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    d.append(os.path.join(os.path.dirname(HERE), "include", "yolonano_hip.h"))
    d.append(os.path.abspath(__file__))
    return d


def _compile(item):
    src, extra = item
    extra = extra + os.environ.get("YN_EXTRA_FLAGS", "").split()      # experiment switches (-DYN_EXP_*)
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    cmd = [HIPCC] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    return obj


def _clean_stale():
    """-save-temps leftovers of THIS recipe (kernels_*.o.0.hipv4-..., *.host-x86_64-...): not build products, they would only ride along to the
    GPU box.  Called on the rebuild path only (never on a plain import: another process's compile may be using its own), exact patterns."""
    for f in os.listdir(CSRC):
        if f.startswith(("kernels_", "yn_api")) and (".o." in f or "hipv4-amdgcn" in f or "host-x86_64" in f):
            try:
                os.remove(os.path.join(CSRC, f))
            except OSError:
                pass


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(p) for p in _deps()):
        return OUT
    _clean_stale()
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    if verbose:
        print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
