"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/yolonano_hip.h declares, the ctypes table covers them, and the host shim mirrors the
reference's module tree.  No compute is launched (there is no GPU in the dev container)."""
import os
import re
import subprocess

import pytest
import torch

from yolo_nano_amd import arch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "yolonano_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yn_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from yolo_nano_amd import build, capi
    build.build()
    return capi.load_library()


def test_library_exports_every_declared_symbol(lib):
    from yolo_nano_amd import capi
    declared = _declared()
    assert len(declared) >= 30
    assert sorted(capi.SIGNATURES) == declared, "ctypes table and header disagree"
    exported = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (yn_[a-z0-9_]+)", exported))
    assert exported == set(declared)
    assert lib.yn_abi_version() == 2


def test_library_is_gfx950_only():
    from yolo_nano_amd import capi
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + capi.LIB_PATH],
                         capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU instead of computing on the CPU."""
    from yolo_nano_amd import capi, YOLONano
    with pytest.raises(capi.YnError):
        capi.Handle(320, 20, arch.MULTI_ANCHOR_SIZE)
    m = YOLONano(torch.device("cuda"), input_size=320, num_classes=20, anchor_size=arch.MULTI_ANCHOR_SIZE)
    with pytest.raises(capi.YnError):
        m(torch.zeros(1, 3, 320, 320))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "yolo-nano_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"import\s+oracle|from\s+oracle|libyn_oracle|yn_oracle|yo_[a-z]+\(", text), \
                    "%s reaches into oracle/" % f


def test_shim_state_dict_matches_reference_keys():
    import json
    from yolo_nano_amd import YOLONano, fuse_conv_bn
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    for tag, C, anchors in (("voc", 20, arch.MULTI_ANCHOR_SIZE), ("coco", 80, arch.MULTI_ANCHOR_SIZE_COCO)):
        m = YOLONano(torch.device("cuda"), input_size=320, num_classes=C, anchor_size=anchors)
        mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        assert mine == ref[tag]
        assert m.stride == [8, 16, 32] and m.num_anchors == 3 and m.input_size == 320
    import copy
    assert len(fuse_conv_bn(copy.deepcopy(m)).state_dict()) == 154
    g, s, a = m.create_grid(320)
    assert tuple(g.shape) == (1, 2100, 1, 2) and tuple(s.shape) == (1, 2100, 3, 2) and tuple(a.shape) == (1, 2100, 3, 2)
    with pytest.raises(Exception):
        YOLONano(torch.device("cuda"), input_size=320, num_classes=20, anchor_size=arch.MULTI_ANCHOR_SIZE, backbone="3.0x")
