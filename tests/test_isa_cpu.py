"""Pins the hand-built software pipelines against the toolchain (round 6; no GPU needed - hipcc cross-compiles gfx950 here).

The LDS-DMA kernels (unit_pipe_kernel, pw_pipe_kernel, stage_pipe_kernel, head_tail_group_kernel, down2_kernel ...) rely on properties hipcc neither
knows about nor promises: hand-counted s_waitcnt around DMA pieces it cannot see, register budgets that leave room for two or three workgroups
per CU, no scratch (a spill's reload is a vector-memory load: its s_waitcnt vmcnt(0) retires every DMA piece in flight).  Five such traps were
found by reading ISA by hand in round 5; these tests read it instead:

  1. from the BUILT library's gfx950 code objects (the .note metadata the loader uses): zero scratch and the VGPR class each kernel's launch
     geometry assumes;
  2. from the assembly of kernels_pipe.hip / kernels_stage.hip: the number of fragment loads between the YN_*_FRAG markers equals the count of
     the hand-placed wait behind them (the first tile's DMA pieces are retired by exactly that wait - ADVICE r5), and inside the tile loops
     no compiler-placed s_waitcnt vmcnt(<= 1) sits between a DMA issue and the hand-placed wait with more than a GEMM tail of MFMAs still to
     come (such a wait ends the overlap the pipeline exists for)."""
import os
import re
import struct
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolo-nano_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


# ---- 1. metadata of the built code objects ---------------------------------------------------------------------------------------
def _code_objects(path):
    """The gfx950 code objects of every translation unit: clang offload bundles inside the shared library's .hip_fatbin section."""
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        o = m.start()
        n = struct.unpack_from("<Q", blob, o + 24)[0]
        p = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl]
            p += tl
            if b"gfx950" in triple and size:
                out.append(blob[o + off:o + off + size])
    return out


@pytest.fixture(scope="module")
def kernel_meta():
    from yolo_nano_amd import build, capi
    build.build()
    meta = {}
    for co in _code_objects(capi.LIB_PATH):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
            fn = f.name
        try:
            txt = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", fn]).decode()
        finally:
            os.unlink(fn)
        cur = {}
        for ln in txt.splitlines():
            m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", ln)
            if not m:
                continue
            cur[m.group(1)] = m.group(2).strip().strip("'")
            if m.group(1) == "wavefront_size":              # the last key of a kernel's entry
                if "name" in cur:
                    meta[cur["name"]] = cur
                cur = {}
    dem = subprocess.run(["c++filt"], input="\n".join(meta), capture_output=True, text=True).stdout.split("\n")
    return {d.replace("ynk::", "").replace("void ", "").split("(")[0]: v for d, v in zip(dem, meta.values())}


def _vgprs(v):
    return int(v["vgpr_count"]) + int(v.get("agpr_count", 0) or 0)


def test_pipeline_kernels_no_scratch_and_their_register_class(kernel_meta):
    assert len(kernel_meta) > 250, "metadata of the library's kernels not found"
    seen = set()
    for name, v in kernel_meta.items():
        fam = name.split("<")[0]
        if fam not in ("unit_pipe_kernel", "pw_pipe_kernel", "stage_pipe_kernel", "down2_kernel", "head_tail_group_kernel", "dwpw_pipe_group_kernel",
                       "down_unit_pipe_kernel", "conv3x3_split_kernel", "stem_pool_kernel"):
            continue
        seen.add(fam)
        assert int(v["private_segment_fixed_size"]) == 0, "%s uses %s bytes of scratch: a reload's vmcnt(0) retires the DMA pieces in flight" % (name, v["private_segment_fixed_size"])
        args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")] if "<" in name else []
        limit = 512                                          # one wavefront per SIMD: the unified file
        if fam in ("unit_pipe_kernel", "down_unit_pipe_kernel", "dwpw_pipe_group_kernel"):
            limit = 256                                      # two wavefronts per SIMD (two four-wavefront workgroups, or one of eight, per CU)
        elif fam == "stage_pipe_kernel":
            limit = 168 if int(args[0]) <= 48 else 256       # narrow branches: three workgroups per CU (launch_stage_pipe's 768 walkers)
        elif fam == "pw_pipe_kernel":
            limit = {"4": 128, "2": 256, "1": 256}[args[3]]  # OCC workgroups of NW wavefronts per CU
        elif fam == "head_tail_group_kernel":
            limit = 168                                      # amdgpu_waves_per_eu(3, 3): three workgroups per CU
        assert _vgprs(v) <= limit, "%s: %d registers, its launch geometry assumes <= %d" % (name, _vgprs(v), limit)
    assert {"unit_pipe_kernel", "pw_pipe_kernel", "stage_pipe_kernel", "down2_kernel", "head_tail_group_kernel"} <= seen, seen


# ---- 2. assembly of the persistent kernels ------------------------------------------------------------------------------------------
def _assemble(src):
    out = os.path.join(tempfile.mkdtemp(prefix="yn_isa_"), src.replace(".hip", ".s"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result",
                           "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                          stderr=subprocess.DEVNULL)
    return out


@pytest.fixture(scope="module")
def kernels_asm():
    """{mangled kernel name: its lines} of kernels_pipe.hip and kernels_stage.hip, with the flags of yolo_nano_amd/build.py"""
    with ThreadPoolExecutor(max_workers=2) as ex:
        paths = list(ex.map(_assemble, ["kernels_pipe.hip", "kernels_stage.hip"]))
    out = {}
    for path in paths:
        name, body = None, []
        for ln in open(path):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                name, body = m.group(1), []
            if name is not None:
                body.append(ln)
                if "s_endpgm" in ln:
                    out[name] = body
                    name = None
    return out


def test_first_tile_wait_counts_the_fragment_loads(kernels_asm):
    """`; YN_*_FRAG_BEGIN` ... N global_load_dwordx4 ... `; YN_*_FRAG_END` / `s_waitcnt vmcnt(N)`: the wait retires exactly what is older than the
    fragment loads - the first tile's DMA pieces (and taps) - in every wavefront.  A toolchain that splits, merges or moves a load changes N."""
    checked = 0
    for name, body in kernels_asm.items():
        cnt, expect = None, None
        for ln in body:
            if "FRAG_BEGIN" in ln:
                cnt = 0
            elif "FRAG_END" in ln:
                expect, cnt = cnt, None
            elif cnt is not None and re.search(r"\b(global|buffer|flat|scratch)_(load|store|atomic)", ln):
                assert "global_load_dwordx4" in ln, "%s: %s between the fragment markers" % (name, ln.strip())
                cnt += 1
            elif expect is not None and "s_waitcnt" in ln:
                m = re.search(r"vmcnt\((\d+)\)", ln)
                assert m and int(m.group(1)) == expect, "%s: %d fragment loads between the markers, the wait behind them says %s" % (name, expect, ln.strip())
                expect = None
                checked += 1
    assert checked >= 22 + 16, "unit_pipe_kernel has 22 instantiations, stage_pipe_kernel 16: only %d marker pairs seen" % checked


def test_no_compiler_drain_under_the_dma_in_the_tile_loops(kernels_asm):
    """Inside a tile loop, between an LDS-DMA issue and the hand-placed wait that retires it, hipcc must not place a s_waitcnt vmcnt(0 / 1) of its own
    while more than a GEMM's tail of MFMAs is still to come (a wait right in front of the hand-placed drain costs nothing; one in front of a
    GEMM serialises the transfer the pipeline hides).  Found this way in round 6: the scratch reloads of unit_pipe_kernel<116,false,8> and the
    panel waits of stage_pipe_kernel's first form.  The control wavefront's ticket / flag code of stage_pipe_kernel waits on purpose (one
    wavefront, `wave == CW` branches): those blocks hold no MFMA and are followed by a barrier, the rule below does not see them as GEMM waits."""
    for name, body in kernels_asm.items():
        in_asm = in_loop = False
        pending = None
        for i, ln in enumerate(body):
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", ln):
                in_loop = "in Loop" in ln
            if ";;#ASMSTART" in ln:
                in_asm = True
            elif ";;#ASMEND" in ln:
                in_asm = False
            if in_asm and "global_load_lds" in ln:
                pending = i if in_loop else None
            if "s_waitcnt" in ln and "vmcnt" in ln:
                if in_asm:
                    pending = None
                elif pending is not None and in_loop and re.search(r"vmcnt\([01]\)", ln):
                    mfma, asm2 = 0, False                  # MFMAs up to the next barrier or hand-placed (inside an asm block) vector-memory wait
                    for nx in body[i + 1:]:
                        if ";;#ASMSTART" in nx:
                            asm2 = True
                        elif ";;#ASMEND" in nx:
                            asm2 = False
                        if "v_mfma" in nx:
                            mfma += 1
                        if "s_barrier" in nx or (asm2 and "s_waitcnt" in nx and "vmcnt" in nx):
                            break
                    assert mfma <= 8, "%s line %d: `%s` with LDS-DMA pieces in flight and %d MFMAs before the next barrier / hand-placed wait" % (
                        name, i, ln.strip(), mfma)
