"""Register / LDS / scratch use of the built library's kernels whose name contains one of the given substrings:  python tools/kernel_meta.py prefilter nms_sweep"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_isa_cpu as t
from yolo_nano_amd import capi
for co in t._code_objects(capi.LIB_PATH):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(co)
        fn = f.name
    txt = subprocess.check_output([t.LLVM + "/llvm-readelf", "--notes", fn]).decode()
    os.unlink(fn)
    cur = {}
    for ln in txt.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", ln)
        if not m:
            continue
        cur[m.group(1)] = m.group(2).strip().strip("'")
        if m.group(1) == "wavefront_size":
            if any(k in cur.get("name", "") for k in sys.argv[1:]):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.split("(")[0]
                print("%-60s vgpr %s agpr %s sgpr %s lds %s scratch %s" % (name[:60], cur["vgpr_count"], cur.get("agpr_count"), cur["sgpr_count"], cur["group_segment_fixed_size"], cur["private_segment_fixed_size"]))
            cur = {}
