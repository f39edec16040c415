#!/usr/bin/env python
"""Register / scratch / LDS use of every kernel in the built objects (code-object metadata; no GPU needed):
    python tools/kernel_regs.py [--filter SUBSTR] [object ...]      default: every more4d_amd/build/*.hip.o
A non-zero scratch / spill count on a production kernel means hipcc spilled: treat it as a build regression
(tests/test_host_logic.py::test_production_kernels_do_not_spill asserts this)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(obj):
    """[{name, vgpr, agpr, sgpr, scratch, lds, spill}] of the gfx950 code object embedded in `obj`."""
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(obj, os.path.join(tmp, "x.o"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "x.o"], cwd=tmp, capture_output=True)
        cos = [f for f in os.listdir(tmp) if f.endswith("gfx950")]
        if not cos:
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, cos[0])], capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        out.append(dict(name=name, vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"), spill=g("vgpr_spill_count")))
    return out


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    objs = args or sorted(glob.glob(os.path.join(ROOT, "more4d_amd", "build", "*.hip.o")))
    for o in objs:
        print("==", os.path.relpath(o, ROOT))
        for k in kernels(o):
            if flt and flt not in k["name"]:
                continue
            nm = k["name"].replace("(anonymous namespace)::", "").replace("void ", "")
            nm = re.sub(r"\((?!.*<).*$", "", nm)        # drop the argument list, keep template arguments
            nm = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", nm)[:64]
            print(f"  {nm:64s} vgpr {k['vgpr']:>4} agpr {k['agpr']:>4} sgpr {k['sgpr']:>4} scratch {k['scratch']:>5} lds {k['lds']:>6} spill {k['spill']:>3}")


if __name__ == "__main__":
    main()
