"""Build libmore4d_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo
snapshot to the GPU box.  Usage: python -m more4d_amd.build [--force] [--ablations]

--ablations additionally builds lib/libmore4d_hip_abl.so with -DM4D_ABLATIONS: the timing ablations of tools/abl*.sh
(kernels that skip work and return wrong results).  The shipping library never contains them; tools select the ablation
build with M4D_LIB=abl (more4d_amd/_lib.py), which bench.py refuses."""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmore4d_hip.so")
OBJDIR = os.path.join(HERE, "build")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", INC,
         "-Wno-unused-result"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, abl=False):
    obj = os.path.join(OBJDIR, src + (".abl.o" if abl else ".o"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(INC, f) for f in os.listdir(INC)]
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + hdrs):
        cmd = [_hipcc()] + FLAGS + (["-DM4D_ABLATIONS"] if abl else []) + (["-x", "hip"] if src.endswith(".cpp") else []) + \
            ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force=False, verbose=False, ablations=False):
    if ablations:
        build(force=force, verbose=verbose)
    lib = LIB.replace(".so", "_abl.so") if ablations else LIB
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            if f.endswith(".abl.o") == ablations:
                os.remove(os.path.join(OBJDIR, f))
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s_: _compile(s_, ablations), srcs))
    if _stale(lib, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", lib)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, ablations="--ablations" in sys.argv)
