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


def source_hash():
    """sha256 over every file of csrc/ and include/ (names + bytes, sorted).  It is compiled into the library
    (m4d_source_hash, api.cpp) and compared by _lib.load(): a shipped binary that does not come from the tracked
    sources cannot be loaded, and build() rebuilds whenever the two differ, whatever the time stamps say."""
    import hashlib
    h = hashlib.sha256()
    for d in (CSRC, INC):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".cpp", ".h", ".inc")):
                h.update(f.encode() + b"\0")
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(fh.read())
                h.update(b"\0")
    return h.hexdigest()


def built_hash(lib):
    """The hash a built library carries (None if it predates m4d_source_hash).  Read from the file's bytes: a dlopen of a
    path this process already has open would return the OLD mapping after a rebuild."""
    import re
    try:
        with open(lib, "rb") as fh:
            m = re.search(rb"M4D_SRC_HASH=([0-9a-f]{64})", fh.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _file_hash(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for q in sorted(paths):
        h.update(os.path.basename(q).encode() + b"\0")
        with open(q, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(src, abl=False):
    """One object per source.  Staleness is decided by CONTENT (source + every header + flags, kept beside the object as
    <obj>.hash), not by time stamps: a snapshot or a checkout does not preserve them."""
    obj = os.path.join(OBJDIR, src + (".abl.o" if abl else ".o"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
           [os.path.join(INC, f) for f in os.listdir(INC)]
    path = os.path.join(CSRC, src)
    stamp = ['-DM4D_SRC_HASH="%s"' % source_hash()] if src == "api.cpp" else []      # api.cpp carries the hash of the whole tree
    cmd = [_hipcc()] + FLAGS + stamp + (["-DM4D_ABLATIONS"] if abl else []) + (["-x", "hip"] if src.endswith(".cpp") else []) + \
        ["-c", path, "-o", obj]
    want = _file_hash([path] + hdrs, " ".join(cmd[1:]))
    tag = obj + ".hash"
    if not os.path.exists(obj) or not os.path.exists(tag) or open(tag).read() != want:
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        with open(tag, "w") as fh:
            fh.write(want)
    return obj


def build(force=False, verbose=False, ablations=False):
    if ablations:
        build(force=force, verbose=verbose)
    lib = LIB.replace(".so", "_abl.so") if ablations else LIB
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            if (f.endswith(".abl.o") or f.endswith(".abl.o.hash")) == ablations:
                os.remove(os.path.join(OBJDIR, f))
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s_: _compile(s_, ablations), srcs))
    if _stale(lib, objs) or (not ablations and built_hash(lib) != source_hash()):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if not ablations:
        got = built_hash(lib)
        if got != source_hash():
            raise RuntimeError(f"{lib} reports source hash {got}, the tree hashes to {source_hash()}")
    if verbose:
        print("built", lib, "source hash", source_hash()[:16])
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, ablations="--ablations" in sys.argv)
