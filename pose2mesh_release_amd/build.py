"""Builds the two native libraries of the package in-tree (pose2mesh_release_amd/lib/):

  libp2m_hip.so   HIP kernels + C ABI (include/p2m.h), cross-compiled for gfx950 with hipcc
  libp2m_host.so  CPU-only helpers for graph preparation (g++)

`python -m pose2mesh_release_amd.build` or __graft_entry__.build() runs this; a rebuild happens
only when a source is newer than the library.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
HIP_SOURCES = ["capi.hip", "basis.hip", "gemm.hip", "bn.hip", "optim.hip", "loss.hip", "chebtile.hip", "posenet.hip"]
# per-source flags.  chebtile.hip: the gather's fmaf chains must stay scalar v_fma_f32 - SLP-packed v_pk_fma_f32 next to the
# MFMA waves measured 9 % slower over the real-row shapes of a train step (17.2 vs 19.0 ms)
HIP_SOURCE_FLAGS = {"chebtile.hip": ["-fno-slp-vectorize"]}
HIP_HEADERS = ["p2m_common.h", "p2m_split.h", os.path.join("..", "..", "include", "p2m.h")]
HIP_LIB = os.path.join(LIBDIR, "libp2m_hip.so")
HOST_LIB = os.path.join(LIBDIR, "libp2m_host.so")


def source_id():
    """Content hash of what the HIP library is built from plus the host-side launch layer (12 hex digits): the identity of
    `the running code` where there is no git (the GPU boxes get a snapshot without .git).  bench.py compares it with the id
    stored in profiles/traffic_latest.json by tools/rocprof_traffic.sh and marks the PMC traffic figures stale when they
    were measured on other code."""
    import hashlib
    h = hashlib.sha1()
    names = [os.path.join(CSRC, n) for n in sorted(os.listdir(CSRC)) if n.endswith((".hip", ".h", ".cpp"))]
    names += [os.path.normpath(os.path.join(CSRC, "..", "..", "include", "p2m.h"))]
    names += [os.path.join(HERE, n) for n in ("ops.py", "meshnet.py", "posenet.py", "cheby_graph_conv.py", "optim.py", "loss.py")]
    for n in names:
        with open(n, "rb") as f:
            h.update(os.path.basename(n).encode() + b"\0" + f.read())
    return h.hexdigest()[:12]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def build_hip(force=False, verbose=False):
    """One object per translation unit (compiled in parallel, only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HIP_HEADERS]
    extra = os.environ.get("P2M_HIPCC_FLAGS", "").split()
    flagfile = os.path.join(objdir, "flags.txt")
    flags_changed = (open(flagfile).read() if os.path.exists(flagfile) else "") != " ".join(extra)
    jobs, objs = [], []
    for src_name in HIP_SOURCES:
        src = os.path.join(CSRC, src_name)
        obj = os.path.join(objdir, src_name.replace(".hip", ".o"))
        objs.append(obj)
        if force or flags_changed or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"]
                        + HIP_SOURCE_FLAGS.get(src_name, []) + extra + ["-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
        open(flagfile, "w").write(" ".join(extra))
    if jobs or not os.path.exists(HIP_LIB):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs)
    return HIP_LIB


def build_host(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "p2m_host.cpp")
    if not force and not _stale(HOST_LIB, [src]):
        return HOST_LIB
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", HOST_LIB, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return HOST_LIB


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    if "--source-id" in sys.argv:
        print(source_id())
    else:
        print(build_all(force="--force" in sys.argv, verbose=True))
