"""Builds libexpv_mi.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and importable as
``python exponentialutilities.jl_amd/build.py``.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libexpv_mi.so")
SOURCES = ["kernels.hip", "fused.hip", "pipe.hip", "lanczos_pl.hip", "engine_core.hip", "engine_drivers.hip", "engine_batch.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Xarch_host", "-mavx2", "-Xarch_host", "-mfma",
         "-Xarch_host", "-fcx-limited-range"]   # complex products of the host small-dense code: plain (ac-bd, ad+bc), no __muldc3 NaN recovery   # host small-dense exp: every MI355X host is x86-64-v3


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X build needs the ROCm toolchain")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "expv_mi.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    _prune(objdir, objs, srcs, headers, verbose)
    return LIB


def _prune(objdir, objs, srcs, headers, verbose):
    """Delete the artefacts of earlier experiments that this build did not produce, so that they cannot travel to the GPU box and be
    loaded by accident: everything compiled in build/ (this package's own object directory) and the A/B libraries tools/build_variant.py
    writes next to the product library (libexpv_mi_<TAG>.so).  Nothing else in the package directory is touched (a
    build must not delete files it does not own by extension match).  EXPV_MI_NO_PRUNE=1 keeps everything.  The trace library (tools
    only) survives while it is newer than every source."""
    if os.environ.get("EXPV_MI_NO_PRUNE"):
        return
    keep = {os.path.abspath(o) for o in objs} | {os.path.abspath(LIB)}
    trace_so, trace_o = os.path.join(HERE, "libexpv_mi_trace.so"), os.path.join(objdir, "pipe_trace.o")
    deps = [os.path.join(CSRC, s) for s in srcs] + headers
    if os.path.exists(trace_so) and not _stale(trace_so, deps):
        keep |= {os.path.abspath(trace_so), os.path.abspath(trace_o)}
    doomed = []
    for f in os.listdir(objdir):      # our own object directory: compiler outputs only
        if f.endswith((".o", ".so", ".a", ".hsaco", ".co", ".s", ".bc", ".hipi", ".out", ".hipfb")) or ".o." in f:
            doomed.append(os.path.join(objdir, f))
    for f in os.listdir(HERE):        # the package directory: variant libraries of the product only
        if f.startswith("libexpv_mi_") and f.endswith(".so"):
            doomed.append(os.path.join(HERE, f))
    for path in map(os.path.abspath, doomed):
        if os.path.isfile(path) and path not in keep:
            if verbose:
                print("[build] removing stale artefact", os.path.relpath(path, HERE), flush=True)
            os.remove(path)


def build_callback_example(verbose=True):
    """tests/c_harness/libstencil_cb.so: a compiled matrix-free operator (one HIP kernel behind the expv_mi_matvec_fn contract) for the
    GPU tests and bench.py's `matrix_free_compiled` entry.  Test infrastructure -- the product never loads it."""
    src = os.path.join(HERE, "..", "tests", "c_harness", "stencil_callback.hip")
    out = os.path.join(HERE, "..", "tests", "c_harness", "libstencil_cb.so")
    if os.path.exists(src) and _stale(out, [src]):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", out]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return os.path.abspath(out)


def build_trace(verbose=True):
    """libexpv_mi_trace.so: the same library with pipe.hip compiled under -DPIPE_TRACE (per-workgroup wall_clock64 stamps of the
    single-pass step; tools/pipe_trace.py).  Profiling aid only -- never loaded by the product or the tests."""
    build(verbose=verbose)
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    tobj = os.path.join(objdir, "pipe_trace.o")
    cmd = [hipcc] + FLAGS + ["-DPIPE_TRACE", "-c", os.path.join(CSRC, "pipe.hip"), "-o", tobj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES if s != "pipe.hip"] + [tobj]
    out = os.path.join(HERE, "libexpv_mi_trace.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    return out


if __name__ == "__main__":
    print(build_trace() if "--trace" in sys.argv else build(force="--force" in sys.argv))
