"""Builds libexpv_mi_<tag>.so with extra -D flags on pipe.hip (A/B builds of the single-pass step; loaded through EXPV_MI_LIB).
usage: python tools/build_variant.py TAG -DPIPE_NT=0 ...        (EXPV_MI_VARIANT_SRC=lanczos_pl.hip: the flags go to that source instead)"""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "exponentialutilities.jl_amd"))
import build as B
tag, defs = sys.argv[1], sys.argv[2:]
B.build(verbose=False)
hipcc = B._hipcc()
objdir = os.path.join(B.HERE, "build")
src = os.environ.get("EXPV_MI_VARIANT_SRC", "pipe.hip")
obj = os.path.join(objdir, "%s_%s.o" % (src.replace(".hip", ""), tag))
subprocess.run([hipcc] + B.FLAGS + defs + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in B.SOURCES if s != src] + [obj]
out = os.path.join(B.HERE, "libexpv_mi_%s.so" % tag)
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
