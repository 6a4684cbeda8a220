"""Build A/B variants of libramses_amd.so that differ in the sweep kernel's compile-time
knobs: scripts/build_ab.py TAG=-DFLAG=1,-DOTHER=2 [TAG2=...]
-> ramses_amd/lib/ab/libramses_amd_TAG.so (load with RAMSES_AMD_LIB=...).  The other
objects come from the regular build (ramses_amd/build/)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramses_amd import build as B  # noqa: E402


def main():
    B.build()
    hipcc = B._hipcc()
    outdir = os.path.join(B.LIBDIR, "ab")
    objdir = os.path.join(B.BUILD, "ab")
    os.makedirs(outdir, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    src = os.path.join(B.CSRC, "hydro_sweep.hip")
    others = [os.path.join(B.BUILD, o) for o, s, f in B.UNITS if s != "hydro_sweep.hip"]
    jobs = []
    for spec in sys.argv[1:]:
        tag, _, flags = spec.partition("=")
        extra = [f for f in flags.split(",") if f]
        objs = []
        for mode, mflags in (("strict", ["-ffp-contract=off"]), ("fast", ["-ffp-contract=off", "-DRAMSES_AMD_FAST=1"])):
            o = os.path.join(objdir, "sweep_%s_%s.o" % (tag, mode))
            objs.append(o)
            jobs.append([hipcc] + B.COMMON + mflags + extra + ["-c", src, "-o", o])
        jobs.append(("link", tag, objs))
    compiles = [j for j in jobs if isinstance(j, list)]
    with ThreadPoolExecutor(max_workers=min(len(compiles), os.cpu_count() or 2)) as ex:
        for r in ex.map(lambda c: subprocess.run(c, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), compiles):
            if r.returncode:
                print(r.stdout)
                sys.exit(1)
    for j in jobs:
        if isinstance(j, tuple):
            _, tag, objs = j
            lib = os.path.join(outdir, "libramses_amd_%s.so" % tag)
            subprocess.check_call([hipcc, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs + others)
            print("built", lib)


if __name__ == "__main__":
    main()
