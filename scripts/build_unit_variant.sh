#!/bin/bash
# scripts/build_unit_variant.sh TAG OBJ "extra flags": ramses_amd/lib/ab/libramses_amd_TAG.so = the regular build with ONE object
# (OBJ: a name of ramses_amd/build.py's UNITS, e.g. mg_kernels.o, mhd_sweep.o, hydro_sweep_fast_st1.o) recompiled with extra
# compiler flags -- for A/B runs of code-generation knobs (RAMSES_AMD_LIB=... selects the library)
set -e
cd "$(dirname "$0")/.."
tag=$1; obj=$2; shift; shift
mkdir -p ramses_amd/lib/ab ramses_amd/build/ab
python - "$tag" "$obj" "$@" <<'PY'
import os, subprocess, sys
sys.path.insert(0, ".")
from ramses_amd import build as B
tag, obj, extra = sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]).split()
names = obj.split(",")          # (several objects: a,b,c)
outs = {}
procs = []
for o in names:
    unit = [u for u in B.UNITS if u[0] == o][0]
    outs[o] = os.path.join(B.BUILD, "ab", "%s_%s" % (tag, o))
    procs.append(subprocess.Popen([B._hipcc()] + B.COMMON + unit[2] + extra + ["-c", os.path.join(B.CSRC, unit[1]), "-o", outs[o]]))
if any(p.wait() for p in procs):
    sys.exit("compile failed")
objs = [outs.get(u[0], os.path.join(B.BUILD, u[0])) for u in B.UNITS]
lib = os.path.join(B.LIBDIR, "ab", "libramses_amd_%s.so" % tag)
subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
print("built", lib)
PY
