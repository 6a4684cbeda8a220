"""Build libramses_amd.so (hand-written HIP kernels + C ABI) for gfx950.

hipcc cross-compiles without a GPU.  Objects are cached on source mtimes under
ramses_amd/build/ and the shared library is written IN-TREE to
ramses_amd/lib/libramses_amd.so so that it travels to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libramses_amd.so")
ARCH = "gfx950"

# --offload-compress: the device code objects of the fat binary are stored compressed (the HIP runtime inflates a translation
# unit's code when its first kernel is launched): libramses_amd.so is 17 MB instead of 110 MB -- the template matrix of the
# option space (amr_sweep: 1080 kernels) compresses tenfold -- and loads and passes the GPU suite the same
# (profiles/r06_compressed_library.txt)
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "--offload-compress",
          "-I", os.path.join(HERE, "..", "include")]

# The fast minmod unit (the flagship kernel of the bench line and the tile sweep of sedov3d.nml's AMR levels) is scheduled for
# instruction-level parallelism instead of occupancy: its occupancy is set by LDS (one 12-wave workgroup per CU) whatever the
# register count, so the default strategy's register economy buys nothing.  Measured at 512^3 (profiles/r06_codegen_ab.txt):
# 3.08 -> 2.98 ms (43.5 -> 45.0 % of the roofline); the shell level in tiles 1.65 -> 1.55 ms; same instructions, same bits.
# Neutral or worse on the strict units (5.40 ms either way), the tree-walking sweep (2.87 -> 3.02), the multigrid and MHD units.
ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-schedule-relaxed-occupancy"]

# (object name, source, extra flags)
UNITS = [
    ("hydro_sweep_strict.o", "hydro_sweep.hip", ["-ffp-contract=off"]),
    ("hydro_sweep_fast.o", "hydro_sweep.hip", ["-ffp-contract=off", "-DRAMSES_AMD_FAST=1"]),
] + [
    # (one unit per slope type and arithmetic; 3 stands for 3, 4, 5, 6: csrc/hydro_sweep.hip)
    ("hydro_sweep_%s_st%d.o" % (mode, st), "hydro_sweep.hip", ["-ffp-contract=off", "-DSWEEP_ST=%d" % st] + flag
     + (ILP if (mode, st) == ("fast", 1) else []))
    for st in (1, 2, 0, 7, 8, 3) for mode, flag in (("strict", []), ("fast", ["-DRAMSES_AMD_FAST=1"]))
] + [
    ("hydro_misc.o", "hydro_misc.hip", ["-ffp-contract=off"]),
    ("mg_kernels.o", "mg_kernels.hip", ["-ffp-contract=off"]),
    ("octree_pack.o", "octree_pack.hip", ["-ffp-contract=off"]),
    ("amr_ops.o", "amr_ops.hip", ["-ffp-contract=off"]),
    ("amr_sweep.o", "amr_sweep.hip", ["-ffp-contract=off"]),
    ("amr_sweep_st0.o", "amr_sweep.hip", ["-ffp-contract=off", "-DAMR_SWEEP_ST=0"]),
    ("amr_sweep_st1.o", "amr_sweep.hip", ["-ffp-contract=off", "-DAMR_SWEEP_ST=1"]),
    ("amr_sweep_st2.o", "amr_sweep.hip", ["-ffp-contract=off", "-DAMR_SWEEP_ST=2"]),
    ("amr_sweep_st3.o", "amr_sweep.hip", ["-ffp-contract=off", "-DAMR_SWEEP_ST=3"]),
    ("amr_sweep_st7.o", "amr_sweep.hip", ["-ffp-contract=off", "-DAMR_SWEEP_ST=7"]),
    ("amr_sweep_st8.o", "amr_sweep.hip", ["-ffp-contract=off", "-DAMR_SWEEP_ST=8"]),
    ("mg_amr.o", "mg_amr.hip", ["-ffp-contract=off"]),
    ("cg_amr.o", "cg_amr.hip", ["-ffp-contract=off"]),
    ("rho_fine.o", "rho_fine.hip", ["-ffp-contract=off"]),
    ("capi.o", "capi.hip", ["-ffp-contract=off"]),
    ("capi_host.o", "capi_host.hip", ["-ffp-contract=off"]),
    ("capi_tree_poisson.o", "capi_tree_poisson.hip", ["-ffp-contract=off"]),
    ("capi_mpi.o", "capi_mpi.hip", ["-ffp-contract=off"]),
    ("capi_amr.o", "capi_amr.hip", ["-ffp-contract=off"]),
    ("pois_amr.o", "pois_amr.hip", ["-ffp-contract=off"]),
    ("mg_dist.o", "mg_dist.hip", ["-ffp-contract=off"]),
    ("mhd_sweep.o", "mhd_sweep.hip", ["-ffp-contract=off"]),
    # (the fast arithmetic of the MHD sweep: csrc/mhd_sweep.hip header; the flags come after COMMON's and win)
    ("mhd_sweep_fast.o", "mhd_sweep.hip", ["-DRAMSES_AMD_MHD_FAST_TU=1", "-fapprox-func", "-ffp-contract=fast"]),
    ("mhd_amr.o", "mhd_amr.hip", ["-ffp-contract=off"]),
]


def _hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        try:
            subprocess.check_output([c, "--version"], stderr=subprocess.STDOUT)
            return c
        except (OSError, subprocess.CalledProcessError):
            continue
    raise RuntimeError("hipcc not found: libramses_amd.so cannot be built")


_INC = None


def _deps(src):
    """the headers a source file reaches through #include "..." (csrc/ and include/), transitively:
    a change of the C ABI header rebuilds only the units that include it, not the ten-minute sweep kernels"""
    import re
    global _INC
    if _INC is None:
        _INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        try:
            text = open(f).read()
        except OSError:
            continue
        for inc in _INC.findall(text):
            for base in (os.path.dirname(f), CSRC, os.path.join(HERE, "..", "include")):
                cand = os.path.normpath(os.path.join(base, inc))
                if os.path.exists(cand):
                    if cand not in seen:
                        seen.add(cand)
                        todo.append(cand)
                    break
    return sorted(seen)


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    global BUILD, LIB
    if os.environ.get("RAMSES_AMD_BUILD_VARIANT"):
        # an A/B build of the whole library with extra compiler flags (RAMSES_AMD_BUILD_FLAGS), objects and library kept apart:
        # ramses_amd/lib/ab/libramses_amd_<variant>.so (load with RAMSES_AMD_LIB=...)
        tag = os.environ["RAMSES_AMD_BUILD_VARIANT"]
        BUILD = os.path.join(HERE, "build", "variant_" + tag)
        os.makedirs(os.path.join(LIBDIR, "ab"), exist_ok=True)
        LIB = os.path.join(LIBDIR, "ab", "libramses_amd_%s.so" % tag)
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    extra = os.environ.get("RAMSES_AMD_BUILD_FLAGS", "").split()
    jobs = []
    objs = []
    for obj, src, flags in UNITS:
        srcp = os.path.join(CSRC, src)
        if not os.path.exists(srcp):
            continue
        objp = os.path.join(BUILD, obj)
        objs.append(objp)
        if force or _stale(objp, [srcp] + _deps(srcp)):
            jobs.append([hipcc] + COMMON + flags + extra + ["-c", srcp, "-o", objp])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            list(ex.map(run, jobs))
    if jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
