"""A/B variants of libramses_amd.so that differ in the compile-time knobs of amr_sweep.hip (development builds: minmod +
LLF + NVAR=5 only, seconds to compile):  scripts/build_ab_amr.py TAG=-DFLAG=1,-DOTHER=2 [TAG2=...]
(the flags are whatever temporary #ifdef a tuning session puts into the kernel; the shipped file carries none)
-> ramses_amd/lib/ab/libramses_amd_amr_TAG.so (load with RAMSES_AMD_LIB=...); the other objects come from the regular build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramses_amd import build as B  # noqa: E402


def main():
    hipcc = B._hipcc()
    outdir = os.path.join(B.LIBDIR, "ab")
    objdir = os.path.join(B.BUILD, "ab")
    os.makedirs(outdir, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    src = os.path.join(B.CSRC, "amr_sweep.hip")
    others = [os.path.join(B.BUILD, o) for o, s, f in B.UNITS if s != "amr_sweep.hip"]
    for spec in sys.argv[1:]:
        tag, _, flags = spec.partition("=")
        extra = [f for f in flags.split(",") if f]
        o = os.path.join(objdir, "amr_sweep_%s.o" % tag)
        log = o + ".log"
        r = subprocess.run([hipcc] + B.COMMON + ["-ffp-contract=off", "-DRAMSES_AMD_AMR_DEV=1", "-Rpass-analysis=kernel-resource-usage"] + extra +
                           ["-c", src, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        open(log, "w").write(r.stdout)
        if r.returncode:
            print(r.stdout[-3000:])
            sys.exit(1)
        lines = r.stdout.splitlines()
        for i, l in enumerate(lines):
            if "Function Name: _ZN10ramses_amd8amrsweep16amr_group_kernelILi1ELi0ELb0ELi5ELi0E" in l:
                res = [x.split("remark: ")[1].split(" [")[0] for x in lines[i:i + 14] if "VGPRs" in x or "Occupancy" in x or "LDS Size" in x]
                print(tag, "|", "; ".join(res))
                break
        lib = os.path.join(outdir, "libramses_amd_amr_%s.so" % tag)
        subprocess.check_call([hipcc, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib, o] + others + ["-ldl"])
        print("built", lib)


if __name__ == "__main__":
    main()
