"""A/B of sweep-kernel variants on the GPU box: for every ramses_amd/lib/ab/libramses_amd_*.so (and the
default library) run bench.py's sweep leg and print kernel ms (fast / strict builds).
scripts/ab_sweep.py [--n 512] [tags...]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    n = "512"
    if args[:1] == ["--n"]:
        n = args[1]
        args = args[2:]
    libs = {"default": None}
    for p in sorted(glob.glob(os.path.join(ROOT, "ramses_amd", "lib", "ab", "libramses_amd_*.so"))):
        tag = os.path.basename(p)[len("libramses_amd_"):-3]
        if not args or tag in args:
            libs[tag] = p
    # AB_CFGS="tag:rows,zchunk tag:rows,zchunk ..." adds runs of a library with other tile rows / z-chunks
    runs = [(tag, path, []) for tag, path in libs.items()]
    for cfg in os.environ.get("AB_CFGS", "").split():
        tag, _, rz = cfg.partition(":")
        rows, _, zc = rz.partition(",")
        if tag in libs:
            runs.append(("%s[%s,%s]" % (tag, rows, zc), libs[tag], ["--tile-rows", rows, "--zchunk", zc or "0"]))
    for rep in range(2):
        for tag, path, extra in runs:
            env = dict(os.environ)
            if path:
                env["RAMSES_AMD_LIB"] = path
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--vcycle-level", "0",
                                "--amr-level", "0", "--stress-steps", "0", "--mhd-level", "0",
                                "--steps", "20", "--warmup", "3", "--n", n] + extra, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, "FAILED", r.stdout[-400:])
                continue
            j = json.loads(line[-1])
            print("%-22s rep%d fast %.3f ms (%.1f%%)  strict %.3f ms (%.1f%%)" % (
                tag, rep, j["roofline"]["kernel_ms"], 100 * j["roofline"]["frac"], j["strict_build"]["kernel_ms"],
                100 * j["strict_build"]["frac"]), flush=True)


if __name__ == "__main__":
    main()
