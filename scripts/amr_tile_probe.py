"""ramses_amd_amrres_godunov of a level in tiles, for rocprofv3 --kernel-trace --stats:  python scripts/amr_tile_probe.py [level] [kind] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    kind = sys.argv[2] if len(sys.argv) > 2 else "covered"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    import torch
    torch.cuda.init()
    out = bench.amr_resident_bench(level, steps=steps, kind=kind)
    print(json.dumps({k: out[k] for k in ("ms_per_sweep", "tree_walking_ms_per_sweep", "cells", "workload")}))
    print("frac", out["roofline"]["frac"])
