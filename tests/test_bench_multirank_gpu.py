"""bench.py's N>1 path end to end on the one-GPU box: the driver's launch line
(`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`) with both ranks on the same device.
  nccl-shared  RCCL is asked for with two ranks on one GPU: it must refuse, the self-test of pick_transport() must notice on
               EVERY rank, and the run must fall back to host-staged gloo and say so in the JSON line;
  gloo         the smoke-test backend.
Neither is a measurement; what is checked is that the line comes out complete -- whole-job value, the census of ranks and
devices, the per-stage breakdown of a step (sweep, pack, send/recv, unpack) and the distributed V-cycle leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("backend", ["nccl-shared", "gloo"])
def test_two_ranks_on_one_gpu(gpu_lib, backend):
    env = dict(os.environ, RAMSES_AMD_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611" if backend == "gloo" else "29612", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--cells", "64", "--vcycle-level", "6", "--spinup-ms", "0", "--deadline", "240", "--vcycle-deadline", "120"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:       # keep the whole story for the session log (pytest shortens long assertion messages)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_multirank_%s.err" % backend), "w") as fo:
            fo.write(r.stdout + "\n==== stderr ====\n" + r.stderr)
    assert lines, (r.stdout[-800:], r.stderr[-1500:])
    j = json.loads(lines[-1])
    assert j.get("error") is None, j
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["value"] > 0 and abs(j["value"] - 2 * 64 ** 3 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    cfg = j["config"]
    assert cfg["ranks"]["world_size"] == 2 and len(cfg["ranks"]["per_rank"]) == 2
    assert cfg["ranks"]["distinct_devices"] == 1            # both ranks on this box's one GPU: said, not hidden
    assert "gloo" in cfg["ranks"]["transport"] and "gloo" in cfg["halo"]
    if backend == "nccl-shared":
        assert "RCCL self-test failed" in cfg["ranks"]["transport"]
    bd = cfg["step_breakdown"]
    for k in ("sweep_ms", "pack_ms", "sendrecv_ms", "unpack_ms"):
        assert bd[k] > 0.0, bd
    assert bd["bytes_sent_per_rank"] == 8 * 5 * ((64 + 4) ** 2 * 2 * 2)      # two x faces of 2 ghost layers incl. the edge regions, 5 variables
    vc = j["vcycle"]
    assert vc.get("error") is None and vc["value"] > 0 and vc["n_gpus"] == 2 and vc["vcycles"] == 10, vc
