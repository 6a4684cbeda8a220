#!/usr/bin/env python
"""Wall time of the reference program with and without the ramses_amd patch
on the GPU box (oracle/_ref binaries; nothing under /root/reference is read).

    python scripts/dropin_timing.py [level] [nstepmax]

Prints one JSON line per configuration: the program's own timer rows
(amr/update_time.f90:77-178) and the total elapsed time."""
import json
import os
import re
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402


def run(tag, binary, level, nstep, env):
    nml = rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=1000)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t0 = time.time()
    try:
        work, out = rs.run_reference(nml, binary=binary, timeout=3000)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.time() - t0
    shutil.rmtree(work, ignore_errors=True)
    rows = {}
    # serial timer table (amr/update_time.f90): "  seconds   %   STEP"
    for line in out.splitlines():
        m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([a-zA-Z].*?)\s*$", line)
        if m and "STEP" not in m.group(3):
            rows[m.group(3)] = float(m.group(1))
    tot = None
    print(json.dumps({"config": tag, "level": level, "steps": nstep, "wall_s": round(wall, 3),
                      "timers_s": rows}), flush=True)


def run_amr(tag, binary, env, lmin, lmax, nstep):
    """AMR + self-gravity (the blob + blast setup of tests/golden/make_golden_amr.py at higher levels)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mkb", os.path.join(ROOT, "tests", "golden", "make_golden_baseline.py"))
    mkb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mkb)
    nml = mkb.amr_grav_namelist(lmin, lmax, nstep, foutput=1000)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t0 = time.time()
    try:
        work, out = rs.run_reference(nml, binary=binary, timeout=3000)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.time() - t0
    shutil.rmtree(work, ignore_errors=True)
    rows = {}
    for line in out.splitlines():
        m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([a-zA-Z].*?)\s*$", line)
        if m and "STEP" not in m.group(3):
            rows[m.group(3)] = float(m.group(1))
    grids = re.findall(r"Level\s+(\d+) has\s+(\d+) grids", out)
    last = {}
    for l, g in grids:
        last[int(l)] = int(g)
    print(json.dumps({"config": tag, "levels": [lmin, lmax], "steps": nstep, "wall_s": round(wall, 3),
                      "octs_per_level": {k: v for k, v in last.items() if k >= lmin}, "timers_s": rows}), flush=True)


GRAV_BLOB = """nregion=2
region_type(1)='square'
region_type(2)='square'
x_center=0.5,0.4
y_center=0.5,0.55
z_center=0.5,0.6
length_x=10.0,0.25
length_y=10.0,0.25
length_z=10.0,0.25
exp_region=10.0,10.0
d_region=1.0,10.0
u_region=0.0,0.0
v_region=0.0,0.0
p_region=1.0,1.0"""


def run_grav(tag, binary, env, level, nstep):
    """BASELINE config C4 stand-in: uniform periodic level, hydro + self-gravity (multigrid_fine +
    force_fine every step), 'square' over-density."""
    nml = rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=1000, boxlen=1.0, poisson=True, init=GRAV_BLOB,
                              extra="&POISSON_PARAMS\nepsilon=1d-6\n/\n")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t0 = time.time()
    try:
        work, out = rs.run_reference(nml, binary=binary, timeout=3000)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.time() - t0
    shutil.rmtree(work, ignore_errors=True)
    rows = {}
    for line in out.splitlines():
        m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([a-zA-Z].*?)\s*$", line)
        if m and "STEP" not in m.group(3):
            rows[m.group(3)] = float(m.group(1))
    solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", out)
    print(json.dumps({"config": tag, "level": level, "steps": nstep, "wall_s": round(wall, 3),
                      "vcycles": [int(b) for _, b, _ in solves], "timers_s": rows}), flush=True)


def run_mpi(tag, binary, env, level, nstep, nproc):
    """sedov3d.nml on nproc MPI ranks (one brick per rank): the MAX column of the reference's MPI timer table"""
    nml = rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=1000, mem_factor=4.0)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t0 = time.time()
    try:
        work, out = rs.run_reference(nml, binary=binary, nproc=nproc, timeout=3000)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.time() - t0
    shutil.rmtree(work, ignore_errors=True)
    rows = {}
    for line in out.splitlines():
        m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+\S+\s+([0-9.]+)\s+(\d+)\s+(\d+)\s+([a-zA-Z].*?)\s*$", line)
        if m:
            rows[m.group(8)] = float(m.group(3))
        m = re.match(r"^\s*([0-9.]+)\s+100\.0\s+TOTAL", line)
        if m:
            rows["TOTAL"] = float(m.group(1))
    note = [l.strip() for l in out.splitlines() if "ramses_amd:" in l]
    print(json.dumps({"config": tag, "level": level, "steps": nstep, "ranks": nproc, "wall_s": round(wall, 3),
                      "timers_max_s": rows, "notes": note[:4]}), flush=True)


def run_grav_mpi(tag, binary, env, level, nstep, nproc):
    """BASELINE config C4 (uniform periodic level, hydro + self-gravity) on nproc MPI ranks: every level takes the AMR
    multigrid path under MPI (the reference's driver, device operators); MAX column of the MPI timer table"""
    nml = rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=1000, boxlen=1.0, poisson=True, init=GRAV_BLOB, mem_factor=4.0,
                              extra="&POISSON_PARAMS\nepsilon=1d-6\n/\n")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t0 = time.time()
    try:
        work, out = rs.run_reference(nml, binary=binary, nproc=nproc, timeout=3000)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.time() - t0
    shutil.rmtree(work, ignore_errors=True)
    rows = {}
    for line in out.splitlines():
        m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+\S+\s+([0-9.]+)\s+(\d+)\s+(\d+)\s+([a-zA-Z].*?)\s*$", line)
        if m:
            rows[m.group(8)] = float(m.group(3))
        m = re.match(r"^\s*([0-9.]+)\s+100\.0\s+TOTAL", line)
        if m:
            rows["TOTAL"] = float(m.group(1))
    solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", out)
    stats = re.findall(r"level arrays across PCIe after the upload:\s*(\d+) bytes in\s*(\d+) copies; halo:\s*(\d+) bytes in\s*(\d+) exchanges", out)
    pcie = {"level_array_bytes": sum(int(a) for a, _, _, _ in stats), "halo_bytes": sum(int(c) for _, _, c, _ in stats),
            "halo_exchanges": sum(int(d) for _, _, _, d in stats)} if stats else None
    sweeps = [l.strip()[11:] for l in out.splitlines() if "godunov_fine of AMR levels" in l]
    # bytes of the Poisson level arrays over PCIe, all ranks (exit lines of RAMSES_AMD_STATS=1; round 6)
    ftr = re.findall(r"acceleration f over PCIe:\s*(\d+) bytes to the device,\s*(\d+) bytes back", out)
    mtr = re.findall(r"distributed multigrid over PCIe: rho\s*(\d+) bytes to the device, phi\s*(\d+) bytes back", out)
    rtr = re.findall(r"density deposit rho over PCIe:\s*(\d+) bytes back", out)
    level_arrays = {"f_up": sum(int(a) for a, _ in ftr), "f_down": sum(int(b) for _, b in ftr), "rho_up": sum(int(a) for a, _ in mtr),
                    "phi_down": sum(int(b) for _, b in mtr), "rho_down": sum(int(a) for a in rtr)} if (ftr or mtr or rtr) else None
    print(json.dumps({"config": tag, "level": level, "steps": nstep, "ranks": nproc, "wall_s": round(wall, 3),
                      "vcycles": [int(b) for _, b, _ in solves], "timers_max_s": rows, "multigrid_pcie_all_ranks": pcie,
                      "poisson_level_array_bytes_all_ranks": level_arrays, "sweeps_of_rank": sweeps[:2]}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gravmpi":
        # python scripts/dropin_timing.py gravmpi LEVEL NSTEP NPROC [all|gpu|ref]
        level, nstep, nproc = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        which = sys.argv[5] if len(sys.argv) > 5 else "all"
        ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
        pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")
        if which == "pcie":
            # round 6: what the Poisson level arrays (rho, phi, f) cost on the bus, steady state on / off, for nstep and 2 nstep steps
            base = {"RAMSES_AMD": "1", "RAMSES_AMD_STATS": "1"}
            for n in (nstep, 2 * nstep):
                run_grav_mpi("patched (default): rho, phi, f resident from the second solve of the steady state on", pat, dict(base), level, n, nproc)
            run_grav_mpi("patched, RAMSES_AMD_PHI_RESIDENT=0 RAMSES_AMD_F_RESIDENT=0: the paths before round 6", pat,
                         dict(base, RAMSES_AMD_PHI_RESIDENT="0", RAMSES_AMD_F_RESIDENT="0"), level, 2 * nstep, nproc)
            run_grav_mpi("reference (MPI)", ref, {"RAMSES_AMD": "0"}, level, 2 * nstep, nproc)
        if which == "tiles":
            base = {"RAMSES_AMD": "1", "RAMSES_AMD_STATS": "1"}
            run_grav_mpi("patched (default): the rank's octs in tiles, dense sweep in place; distributed dense V-cycles", pat, dict(base), level, nstep, nproc)
            run_grav_mpi("patched, RAMSES_AMD_DEVICE_ORDER=0: host numbering on the device, tree-walking sweep (the round-4 layout)", pat,
                         dict(base, RAMSES_AMD_DEVICE_ORDER="0"), level, nstep, nproc)
        if which in ("all", "gpu"):
            run_grav_mpi("patched: dense V-cycles distributed over one brick per rank, one deep-halo exchange per smoother launch (csrc/mg_dist.hip; default)", pat,
                         {"RAMSES_AMD": "1", "RAMSES_AMD_MG_STATS": "1"}, level, nstep, nproc)
            run_grav_mpi("patched: multigrid of AMR levels, levels resident on the device, virtual boundaries exchanged from there (RAMSES_AMD_MG_DIST=0)", pat,
                         {"RAMSES_AMD": "1", "RAMSES_AMD_MG_STATS": "1", "RAMSES_AMD_MG_DIST": "0"}, level, nstep, nproc)
            if which == "all":
                run_grav_mpi("patched: arrays across PCIe around every multigrid routine, host exchanges (RAMSES_AMD_MG_DIST=0 RAMSES_AMD_MG_MPI_SYNC=1, round 2)", pat,
                             {"RAMSES_AMD": "1", "RAMSES_AMD_MG_STATS": "1", "RAMSES_AMD_MG_DIST": "0", "RAMSES_AMD_MG_MPI_SYNC": "1"}, level, nstep, nproc)
        if which in ("all", "ref"):
            run_grav_mpi("reference (MPI)", ref, {"RAMSES_AMD": "0"}, level, nstep, nproc)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mpi":
        level, nstep, nproc = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        which = sys.argv[5] if len(sys.argv) > 5 else "all"
        ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
        pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")
        if which in ("all", "gpu"):
            run_mpi("patched, one brick per rank resident on the GPU, exchange behind the interior sweep", pat, {"RAMSES_AMD": "1"}, level, nstep, nproc)
            run_mpi("patched, resident bricks, exchange after the sweep", pat, {"RAMSES_AMD": "1", "RAMSES_AMD_OVERLAP": "0"}, level, nstep, nproc)
            run_mpi("patched, tree-walking sweep + the reference's host MPI halo (round 1 path)", pat, {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT": "0"}, level, nstep, nproc)
        if which in ("all", "ref"):
            run_mpi("reference (MPI)", ref, {"RAMSES_AMD": "0"}, level, nstep, nproc)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("c5", "c5walls"):
        # BASELINE config C5 (sedov3d.nml, AMR, hydro only) at levels lmin..lmax, nstep coarse steps;
        # c5walls: the same run between six physical boundaries (x, z reflexive, y free): make_boundary_hydro on the device
        walls = sys.argv[1] == "c5walls"
        lmin, lmax, nstep = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        which = sys.argv[5] if len(sys.argv) > 5 else "all"
        import importlib.util
        spec = importlib.util.spec_from_file_location("mkb", os.path.join(ROOT, "tests", "golden", "make_golden_baseline.py"))
        mkb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mkb)
        ngt = {7: 900000, 8: 4000000}.get(lmin, 300000)
        nml = mkb.c5_namelist(lmin, lmax, nstep, ngt).replace("foutput=%d" % nstep, "foutput=1000")
        if walls:
            nml += """
&BOUNDARY_PARAMS
nboundary=6
ibound_min=-1,+1,-1,-1,-1,-1
ibound_max=-1,+1,+1,+1,+1,+1
jbound_min= 0, 0,-1,+1,-1,-1
jbound_max= 0, 0,-1,+1,+1,+1
kbound_min= 0, 0, 0, 0,-1,+1
kbound_max= 0, 0, 0, 0,-1,+1
bound_type= 1, 1, 2, 2, 1, 1
/
"""

        def run_c5(tag, binary, env):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            t0 = time.time()
            try:
                work, out = rs.run_reference(nml, binary=binary, timeout=3000)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            wall = time.time() - t0
            shutil.rmtree(work, ignore_errors=True)
            rows = {}
            for line in out.splitlines():
                m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([a-zA-Z].*?)\s*$", line)
                if m and "STEP" not in m.group(3):
                    rows[m.group(3)] = float(m.group(1))
            last = {}
            for l, g in re.findall(r"Level\s+(\d+) has\s+(\d+) grids", out):
                last[int(l)] = int(g)
            rec = {"config": tag, "levels": [lmin, lmax], "steps": nstep, "wall_s": round(wall, 3),
                   "octs_per_level": {k: v for k, v in last.items() if k >= lmin}, "timers_s": rows,
                   "notes": [ln.strip() for ln in out.splitlines() if "ramses_amd: godunov_fine of AMR levels" in ln][:2]}
            if env.get("RAMSES_AMD_PROFILE"):
                # the shims' own wall-clock table (ramses_amd_tic / ramses_amd_toc): what of the reference's "flag" timer is the shim
                rec["shim_profile"] = [ln.strip() for ln in out.splitlines() if "hydro_flag" in ln or "godunov" in ln or "ramses_amd profile" in ln][:16]
            print(json.dumps(rec), flush=True)

        ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
        pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
        if which == "prof":
            run_c5("patched, state and tree resident on the GPU, RAMSES_AMD_PROFILE=1", pat, {"RAMSES_AMD": "1", "RAMSES_AMD_PROFILE": "1"})
        if which == "tiles":
            # round 5: the device's own oct numbering -- levels in tiles + the dense sweep in place; the same layout with the
            # tree-walking sweep; the host's numbering with the tree-walking sweep (what round 4 ran on the partial levels)
            base = {"RAMSES_AMD": "1", "RAMSES_AMD_STATS": "1", "RAMSES_AMD_PROFILE": "1"}
            run_c5("resident, levels in tiles, dense sweep in place (default: levels below 32768 octs walk the tree)", pat, dict(base))
            run_c5("resident, levels in tiles, dense sweep in place on every level (RAMSES_AMD_TILE_MIN_OCTS=0)", pat, dict(base, RAMSES_AMD_TILE_MIN_OCTS="0"))
            run_c5("resident, levels in tiles, tree-walking sweep (RAMSES_AMD_TILE_DENSE=0 RAMSES_AMD_COVERED_DENSE=0)", pat,
                   dict(base, RAMSES_AMD_TILE_DENSE="0", RAMSES_AMD_COVERED_DENSE="0"))
            run_c5("resident, host numbering on the device, tree-walking sweep (RAMSES_AMD_DEVICE_ORDER=0)", pat,
                   dict(base, RAMSES_AMD_DEVICE_ORDER="0"))
        if which in ("all", "gpu"):
            run_c5("patched, state and tree resident on the GPU", pat, {"RAMSES_AMD": "1"})
            run_c5("patched, arrays staged around every godunov_fine (round 1 path)", pat,
                   {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_WALLS" if walls else "RAMSES_AMD_RESIDENT_AMR": "0"})
        if which in ("all", "ref"):
            run_c5("reference (1 core)", ref, {"RAMSES_AMD": "0"})
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c5mpi":
        # config C5's shape on several ranks (sedov3d.nml, AMR, hydro only): lmin lmax nstep nproc
        lmin, lmax, nstep, nproc = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        which = sys.argv[6] if len(sys.argv) > 6 else "all"
        import importlib.util
        spec = importlib.util.spec_from_file_location("mkb", os.path.join(ROOT, "tests", "golden", "make_golden_baseline.py"))
        mkb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mkb)
        ngt = {7: 2000000, 8: 8000000}.get(lmin, 600000)
        nml = mkb.c5_namelist(lmin, lmax, nstep, ngt).replace("foutput=%d" % nstep, "foutput=1000")
        walls = False
        if walls:
            nml += """
&BOUNDARY_PARAMS
nboundary=6
ibound_min=-1,+1,-1,-1,-1,-1
ibound_max=-1,+1,+1,+1,+1,+1
jbound_min= 0, 0,-1,+1,-1,-1
jbound_max= 0, 0,-1,+1,+1,+1
kbound_min= 0, 0, 0, 0,-1,+1
kbound_max= 0, 0, 0, 0,-1,+1
bound_type= 1, 1, 2, 2, 1, 1
/
"""

        def run_c5mpi(tag, binary, env):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            t0 = time.time()
            try:
                work, out = rs.run_reference(nml, binary=binary, nproc=nproc, timeout=3000)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            wall = time.time() - t0
            shutil.rmtree(work, ignore_errors=True)
            rows = {}
            for line in out.splitlines():
                m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+\S+\s+([0-9.]+)\s+(\d+)\s+(\d+)\s+([a-zA-Z].*?)\s*$", line)
                if m:
                    rows[m.group(8)] = float(m.group(3))
                m = re.match(r"^\s*([0-9.]+)\s+100\.0\s+TOTAL", line)
                if m:
                    rows["TOTAL"] = float(m.group(1))
            last = {}
            for l, g in re.findall(r"Level\s+(\d+) has\s+(\d+) grids", out):
                last[int(l)] = int(g)
            note = [l.strip() for l in out.splitlines() if "ramses_amd:" in l and "godunov_fine of AMR levels" not in l]
            sweeps = [l.strip()[11:] for l in out.splitlines() if "godunov_fine of AMR levels" in l]
            print(json.dumps({"config": tag, "levels": [lmin, lmax], "steps": nstep, "ranks": nproc, "wall_s": round(wall, 3),
                              "octs_per_level": {k: v for k, v in last.items() if k >= lmin}, "timers_max_s": rows, "notes": note[:4],
                              "sweeps_of_rank": sweeps[:2]}), flush=True)

        ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
        pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")
        if which == "tiles":
            base = {"RAMSES_AMD": "1", "RAMSES_AMD_STATS": "1"}
            run_c5mpi("resident, every rank's levels in tiles, dense sweep in place (default)", pat, dict(base))
            run_c5mpi("resident, host numbering on the device, tree-walking sweep (RAMSES_AMD_DEVICE_ORDER=0: the round-4 layout)", pat,
                      dict(base, RAMSES_AMD_DEVICE_ORDER="0"))
        if which in ("all", "gpu"):
            run_c5mpi("patched, every rank's cell vectors and tree resident on the GPU, virtual boundaries on the device", pat, {"RAMSES_AMD": "1"})
            run_c5mpi("patched, arrays staged around every godunov_fine + the reference's host MPI halo (round 1 path)", pat,
                      {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_AMR_MPI": "0"})
        if which in ("all", "ref"):
            run_c5mpi("reference (MPI)", ref, {"RAMSES_AMD": "0"})
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mhd":
        # SOLVER=mhd: a magnetised blast on a uniform periodic level (tests/mhd_common.py), hlld + hlld
        level, nstep = int(sys.argv[2]), int(sys.argv[3])
        which = sys.argv[4] if len(sys.argv) > 4 else "all"
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from mhd_common import mhd_namelist
        nml = mhd_namelist(level, nstep, "hlld", "hlld", 2).replace("foutput=%d" % nstep, "foutput=1000")

        def run_mhd(tag, binary, env):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            t0 = time.time()
            try:
                work, out = rs.run_reference(nml, binary=binary, timeout=3000)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            wall = time.time() - t0
            shutil.rmtree(work, ignore_errors=True)
            rows = {}
            for line in out.splitlines():
                m = re.match(r"^\s*([0-9.]+)\s+([0-9.]+)\s+([a-zA-Z].*?)\s*$", line)
                if m and "STEP" not in m.group(3):
                    rows[m.group(3)] = float(m.group(1))
            print(json.dumps({"config": tag, "level": level, "steps": nstep, "wall_s": round(wall, 3), "timers_s": rows}), flush=True)

        ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mhd")
        pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch_mhd_mhd")
        if which in ("all", "gpu"):
            run_mhd("patched (SOLVER=mhd), level resident on the GPU", pat, {"RAMSES_AMD": "1"})
            run_mhd("patched (SOLVER=mhd), godunov_fine staged", pat, {"RAMSES_AMD": "1", "RAMSES_AMD_MHD_RESIDENT": "0"})
        if which in ("all", "ref"):
            run_mhd("reference SOLVER=mhd (1 core)", ref, {"RAMSES_AMD": "0"})
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "grav":
        level, nstep = int(sys.argv[2]), int(sys.argv[3])
        which = sys.argv[4] if len(sys.argv) > 4 else "all"
        ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
        pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
        if which in ("all", "gpu"):
            run_grav("patched, level and acceleration resident on the GPU", pat, {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_GRAV": "1"}, level, nstep)
            run_grav("patched, arrays staged per call", pat, {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_GRAV": "0"}, level, nstep)
        if which in ("all", "ref"):
            run_grav("reference (1 core)", ref, {"RAMSES_AMD": "0"}, level, nstep)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "amr":
        lmin, lmax, nstep = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        which = sys.argv[5] if len(sys.argv) > 5 else "all"
        ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
        pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
        if which in ("all", "gpu"):
            run_amr("patched: hydro state, tree and acceleration resident; multigrid driver + setup and force_fine of the AMR levels on the device",
                    pat, {"RAMSES_AMD": "1"}, lmin, lmax, nstep)
            run_amr("patched, hydro arrays staged per call (RAMSES_AMD_RESIDENT_GRAV=0), device multigrid driver", pat,
                    {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_GRAV": "0"}, lmin, lmax, nstep)
            run_amr("patched, staged + the reference's multigrid driver / setup / force_fine with device operators (round 1 path)", pat,
                    {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_GRAV": "0", "RAMSES_AMD_MG_DRIVER": "host"}, lmin, lmax, nstep)
        if which in ("all", "ref"):
            run_amr("reference (1 core)", ref, {"RAMSES_AMD": "0"}, lmin, lmax, nstep)
        sys.exit(0)
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
    pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
    if which in ("all", "gpu"):
        run("patched, device-resident", pat, level, nstep, {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT": "1"})
        run("patched, staged per sweep", pat, level, nstep, {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT": "0"})
    if which in ("all", "ref"):
        run("reference (1 core)", ref, level, nstep, {"RAMSES_AMD": "0"})
