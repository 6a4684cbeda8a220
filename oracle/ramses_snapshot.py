"""ramses_snapshot.py -- TEST INFRASTRUCTURE ONLY.

Minimal reader of the reference's Fortran-unformatted snapshot files
(output_NNNNN/{amr,hydro,grav}_NNNNN.outCCCCC) and a runner for the reference
binary built by oracle/build_ref.sh.  Record layouts follow the writers:
backup_amr (amr/output_amr.f90:211-400), backup_hydro
(hydro/output_hydro.f90:1-179), backup_poisson (poisson/output_poisson.f90:1-103).

Used to pin the oracle and the HIP path against END-TO-END runs of the
reference itself (same namelist in, per-cell fields out).
"""
import glob
import os
import shutil
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class FortranFile:
    def __init__(self, path):
        self.f = open(path, "rb")

    def record(self):
        head = self.f.read(4)
        if len(head) < 4:
            raise EOFError
        n = struct.unpack("<i", head)[0]
        data = self.f.read(n)
        tail = struct.unpack("<i", self.f.read(4))[0]
        assert tail == n
        return data

    def ints(self):
        return np.frombuffer(self.record(), dtype="<i4")

    def reals(self):
        return np.frombuffer(self.record(), dtype="<f8")

    def skip(self, n=1):
        for _ in range(n):
            self.record()

    def close(self):
        self.f.close()


def read_amr(path):
    """-> dict(header..., levels=[list over level of list over domain of
    dict(xg[ngrid,ndim], son[ngrid,2^ndim])])"""
    ff = FortranFile(path)
    ncpu = int(ff.ints()[0]); ndim = int(ff.ints()[0])
    nx, ny, nz = ff.ints()[:3]
    nlevelmax = int(ff.ints()[0]); ngridmax = int(ff.ints()[0]); nboundary = int(ff.ints()[0])
    ff.skip(1)  # ngrid_current
    boxlen = float(ff.reals()[0])
    ff.skip(1)  # noutput, iout, ifout
    ff.skip(2)  # tout, aout
    t = float(ff.reals()[0])
    ff.skip(1)  # dtold
    dtnew = ff.reals().copy()
    nstep = ff.ints().copy()
    einit, mass_tot_0, rho_tot = ff.reals()[:3]
    ff.skip(3)  # cosmology, aexp..., mass_sph
    ff.skip(2)  # headl, taill
    numbl = ff.ints().reshape(nlevelmax, ncpu)   # Fortran (ncpu,nlevelmax)
    ff.skip(1)  # numbtot
    numbb = None
    if nboundary > 0:
        ff.skip(2)
        numbb = ff.ints().reshape(nlevelmax, nboundary)
    ff.skip(1)  # free memory
    ordering = ff.record().decode().strip()
    if ordering == "bisection":
        ff.skip(5)
    else:
        ff.skip(1)  # bound_key
    ff.skip(3)  # coarse son, flag1, cpu_map
    twotondim, twondim = 2 ** ndim, 2 * ndim
    levels = []
    for il in range(nlevelmax):
        doms = []
        for ib in range(ncpu + nboundary):
            ncache = int(numbl[il, ib]) if ib < ncpu else int(numbb[il, ib - ncpu])
            if ncache == 0:
                doms.append(None)
                continue
            ff.skip(3)  # ind_grid, next, prev
            xg = np.stack([ff.reals() for _ in range(ndim)], axis=1)
            ff.skip(1)  # father
            ff.skip(twondim)  # nbor
            son = np.stack([ff.ints() for _ in range(twotondim)], axis=1)
            ff.skip(twotondim)  # cpu_map
            ff.skip(twotondim)  # flag1
            doms.append(dict(xg=xg, son=son))
        levels.append(doms)
    ff.close()
    return dict(ncpu=ncpu, ndim=ndim, nx=(int(nx), int(ny), int(nz)), nlevelmax=nlevelmax, ngridmax=ngridmax,
                nboundary=nboundary, boxlen=boxlen, t=t, dtnew=dtnew, nstep=nstep, rho_tot=float(rho_tot),
                levels=levels)


def _read_cellfile(path):
    """hydro_/grav_ files: -> (header dict, levels[il][ib] = array[nvar, 2^ndim, ncache])"""
    ff = FortranFile(path)
    ncpu = int(ff.ints()[0])
    nvar = int(ff.ints()[0])
    if os.path.basename(path).startswith("hydro"):
        ndim = int(ff.ints()[0])
        nlevelmax = int(ff.ints()[0])
        nboundary = int(ff.ints()[0])
        gamma = float(ff.reals()[0])
    else:
        # grav: nvar = ndim+1 (phi, f) or ndim+2 with -DOUTPUT_PARTICLE_DENSITY (rho, phi, f)
        nlevelmax = int(ff.ints()[0])
        nboundary = int(ff.ints()[0])
        gamma = None
        ndim = None
    levels = []
    for il in range(nlevelmax):
        doms = []
        for ib in range(ncpu + nboundary):
            ff.skip(1)  # ilevel
            ncache = int(ff.ints()[0])
            if ncache == 0:
                doms.append(None)
                continue
            # the number of octants is not in the grav header: read until the
            # next record is the 4-byte "ilevel" of the following domain
            octs = []
            while True:
                pos = ff.f.tell()
                head = ff.f.read(4)
                ff.f.seek(pos)
                if len(head) < 4 or struct.unpack("<i", head)[0] != 8 * ncache:
                    break
                octs.append(np.stack([ff.reals() for _ in range(nvar)], axis=0))
            doms.append(np.stack(octs, axis=1))  # [nvar, 2^ndim, ncache]
        levels.append(doms)
    ff.close()
    return dict(ncpu=ncpu, nvar=nvar, ndim=ndim, nlevelmax=nlevelmax, gamma=gamma), levels


def load_uniform_level(outdir, level, with_grav=False, grav_has_rho=False):
    """Assemble dense bricks [nvar, nz, ny, nx] of a fully refined level from all
    cpu files of one output directory.  Cell (ix,iy,iz) = floor(x*2^level)."""
    num = os.path.basename(outdir.rstrip("/")).split("_")[-1]
    amr_files = sorted(glob.glob(os.path.join(outdir, "amr_%s.out*" % num)))
    n = 2 ** level
    prim = None
    grav = None
    info = None
    for af in amr_files:
        cpu = af.split(".out")[-1]
        amr = read_amr(af)
        info = amr
        ndim = amr["ndim"]
        hh, hl = _read_cellfile(os.path.join(outdir, "hydro_%s.out%s" % (num, cpu)))
        if prim is None:
            shape = (n if ndim > 2 else 1, n if ndim > 1 else 1, n)
            prim = np.full((hh["nvar"],) + shape, np.nan)
        gl = None
        if with_grav:
            gh, gl = _read_cellfile(os.path.join(outdir, "grav_%s.out%s" % (num, cpu)))
            if grav is None:
                grav = np.full((gh["nvar"],) + prim.shape[1:], np.nan)
        icpu = int(cpu) - 1
        dom = amr["levels"][level - 1][icpu]
        if dom is None:
            continue
        xg = dom["xg"]  # oct centres in units of the coarse box (nx=1: [0,1))
        # coarse-grid offset: icoarse_min = 0 for nx=1 periodic boxes
        for ind in range(2 ** ndim):
            off = [(ind >> d) & 1 for d in range(ndim)]
            idx = []
            for d in range(ndim):
                xc = xg[:, d] + (off[d] - 0.5) * 0.5 ** level
                idx.append(np.floor(xc * n).astype(int))
            while len(idx) < 3:
                idx.append(np.zeros_like(idx[0]))
            prim[:, idx[2], idx[1], idx[0]] = hl[level - 1][icpu][:, ind, :]
            if with_grav:
                grav[:, idx[2], idx[1], idx[0]] = gl[level - 1][icpu][:, ind, :]
    assert not np.isnan(prim).any(), "level %d is not fully refined / not all cpus found" % level
    return dict(prim=prim, grav=grav, info=info)


def load_leaf_cells(outdir, with_grav=False):
    """All leaf cells (son == 0) of a snapshot, every level and cpu:
    -> dict(level, x[ncell,ndim] in box units, prim[nvar,ncell], info[, grav[ngv,ncell]])."""
    num = os.path.basename(outdir.rstrip("/")).split("_")[-1]
    levels, xs, prims, gravs = [], [], [], []
    info = None
    for af in sorted(glob.glob(os.path.join(outdir, "amr_%s.out*" % num))):
        cpu = af.split(".out")[-1]
        amr = read_amr(af)
        info = amr
        ndim = amr["ndim"]
        hh, hl = _read_cellfile(os.path.join(outdir, "hydro_%s.out%s" % (num, cpu)))
        gl = None
        if with_grav:
            gh, gl = _read_cellfile(os.path.join(outdir, "grav_%s.out%s" % (num, cpu)))
        icpu = int(cpu) - 1
        # coarse-grid offset of boxes with physical boundaries (nx = 3: the domain is the middle cell)
        skip = [1.0 if amr["nx"][d] > 1 else 0.0 for d in range(ndim)]
        scale = amr["boxlen"] / max(1, amr["nx"][0] - 2 * int(skip[0]))
        for il in range(amr["nlevelmax"]):
            dom = amr["levels"][il][icpu]
            if dom is None:
                continue
            dxl = 0.5 ** (il + 1)
            for ind in range(2 ** ndim):
                leaf = dom["son"][:, ind] == 0
                if not leaf.any():
                    continue
                pos = np.stack([(dom["xg"][leaf, d] + (((ind >> d) & 1) - 0.5) * dxl - skip[d]) * scale
                                for d in range(ndim)], axis=1)
                xs.append(pos)
                prims.append(hl[il][icpu][:, ind, :][:, leaf])
                if with_grav:
                    gravs.append(gl[il][icpu][:, ind, :][:, leaf])
                levels.append(np.full(int(leaf.sum()), il + 1))
    out = dict(level=np.concatenate(levels), x=np.concatenate(xs), prim=np.concatenate(prims, axis=1), info=info)
    if with_grav:
        out["grav"] = np.concatenate(gravs, axis=1)
    return out


def prim_to_cons(prim, gamma):
    """Inverse of backup_hydro's conversion (hydro/output_hydro.f90:83-129),
    only used to seed runs; parity checks go cons -> prim instead."""
    ndim = prim.shape[0] - 2
    u = np.zeros_like(prim)
    u[0] = prim[0]
    ek = 0.0
    for d in range(ndim):
        u[1 + d] = prim[0] * prim[1 + d]
        ek = ek + 0.5 * prim[0] * prim[1 + d] ** 2
    u[ndim + 1] = prim[ndim + 1] / (gamma - 1.0) + ek
    return u


def cons_to_prim(u, gamma, smallr=1e-10):
    """backup_hydro's conversion, same operation order (output_hydro.f90:83-129)."""
    ndim = u.shape[0] - 2
    q = np.zeros_like(u)
    q[0] = u[0]
    d = np.maximum(u[0], smallr)
    e = u[ndim + 1].copy()
    for k in range(ndim):
        q[1 + k] = u[1 + k] / d
        e = e - 0.5 * u[1 + k] ** 2 / d
    q[ndim + 1] = (gamma - 1.0) * e
    return q


def run_reference(namelist_text, ndim=3, nproc=1, binary=None, keep=False, timeout=3600):
    """Run oracle/_ref/ramses{ndim}d on a namelist in a scratch directory.
    Returns (workdir, stdout).  Caller removes workdir unless keep."""
    if binary is None:
        binary = os.path.join(HERE, "_ref", "ramses%dd" % ndim)
    if not os.path.exists(binary):
        raise FileNotFoundError(binary)
    work = tempfile.mkdtemp(prefix="ramses_ref_")
    with open(os.path.join(work, "run.nml"), "w") as fh:
        fh.write(namelist_text)
    cmd = [binary, "run.nml"]
    if nproc > 1:
        cmd = ["/opt/conda/bin/mpiexec", "-n", str(nproc)] + cmd
    out = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=timeout)
    if out.returncode != 0 and "Run completed" not in out.stdout:
        if not keep:
            shutil.rmtree(work, ignore_errors=True)
        raise RuntimeError("reference run failed (rc=%d):\n%s" % (out.returncode, out.stdout[-3000:]))
    return work, out.stdout


SEDOV3D_NML = """
&RUN_PARAMS
hydro=.true.
{poisson}
ncontrol=1
nrestart=0
nremap=0
nsubcycle=10*1
nstepmax={nstepmax}
verbose=.false.
/

&AMR_PARAMS
levelmin={level}
levelmax={level}
ngridtot={ngridtot}
nexpand=1
boxlen={boxlen}
/

&INIT_PARAMS
{init}
/

&OUTPUT_PARAMS
foutput={foutput}
noutput=1
tout=1000.0
/

&HYDRO_PARAMS
gamma=1.4
courant_factor=0.8
scheme='{scheme}'
slope_type={slope_type}
riemann='{riemann}'
/
{extra}
"""

SEDOV_INIT = """nregion=2
region_type(1)='square'
region_type(2)='point'
x_center=0.5,0.0
y_center=0.5,0.0
z_center=0.5,0.0
length_x=10.0,1.0
length_y=10.0,1.0
length_z=10.0,1.0
exp_region=10.0,10.0
d_region=1.0,0.0
u_region=0.0,0.0
v_region=0.0,0.0
p_region=1e-5,0.4"""


def sedov3d_namelist(level, nstepmax, foutput=1, riemann="llf", slope_type=1, scheme="muscl", boxlen=0.5,
                     poisson=False, init=SEDOV_INIT, extra="", mem_factor=1.3):
    """mem_factor: ngridtot / number of octs; use ~3 for MPI runs (ghost octs per rank)."""
    ngridtot = int(mem_factor * sum(8 ** l for l in range(level))) + 1000
    return SEDOV3D_NML.format(level=level, nstepmax=nstepmax, foutput=foutput, riemann=riemann,
                              slope_type=slope_type, scheme=scheme, boxlen=boxlen, ngridtot=ngridtot,
                              poisson="poisson=.true." if poisson else "", init=init, extra=extra)
