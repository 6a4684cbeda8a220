"""CPU tests of the host-side halo plan behind the C ABI (ramses_amd_halo_plan: no device needed).

A uniform level is cut into rank boxes the way the reference's Hilbert split does for 2^k ranks;
emission / reception oct lists are built the way build_comm leaves them (amr/virtual_boundaries.f90:
1286-1648: reception(icpu)%igrid(i) on the receiver <-> emission(myid)%igrid(i) on the owner, same
order).  The plan must turn them into brick offsets such that pack -> exchange -> unpack -> periodic
self-fill leaves in every ghost cell the value of the periodic neighbour cell (the contract of
make_virtual_fine_dp, :373-528).  The device kernels index exactly like the numpy emulation here
(capi_mpi.hip: buf[(m*nvar+v)*8+ind] <-> brick[org+...]).  One variant runs the exchange between two
real processes over gloo."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    import ramses_amd
    return ramses_amd.lib()


def make_ranks(level, pgrid, seed=0, extra_layer=False):
    """-> list over ranks of dict(igrid, xg, ngridmax, em_n, em_ig, rc_n, rc_ig, box=(lo, dim), slot_pos)
    Octs of the level get random slots per rank (own octs and ghost octs of the local tree)."""
    rng = np.random.default_rng(seed)
    no = 2 ** (level - 1)
    px, py, pz = pgrid
    assert no % px == 0 and no % py == 0 and no % pz == 0
    dims = (no // px, no // py, no // pz)
    ncpu = px * py * pz
    owner = {}
    boxes = []
    for r in range(ncpu):
        rx, ry, rz = r % px, (r // px) % py, r // (px * py)
        lo = (rx * dims[0], ry * dims[1], rz * dims[2])
        boxes.append((lo, dims))
    def owner_of(o):
        return (o[0] // dims[0]) + px * ((o[1] // dims[1]) + py * (o[2] // dims[2]))
    ranks = []
    for r in range(ncpu):
        lo, dim = boxes[r]
        own = [(lo[0] + i, lo[1] + j, lo[2] + k) for k in range(dim[2]) for j in range(dim[1]) for i in range(dim[0])]
        width = 2 if extra_layer else 1
        ghosts = set()
        for k in range(-width, dim[2] + width):
            for j in range(-width, dim[1] + width):
                for i in range(-width, dim[0] + width):
                    o = ((lo[0] + i) % no, (lo[1] + j) % no, (lo[2] + k) % no)
                    if owner_of(o) != r:
                        ghosts.add(o)
        ghosts = sorted(ghosts)
        ngridmax = len(own) + len(ghosts) + 11
        slots = rng.permutation(ngridmax)[:len(own) + len(ghosts)] + 1
        pos_slot = {}
        xg = np.zeros((3, ngridmax))
        for o, sl in zip(own + ghosts, slots):
            pos_slot[o] = int(sl)
            for d in range(3):
                xg[d, sl - 1] = (o[d] + 0.5) / no
        igrid = np.array([pos_slot[own[t]] for t in rng.permutation(len(own))], np.int32)
        ranks.append(dict(igrid=igrid, xg=xg, ngridmax=ngridmax, pos_slot=pos_slot, own=own, ghosts=ghosts, box=(lo, dim)))
    # communicators: for every (receiver r, owner c): the ghost octs of r owned by c, in one order shared by both ends
    for r in range(ncpu):
        ranks[r]["em"] = [[] for _ in range(ncpu)]
        ranks[r]["rc"] = [[] for _ in range(ncpu)]
    for r in range(ncpu):
        by_owner = {}
        for o in ranks[r]["ghosts"]:
            by_owner.setdefault(owner_of(o), []).append(o)
        for c, octs in by_owner.items():
            octs = [octs[t] for t in rng.permutation(len(octs))]
            ranks[r]["rc"][c] = [ranks[r]["pos_slot"][o] for o in octs]
            ranks[c]["em"][r] = [ranks[c]["pos_slot"][o] for o in octs]
    for R in ranks:
        R["em_n"] = np.array([len(x) for x in R["em"]], np.int32)
        R["rc_n"] = np.array([len(x) for x in R["rc"]], np.int32)
        R["em_ig"] = np.array([s for x in R["em"] for s in x] or [0], np.int32)
        R["rc_ig"] = np.array([s for x in R["rc"] for s in x] or [0], np.int32)
    return ranks, no


def plan(level, R, ncpu):
    L = _lib()
    box = (C.c_int * 8)()
    nem, nrc = int(R["em_n"].sum()), int(R["rc_n"].sum())
    act = np.zeros(len(R["igrid"]), np.int64)
    em = np.zeros(max(nem, 1), np.int64)
    cap = 8 * max(nrc, 1)
    rsrc = np.zeros(cap, np.int32)
    rorg = np.zeros(cap, np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.ramses_amd_halo_plan(level, len(R["igrid"]), vp(R["igrid"]), vp(R["xg"]), R["ngridmax"], ncpu,
                                vp(R["em_n"]), vp(R["em_ig"]), vp(R["rc_n"]), vp(R["rc_ig"]), box, vp(act), vp(em),
                                vp(rsrc), vp(rorg), cap)
    if rc != 0:
        return None, L.ramses_amd_last_error().decode()
    b = list(box)
    return dict(olo=b[0:3], odim=b[3:6], self_axes=b[6], nrc=b[7], act=act, em=em[:nem], rsrc=rsrc[:b[7]], rorg=rorg[:b[7]]), ""


def cell_value(x, y, z, v, n):
    return ((x % n) + 1000.0 * (y % n) + 1e6 * (z % n)) * 10.0 + v


class Brick:
    """numpy twin of the device brick: flat [nvar][pitch_var], offsets as the plan gives them"""

    def __init__(self, P, nvar, n):
        self.P, self.nvar, self.n = P, nvar, n
        self.nx, self.ny, self.nz = (2 * d for d in P["odim"])
        self.py = self.nx + 4
        self.pz = self.py * (self.ny + 4)
        self.pv = self.pz * (self.nz + 4)
        self.u = np.full((nvar, self.pv), np.nan)
        self.off = np.array([(i & 1) + self.py * ((i >> 1) & 1) + self.pz * (i >> 2) for i in range(8)])

    def fill_interior(self):
        lo = [2 * o for o in self.P["olo"]]
        k, j, i = np.meshgrid(np.arange(self.nz), np.arange(self.ny), np.arange(self.nx), indexing="ij")
        idx = (i + 2) + self.py * (j + 2) + self.pz * (k + 2)
        for v in range(self.nvar):
            self.u[v, idx] = cell_value(lo[0] + i, lo[1] + j, lo[2] + k, v, self.n)

    def pack(self):
        org = self.P["em"]
        buf = np.zeros((len(org), self.nvar, 8))
        for v in range(self.nvar):
            buf[:, v, :] = self.u[v][org[:, None] + self.off[None, :]]
        return buf

    def unpack(self, buf):
        src, org = self.P["rsrc"], self.P["rorg"]
        for v in range(self.nvar):
            self.u[v][org[:, None] + self.off[None, :]] = buf[src, v, :]

    def self_fill(self):
        n = [self.nx, self.ny, self.nz]
        u = self.u.reshape(self.nvar, self.nz + 4, self.ny + 4, self.nx + 4)
        for d in range(3):
            if not (self.P["self_axes"] >> d) & 1:
                continue
            sl = [slice(None)] * 3   # (x, y, z) extents
            for e in range(3):
                full = (not (self.P["self_axes"] >> e) & 1) or e < d
                sl[e] = slice(0, n[e] + 4) if full else slice(2, n[e] + 2)
            for hi in (0, 1):
                s, t = list(sl), list(sl)
                t[d] = slice(n[d] + 2, n[d] + 4) if hi else slice(0, 2)
                s[d] = slice(2, 4) if hi else slice(n[d], n[d] + 2)
                u[:, t[2], t[1], t[0]] = u[:, s[2], s[1], s[0]]

    def check(self):
        lo = [2 * o for o in self.P["olo"]]
        u = self.u.reshape(self.nvar, self.nz + 4, self.ny + 4, self.nx + 4)
        k, j, i = np.meshgrid(np.arange(-2, self.nz + 2), np.arange(-2, self.ny + 2), np.arange(-2, self.nx + 2), indexing="ij")
        for v in range(self.nvar):
            want = cell_value(lo[0] + i, lo[1] + j, lo[2] + k, v, self.n)
            assert np.array_equal(u[v], want), "ghost layer wrong for variable %d" % v


@pytest.mark.parametrize("level,pgrid", [(4, (2, 1, 1)), (4, (2, 2, 1)), (4, (2, 2, 2)), (3, (4, 1, 1)), (2, (2, 1, 1)),
                                          (3, (1, 2, 4)), (5, (2, 2, 2))])
@pytest.mark.parametrize("extra_layer", [False, True])
def test_plan_fills_every_ghost_cell(level, pgrid, extra_layer):
    if extra_layer and min(2 ** (level - 1) // p for p in pgrid if p > 1) < 2:
        pytest.skip("two ghost layers do not fit")
    ranks, no = make_ranks(level, pgrid, seed=level, extra_layer=extra_layer)
    ncpu = len(ranks)
    n = 2 * no
    nvar = 3
    plans = []
    for R in ranks:
        P, err = plan(level, R, ncpu)
        assert P is not None, err
        assert tuple(P["olo"]) == R["box"][0] and tuple(P["odim"]) == R["box"][1]
        plans.append(P)
    bricks = [Brick(P, nvar, n) for P in plans]
    for b in bricks:
        b.fill_interior()
    sent = [b.pack() for b in bricks]
    for r, b in enumerate(bricks):
        # message of owner c for receiver r = rows em_first[r] .. of c's pack, in list order = r's reception order
        recv = np.zeros((int(ranks[r]["rc_n"].sum()), nvar, 8))
        pos = 0
        for c in range(ncpu):
            cnt = int(ranks[r]["rc_n"][c])
            if cnt:
                first = int(ranks[c]["em_n"][:r].sum())
                recv[pos:pos + cnt] = sent[c][first:first + cnt]
            pos += cnt
        b.unpack(recv)
        b.self_fill()
        b.check()


def test_plan_refuses_domains_that_are_not_boxes():
    ranks, no = make_ranks(4, (2, 1, 1), seed=1)
    R = ranks[0]
    R2 = dict(R)
    R2["igrid"] = R["igrid"][:-1].copy()     # one oct missing: not a box
    P, err = plan(4, R2, 2)
    assert P is None and "do not fill a box" in err
    R3 = dict(R)
    R3["rc_n"] = R["rc_n"].copy()
    R3["rc_n"][1] -= 1                        # one ghost oct missing from the reception list
    P, err = plan(4, R3, 2)
    assert P is None and "ghost octs" in err


def _gloo_worker(rank, world, port, level, pgrid, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ranks, no = make_ranks(level, pgrid, seed=7)     # every process builds the same decomposition
        R = ranks[rank]
        P, err = plan(level, R, world)
        assert P is not None, err
        nvar = 5
        b = Brick(P, nvar, 2 * no)
        b.fill_interior()
        sendbuf = torch.from_numpy(b.pack().copy())
        recvbuf = torch.zeros((int(R["rc_n"].sum()), nvar, 8), dtype=torch.float64)
        ops, so, ro = [], 0, 0
        for c in range(world):
            ns, nr = int(R["em_n"][c]), int(R["rc_n"][c])
            if nr:
                ops.append(dist.P2POp(dist.irecv, recvbuf[ro:ro + nr], c))
            if ns:
                ops.append(dist.P2POp(dist.isend, sendbuf[so:so + ns], c))
            so += ns
            ro += nr
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        b.unpack(recvbuf.numpy())
        b.self_fill()
        b.check()
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pgrid", [(2, 1, 1)])
def test_two_processes_exchange_over_gloo(pgrid):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = pgrid[0] * pgrid[1] * pgrid[2]
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, 4, pgrid, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res
