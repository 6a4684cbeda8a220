"""CPU test of the multi-rank protocol of the distributed multigrid driver (csrc/mg_dist.hip): the deep-halo plan the
C library builds for every rank (ramses_amd_mgdist_plan: host only) is played through with numpy -- every rank packs
its 26 send regions, every message reaches its peer at the offset the RECEIVER expects, every rank unpacks -- and the
ghost layers must then hold the periodic continuation of the global field.  Rank grids of 2, 4, 8 and 16 ranks in a
cubic box (bricks that are not cubes), brick <-> rank maps that are not the identity (the reference numbers its
domains along the Hilbert curve), a brick that wraps onto itself along the uncut axes."""
import ctypes as C

import numpy as np
import pytest


def _plan(L, pgrid, rank, rob, dims, ng):
    I, I64 = C.c_int, C.c_int64
    sb, rb = (I * 156)(), (I * 156)()
    so, ro = (I64 * 26)(), (I64 * 26)()
    npeer, total = I(), I64()
    peer = (I * 26)()
    sso, ssc, sro, src = (I64 * 26)(), (I64 * 26)(), (I64 * 26)(), (I64 * 26)()
    rc = L.ramses_amd_mgdist_plan((I * 3)(*pgrid), rank, (I * len(rob))(*rob) if rob is not None else None, (I * 3)(*dims), ng,
                                  sb, so, rb, ro, C.byref(npeer), peer, sso, ssc, sro, src, C.byref(total))
    assert rc == 0, L.ramses_amd_last_error()
    n = npeer.value
    return dict(send_boxes=np.array(sb).reshape(26, 6), send_offs=np.array(so), recv_boxes=np.array(rb).reshape(26, 6),
                recv_offs=np.array(ro), peers=list(peer[:n]), send_off=list(sso[:n]), send_cnt=list(ssc[:n]),
                recv_off=list(sro[:n]), recv_cnt=list(src[:n]), total=total.value)


@pytest.mark.parametrize("pgrid,permute", [((1, 1, 2), False), ((1, 2, 2), True), ((2, 2, 2), True), ((2, 1, 1), True),
                                          ((4, 2, 2), True), ((1, 1, 1), False)])
def test_the_library_s_halo_plan_fills_every_ghost_layer(pgrid, permute):
    from ramses_amd import _capi
    L = _capi.lib()
    N, ng = 16, 3
    world = pgrid[0] * pgrid[1] * pgrid[2]
    dims = tuple(N // p for p in pgrid)
    rng = np.random.default_rng(world)
    rob = list(rng.permutation(world)) if permute else None          # rank of brick b
    rank_of = (lambda b: int(rob[b])) if permute else (lambda b: b)
    G = rng.normal(size=(N, N, N))                                    # [z][y][x]
    plans, bricks, sendbuf, recvbuf = {}, {}, {}, {}
    nx, ny, nz = dims
    for b in range(world):
        r = rank_of(b)
        c = (b % pgrid[0], (b // pgrid[0]) % pgrid[1], b // (pgrid[0] * pgrid[1]))
        plans[r] = _plan(L, pgrid, r, rob, dims, ng)
        u = np.full((nz + 2 * ng, ny + 2 * ng, nx + 2 * ng), np.nan)
        u[ng:ng + nz, ng:ng + ny, ng:ng + nx] = G[c[2] * nz:(c[2] + 1) * nz, c[1] * ny:(c[1] + 1) * ny, c[0] * nx:(c[0] + 1) * nx]
        bricks[r] = (u, c)
    # pack
    for r, P in plans.items():
        u = bricks[r][0]
        buf = np.full(P["total"], np.nan)
        for (ox, oy, oz, ex, ey, ez), off in zip(P["send_boxes"], P["send_offs"]):
            buf[off:off + ex * ey * ez] = u[oz:oz + ez, oy:oy + ey, ox:ox + ex].reshape(-1)
        assert not np.isnan(buf).any()
        sendbuf[r] = buf
        recvbuf[r] = np.full(P["total"], np.nan)
    # one message per peer: what r sends to q lands where q expects r's message
    for r, P in plans.items():
        assert len(set(P["peers"])) == len(P["peers"])                # ONE message per peer
        for q, off, cnt in zip(P["peers"], P["send_off"], P["send_cnt"]):
            Q = plans[q]
            i = Q["peers"].index(r)
            assert Q["recv_cnt"][i] == cnt
            recvbuf[q][Q["recv_off"][i]:Q["recv_off"][i] + cnt] = sendbuf[r][off:off + cnt]
    # unpack and compare with the periodic continuation of the global field
    for r, P in plans.items():
        u, c = bricks[r]
        assert not np.isnan(recvbuf[r]).any()
        for (ox, oy, oz, ex, ey, ez), off in zip(P["recv_boxes"], P["recv_offs"]):
            u[oz:oz + ez, oy:oy + ey, ox:ox + ex] = recvbuf[r][off:off + ex * ey * ez].reshape(ez, ey, ex)
        idx = lambda cc, n: (np.arange(cc * n - ng, (cc + 1) * n + ng)) % N      # noqa: E731
        want = G[idx(c[2], nz)][:, idx(c[1], ny)][:, :, idx(c[0], nx)]
        assert np.array_equal(u, want), (pgrid, r)


def test_plan_rejects_a_rank_without_a_brick():
    from ramses_amd import _capi
    L = _capi.lib()
    I, I64 = C.c_int, C.c_int64
    args = [(I * 156)(), (I64 * 26)(), (I * 156)(), (I64 * 26)(), C.byref(I()), (I * 26)(), (I64 * 26)(), (I64 * 26)(), (I64 * 26)(),
            (I64 * 26)(), C.byref(I64())]
    assert L.ramses_amd_mgdist_plan((I * 3)(2, 1, 1), 5, None, (I * 3)(8, 16, 16), 3, *args) != 0


def _two_process_worker(rank, world, pgrid, port, ret):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ramses_amd import _capi
        from ramses_amd.poisson_parallel import _Callbacks
        from ramses_amd.transport import DistTransport
        L = _capi.lib()
        N, ng = 16, 5
        dims = tuple(N // p for p in pgrid)
        nx, ny, nz = dims
        G = np.random.default_rng(4).normal(size=(N, N, N))
        c = (rank % pgrid[0], (rank // pgrid[0]) % pgrid[1], rank // (pgrid[0] * pgrid[1]))
        P = _plan(L, pgrid, rank, None, dims, ng)
        u = np.full((nz + 2 * ng, ny + 2 * ng, nx + 2 * ng), np.nan)
        u[ng:ng + nz, ng:ng + ny, ng:ng + nx] = G[c[2] * nz:(c[2] + 1) * nz, c[1] * ny:(c[1] + 1) * ny, c[0] * nx:(c[0] + 1) * nx]
        send = np.full(P["total"], np.nan)
        recv = np.full(P["total"], np.nan)
        for (ox, oy, oz, ex, ey, ez), off in zip(P["send_boxes"], P["send_offs"]):
            send[off:off + ex * ey * ez] = u[oz:oz + ez, oy:oy + ey, ox:ox + ex].reshape(-1)
        # what csrc/mg_dist.hip does with a callback transport: self messages copied, the others through the callback
        cb = _Callbacks(DistTransport(), "cpu")
        DP = C.POINTER(C.c_double)
        peers, so, sc, ro, rcn = [], [], [], [], []
        for q, a, n, b, m in zip(P["peers"], P["send_off"], P["send_cnt"], P["recv_off"], P["recv_cnt"]):
            if q == rank:
                recv[b:b + m] = send[a:a + n]
            else:
                peers.append(q); so.append(a); sc.append(n); ro.append(b); rcn.append(m)
        k = len(peers)
        rc = cb.table.exchange(None, k, (C.c_int * k)(*peers), send.ctypes.data_as(DP), (C.c_int64 * k)(*so), (C.c_int64 * k)(*sc),
                               recv.ctypes.data_as(DP), (C.c_int64 * k)(*ro), (C.c_int64 * k)(*rcn))
        ok = rc == 0 and cb.error is None
        for (ox, oy, oz, ex, ey, ez), off in zip(P["recv_boxes"], P["recv_offs"]):
            u[oz:oz + ez, oy:oy + ey, ox:ox + ex] = recv[off:off + ex * ey * ez].reshape(ez, ey, ex)
        idx = lambda cc, n: (np.arange(cc * n - ng, (cc + 1) * n + ng)) % N      # noqa: E731
        ok = ok and np.array_equal(u, G[idx(c[2], nz)][:, idx(c[1], ny)][:, :, idx(c[0], nx)])
        # the replicated level's all-gather and the norm's all-reduce
        mine = np.full(6, float(rank + 1))
        parts = np.zeros(6 * world)
        ok = ok and cb.table.allgather(None, mine.ctypes.data_as(DP), 6, parts.ctypes.data_as(DP)) == 0
        ok = ok and np.array_equal(parts.reshape(world, 6), np.arange(1, world + 1)[:, None] * np.ones(6))
        val = C.c_double(float(rank + 1))
        ok = ok and cb.table.allreduce_sum(None, C.byref(val)) == 0 and val.value == world * (world + 1) / 2
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pgrid", [(1, 1, 2), (2, 1, 1)])
def test_callback_transport_between_two_processes(pgrid):
    """World size 2 on CPU (gloo): the three callbacks of struct ramses_amd_mg_transport as ramses_amd/poisson_parallel.py
    hands them to the library, driven the way csrc/mg_dist.hip drives them (one message per peer at the library's offsets
    into its host buffers, the all-gather of a replicated level, the sum of a residual norm), on the plan the library built."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_two_process_worker, args=(2, pgrid, port, ret), nprocs=2, join=True)
    assert len(ret) == 2 and all(ret.values()), dict(ret)
