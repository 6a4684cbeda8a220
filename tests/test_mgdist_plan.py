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
