"""The RCCL transport of the C ABI EXECUTES: on the one-GPU box a communicator of ONE rank is
brought up exactly as the Fortran shim / bench.py bring up theirs (ramses_amd_rccl_probe -> _unique_id -> _init), and the
grouped neighbour exchange, the pointer-per-message variant, the all-reduce and the AMR virtual-boundary exchange
(ramses_amd_amrres_halo_rccl: make_virtual_fine_dp amr/virtual_boundaries.f90:373-528 and make_virtual_reverse_dp :693-983 on
the resident cell vectors) run with the rank as its own peer -- a send to self matched by the receive from self inside one
ncclGroupStart/End: init, group semantics, stream ordering and the buffer offsets are the multi-rank code's.  The result
must equal (bit for bit) what the host-staged transport (halo_stage_out -> the caller's MPI -> halo_stage_in) produces, and
the numpy statement of the exchange."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl(gpu_lib):
    import torch
    torch.cuda.init()
    L = gpu_lib
    from ramses_amd._capi import check
    check(L.ramses_amd_rccl_probe())
    buf = (C.c_char * 128)()
    check(L.ramses_amd_rccl_unique_id(buf))
    check(L.ramses_amd_rccl_init(C.c_char_p(bytes(buf)), 1, 0))
    assert L.ramses_amd_rccl_ready() == 1
    yield L
    check(L.ramses_amd_rccl_finalize())
    assert L.ramses_amd_rccl_ready() == 0


def _i64(*v):
    return (C.c_int64 * len(v))(*v)


def _i32(*v):
    return (C.c_int * len(v))(*v)


def test_grouped_exchange_with_self_peer(rccl):
    """ramses_amd_rccl_exchange: two messages to self at different offsets inside ONE group, on a side stream."""
    import torch
    from ramses_amd._capi import check
    L = rccl
    rng = np.random.default_rng(7)
    send = torch.from_numpy(rng.standard_normal(5000)).cuda()
    recv = torch.zeros(6000, dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        # message 0: send[100:1100] -> recv[2000:3000]; message 1: send[3000:5000] -> recv[10:2010)... kept disjoint
        check(L.ramses_amd_rccl_exchange(2, _i32(0, 0), C.c_void_p(send.data_ptr()), _i64(100, 3000), _i64(1000, 2000),
                                         C.c_void_p(recv.data_ptr()), _i64(4000, 10), _i64(1000, 2000), C.c_void_p(st.cuda_stream)))
        # ordered after the exchange on the same stream: a kernel that reads what arrived
        twice = recv * 2.0
    st.synchronize()
    exp = np.zeros(6000)
    s = send.cpu().numpy()
    exp[4000:5000] = s[100:1100]
    exp[10:2010] = s[3000:5000]
    assert np.array_equal(recv.cpu().numpy(), exp)
    assert np.array_equal(twice.cpu().numpy(), exp * 2.0)
    # a message to self whose two ends disagree is refused, and the group is closed again (the next exchange works)
    rc = L.ramses_amd_rccl_exchange(1, _i32(0), C.c_void_p(send.data_ptr()), _i64(0), _i64(10), C.c_void_p(recv.data_ptr()),
                                    _i64(0), _i64(12), None)
    assert rc != 0 and b"message to self" in L.ramses_amd_last_error()
    rc = L.ramses_amd_rccl_exchange(1, _i32(3), C.c_void_p(send.data_ptr()), _i64(0), _i64(10), C.c_void_p(recv.data_ptr()),
                                    _i64(0), _i64(10), None)
    assert rc != 0 and b"bad peer rank" in L.ramses_amd_last_error()
    check(L.ramses_amd_rccl_exchange(1, _i32(0), C.c_void_p(send.data_ptr()), _i64(0), _i64(10), C.c_void_p(recv.data_ptr()),
                                     _i64(0), _i64(10), None))
    torch.cuda.synchronize()
    assert np.array_equal(recv[:10].cpu().numpy(), s[:10])
    # an empty exchange (a rank without neighbours on a level) is a no-op
    check(L.ramses_amd_rccl_exchange(0, None, None, None, None, None, None, None, None))


def test_pointer_per_message_and_allreduce(rccl):
    """ramses_amd_rccl_sendrecv (the Python mirror's tensors; bench.py --gpus N) and ramses_amd_rccl_allreduce."""
    import torch
    from ramses_amd._capi import check
    L = rccl
    a = torch.arange(0, 300, dtype=torch.float64, device="cuda")
    b = torch.arange(1000, 1200, dtype=torch.float64, device="cuda")
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    sp = (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    rp = (C.c_void_p * 2)(ra.data_ptr(), rb.data_ptr())
    cnt = _i64(300, 200)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(L.ramses_amd_rccl_sendrecv(2, sp, cnt, _i32(0, 0), 2, rp, cnt, _i32(0, 0), stream))
    torch.cuda.synchronize()
    assert torch.equal(ra, a) and torch.equal(rb, b)     # per peer, messages match in posting order
    for op in (0, 1, 2):                                 # sum / min / max over one rank: the identity, executed by RCCL
        v = torch.tensor([1.5, -2.25, 1e300, -0.0], dtype=torch.float64, device="cuda")
        w = v.clone()
        check(L.ramses_amd_rccl_allreduce(C.c_void_p(w.data_ptr()), 4, op, stream))
        torch.cuda.synchronize()
        assert np.array_equal(w.cpu().numpy().view(np.int64), v.cpu().numpy().view(np.int64))
    assert L.ramses_amd_rccl_allreduce(C.c_void_p(v.data_ptr()), 4, 7, stream) != 0


def _amr_state(nvar, ngridmax, seed):
    rng = np.random.default_rng(seed)
    ncoarse = 1
    ncell = ncoarse + 8 * ngridmax
    uold = rng.standard_normal((nvar, ncell))
    uold[0] = np.abs(uold[0]) + 0.5
    son = np.zeros(ncell, np.int32)
    nbor = np.ones((6, ngridmax), np.int32)
    father = np.ones(ngridmax, np.int32)
    return ncoarse, ncell, uold, son, nbor, father


def _cells(ncoarse, ngridmax, octs):
    """cell indices (0-based) of the 8 cells of each oct, [8, n]"""
    return ncoarse + np.arange(8)[:, None] * ngridmax + (np.asarray(octs)[None, :] - 1)


@pytest.mark.parametrize("nvar", [5, 6])
def test_amr_virtual_boundaries_over_rccl_equal_host_transport(rccl, nvar):
    """Forward (uold, all nvar at once) and reverse (unew, accumulated) exchange of an AMR level on the resident cell
    vectors, the rank being its own peer: RCCL result == host-staged result == numpy statement, bit for bit."""
    from ramses_amd._capi import check
    L = rccl
    ngridmax, lvl = 4096, 5
    ncoarse, ncell, uold0, son, nbor, father = _amr_state(nvar, ngridmax, 11 + nvar)
    rng = np.random.default_rng(3)
    perm = rng.permutation(ngridmax)[:1400] + 1
    em, rc_ = perm[:700].astype(np.int32), perm[700:].astype(np.int32)     # disjoint emission / reception octs
    allocts = np.arange(1, ngridmax + 1, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(transport):
        u = uold0.copy()
        check(L.ramses_amd_amrres_load(nvar, ngridmax, ncoarse, vp(u), vp(son), vp(nbor), vp(father)))
        check(L.ramses_amd_amrres_comm_set(lvl, 1, 1, _i32(len(em)), vp(em), _i32(len(rc_)), vp(rc_)))
        assert L.ramses_amd_amrres_comm_epoch(lvl) == 1

        def halo(d):
            if transport == "rccl":
                check(L.ramses_amd_amrres_halo_rccl(lvl, d, 1))
                return
            hs, hr = C.c_int64(0), C.c_int64(0)
            so, ro = _i64(0, 0), _i64(0, 0)
            check(L.ramses_amd_amrres_halo_stage_out(lvl, d, 1, C.byref(hs), C.byref(hr), so, ro))
            n = so[1] - so[0]
            assert n == ro[1] - ro[0] and n == 8 * nvar * 700
            C.memmove(hr.value, hs.value, 8 * n)        # what MPI_ISEND/IRECV between the two ends does
            check(L.ramses_amd_amrres_halo_stage_in(lvl, d))

        check(L.ramses_amd_amrres_set_unew(ngridmax, vp(allocts)))          # unew = uold
        halo(1)                                                             # unew[emission] += unew[reception]
        p = __import__("ramses_amd").make_params(nvar=nvar)
        check(L.ramses_amd_amrres_set_uold(C.byref(p), ngridmax, vp(allocts)))   # uold = unew
        halo(0)                                                             # uold[reception] = uold[emission]
        check(L.ramses_amd_amrres_sync_all(vp(u)))
        check(L.ramses_amd_amrres_invalidate())
        return u

    got_rccl, got_host = run("rccl"), run("host")
    exp = uold0.copy()
    ce, cr = _cells(ncoarse, ngridmax, em), _cells(ncoarse, ngridmax, rc_)
    exp[:, ce] = exp[:, ce] + exp[:, cr]
    exp[:, cr] = exp[:, ce]
    assert np.array_equal(got_host.view(np.int64), exp.view(np.int64))
    assert np.array_equal(got_rccl.view(np.int64), got_host.view(np.int64))


@pytest.mark.parametrize("n", [64, 128])
def test_distributed_multigrid_in_library_rccl_leg_equals_the_callback_transport(rccl, monkeypatch, n):
    """What `bench.py --gpus N`'s V-cycle leg and the Fortran drop-in on N GPUs execute first:
    ramses_amd_mgdist_create with transport == NULL, i.e. the library's own RCCL communicator inside the distributed
    multigrid (csrc/mg_dist.hip: the deep-halo exchange through ramses_amd_rccl_exchange, the all-reduce of the residual
    norm, the all-gather that assembles the replicated coarse levels).  On the one-GPU box the communicator has ONE rank;
    RAMSES_AMD_MGDIST_RCCL_SELF=1 sends the rank's periodic wrap-around messages, the all-reduce and the all-gather
    through RCCL all the same (26 messages to self per exchange inside one group).  phi, f, the V-cycle count and the
    error must equal the solve through the three host callbacks bit for bit."""
    import torch
    from ramses_amd._capi import check
    from ramses_amd.poisson_parallel import PoissonDecomposition
    L = rccl
    rng = np.random.default_rng(5)
    rho = 1.0 + 0.5 * rng.random((n, n, n))
    rho[n // 8:n // 3, n // 4:n // 2, -n // 10:] += 15.0          # straddles the periodic boundary
    rho_tot = float(rho.mean())
    level = int(np.log2(n))
    # the callback transport (what the one-GPU tests and the MPI shim use)
    pd = PoissonDecomposition((1, 1, 1), 0, n, boxlen=1.0, epsilon=1e-6)
    pd.rho.copy_(torch.from_numpy(rho).cuda())
    it0, err0 = pd.multigrid_fine(rho_tot)
    pd.force_fine()
    torch.cuda.synchronize()
    phi0, f0, ex0 = pd.phi_interior().cpu().numpy(), pd.f.cpu().numpy(), pd.exchanges
    # the in-library RCCL leg
    monkeypatch.setenv("RAMSES_AMD_MGDIST_RCCL_SELF", "1")
    ctx = C.c_void_p()
    check(L.ramses_amd_mgdist_create(level, _i32(1, 1, 1), 0, None, None, C.byref(ctx)))
    try:
        d_rho = torch.from_numpy(rho).cuda()
        phi = torch.zeros(n, n, n, dtype=torch.float64, device="cuda")
        f = torch.zeros(3, n, n, n, dtype=torch.float64, device="cuda")
        it, err = C.c_int(), C.c_double()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(L.ramses_amd_mgdist_solve(ctx, C.c_void_p(d_rho.data_ptr()), rho_tot, pd.fourpi, 1e-6, C.byref(it), C.byref(err), st))
        check(L.ramses_amd_mgdist_get_phi(ctx, C.c_void_p(phi.data_ptr()), st))
        check(L.ramses_amd_mgdist_force(ctx, C.c_void_p(f.data_ptr()), st))
        torch.cuda.synchronize()
        nex = C.c_int64()
        check(L.ramses_amd_mgdist_info(ctx, None, None, None, None, None, C.byref(nex)))
    finally:
        check(L.ramses_amd_mgdist_destroy(ctx))
    assert it.value == it0 and err.value == err0
    assert nex.value >= 10 and ex0 >= 10                    # both legs exchanged halos (26 regions each, all to self)
    assert np.array_equal(phi.cpu().numpy(), phi0)
    assert np.array_equal(f.cpu().numpy(), f0)
