"""Message transport under the halo exchange and the multigrid driver.

DistTransport    torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
                 box, "gloo" in the CPU protocol tests): grouped send/recv, the
                 scalar reductions and the coarse-level all-gather.
LocalWorld       several virtual ranks inside ONE process (one Python thread
                 each, mailboxes instead of a network).  Used by the tests to
                 run the multi-rank code paths on a single GPU and compare them
                 bit for bit with the single-brick result.

The reference's counterpart is the MPI layer of amr/virtual_boundaries.f90
(isend/irecv per peer, MPI_ALLREDUCE of scalars).
"""
import threading

import torch
import torch.distributed as dist


class DistTransport:
    """backend "nccl" (= RCCL): device tensors go straight to the collective.
    backend "gloo": device tensors are staged through host memory, which lets
    the multi-process paths be exercised where RCCL cannot run (CPU container,
    several ranks sharing one GPU)."""

    def __init__(self, group=None, staged=None):
        """group: a process group spanning all ranks (default: the world group); staged: force
        the host-staged path (default: staged iff the group's backend is gloo)."""
        self.group = group
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if staged is None:
            staged = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self.staged = bool(staged)

    def sendrecv(self, sends, recvs):
        """sends/recvs: lists of (tensor, peer).  One grouped launch
        (ncclGroupStart/End); per peer, messages match in posting order."""
        if not sends and not recvs:
            return
        if self.staged and (sends + recvs)[0][0].is_cuda:
            hs = [(t.cpu(), p) for t, p in sends]
            hr = [(torch.empty(t.shape, dtype=t.dtype), p) for t, p in recvs]
            g = self.group
            ops = [dist.P2POp(dist.isend, t, p, g) for t, p in hs] + [dist.P2POp(dist.irecv, t, p, g) for t, p in hr]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for (t, _), (h, _) in zip(recvs, hr):
                t.copy_(h)
            return
        g = self.group
        ops = [dist.P2POp(dist.isend, t, p, g) for t, p in sends] + [dist.P2POp(dist.irecv, t, p, g) for t, p in recvs]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def allreduce(self, value, device, op="sum"):
        t = torch.tensor([value], dtype=torch.float64, device="cpu" if self.staged else device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN if op == "min" else (dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM),
                            group=self.group)
        return float(t.item())

    def allgather(self, t):
        """t: contiguous tensor; returns a tensor [world, *t.shape] in rank order."""
        if self.world == 1:
            return t.contiguous().unsqueeze(0).clone()
        src = t.contiguous().cpu() if self.staged else t.contiguous()
        # concatenated along dim 0 (the layout every backend accepts), viewed per rank
        out = torch.empty((self.world * src.shape[0],) + tuple(src.shape[1:]), dtype=t.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=self.group)
        return out.view((self.world,) + tuple(src.shape)).to(t.device)

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)


class RcclTransport(DistTransport):
    """Neighbour send/recv through the C ABI of libramses_amd.so (ramses_amd_rccl_sendrecv: one grouped
    ncclSend/ncclRecv on the current HIP stream) -- the same entry the Fortran shim
    ramses_amd/patch/virtual_boundaries.f90 reaches through ramses_amd_mpires_halo_forward.  The 128-byte
    unique id travels over the torch.distributed group that launched the ranks; the scalar reductions,
    the coarse-level all-gather and the barrier stay with torch.distributed."""

    def __init__(self, group=None):
        super().__init__(group=group, staged=False)
        import ctypes as C
        from ._capi import check, lib
        self._C, self._check, self._lib = C, check, lib()
        on_gpu = dist.get_backend(group) == "nccl"
        # local probe first (dlopen + symbols), agreed on by every rank BEFORE the collective init
        idt = torch.zeros(128, dtype=torch.uint8)
        ok = 1 if self._lib.ramses_amd_rccl_probe() == 0 else 0
        if ok and self.rank == 0:
            buf = (C.c_char * 128)()
            if self._lib.ramses_amd_rccl_unique_id(buf) == 0:
                idt = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
            else:
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        flag = flag.cuda() if on_gpu else flag
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            raise RuntimeError("RCCL cannot be brought up on every rank: " +
                               (self._lib.ramses_amd_last_error() or b"").decode())
        dev = idt.cuda() if on_gpu else idt
        dist.broadcast(dev, src=0, group=group)
        raw = bytes(dev.cpu().numpy().tobytes())
        check(self._lib.ramses_amd_rccl_init(C.c_char_p(raw), self.world, self.rank))

    def sendrecv(self, sends, recvs):
        if not sends and not recvs:
            return
        C = self._C
        for t, _ in sends + recvs:
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        ns, nr = len(sends), len(recvs)
        sp = (C.c_void_p * max(ns, 1))(*[t.data_ptr() for t, _ in sends])
        sc = (C.c_int64 * max(ns, 1))(*[t.numel() for t, _ in sends])
        sq = (C.c_int * max(ns, 1))(*[p for _, p in sends])
        rp = (C.c_void_p * max(nr, 1))(*[t.data_ptr() for t, _ in recvs])
        rc = (C.c_int64 * max(nr, 1))(*[t.numel() for t, _ in recvs])
        rq = (C.c_int * max(nr, 1))(*[p for _, p in recvs])
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self._lib.ramses_amd_rccl_sendrecv(ns, sp, sc, sq, nr, rp, rc, rq, stream))


class LocalWorld:
    """world virtual ranks in this process: run(fn) calls fn(transport) on one
    thread per rank and returns the list of results (re-raises the first error)."""

    def __init__(self, world):
        self.world = world
        self._cv = threading.Condition()
        self._mail = {}                      # (src, dst) -> list of tensors, FIFO
        self._coll = {}                      # collective round -> contributions
        self._round = [0] * world

    def transport(self, rank):
        return _LocalTransport(self, rank)

    def run(self, fn):
        res, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                res[r] = fn(self.transport(r))
            except BaseException as e:          # noqa: BLE001
                err[r] = e
                with self._cv:
                    self._cv.notify_all()

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return res


class _LocalTransport:
    def __init__(self, world, rank):
        self.w = world
        self.rank = rank
        self.world = world.world

    def sendrecv(self, sends, recvs):
        w = self.w
        if sends and sends[0][0].is_cuda:
            torch.cuda.current_stream().synchronize()
        with w._cv:
            for t, p in sends:
                w._mail.setdefault((self.rank, p), []).append(t.clone())
            w._cv.notify_all()
        for t, p in recvs:
            with w._cv:
                while not w._mail.get((p, self.rank)):
                    w._cv.wait(timeout=60)
                    if not w._mail.get((p, self.rank)) and not any(th.is_alive() for th in threading.enumerate() if th is not threading.current_thread() and th is not threading.main_thread()):
                        raise RuntimeError("LocalWorld: peer died")
                m = w._mail[(p, self.rank)].pop(0)
            t.copy_(m)
        if recvs and recvs[0][0].is_cuda:
            torch.cuda.current_stream().synchronize()

    def _collect(self, item):
        w = self.w
        with w._cv:
            rnd = w._round[self.rank]
            w._round[self.rank] += 1
            slot = w._coll.setdefault(rnd, {})
            slot[self.rank] = item
            w._cv.notify_all()
            while len(w._coll[rnd]) < self.world:
                w._cv.wait(timeout=60)
            return [w._coll[rnd][r] for r in range(self.world)]

    def allreduce(self, value, device, op="sum"):
        vals = self._collect(float(value))
        if op == "min":
            return min(vals)
        if op == "max":
            return max(vals)
        s = 0.0
        for v in vals:                       # rank order: deterministic
            s += v
        return s

    def allgather(self, t):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        parts = self._collect(t.clone())
        return torch.stack(parts, 0)

    def barrier(self):
        self._collect(None)
