"""Row-slab decomposition of ONE large frame over several GPUs (BASELINE config C5).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Rank r owns the contiguous rows [lo_r, hi_r) of the frame and holds them plus
`halo` extra rows above and below in its own solver context.  The Jacobi update at row y needs
u, v at rows y-1 and y+1 only, so after k sweeps the rows closer than k to a slab's artificial edge
are stale; the driver therefore runs the sweeps in chunks of at most `halo`, and after every chunk
neighbouring ranks swap `halo` rows of u and v (send/recv point to point, no collective).  The
derivative pass needs frame rows y-1..y+1, which the overlapping upload already provides.  The
result is bit-identical to solving the whole frame on one GPU.

The module contains no arithmetic: the per-slab solver is any object with the small interface of
`HSFlowSlabBackend` below (the product one wraps the HIP context; the CPU tests inject one that
wraps the oracle, which is test infrastructure and never imported from here).
"""
import numpy as np


def slab_rows(height, world, rank):
    """Rows [lo, hi) owned by `rank`: contiguous, sizes differ by at most one."""
    base, extra = divmod(height, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def slab_extent(height, world, rank, halo):
    """(lo, hi, top, bot): owned rows and the halo depths actually available above / below."""
    lo, hi = slab_rows(height, world, rank)
    top = min(halo, lo)
    bot = min(halo, height - hi)
    return lo, hi, top, bot


def chunks(total, size):
    """Sweep counts of the successive chunks: as many `size` as fit, then the remainder."""
    out = [size] * (total // size)
    if total % size:
        out.append(total % size)
    return out


def _make(make_backend, width, local_height, first_row):
    """make_backend(width, local_height[, first_row]): the frame row of the slab's row 0 is passed where the
    factory takes it (the HIP backend needs its parity, see hsflow_set_row_origin)."""
    import inspect
    try:
        n = len(inspect.signature(make_backend).parameters)
    except (TypeError, ValueError):
        n = 2
    return make_backend(width, local_height, first_row) if n >= 3 else make_backend(width, local_height)


class HSFlowSlabBackend(object):
    """Per-rank solver over the local rows (product implementation: the HIP context).

    stream = None: the context works on a private stream and every hand-over to torch.distributed is
    bracketed by host synchronisation (simple, used by the thread-based tests).
    stream = a HIP stream handle that is ALSO torch's current stream (e.g.
    `torch.cuda.current_stream().cuda_stream` after `torch.cuda.set_stream`): sweeps are only enqueued
    (`hsflow_solve_async`), the halo copies ride the same stream, and RCCL orders itself against that
    stream -- the host never waits between chunks, so the GPU runs the chunks back to back."""

    def __init__(self, hs, width, local_height, device, stream=None, torch_stream=None, first_row=0):
        import torch
        self.torch = torch
        self.hs = hs
        # torch_stream: a torch.cuda.Stream the context shares; `activate()` then makes it torch's current
        # stream, so that several backends of one rank (OverlappedSlabSolver) each order their own work
        self.tstream = torch_stream
        if torch_stream is not None:
            stream = torch_stream.cuda_stream
        self.ctx = hs.HSFlow(width, local_height, 1, device=device, stream=stream, own_stream=stream is None)
        self.ctx.set_row_origin(first_row)  # the slab's rows keep the frame's checkerboard phase
        self.width, self.height, self.device = width, local_height, device
        self.enqueue_only = stream is not None

    def activate(self):
        """Context manager under which this backend's calls and the P2P calls for it must be made."""
        import contextlib
        return self.torch.cuda.stream(self.tstream) if self.tstream is not None else contextlib.nullcontext()

    def record(self):
        """A marker after everything enqueued so far (None when every call is synchronous anyway)."""
        if self.tstream is None:
            return None
        ev = self.torch.cuda.Event()
        ev.record(self.tstream)
        return ev

    def wait_event(self, ev):
        if ev is not None and self.tstream is not None:
            self.tstream.wait_event(ev)

    def set_frames(self, prev, curr):
        self.ctx.set_frames(prev, curr)

    def sweep(self, n, lam, first, from_zero=None, eps=None):
        """n sweeps.  first: the first chunk of a solve (derivative pass); from_zero: start from zero flow (default: first).
        eps: run them as ITER|EPS witness launches -- enqueued only, the early-stop check stays owed until `verdict()`."""
        from_zero = first if from_zero is None else from_zero
        # derivatives are computed by the first chunk only; later chunks continue from the flow
        kw = dict(lam=lam, max_iter=n, term_type=self.hs.TERM_ITER, use_previous=not from_zero, reuse_derivatives=not first)
        if eps is not None:
            kw.update(term_type=self.hs.TERM_ITER | self.hs.TERM_EPS, epsilon=eps)
            self.ctx.solve_async(**kw)
        elif self.enqueue_only:
            self.ctx.solve_async(**kw)
        else:
            self.ctx.solve(**kw)

    def set_eps_rows(self, first_row, rows):
        """Only these local rows count for Eps / the witness: the slab's OWNED rows (hsflow_set_eps_rows)."""
        self.ctx.set_eps_rows(first_row, rows)

    def verdict(self):
        """Did the witness launches of the last `sweep(..., eps=...)` prove "Eps stayed >= eps in my rows"?  Nothing is re-run."""
        return self.ctx.take_verdict()

    def probe(self, n, lam, first, from_zero=None):
        """n sweeps, the Eps of each (over the rows of set_eps_rows) as an fp32 array; synchronous."""
        from_zero = first if from_zero is None else from_zero
        return self.ctx.solve_probe(lam=lam, max_iter=n, use_previous=not from_zero, reuse_derivatives=not first)

    def solve_whole(self, lam, iters, eps, use_previous):
        """The whole solve on this one context (world size 1): its own stop rule.  Returns (sweeps done, measured?)."""
        kw = dict(lam=lam, max_iter=iters, use_previous=use_previous)
        kw.update(dict(term_type=self.hs.TERM_ITER) if eps is None else dict(term_type=self.hs.TERM_ITER | self.hs.TERM_EPS, epsilon=eps))
        i = self.ctx.solve(**kw)
        return i["iterations_done"], bool(i["eps_rerun"])

    def save(self):
        """A copy of the slab's flow, halos included (device tensors)."""
        u, v = self.new_rows(self.height)
        self.get_rows(0, u, v)
        return u, v

    def restore(self, saved):
        self.put_rows(0, saved[0], saved[1])

    def new_rows(self, nrows):
        t = self.torch
        dev = "cuda:%d" % self.device
        return (t.empty((nrows, self.width), dtype=t.float32, device=dev),
                t.empty((nrows, self.width), dtype=t.float32, device=dev))

    def get_rows(self, row0, u, v):
        self.ctx.flow_rows_to(u, v, row0, u.shape[0])
        if not self.enqueue_only:
            self.ctx.synchronize()

    def put_rows(self, row0, u, v):
        if not self.enqueue_only:
            self.torch.cuda.synchronize(self.device)  # the received rows were written on torch's stream
        self.ctx.set_flow_rows_from(u, v, row0, u.shape[0])
        if not self.enqueue_only:
            self.ctx.synchronize()

    def flow(self):
        return self.ctx.flow()

    def close(self):
        self.ctx.close()


class SlabSolver(object):
    """Drives one rank's slab: chunked sweeps + halo exchange with the ranks above and below."""

    def __init__(self, dist, rank, world, width, height, halo, make_backend, stage_on_host=False):
        # stage_on_host: exchange through host copies of the halo rows (gloo rehearsals of the GPU
        # path on a box whose ranks share one card; RCCL sends device buffers directly)
        if halo < 1:
            raise ValueError("halo must be >= 1")
        if world > 1 and height // world < halo:
            raise ValueError("slabs of %d rows are thinner than the halo (%d)" % (height // world, halo))
        self.dist, self.rank, self.world = dist, rank, world
        self.width, self.height, self.halo = width, height, halo
        self.lo, self.hi, self.top, self.bot = slab_extent(height, world, rank, halo)
        self.row0 = self.lo - self.top                    # first frame row held locally
        self.local_height = (self.hi + self.bot) - self.row0
        self.backend = _make(make_backend, width, self.local_height, self.row0)
        if hasattr(self.backend, "set_eps_rows"):  # Eps of the frame = the maximum over the ranks of the Eps of their OWNED rows
            self.backend.set_eps_rows(self.top, self.hi - self.lo)
        self.stage_on_host = stage_on_host
        self._bufs = None
        self._solved = False
        self.iterations_done = 0
        self.eps_measured = False

    def local_frame_rows(self):
        """Frame rows [row0, row1) this rank must upload (owned rows + halos)."""
        return self.row0, self.row0 + self.local_height

    def set_frames(self, prev_local, curr_local):
        if prev_local.shape != (self.local_height, self.width):
            raise ValueError("local frames must have shape (%d, %d)" % (self.local_height, self.width))
        self.backend.set_frames(prev_local, curr_local)
        self._solved = False

    def _all_max(self, values):
        """Element-wise maximum over the ranks of a small fp32 vector (the one collective of the EPS path)."""
        import torch
        d = self.dist
        dev = "cuda" if d.get_backend() == "nccl" else "cpu"
        t = torch.as_tensor(np.asarray(values, dtype=np.float32), device=dev)
        d.all_reduce(t, op=d.ReduceOp.MAX)
        return t.cpu().numpy()

    def _exchange(self):
        """Swap `halo` rows of u, v with the neighbouring ranks (2 sends + 2 recvs per neighbour)."""
        d, b, h = self.dist, self.backend, self.halo
        if self._bufs is None:
            self._bufs = {k: b.new_rows(h) for k in ("send_up", "recv_up", "send_dn", "recv_dn")}
        ops = []
        up, dn = self.rank - 1, self.rank + 1
        host = self.stage_on_host
        wire = {}

        def out(name):  # tensor that goes on the wire for send buffer `name`
            t = self._bufs[name]
            wire[name] = (t[0].cpu(), t[1].cpu()) if host else t
            return wire[name]

        def inn(name):  # tensor the wire writes into for recv buffer `name`
            t = self._bufs[name]
            wire[name] = (t[0].cpu(), t[1].cpu()) if host else t
            return wire[name]

        if up >= 0:      # my first owned rows become the bottom halo of the rank above
            b.get_rows(self.top, *self._bufs["send_up"])
            su, sv = out("send_up")
            ru, rv = inn("recv_up")
            ops += [d.P2POp(d.isend, su, up), d.P2POp(d.isend, sv, up),
                    d.P2POp(d.irecv, ru, up), d.P2POp(d.irecv, rv, up)]
        if dn < self.world:  # my last owned rows become the top halo of the rank below
            b.get_rows(self.top + (self.hi - self.lo) - h, *self._bufs["send_dn"])
            su, sv = out("send_dn")
            ru, rv = inn("recv_dn")
            ops += [d.P2POp(d.isend, su, dn), d.P2POp(d.isend, sv, dn),
                    d.P2POp(d.irecv, ru, dn), d.P2POp(d.irecv, rv, dn)]
        if ops:
            for w in d.batch_isend_irecv(ops):
                w.wait()
        if host:
            for name in ("recv_up", "recv_dn"):
                if name in wire:
                    self._bufs[name][0].copy_(wire[name][0])
                    self._bufs[name][1].copy_(wire[name][1])
        if up >= 0:
            b.put_rows(0, *self._bufs["recv_up"])                               # rows [lo-halo, lo)
        if dn < self.world:
            b.put_rows(self.top + (self.hi - self.lo), *self._bufs["recv_dn"])  # rows [hi, hi+halo)

    def solve(self, lam, iters, eps=None, use_previous=False):
        """`iters` Jacobi sweeps on the whole frame, from zero flow or (use_previous) from the flow of the last solve;
        returns the number of halo exchanges.  eps: ITER|EPS, the reference's own criteria (OpticalFlowOpenCV.cpp:29) --
        the frame's Eps is the maximum over the ranks of the Eps of their owned rows (cv210.dll@0x1012ed2f-0x1012eda5), so
          * a chunk that ANY rank's witness vouches for (all_reduce MAX of the verdicts) holds no stop;
          * when none does, every rank replays the solve to that chunk's start (it is deterministic) and from there on
            each chunk is measured: per-sweep Eps per rank, all_reduce MAX, first sweep below eps; the chunk is then
            repeated up to that sweep from a copy of its starting flow.
        `iterations_done` / `eps_measured` afterwards.  Bit-identical to the one-context solve, stopping sweep included."""
        b = self.backend
        self.eps_measured = False
        if self.world == 1:
            self.iterations_done, self.eps_measured = b.solve_whole(lam, iters, eps, use_previous) if eps is not None or use_previous else (None, False)
            if self.iterations_done is None:
                b.sweep(iters, lam, first=True)
                self.iterations_done = iters
            self._solved = True
            return 0
        n_ex = 0
        if use_previous and self._solved:   # the halos the last solve left are stale
            self._exchange()
        start = b.save() if (use_previous and eps is not None) else None
        plan = chunks(iters, self.halo)

        def run(i, n, witness):
            b.sweep(n, lam, first=(i == 0), from_zero=(i == 0 and not use_previous), eps=eps if witness else None)

        done, i, measure = 0, 0, False
        while i < len(plan):
            n = plan[i]
            if eps is None:
                run(i, n, False)
            elif not measure:
                try:
                    run(i, n, True)
                    mine = 1.0 if b.verdict() else 0.0
                except Exception as e:  # noqa: BLE001 -- a launch plan that cannot run witness launches (a very thin slab): no voucher
                    if getattr(e, "status", None) != 1:   # HSFLOW_E_ARG
                        raise
                    mine = 0.0
                if self._all_max([mine])[0] < 0.5:   # nobody vouches for this chunk
                    measure = self.eps_measured = True
                    if start is not None:
                        b.restore(start)
                    for j in range(i):                                        # replay to its start
                        run(j, plan[j], False)
                        self._exchange()
                    continue
            else:
                saved = b.save()
                e = self._all_max(b.probe(n, lam, first=(i == 0), from_zero=(i == 0 and not use_previous)))
                hit = np.nonzero(e.astype(np.float64) < eps)[0]
                if len(hit):
                    k = int(hit[0]) + 1
                    if k < n:
                        b.restore(saved)
                        b.sweep(k, lam, first=False, from_zero=(i == 0 and not use_previous))
                    done += k
                    break
            done += n
            i += 1
            if i < len(plan):
                self._exchange()
                n_ex += 1
        self.iterations_done = done
        self._solved = True
        return n_ex

    def owned_flow(self):
        """(u, v) of the rows this rank owns, as host arrays of shape (hi - lo, width)."""
        u, v = self.backend.flow()
        u, v = np.asarray(u), np.asarray(v)
        return u[self.top:self.top + (self.hi - self.lo)], v[self.top:self.top + (self.hi - self.lo)]

    def close(self):
        self.backend.close()


class OverlappedSlabSolver(object):
    """Two sub-slabs per rank, worked alternately, so that the halo exchange with the neighbouring RANK
    runs under the sweeps of the other sub-slab ("overlapped with interior-row compute").

    Rank r owns the virtual slabs 2r (A, upper) and 2r+1 (B, lower) of a 2*world partition.  Per chunk:
        sweep(A); post A's top rows <-> rank r-1      (in flight while B is being swept)
        sweep(B); post B's bottom rows <-> rank r+1   (in flight while A's next chunk is swept)
        A's bottom rows <-> B's top rows locally (device copies ordered by stream events)
    and the rows received from a neighbouring rank are written into the halo right before that sub-slab's
    next sweep.  Same arithmetic as SlabSolver with 2*world ranks: bit-identical to the whole-frame solve.
    Backends need the interface of HSFlowSlabBackend incl. activate() / record() / wait_event()."""

    def __init__(self, dist, rank, world, width, height, halo, make_backend, stage_on_host=False):
        if halo < 1:
            raise ValueError("halo must be >= 1")
        vworld = 2 * world
        if height // vworld < halo:
            raise ValueError("sub-slabs of %d rows are thinner than the halo (%d)" % (height // vworld, halo))
        self.dist, self.rank, self.world = dist, rank, world
        self.width, self.height, self.halo = width, height, halo
        self.stage_on_host = stage_on_host
        self.subs = []
        for j in range(2):
            lo, hi, top, bot = slab_extent(height, vworld, 2 * rank + j, halo)
            sub = dict(lo=lo, hi=hi, top=top, bot=bot, row0=lo - top, local_height=(hi + bot) - (lo - top))
            sub["backend"] = _make(make_backend, width, sub["local_height"], sub["row0"])
            sub["pending"] = None   # (work handles, wire buffers, device buffers, halo row) of a posted exchange
            sub["bufs"] = None
            self.subs.append(sub)
        self.lo, self.hi = self.subs[0]["lo"], self.subs[1]["hi"]

    def local_frame_rows(self):
        """Frame rows [row0, row1) this rank must provide (both sub-slabs with their halos)."""
        return self.subs[0]["row0"], self.subs[1]["row0"] + self.subs[1]["local_height"]

    def set_frames(self, prev_local, curr_local):
        r0, r1 = self.local_frame_rows()
        if prev_local.shape != (r1 - r0, self.width):
            raise ValueError("local frames must have shape (%d, %d)" % (r1 - r0, self.width))
        for sub in self.subs:
            a = sub["row0"] - r0
            sub["backend"].set_frames(prev_local[a:a + sub["local_height"]], curr_local[a:a + sub["local_height"]])

    def _bufs(self, sub):
        if sub["bufs"] is None:
            sub["bufs"] = {k: sub["backend"].new_rows(self.halo) for k in ("send", "recv", "local")}
        return sub["bufs"]

    def _post_remote(self, sub, peer, own_row, halo_row):
        """Owned rows [own_row, own_row+halo) go to `peer`, its rows come back for the halo at halo_row."""
        d, b = self.dist, sub["backend"]
        bufs = self._bufs(sub)
        b.get_rows(own_row, *bufs["send"])
        if self.stage_on_host:
            send = (bufs["send"][0].cpu(), bufs["send"][1].cpu())
            recv = (bufs["recv"][0].cpu(), bufs["recv"][1].cpu())
        else:
            send, recv = bufs["send"], bufs["recv"]
        ops = [d.P2POp(d.isend, send[0], peer), d.P2POp(d.isend, send[1], peer),
               d.P2POp(d.irecv, recv[0], peer), d.P2POp(d.irecv, recv[1], peer)]
        sub["pending"] = (d.batch_isend_irecv(ops), (send, recv), halo_row)

    def _finish_remote(self, sub):
        if sub["pending"] is None:
            return
        works, (send, recv), halo_row = sub["pending"]
        sub["pending"] = None
        for w in works:
            w.wait()
        bufs = self._bufs(sub)
        if self.stage_on_host:
            bufs["recv"][0].copy_(recv[0])
            bufs["recv"][1].copy_(recv[1])
        sub["backend"].put_rows(halo_row, *bufs["recv"])

    def _local_exchange(self):
        """A's last owned rows -> B's top halo, B's first owned rows -> A's bottom halo."""
        A, B = self.subs
        h = self.halo
        ba, bb = A["backend"], B["backend"]
        with ba.activate():
            ba.get_rows(A["top"] + (A["hi"] - A["lo"]) - h, *self._bufs(A)["local"])
            ea = ba.record()
        with bb.activate():
            bb.get_rows(B["top"], *self._bufs(B)["local"])
            eb = bb.record()
            bb.wait_event(ea)
            bb.put_rows(0, *self._bufs(A)["local"])                       # rows [B.lo - halo, B.lo)
            eb2 = bb.record()
        with ba.activate():
            ba.wait_event(eb)
            ba.put_rows(A["top"] + (A["hi"] - A["lo"]), *self._bufs(B)["local"])  # rows [A.hi, A.hi + halo)
            ba.wait_event(eb2)   # A's buffer is free again only once B has read it
            ea2 = ba.record()
        with bb.activate():
            bb.wait_event(ea2)

    def solve(self, lam, iters, eps=None, use_previous=False):
        """`iters` Jacobi sweeps from zero flow on the whole frame; returns the number of exchange rounds.
        (ITER from zero only: ITER|EPS and warm starts are SlabSolver's.)"""
        if eps is not None or use_previous:
            raise NotImplementedError("OverlappedSlabSolver runs ITER from zero flow; use SlabSolver for ITER|EPS / use_previous")
        A, B = self.subs
        plan = chunks(iters, self.halo)
        n_ex = 0
        for i, n in enumerate(plan):
            last = i + 1 == len(plan)
            with A["backend"].activate():
                self._finish_remote(A)
                A["backend"].sweep(n, lam, first=(i == 0))
                if not last and self.rank > 0:
                    self._post_remote(A, self.rank - 1, A["top"], 0)
            with B["backend"].activate():
                self._finish_remote(B)
                B["backend"].sweep(n, lam, first=(i == 0))
                if not last and self.rank + 1 < self.world:
                    self._post_remote(B, self.rank + 1, B["top"] + (B["hi"] - B["lo"]) - self.halo, B["top"] + (B["hi"] - B["lo"]))
            if not last:
                self._local_exchange()
                n_ex += 1
        return n_ex

    def owned_flow(self):
        """(u, v) of the rows this rank owns, as host arrays of shape (hi - lo, width)."""
        parts = []
        for sub in self.subs:
            u, v = sub["backend"].flow()
            u, v = np.asarray(u), np.asarray(v)
            parts.append((u[sub["top"]:sub["top"] + (sub["hi"] - sub["lo"])], v[sub["top"]:sub["top"] + (sub["hi"] - sub["lo"])]))
        return np.concatenate([parts[0][0], parts[1][0]]), np.concatenate([parts[0][1], parts[1][1]])

    def close(self):
        for sub in self.subs:
            sub["backend"].close()


def shard_pairs(n_pairs, world, rank):
    """Independent pairs (BASELINE config C4): pair i runs on rank i mod world; no collective."""
    return list(range(rank, n_pairs, world))
