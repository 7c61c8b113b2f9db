"""The single-context building blocks added for drivers that spread one solve over several contexts or overlap several
solves (include/hsflow.h: hsflow_set_eps_rows, hsflow_solve_probe, hsflow_take_verdict, hsflow_flow_view_device,
hsflow_set_cu_share, hsflow_set_async_reduce).  Bar: per-sweep Eps equal to the oracle's sweep by sweep (fp32 differences
of fp32 flows: a few ulp of the flow), flows bit-identical whatever the switches."""
import ctypes

import numpy as np
import pytest

from opticalflowhs_amd import synth

pytestmark = pytest.mark.gpu
ITER, EPS = 1, 2


def oracle_sweep_eps(oracle, A, B, lam, n, rows=None):
    """Eps of every sweep (max |change| of u, v over `rows`) and the flow after n sweeps, by the CPU oracle."""
    H, W = A.shape
    a, b = rows if rows else (0, H)
    u, v = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
    out = []
    for k in range(n):
        u1, v1 = oracle.calc_optical_flow_hs(A, B, lam, 1, term_type=ITER, use_previous=k > 0, velx=u, vely=v)
        out.append(max(float(np.abs(u1[a:b] - u[a:b]).max()), float(np.abs(v1[a:b] - v[a:b]).max())))
        u, v = u1, v1
    return np.array(out), u, v


@pytest.mark.parametrize("kernel", ["auto", "strip", "simple"])
def test_probe_reports_every_sweeps_eps_over_the_row_window(hs, oracle, gpu_ok, kernel):
    W, H, n, lam = 300, 170, 23, 0.8
    A, B = synth.translating_pair(W, H, seed=9)
    k = {"auto": hs.KERNEL_AUTO, "strip": hs.KERNEL_STRIP, "simple": hs.KERNEL_SIMPLE}[kernel]
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for rows in (None, (40, 97), (0, 1), (169, 170)):
            ctx.set_eps_rows(*(rows[0], rows[1] - rows[0]) if rows else (0, 0))
            e = ctx.solve_probe(lam=lam, max_iter=n, term_type=ITER, kernel=k)
            eo, uo, vo = oracle_sweep_eps(oracle, A, B, lam, n, rows)
            u, v = ctx.flow()
            fmax = max(1.0, float(np.abs(uo).max()), float(np.abs(vo).max()))
            assert e.shape == (n,) and np.all(np.abs(e - eo) <= 4e-7 * fmax), (kernel, rows, np.abs(e - eo).max())
            assert np.sqrt(np.mean((u - uo) ** 2)) <= 1e-4 and np.sqrt(np.mean((v - vo) ** 2)) <= 1e-4
        # the window is the strip / simple kernels': the folded and the LDS-tile kernel refuse EPS over a window
        ctx.set_eps_rows(40, 57)
        for bad in (hs.KERNEL_FOLD, hs.KERNEL_FUSED):
            with pytest.raises(hs.HsflowError) as err:
                ctx.solve(lam=lam, max_iter=n, term_type=ITER | EPS, epsilon=1e-6, kernel=bad)
            assert err.value.status == hs._lib.E_ARG
        ctx.solve(lam=lam, max_iter=n, term_type=ITER, kernel=hs.KERNEL_FOLD)   # (ITER does not look at Eps)
        with pytest.raises(hs.HsflowError):
            ctx.set_eps_rows(160, 20)   # outside the frame


def test_take_verdict_looks_without_acting(hs, gpu_ok):
    W, H = 512, 160
    flat = np.full((H, W), 90, np.uint8)
    a, b = flat.copy(), flat.copy()
    a[40:120, 100:400], b[40:120, 100:400] = 120, 121
    moving = synth.translating_pair(W, H, seed=11)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        with pytest.raises(hs.HsflowError) as err:   # nothing owed
            ctx.take_verdict()
        assert err.value.status == hs._lib.E_STATE
        # a textured pair: the witness holds
        ctx.set_frames(*moving)
        ctx.solve(lam=1.0, max_iter=40, term_type=ITER)
        ref = ctx.flow()
        ctx.solve_async(lam=1.0, max_iter=40, term_type=ITER | EPS, epsilon=1e-6)
        assert ctx.take_verdict() is True
        i = ctx.info()
        assert i["iterations_done"] == 40 and i["eps_rerun"] == 0
        assert np.array_equal(ctx.flow()[0], ref[0])
        # a converging pair: the witness fails, and NOTHING is re-run -- the flow of the whole budget stands
        ctx.set_frames(a, b)
        ctx.solve(lam=1e-3, max_iter=300, term_type=ITER)
        ref = ctx.flow()
        stop = ctx.solve(lam=1e-3, max_iter=300, term_type=ITER | EPS, epsilon=1e-4)["iterations_done"]
        assert stop < 300
        ctx.solve_async(lam=1e-3, max_iter=300, term_type=ITER | EPS, epsilon=1e-4)
        assert ctx.take_verdict() is False
        i = ctx.info()
        assert i["iterations_done"] == 300 and i["eps_rerun"] == 0
        u, v = ctx.flow()
        same = lambda x, y: bool(np.all((x == y) | ((np.abs(x) < 1e-30) & (np.abs(y) < 1e-30))))
        assert same(u, ref[0]) and same(v, ref[1])


def test_switches_do_not_change_a_bit_and_the_flow_view_is_the_flow(hs, gpu_ok):
    import torch
    W, H, it = 640, 360, 37
    A, B = synth.translating_pair(W, H, seed=13)
    lib = hs._lib.load()
    with hs.HSFlow(W, H, 2, own_stream=True) as ctx:
        ctx.set_frames(A, B, pair=0)
        ctx.set_frames(B, A, pair=1)
        ctx.solve(lam=0.6, max_iter=it, term_type=ITER | EPS, epsilon=1e-6)
        ref = [ctx.flow(pair=p) for p in (0, 1)]
        shapes = set()
        for share, reduce_ in ((0, 0), (32, 0), (8, 1), (200, 1), (0, 1)):
            assert lib.hsflow_set_cu_share(ctx._h, share) == 0 and lib.hsflow_set_async_reduce(ctx._h, reduce_) == 0
            ctx.solve_async(lam=0.6, max_iter=it, term_type=ITER | EPS, epsilon=1e-6, use_graph=bool(reduce_))
            ctx.synchronize()
            i = ctx.info()
            shapes.add((i["kernel"], i["tiles"], i["fuse_steps"]))
            assert i["iterations_done"] == it and i["eps_rerun"] == 0
            for p in (0, 1):
                u, v = ctx.flow(pair=p)
                assert np.array_equal(u, ref[p][0]) and np.array_equal(v, ref[p][1]), (share, reduce_, p)
        assert len(shapes) >= 2, shapes   # planning for a share of the chip did pick other launch shapes
        assert lib.hsflow_set_cu_share(ctx._h, -1) == hs._lib.E_ARG
        # the flow in place: device pointers + pitch
        for p in (0, 1):
            pu, pv, sb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
            assert lib.hsflow_flow_view_device(ctx._h, p, ctypes.byref(pu), ctypes.byref(pv), ctypes.byref(sb)) == 0
            assert sb.value == ctx.info()["pitch"] * 4
            from opticalflowhs_amd.pipeline import _DeviceView
            u = torch.as_tensor(_DeviceView(pu.value, (H, W), (sb.value, 4)), device="cuda").cpu().numpy()
            v = torch.as_tensor(_DeviceView(pv.value, (H, W), (sb.value, 4)), device="cuda").cpu().numpy()
            assert np.array_equal(u, ref[p][0]) and np.array_equal(v, ref[p][1])
        assert lib.hsflow_flow_view_device(ctx._h, 2, ctypes.byref(pu), ctypes.byref(pv), ctypes.byref(sb)) == hs._lib.E_ARG


def test_cached_plans_do_not_carry_a_callers_flags(hs, gpu_ok):
    """A context plans once per parameter set (the planner's sweep costs tens of microseconds per solve); what does not enter the
    plan -- use_previous, reuse_derivatives, use_graph -- must still be honoured on every call."""
    W, H, it = 700, 300, 24
    A, B = synth.translating_pair(W, H, seed=17)
    A2, B2 = synth.translating_pair(W, H, seed=18)

    def fresh(frames, steps):
        with hs.HSFlow(W, H, 1, own_stream=True) as c:
            c.set_frames(*frames)
            for kw in steps:
                c.solve(lam=0.9, max_iter=it, term_type=ITER, **kw)
            return c.flow()

    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        seq = [dict(), dict(use_previous=True), dict(use_previous=True, use_graph=True), dict(), dict(use_graph=True), dict(use_previous=True, reuse_derivatives=True)]
        for k in range(len(seq)):
            ctx.solve(lam=0.9, max_iter=it, term_type=ITER, **seq[k])
            u, v = ctx.flow()
            j = max(i for i in range(k + 1) if not seq[i].get("use_previous"))   # the last cold start
            uo, vo = fresh((A, B), seq[j:k + 1])
            assert np.array_equal(u, uo) and np.array_equal(v, vo), k
        # new frames, same parameters: reuse_derivatives must not be inherited from the cached call
        ctx.set_frames(A2, B2)
        ctx.solve(lam=0.9, max_iter=it, term_type=ITER)
        u, v = ctx.flow()
        uo, vo = fresh((A2, B2), [dict()])
        assert np.array_equal(u, uo) and np.array_equal(v, vo)


@pytest.mark.parametrize("shape", [(1920, 1080), (600, 480), (333, 150), (258, 81)])
def test_a_solve_from_zero_flow_equals_a_warm_start_from_zeroed_planes(hs, gpu_ok, shape):
    """A solve from zero flow hands the strip / fold / classic strip kernels ONE row of zeros in place of the two flow
    planes (StripGeom::zero_in, hsflow_ctx::dZero); a warm start (`use_previous`) from planes that were set to zero
    reads the planes themselves: the same bits, with and without the derivative pass in the first launch, ITER and
    ITER|EPS, and in the classic mode."""
    import torch
    W, H = shape
    A, B = synth.translating_pair(W, H, seed=11)
    z = torch.zeros((H, W), dtype=torch.float32, device="cuda")
    cases = [dict(kernel=hs.KERNEL_STRIP), dict(kernel=hs.KERNEL_FOLD), dict(kernel=hs.KERNEL_AUTO),
             dict(kernel=hs.KERNEL_STRIP, reuse_derivatives=True), dict(mode=hs.MODE_CLASSIC, alpha=15.0)]
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        for kw in cases:
            for tt in (ITER, ITER | EPS):
                if kw.get("mode") == hs.MODE_CLASSIC and tt != ITER:
                    continue
                base = dict(lam=1.0, max_iter=37, term_type=tt, epsilon=1e-9, use_graph=True)
                base.update(kw)
                try:
                    ctx.set_frames(A, B)
                    if base.get("reuse_derivatives"):
                        ctx.solve(**dict(base, reuse_derivatives=False, max_iter=1))  # leaves the derivative plane behind
                    ctx.solve(**base)
                except hs.HsflowError:
                    continue  # a shape this frame cannot take (e.g. the folded kernel on a sliver)
                u0, v0 = ctx.flow()
                ctx.set_flow_rows_from(z, z, 0, H)
                ctx.solve(**dict(base, use_previous=True))
                u1, v1 = ctx.flow()
                assert np.array_equal(u0, u1) and np.array_equal(v0, v1), (shape, kw, tt)
