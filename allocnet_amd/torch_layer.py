"""The QP as a differentiable torch layer on the device: what OsqpLayer is to the reference's training loop
(network/utils/learning/layers.py:51-151 -- forward = solve with OSQP, backward = the -J^-1 grad KKT hook), with the
gradient carried through to the segment times.

    coeffs, obj, status = qp_layer(times, state, hpolys, order=4)       # times.requires_grad -> d loss / d times

forward: anet_qp_solve_time_grad_dev (interior point); backward: anet_qp_solve_vjp_dev for the part of the loss that
reaches the coefficients plus the envelope-theorem gradient for the part that is the optimal cost itself.  Problems the
solver reports unsolved (status != 1: infeasible corridor / limits) get a zero gradient; `status` tells which."""
import torch

from .qp import qp_solve_dev, qp_solve_vjp_dev


class _QPSolve(torch.autograd.Function):
    @staticmethod
    def forward(fctx, times, state, hpolys, order, res, max_vel, max_acc, m34, anet_ctx):
        t = times.detach().contiguous()
        out = qp_solve_dev(order, state, t, hpolys, res=res, max_vel=max_vel, max_acc=max_acc, m34=m34, time_grad=True,
                           ctx=anet_ctx)
        fctx.save_for_backward(t, state, hpolys, out["grad_T"], out["status"])
        fctx.meta = (order, res, max_vel, max_acc, m34, anet_ctx)
        fctx.mark_non_differentiable(out["status"])
        # undefined output gradients stay None: a loss that only uses `obj` must not pay the second QP solve of the
        # backward pass through the coefficients
        fctx.set_materialize_grads(False)
        return out["coeffs"], out["obj"], out["status"]

    @staticmethod
    def backward(fctx, g_coeffs, g_obj, _g_status):
        t, state, hpolys, env, status = fctx.saved_tensors
        order, res, max_vel, max_acc, m34, anet_ctx = fctx.meta
        if not fctx.needs_input_grad[0]:
            return (None,) * 9
        grad = torch.zeros_like(t)
        if g_coeffs is not None:
            back = qp_solve_vjp_dev(order, state, t, hpolys, g_coeffs.contiguous(), res=res, max_vel=max_vel,
                                    max_acc=max_acc, m34=m34, ctx=anet_ctx)
            ok = (back["status"] == 1) & (status == 1)
            grad = torch.where(ok[:, None], back["grad_T"], grad)
        if g_obj is not None:
            grad = grad + torch.where((status == 1)[:, None], g_obj[:, None] * env, torch.zeros_like(env))
        return grad, None, None, None, None, None, None, None, None


def qp_layer(times, state, hpolys, order=4, res=20, max_vel=4.0, max_acc=6.0, m34=1400.0, ctx=None):
    """times (B,N) float64 CUDA (may require grad); state (B,2,3,3) = [start PVA, end PVA]; hpolys (B,N,M,4) planner form,
    zero rows as padding.  Returns (coeffs (B,N,3,2*order), obj (B,), status (B,) int32)."""
    return _QPSolve.apply(times, state.contiguous(), hpolys.contiguous(), int(order), int(res), float(max_vel),
                          float(max_acc), float(m34), ctx)


class _MincoSolve(torch.autograd.Function):
    """coefficients and energy of the minimum-control trajectory through (wps, T) as a differentiable function of the
    interior waypoints and the durations: forward = anet_minco_solve_dev, backward = MINCO's propogateGrad
    (anet_minco_propagate_grad_dev) fed with d loss / d coeffs plus d loss / d energy times the energy's own partials
    (anet_minco_partial_grads_dev).  Batch-minor tensors, common row stride."""

    @staticmethod
    def forward(fctx, wps, T, head, tail, s, c, N, B, anet_ctx):
        import ctypes
        from .context import default_context
        from .minco import minco_solve_dev
        actx = anet_ctx or default_context(T.device.index or 0)
        ld = T.stride(0)
        coeffs = torch.empty(N * 3 * 2 * s, ld, device=T.device, dtype=torch.float64)
        energy = torch.empty(ld, device=T.device, dtype=torch.float64)
        minco_solve_dev(head, tail, wps.detach() if wps is not None else None, T.detach(), s, c, N, B, coeffs=coeffs,
                        energy=energy, ctx=actx)            # (one piece: no interior waypoints, wps may be None)
        fctx.save_for_backward(T.detach(), coeffs)
        fctx.meta = (s, c, N, B, actx)
        fctx.set_materialize_grads(False)
        return coeffs, energy

    @staticmethod
    def backward(fctx, g_coeffs, g_energy):
        import ctypes
        T, coeffs = fctx.saved_tensors
        s, c, N, B, actx = fctx.meta
        ld = T.stride(0)
        dev = T.device
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        q = lambda t: ctypes.c_void_p(t.data_ptr())
        gdC = torch.zeros_like(coeffs) if g_coeffs is None else g_coeffs.contiguous().clone()
        gdT = torch.zeros(N, ld, device=dev, dtype=torch.float64)
        if g_energy is not None:
            eC = torch.empty_like(coeffs); eT = torch.empty(N, ld, device=dev, dtype=torch.float64)
            actx.check(actx.lib.anet_minco_partial_grads_dev(actx.handle, s, N, B, ld, q(coeffs), q(T), None, None, 1, q(eC), q(eT),
                                                            None, st))
            gdC = gdC + g_energy[None, :] * eC
            gdT = gdT + g_energy[None, :] * eT
        gP = torch.zeros(max(3 * (N - 1), 1), ld, device=dev, dtype=torch.float64)
        gT = torch.zeros(N, ld, device=dev, dtype=torch.float64)
        actx.check(actx.lib.anet_minco_propagate_grad_dev(actx.handle, s, c, N, B, ld, q(T), q(coeffs), q(gdC), q(gdT), q(gP), q(gT), st))
        return (gP if N > 1 else None), gT, None, None, None, None, None, None, None


def minco_layer(wps, T, head, tail, s, c, N, B, ctx=None):
    """Batch-minor float64 CUDA tensors with a common row stride ld = T.stride(0): wps ((N-1)*3, ld), T (N, ld),
    head / tail (3c, ld).  Returns coeffs (N*3*2s, ld) and energy (ld,); gradients flow to wps and T."""
    return _MincoSolve.apply(wps, T, head, tail, int(s), int(c), int(N), int(B), ctx)
