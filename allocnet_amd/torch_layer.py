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
        return out["coeffs"], out["obj"], out["status"]

    @staticmethod
    def backward(fctx, g_coeffs, g_obj, _g_status):
        t, state, hpolys, env, status = fctx.saved_tensors
        order, res, max_vel, max_acc, m34, anet_ctx = fctx.meta
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
