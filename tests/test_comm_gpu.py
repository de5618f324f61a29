"""Native RCCL all-gather of costs (anet_comm_*) with a single rank: the code path bench-independent
C/C++ hosts use for multi-GPU.  (2/4/8-rank runs need as many GPUs; the box has one.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_native_allgather_single_rank(anet_ctx):
    import torch
    import allocnet_amd as aa
    from allocnet_amd.distributed import NativeComm
    ctx = aa.Context(0)              # own context: the communicator lives in it
    comm = NativeComm(ctx, 1, 0)
    assert len(comm.unique_id) == 128
    send = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    recv = torch.full((1000,), -1.0, dtype=torch.float64, device="cuda")
    comm.allgather_costs(send, recv, 1000)
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    with pytest.raises(aa.AnetError):            # second init on the same context is refused
        NativeComm(ctx, 1, 0)
    comm.close()
    ctx.close()


def test_config5_sharded_cost_grad_full_size(anet_ctx):
    """SURVEY 8(d) config 5 at its full size: B = 32768 8-segment snap problems with corridor and
    dynamic-limit penalties, seed 3, split into the 8 contiguous shards 8 ranks would own; each
    shard's costs go through the native all-gather.  Size-independent properties: the gathered costs
    and the gradients equal the unsharded evaluation to rounding (no cross-trajectory coupling, padding
    rows inert; the shards run the small-batch kernel shapes), ragged shard sizes included; a sample is checked against the numpy
    oracle."""
    import torch
    import allocnet_amd as aa
    from allocnet_amd.distributed import NativeComm, shard_bounds
    from oracle import minco_np as onp
    from tests.util import corridor_problem
    B, s, c, N, M, world = 32768, 4, 3, 8, 16, 8
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(3), B, N, c, M)
    kw = dict(res=20, vmax=4.0, amax=6.0, wc=1e4, wv=1e3, wa=1e3, mu=1e-2)
    pen = aa.make_penalty(rho=50.0, w_corridor=kw["wc"], w_vel=kw["wv"], w_acc=kw["wa"], smooth_mu=kw["mu"],
                          max_vel=kw["vmax"], max_acc=kw["amax"], res=kw["res"], poly_rows=M)
    cost, gP, gT = aa.minco_cost_grad(head, tail, wps, T, s, hpolys=hp, penalty=pen, ctx=anet_ctx)
    assert np.isfinite(cost).all() and np.isfinite(gP).all() and np.isfinite(gT).all()
    ctx = aa.Context(0)
    comm = NativeComm(ctx, 1, 0)
    gathered = np.empty(B)
    for total in (B, B - 5):                      # even and ragged shards
        for r in range(world):
            lo, hi = shard_bounds(total, world, r)
            c_r, gP_r, gT_r = aa.minco_cost_grad(head[lo:hi], tail[lo:hi], wps[lo:hi], T[lo:hi], s,
                                                  hpolys=hp[lo:hi], penalty=pen, ctx=anet_ctx)
            send = torch.from_numpy(c_r).cuda()
            recv = torch.empty_like(send)
            comm.allgather_costs(send, recv, hi - lo)
            torch.cuda.synchronize()
            gathered[lo:hi] = recv.cpu().numpy()
            # a 4096-trajectory shard runs the small-batch kernels (two lanes per piece in the penalty kernel, one
            # lane per axis in the propagate kernel), the full batch the lane-per-trajectory ones: sums are
            # taken in a different order
            assert np.abs(gP_r - gP[lo:hi]).max() <= 1e-11 * np.abs(gP).max()
            assert np.abs(gT_r - gT[lo:hi]).max() <= 1e-11 * np.abs(gT).max()
        assert np.abs(gathered[:total] - cost[:total]).max() <= 1e-12 * np.abs(cost).max()
    comm.close()
    ctx.close()
    for b in (0, 4097, 20000, B - 1):
        hpb = np.transpose(hp[b], (1, 2, 0))
        co0, e0, *_ = onp.minco_dense_solve(s, head[b], tail[b], wps[b].T, T[b])
        jp, gC, gTp, _ = onp.penalty_partials(s, co0, T[b], hpb, **kw)
        eC, eT = onp.energy_partials(s, co0, T[b])
        gP0, gT0 = onp.minco_dense_propagate(s, head[b], tail[b], wps[b].T, T[b], gC + eC, gTp + eT + 50.0)
        c0 = e0 + 50.0 * T[b].sum() + jp
        assert abs(cost[b] - c0) <= 1e-9 * abs(c0)
        assert np.abs(gP[b].T - gP0).max() <= 1e-7 * max(1.0, np.abs(gP0).max())
        assert np.abs(gT[b] - gT0).max() <= 1e-7 * max(1.0, np.abs(gT0).max())
