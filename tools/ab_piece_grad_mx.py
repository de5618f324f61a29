"""A/B of the large-batch penalty kernel: k_piece_grad (ANET_PG_MX=0) against k_piece_grad_mx (ANET_PG_MX=1, the basis-table
contractions on the FP64 matrix instructions, csrc/piece_grad_mx.h) on bench.py's config3 generator -- same process image, the
variant chosen by the environment variable the library reads once, so each side runs in a process of its own:

    python tools/ab_piece_grad_mx.py [--batch 131072] [--order 4] [--pieces 8]

Prints per side the median kernel time of anet_minco_partial_grads_dev (events on the stream, 5 x 20 launches) and of the whole
cost + gradient evaluation, then the largest relative differences of the partial gradients, dJ/dT partials and piece costs between
the two sides (each side's parity with the C restatement is tests/test_grad_gpu.py's business)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np
    import torch
    import allocnet_amd as aa
    from allocnet_amd.synth import corridor_problem
    from bench import PEN, _to_bm
    s, c, N, M, B = args.order, 3, args.pieces, 16, args.batch
    dev = torch.device("cuda:0")
    ctx = aa.Context(0)
    pen = aa.make_penalty(poly_rows=M, **PEN)
    head, tail, wps, T, hp = corridor_problem(np.random.default_rng(1), B, N, c, M)
    ld = aa.recommended_ld(B)
    th, tt, tw, tT, thp = (_to_bm(torch, x, B, ld, dev) for x in (head, tail, wps, T, hp))
    cost, gP, gT, work = aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, ctx=ctx)
    torch.cuda.synchronize()
    D = 2 * s
    coeffs = torch.empty((N * 3 * D, ld), dtype=torch.float64, device=dev)
    energy = torch.empty((ld,), dtype=torch.float64, device=dev)
    aa.minco_solve_dev(th, tt, tw, tT, s, c, N, B, coeffs=coeffs, energy=energy, ctx=ctx)
    gdC = torch.zeros_like(coeffs)
    gdT = torch.zeros((N, ld), dtype=torch.float64, device=dev)
    pcs = torch.zeros((N, ld), dtype=torch.float64, device=dev)

    import ctypes
    q = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pp = ctypes.cast(ctypes.pointer(pen), ctypes.c_void_p)

    def pg():
        ctx.check(ctx.lib.anet_minco_partial_grads_dev(ctx.handle, s, N, B, ld, q(coeffs), q(tT), q(thp), pp, 1, q(gdC), q(gdT), q(pcs), st))

    def ev():
        aa.minco_cost_grad_dev(th, tt, tw, tT, s, c, N, B, hpolys=thp, penalty=pen, work=work, cost=cost, gradP=gP, gradT=gT, ctx=ctx)

    def timed(fn, K=20, reps=5):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        out = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(K):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / K * 1e3)
        return sorted(out)[len(out) // 2]
    res = {"piece_grad_us": timed(pg), "evaluation_us": timed(ev)}
    pg()
    ev()
    torch.cuda.synchronize()
    np.savez(args.dump, gdC=gdC[:, :B].cpu().numpy(), gdT=gdT[:, :B].cpu().numpy(), pcs=pcs[:, :B].cpu().numpy(),
             cost=cost[:B].cpu().numpy(), gT=gT[:, :B].cpu().numpy(), gP=gP[:, :B].cpu().numpy())
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1 << 17)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--pieces", type=int, default=8)
    ap.add_argument("--dump", default=None)
    args = ap.parse_args()
    if args.dump:
        return child(args)
    import numpy as np
    with tempfile.TemporaryDirectory() as td:
        sides = {}
        for mx in (0, 1):
            dump = os.path.join(td, f"mx{mx}.npz")
            env = dict(os.environ, ANET_PG_MX=str(mx))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", str(args.batch), "--order", str(args.order),
                                "--pieces", str(args.pieces), "--dump", dump],
                               capture_output=True, text=True, env=env, cwd=ROOT)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                raise SystemExit(f"ANET_PG_MX={mx} failed")
            sides[mx] = (json.loads(r.stdout.strip().splitlines()[-1]), dict(np.load(dump)))
            print(f"ANET_PG_MX={mx}:", json.dumps(sides[mx][0]))
        a, b = sides[0][1], sides[1][1]
        for k in a:
            sc = np.abs(a[k]).max()
            print(f"  {k}: max |mx - plain| / max |plain| = {np.abs(a[k] - b[k]).max() / sc:.3e}")


if __name__ == "__main__":
    main()
