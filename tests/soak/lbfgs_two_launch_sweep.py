"""One launch against two (lbfgs_minco_persistent.h PersistArgs::park) over shapes the -m gpu test leaves out: variable sets, history
lengths, `past`, no corridor rows, one piece, small and ragged batches, early split points -- hashes of every output must be equal.
   gpurun -- 'python tests/soak/lbfgs_two_launch_sweep.py'        (runs itself twice with different environments)"""
import hashlib, json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    sys.path.insert(0, ROOT)
    import allocnet_amd as aa
    from allocnet_amd.synth import corridor_problem
    rng = np.random.default_rng(77)
    out = []
    for trial in range(36):
        s = int(rng.choice([3, 4])); N = int(rng.choice([1, 2, 3, 5, 8, 12, 16])); B = int(rng.choice([1, 63, 200, 777]))
        M = int(rng.choice([0, 6, 16])); opt = int(rng.choice([1, 2, 3])) if N > 1 else 2
        head, tail, wps, T, hp = corridor_problem(rng, B, N, 3, max(M, 6))
        pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=int(rng.choice([7, 20])), poly_rows=max(M, 6))
        prm = aa.lbfgs_parameter_t(mem_size=int(rng.choice([1, 3, 8])), past=int(rng.choice([0, 1, 3])))
        kw = dict(min_duration=0.3) if (rng.random() < 0.3 and opt != 1) else {}
        r = aa.lbfgs_minco(head, tail, wps, T, s, hpolys=hp if M else None, penalty=pen, param=prm, opt=opt, max_evals=6000, **kw)
        h = hashlib.sha256()
        for k in ("wps", "T", "cost", "status", "iters", "evals"): h.update(np.ascontiguousarray(r[k]).tobytes())
        out.append(dict(trial=trial, s=s, N=N, B=B, M=M, opt=opt, sha=h.hexdigest(), evals_max=int(r["evals"].max()), evals_min=int(r["evals"].min())))
    print(json.dumps(out))
    sys.exit(0)
res = {}
for name, env_extra in (("one", dict(ANET_LBFGS_SPLIT_EVALS="0")), ("two@60", dict(ANET_LBFGS_SPLIT_EVALS="60")), ("two@400", dict(ANET_LBFGS_SPLIT_EVALS="400"))):
    env = dict(os.environ, ANET_LBFGS_SPLIT_MIN_BATCH="1", ANET_LBFGS_SPLIT_MIN_VARS="1", **env_extra)
    p = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    res[name] = json.loads(p.stdout.strip().splitlines()[-1])
bad = 0
for a, b, c in zip(res["one"], res["two@60"], res["two@400"]):
    same = a["sha"] == b["sha"] == c["sha"]
    bad += not same
    print(("ok  " if same else "DIFF"), {k: a[k] for k in ("trial", "s", "N", "B", "M", "opt", "evals_min", "evals_max")})
print("different:", bad)
sys.exit(1 if bad else 0)
