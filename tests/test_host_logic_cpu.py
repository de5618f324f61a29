"""Host-side data formats of the path's callers (no GPU): the network-input packing of LearningPlanner::callModel
(learning_planner.hpp:147-170) and the planner form of a corridor (learning_planner.hpp:293-299)."""
import numpy as np
import pytest


def test_pack_model_inputs_layout_and_round_trip():
    import allocnet_amd as aa
    rng = np.random.default_rng(0)
    raw = [np.c_[rng.normal(size=(k, 3)) * rng.uniform(0.5, 3.0, size=(k, 1)), rng.normal(size=k)] for k in (7, 12, 9)]
    n_rows = [len(h) for h in raw]
    H = max(n_rows)
    padded = np.zeros((3, H, 4))
    for i, h in enumerate(raw):
        padded[i, :len(h)] = h
    planner = aa.to_planner_form(padded, n_rows)            # unit normals, offset negated: a.x <= b
    for i, h in enumerate(raw):
        nrm = np.linalg.norm(h[:, :3], axis=1)
        assert np.allclose(np.linalg.norm(planner[i, :n_rows[i], :3], axis=1), 1.0)
        assert np.allclose(planner[i, :n_rows[i], 3], -h[:, 3] / nrm) and (planner[i, n_rows[i]:] == 0).all()
    ini = np.array([[0.0, 0.1, 0.2], [1.0, 1.1, 1.2], [2.0, 2.1, 2.2]])      # row = axis, columns p, v, a
    fin = ini + 10.0
    state, hp = aa.pack_model_inputs(ini, fin, [planner[i, :n_rows[i]] for i in range(3)])
    assert state.dtype == np.float32 and state.shape == (9, 2) and hp.dtype == np.float32 and hp.shape == (50, 4, 5)
    assert np.allclose(state[:, 0], [0.0, 0.1, 0.2, 1.0, 1.1, 1.2, 2.0, 2.1, 2.2]) and np.allclose(state[:, 1], state[:, 0] + 10.0)
    # tensor[row][comp][seg] = hPolys[seg](row, comp): what from_blob makes of the column-major Eigen buffer
    for i in range(3):
        assert np.allclose(hp[:n_rows[i], :, i], planner[i, :n_rows[i]].astype(np.float32))
        assert (hp[n_rows[i]:, :, i] == 0).all()
    assert (hp[:, :, 3:] == 0).all()
    # the consumer's side (min_traj_opt.py:75-90): polytopes until the first all-zero one, rows until the first zero row
    got = []
    for i in range(5):
        poly = hp[:, :, i]
        if np.linalg.norm(poly) <= 1.0:
            break
        k = next((j for j in range(poly.shape[0]) if np.linalg.norm(poly[j]) <= 0.0), poly.shape[0])
        got.append(poly[:k])
    assert [len(g) for g in got] == n_rows
    with pytest.raises(ValueError):
        aa.pack_model_inputs(ini, fin, [planner[0, :7]] * 6)
    with pytest.raises(ValueError):
        aa.pack_model_inputs(ini, fin, [np.ones((51, 4))])
