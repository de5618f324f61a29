"""allocnet_amd -- MI355X-native batched MINCO / min-jerk / min-snap trajectory solver.

Host-side mirror (Python) of the reference's operator surface for the hot path; all arithmetic
runs in hand-written HIP kernels behind the C ABI in include/allocnet_amd.h.
"""
from ._lib import AnetError, load, LIB_PATH  # noqa: F401
from .context import Context, default_context  # noqa: F401
from .minco import MINCO, MINCO_S2NU, MINCO_S3NU, MINCO_S4NU, minco_solve, minco_solve_dev, bind_minco_solve, BoundMincoSolve, recommended_ld, minco_cost_grad, minco_cost_grad_dev, minco_cost_grad_launches, minco_piece_grad_shape, make_penalty, minco_sample_costs, minco_sample_costs_dev  # noqa: F401

from .trajectory import Piece, Trajectory, traj_eval, traj_cost, traj_cost_grad_T, traj_max_rate  # noqa: F401
from . import lbfgs  # noqa: F401
from .lbfgs import (lbfgs_parameter_t, lbfgs_strerror, lbfgs_mvie, lbfgs_minco, lbfgs_minco_dev,  # noqa: F401
                    launch_order_from_counts, lbfgs_optimize_dev, lbfgs_optimize)
from . import qp  # noqa: F401
from .qp import (qp_assemble, qp_dims, qp_solve, qp_solve_vjp, qp_solve_dev, qp_solve_vjp_dev, qp_settings,  # noqa: F401
                 QPSolver, QPConfig)
from .min_traj_opt import MinTrajOpt, OsqpLayer  # noqa: F401
from . import firi as _firi_mod  # noqa: F401
from .firi import (firi, firi_dev, firi_params, convex_cover, polytope_depth, find_interior, overlap,  # noqa: F401
                   overlap_pt, short_cut, pack_model_inputs, to_planner_form)

__version__ = "0.1.0"
