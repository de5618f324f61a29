"""ctypes binding of include/allocnet_amd.h.  The library is the product; there is no Python or
CPU fallback: if the .so is missing or no GPU is visible, calls fail loudly."""
import ctypes
import os
from ctypes import c_int, c_int64, c_void_p, c_char_p, c_double, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liballocnet_amd.so")

ANET_OK = 0
ANET_ERR_INVALID = -1
ANET_ERR_HIP = -2
ANET_ERR_UNSUPPORTED = -3
ANET_ERR_NOMEM = -4
ANET_ERR_NODEVICE = -5

_dp = POINTER(c_double)

# name -> (restype, argtypes); must list every symbol include/allocnet_amd.h declares
PROTOTYPES = {
    "anet_abi_version": (c_int, []),
    "anet_device_count": (c_int, []),
    "anet_create": (c_int, [c_int, POINTER(c_void_p)]),
    "anet_destroy": (None, [c_void_p]),
    "anet_last_error": (c_char_p, [c_void_p]),
    "anet_compute_units": (c_int, [c_void_p]),
    "anet_stream": (c_void_p, [c_void_p]),
    "anet_synchronize": (c_int, [c_void_p]),
    "anet_recommended_ld": (c_int64, [c_int64]),
    "anet_dev_alloc": (c_int, [c_void_p, ctypes.c_size_t, POINTER(c_void_p)]),
    "anet_dev_free": (None, [c_void_p]),
    "anet_dev_upload": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]),
    "anet_dev_download": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]),
    "anet_to_batch_minor_dev": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "anet_to_traj_major_dev": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "anet_minco_solve_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_minco_solve_wide_spread_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "anet_minco_solve": (c_int, [c_void_p, c_int, c_int, c_int, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_traj_eval_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                   c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "anet_traj_eval": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                               c_int, c_void_p, c_int, c_void_p]),
    "anet_traj_max_rate_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p]),
    "anet_traj_max_rate": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "anet_traj_cost_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                   c_double, c_void_p, c_void_p]),
    "anet_traj_cost": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_double, c_void_p]),
    "anet_traj_cost_grad_T_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                          c_double, c_void_p, c_void_p]),
    "anet_traj_cost_grad_T": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_double, c_void_p]),
    "anet_minco_spread_flags_dev": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_double, c_void_p, c_void_p]),
    "anet_minco_sample_costs_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "anet_minco_sample_costs": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_double, c_void_p]),
    "anet_minco_partial_grads_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_minco_propagate_grad_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_minco_cost_grad_workspace": (c_int64, [c_int, c_int, c_int64]),
    "anet_minco_cost_grad_launches": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "anet_minco_piece_grad_shape": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p]),
    "anet_minco_cost_grad_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64] + [c_void_p] * 12),
    "anet_minco_cost_grad": (c_int, [c_void_p, c_int, c_int, c_int, c_int64] + [c_void_p] * 10),
    "anet_qp_dims_of": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "anet_qp_assemble_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double,
                                     c_int, c_int] + [c_void_p] * 10),
    "anet_qp_assemble": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double,
                                 c_int, c_int] + [c_void_p] * 9),
    "anet_qp_default_settings": (None, [c_void_p]),
    "anet_qp_solve": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double]
                      + [c_void_p] * 9),
    "anet_qp_solve_workspace": (c_int64, [c_int, c_int, c_int64, c_int, c_int]),
    "anet_qp_solve_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double]
                          + [c_void_p] * 11),
    "anet_qp_solve_ordered_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double]
                                  + [c_void_p] * 12),
    "anet_qp_solve_time_grad": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double]
                                + [c_void_p] * 10),
    "anet_qp_solve_time_grad_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double,
                                            c_double] + [c_void_p] * 12),
    "anet_qp_solve_vjp": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double]
                          + [c_void_p] * 11),
    "anet_qp_solve_vjp_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_double, c_double, c_double]
                              + [c_void_p] * 13),
    "anet_polytope_depth": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "anet_polytope_depth_dev": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "anet_firi_default_params": (None, [c_void_p]),
    "anet_firi": (c_int, [c_void_p, c_int64, c_int, c_int, c_int] + [c_void_p] * 10),
    "anet_firi_var": (c_int, [c_void_p, c_int64, c_int, c_int, c_int] + [c_void_p] * 11),
    "anet_firi_var_dev": (c_int, [c_void_p, c_int64, c_int, c_int, c_int] + [c_void_p] * 13),
    "anet_firi_workspace": (c_int64, [c_int64, c_int, c_int]),
    "anet_firi_dev": (c_int, [c_void_p, c_int64, c_int, c_int, c_int] + [c_void_p] * 12),
    "anet_comm_unique_id": (c_int, [c_void_p, c_void_p]),
    "anet_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "anet_comm_allgather_costs_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "anet_comm_destroy": (c_int, [c_void_p]),
    "anet_lbfgs_default_params": (None, [c_void_p]),
    "anet_lbfgs_check_params": (c_int, [c_int, c_void_p]),
    "anet_lbfgs_strerror": (c_char_p, [c_int]),
    "anet_lbfgs_mvie": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_double, c_double, c_void_p, c_void_p,
                                c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "anet_lbfgs_workspace": (c_int64, [c_int, c_int64, c_void_p]),
    "anet_lbfgs_optimize_dev": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_piece_normalized_coeffs": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "anet_piece_normalized_coeffs_dev": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "anet_lbfgs_optimize_host": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "anet_lbfgs_minco": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "anet_lbfgs_minco_workspace": (c_int64, [c_int, c_int, c_int64, c_void_p]),
    "anet_lbfgs_minco_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_launch_order_from_counts_dev": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_launch_order_from_steps_dev": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_lbfgs_minco_ordered_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_lbfgs_minco_bounded_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anet_set_cancel_flag": (c_int, [c_void_p, c_void_p]),
    "anet_lbfgs_minco_bounded": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
}


class QpSettings(ctypes.Structure):
    """struct anet_qp_settings (OSQP defaults)."""
    _fields_ = [("rho", c_double), ("sigma", c_double), ("alpha", c_double), ("eps_abs", c_double),
                ("eps_rel", c_double), ("max_iter", ctypes.c_int32), ("check_termination", ctypes.c_int32),
                ("adaptive_rho_interval", ctypes.c_int32), ("scaled_termination", ctypes.c_int32),
                ("method", ctypes.c_int32)]


class FiriParams(ctypes.Structure):
    """struct anet_firi_params (defaults of firi::firi / maxVolInsEllipsoid)."""
    _fields_ = [("iterations", ctypes.c_int32), ("epsilon", c_double), ("smooth_eps", c_double),
                ("penalty_wt", c_double), ("mvie_max_evals", ctypes.c_int32)]


class QpDims(ctypes.Structure):
    _fields_ = [("n", c_int64), ("m_e", c_int64), ("m_g", c_int64)]


class LbfgsParams(ctypes.Structure):
    """struct anet_lbfgs_params == lbfgs::lbfgs_parameter_t (lbfgs.hpp:15-129)."""
    _fields_ = [("mem_size", ctypes.c_int32), ("g_epsilon", c_double), ("past", ctypes.c_int32),
                ("delta", c_double), ("max_iterations", ctypes.c_int32), ("max_linesearch", ctypes.c_int32),
                ("min_step", c_double), ("max_step", c_double), ("f_dec_coeff", c_double),
                ("s_curv_coeff", c_double), ("cautious_factor", c_double), ("machine_prec", c_double)]


class Penalty(ctypes.Structure):
    """struct anet_penalty (include/allocnet_amd.h)."""
    _fields_ = [("rho", c_double), ("w_corridor", c_double), ("w_vel", c_double), ("w_acc", c_double),
                ("smooth_mu", c_double), ("max_vel", c_double), ("max_acc", c_double),
                ("res", ctypes.c_int32), ("poly_rows", ctypes.c_int32)]


_lib = None


class AnetError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"allocnet_amd error {code}: {msg}")
        self.code = code


def _preload_shared_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64; if this library
    pulled in /opt/rocm's copy first, a later `import torch` in the same process would find "No HIP
    GPUs".  When a torch installation is present (it is NOT imported here) its runtime is loaded first,
    so the import order of torch and allocnet_amd does not matter."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the shared library (building is a separate, explicit step: allocnet_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -m allocnet_amd.build` (hipcc, gfx950). "
            "allocnet_amd has no CPU fallback.")
    _preload_shared_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc):
    if rc != ANET_OK:
        msg = load().anet_last_error(ctx)
        raise AnetError(rc, msg.decode() if msg else "?")
