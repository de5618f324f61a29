/*
 * allocnet_amd -- C ABI of the MI355X-native batched MINCO / min-jerk / min-snap trajectory
 * solver.  Plain C: opaque handle, POD arrays, no C++ or torch types cross this line.
 *
 * What each entry point replaces in the reference (KumarRobotics/AllocNet, paths relative to
 * the reference root) is cited per function.  `minco.hpp` itself is NOT part of the reference
 * tree; the MINCO entry points follow the upstream GCOPTER method names
 * the north star asks for (setConditions / setParameters / getEnergy / getCoeffs /
 * getEnergyPartialGradBy{Coeffs,Times} / propogateGrad).
 *
 * Conventions (the reference's own):
 *   - s      : order, 3 = min-jerk (degree 5), 4 = min-snap (degree 7); 2 = min-acc also works.
 *   - D      : 2*s coefficients per axis per piece, HIGHEST POWER FIRST
 *              (src/planner/include/gcopter/trajectory.hpp:75-133).
 *   - coeffs : piece-major, then axis x,y,z, then the D coefficients
 *              (src/planner/include/planner/learning_planner.hpp:212,227).
 *   - head/tail : 3 x c, row = axis, columns p,v,a[,j] (src/planner/src/learning_planning.cpp:150-151);
 *              c = number of boundary derivatives fixed per end: 3 = reference convention
 *              (qp_solver.hpp:37,152-158; for snap the end jerk is then free -> natural
 *              condition p''''=0), c = s = classic MINCO convention.
 *   - wps    : interior waypoints, (N-1) x 3 (waypoint-major, like GCOPTER's 3 x (N-1)
 *              column-major inPs).
 *   - energy : int (p^(s))^2 dt summed over axes (MINCO getEnergy convention, no 1/2);
 *              the reference's Trajectory::getTrajCost convention (1/2, m_34=1400) is
 *              available through anet_traj_cost*.
 *
 * Two families of entry points:
 *   *_dev : pointers are DEVICE pointers in the batch-minor layout: value f of trajectory b
 *           lives at ptr[f*ld + b] (f = the flattened per-trajectory index in the order given
 *           above, ld >= batch).  Asynchronous on `stream` (a hipStream_t, NULL = default).
 *   plain : pointers are HOST pointers, trajectory-major (each trajectory's values contiguous,
 *           exactly the reference's flattening).  Synchronous.
 *
 * All functions return ANET_OK (0) or a negative error code; anet_last_error() gives text.
 * One context per host thread per device; a context is not re-entrant: in particular ONE L-BFGS call in flight per
 * context (its completion flag lives in the context), whatever the streams (same as the
 * reference's QPSolver, qp_solver.hpp:43-45).
 */
#ifndef ALLOCNET_AMD_H
#define ALLOCNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANET_ABI_VERSION 2 /* 2: QP default method = interior point, QP max-iter status -2 (OSQP), FIRI ok = 2, new entry points */

enum {
  ANET_OK = 0,
  ANET_ERR_INVALID = -1,     /* bad argument (order, piece count, batch, NULL pointer ...) */
  ANET_ERR_HIP = -2,         /* a HIP runtime call failed; text in anet_last_error()       */
  ANET_ERR_UNSUPPORTED = -3, /* valid request this build has no kernel for                 */
  ANET_ERR_NOMEM = -4,
  ANET_ERR_NODEVICE = -5     /* no usable gfx950 device: the product path has NO CPU fallback */
};

#define ANET_MAX_PIECES 16   /* register-resident kernels are instantiated for N <= 16      */
#define ANET_MAX_POLY_ROWS 50 /* learning_planner.hpp:40 (eigen_stacked_hpolys 4*50)        */

typedef struct anet_ctx anet_ctx;

int anet_abi_version(void);
int anet_device_count(void);
int anet_create(int device, anet_ctx **out);
void anet_destroy(anet_ctx *ctx);
const char *anet_last_error(const anet_ctx *ctx);
/* Compute units of the context's device (hipDeviceAttributeMultiprocessorCount, read once by anet_create: 256 on an
 * MI355X in SPX mode, 32 per logical device in CPX).  Every launch-shape threshold of the library -- lane per trajectory or
 * per (trajectory, axis), one launch or three per cost + gradient evaluation, workgroups per CU of the QP, one or two
 * launches of the batched optimisers -- is a number of rounds of workgroups per compute unit and scales with it.           */
int anet_compute_units(const anet_ctx *ctx);
/* hipStream_t the plain (host) entry points run on. */
void *anet_stream(anet_ctx *ctx);
int anet_synchronize(anet_ctx *ctx);

/* ---- device buffers for callers that do not link the HIP runtime themselves ---------------- */
int anet_dev_alloc(anet_ctx *ctx, size_t n_doubles, double **out);
void anet_dev_free(double *p);
int anet_dev_upload(anet_ctx *ctx, double *dst_dev, const double *src_host, size_t n_doubles);   /* synchronous */
int anet_dev_download(anet_ctx *ctx, double *dst_host, const double *src_dev, size_t n_doubles); /* synchronous */

/* Recommended row stride for `batch` trajectories in the batch-minor layout: a multiple of 64 that
 * is >= batch and NOT a multiple of 4 KiB worth of doubles.  A power-of-two row stride puts the same
 * column of every field on the same HBM channel/bank; measured on MI355X the 8-segment solve runs
 * 59 % of the HBM roofline at ld = 2^20 and 63-65 % at ld = 2^20 + 576.      */
int64_t anet_recommended_ld(int64_t batch);

/* ---- layout helpers: trajectory-major host/device <-> batch-minor device ---------------- */
/* dst[f*ld + b] = src[b*nfield + f]  (both device pointers). */
int anet_to_batch_minor_dev(anet_ctx *ctx, int64_t batch, int64_t nfield, int64_t ld,
                            const double *src, double *dst, void *stream);
/* dst[b*nfield + f] = src[f*ld + b]. */
int anet_to_traj_major_dev(anet_ctx *ctx, int64_t batch, int64_t nfield, int64_t ld,
                           const double *src, double *dst, void *stream);

/* ---- MINCO coefficient solve + energy --------------------------------------------------- */
/* Replaces (north star; upstream GCOPTER minco.hpp, absent from the reference tree):
 *   MINCO_S{2,3,4}NU::setConditions(head,tail,N) + setParameters(inPs,ts) + getCoeffs +
 *   getEnergy, batched.  Fixed waypoints and durations; minimises int (p^(s))^2.
 * coeffs may be NULL (energy only); energy may be NULL.                                      */
int anet_minco_solve_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                         const double *head,  /* [3*c][ld]        */
                         const double *tail,  /* [3*c][ld]        */
                         const double *wps,   /* [(N-1)*3][ld]    (ignored when N == 1) */
                         const double *T,     /* [N][ld]          */
                         double *coeffs,      /* [N*3*2s][ld]     */
                         double *energy,      /* [batch]          */
                         void *stream);
/* Trajectories whose durations spread widely (max T / min T > min_spread; <= 1 selects all) solved again by the classic
 * formulation -- one 2sN x 2sN banded collocation system, LU with partial pivoting -- and their coeffs / energy
 * overwritten; the others are left as they are.  The fast kernel solves a reduced system whose conditioning is the
 * square of this one's: accurate to 1e-8 up to a spread of 100, 5e-4 at 10^3 (snap).  The host entry point
 * anet_minco_solve applies this by itself above a spread of 50; device callers call it when their durations can spread
 * (about 10^3 times slower per trajectory that is redone).                                                    */
int anet_minco_solve_wide_spread_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                     const double *head, const double *tail, const double *wps, const double *T,
                                     double min_spread, double *coeffs, double *energy, void *stream);
/* flags[b] (device int32 [batch]) = 1 where the durations of trajectory b spread over more than min_spread (max T >
 * min_spread * min T; min_spread <= 0 selects the library's own threshold of 50), 0 elsewhere: which trajectories the
 * call above redoes, and which results of anet_lbfgs_minco[_dev] had their returned coefficients re-solved that way. */
int anet_minco_spread_flags_dev(anet_ctx *ctx, int n_pieces, int64_t batch, int64_t ld, const double *T /* [N][ld] */,
                                double min_spread, int32_t *flags, void *stream);

/* Time-allocation sampling (north star: "batch of candidate trajectories / time-allocation samples"): MANY candidate
 * duration vectors for FEW problems in ONE launch.  A sampler does not have 1024 different problems -- the literal
 * BASELINE configs[1] batch, launch-bound at 2 MB per launch -- it has one problem (boundary states, waypoints) and K
 * candidate duration vectors.  Sample b (0 <= b < problems * samples_per_problem) belongs to problem b / samples_per_problem;
 * head / tail / wps are per PROBLEM (batch-minor with row stride ldp >= problems), T per SAMPLE (row stride ld); the only
 * output is cost[b] = int (p^(s))^2 + rho * sum T (rho = 0: the energy of MINCO_S*NU::getEnergy), 8 (N + 1) bytes of
 * traffic per sample instead of the 1920 of a full solve.  No counterpart in the reference (it solves one trajectory per
 * call, learning_planner.hpp:196); upstream's use is a loop over setParameters / getEnergy.
 * ACCURACY ENVELOPE: the costs come from the reduced (block-tridiagonal) system only -- there is no pivoted re-solve
 * here as anet_minco_solve_wide_spread_dev / anet_lbfgs_minco apply above a spread of 50.  Relative error of the energy:
 * <= 1e-8 while max T / min T of a sample stays below ~100, degrading beyond it (order 1e-4 at 10^3 for snap).  A sampler
 * that draws wider spreads should flag those samples with anet_minco_spread_flags_dev (same T layout) and re-evaluate
 * them with anet_minco_solve_wide_spread_dev.                                                                        */
int anet_minco_sample_costs_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t problems, int64_t samples_per_problem,
                                int64_t ld, int64_t ldp, const double *head /* [3c][ldp] */, const double *tail,
                                const double *wps /* [(N-1)*3][ldp] */, const double *T /* [N][ld] */, double rho,
                                double *cost /* [problems * samples_per_problem] */, void *stream);
/* Host variant for ONE problem: head / tail [3][c], wps [N-1][3], T [samples][N] (trajectory-major), cost [samples]. */
int anet_minco_sample_costs(anet_ctx *ctx, int s, int c, int n_pieces, int64_t samples, const double *head,
                            const double *tail, const double *wps, const double *T, double rho, double *cost);
int anet_minco_solve(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch,
                     const double *head,  /* [batch][3][c]     */
                     const double *tail,  /* [batch][3][c]     */
                     const double *wps,   /* [batch][N-1][3]   */
                     const double *T,     /* [batch][N]        */
                     double *coeffs,      /* [batch][N][3][2s] */
                     double *energy);     /* [batch]           */

/* ---- Trajectory<D> evaluation --------------------------------------------------------- */
/* Replaces Trajectory<D>::getPos/getVel/getAcc/getJer (gcopter/trajectory.hpp:516-538) =
 * locatePieceIdx (:496-514, including its clamp to the last piece for t beyond the total
 * duration) + Piece<D>::getPos/getVel/getAcc/getJer (:75-133), and network/utils/trajectory.py
 * get_pos/get_vel/get_acc (:47-98), batched: nq query times per trajectory.
 * deriv: 0 position, 1 velocity, 2 acceleration, 3 jerk.                                      */
int anet_traj_eval_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                       const double *coeffs, /* [N*3*2s][ld] */
                       const double *T,      /* [N][ld]      */
                       int nq, const double *tq, /* [nq][ld] absolute times from trajectory start */
                       int deriv, double *out,   /* [nq*3][ld]  (query-major, then axis) */
                       void *stream);
int anet_traj_eval(anet_ctx *ctx, int s, int n_pieces, int64_t batch,
                   const double *coeffs, /* [batch][N][3][2s] */
                   const double *T,      /* [batch][N]        */
                   int nq, const double *tq, /* [batch][nq]   */
                   int deriv, double *out);  /* [batch][nq][3] */

/* Replaces Trajectory<D>::getTrajCost(order) (gcopter/trajectory.hpp:354-427):
 * sum over pieces and axes of 1/2 z' Q_s(T) z on the s highest coefficients.  m34 is the (3,4)
 * entry constant of the snap block: 1400.0 reproduces the reference (qp_solver.hpp:212,
 * trajectory.hpp:385, min_traj_opt.py:493), 1440.0 is the true integral.  Ignored for s != 4. */
int anet_traj_cost_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                       const double *coeffs, const double *T, double m34, double *cost /* [batch] */,
                       void *stream);
int anet_traj_cost(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                   const double *T, double m34, double *cost);
/* d(getTrajCost)/dT_i with the coefficients held fixed: 1/2 z_i' (dQ_s/dT)(T_i) z_i summed over axes.
 * This is the time-allocation gradient the reference's training actually back-propagates: in
 * OsqpLayer the QP solution is a detached leaf (network/utils/learning/layers.py:121,222), so the loss
 * 1/2 z'Q(T)z / path_length (:143-147, :245) reaches the segment times only through Q(T); the KKT hook
 * (:136-141, :238-243) rewrites the gradient of that leaf and never reaches the network.
 * gradT: [N][ld] (dev) / [batch][N] (host).                                                        */
int anet_traj_cost_grad_T_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                              const double *coeffs, const double *T, double m34, double *gradT, void *stream);
int anet_traj_cost_grad_T(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                          const double *T, double m34, double *gradT);

/* Replaces Piece<D>::getMaxVelRate / getMaxAccRate (gcopter/trajectory.hpp:177-273; root isolation in
 * gcopter/root_finder.hpp) batched: the maximum of ||v|| (which = 1) or ||a|| (which = 2) over every piece.
 * Trajectory<D>::getMaxVelRate/getMaxAccRate (:576-604) is the maximum over the pieces, and
 * checkMaxVelRate/checkMaxAccRate(bound) (:275-314, 606-630) is "max < bound" (both ends and interior).
 * rate: [N][ld] (dev) / [batch][N] (host).                                                          */
int anet_traj_max_rate_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                           const double *coeffs, const double *T, int which, double *rate, void *stream);
int anet_traj_max_rate(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const double *coeffs,
                       const double *T, int which, double *rate);

/* Replaces Piece<D>::normalizePosCoeffMat / normalizeVelCoeffMat / normalizeAccCoeffMat (gcopter/trajectory.hpp:135-171) for a
 * batch of independent pieces: the coefficients of the position (deriv = 0), velocity (1) or acceleration (2) polynomial of each
 * piece in normalised time tau = t / duration, highest power first: column i = (falling factorial of its power) * coeffMat.col(i) *
 * duration^power.  coeffs [pieces][3][2s], T [pieces], out [pieces][3][2s - deriv]; HOST pointers (the _dev variant: device
 * pointers, the same piece-major layout, asynchronous on `stream`). */
int anet_piece_normalized_coeffs(anet_ctx *ctx, int s, int64_t pieces, const double *coeffs, const double *T, int deriv,
                                 double *out);
int anet_piece_normalized_coeffs_dev(anet_ctx *ctx, int s, int64_t pieces, const double *coeffs, const double *T, int deriv,
                                     double *out, void *stream);

/* ---- cost + analytic gradients ----------------------------------------------------------- */
/* Penalty functional on the reference's own inequality rows (QPSolver::solve step three,
 * planner/qp_solver.hpp:244-296; MinTrajOpt.fill_ineq, network/utils/min_traj_opt.py:535-613):
 * for piece i and sample j in [0,res) at t = j*T_i/res
 *     corridor rows   a_r . p(t) - b_r            (hPolys[i] rows, a.x <= b form)
 *     box rows        +-v_axis(t) - max_vel,  +-a_axis(t) - max_acc     (per axis)
 * the reference imposes them as hard constraints of a QP; here each row g enters the cost as
 *     J_pen = sum_i (T_i/res) sum_j sum_rows w_row * smoothedL1(smooth_mu, g)
 * with firi::smoothedL1 (gcopter/firi.hpp:60-84), the MINCO/GCOPTER penalty-functional form the
 * north star asks for.  Total cost: J = int (p^(s))^2 + rho*sum(T) + J_pen.                    */
typedef struct anet_penalty {
  double rho;        /* weight of sum(T)                                               */
  double w_corridor; /* weight of corridor rows                                        */
  double w_vel;      /* weight of velocity box rows                                    */
  double w_acc;      /* weight of acceleration box rows                                */
  double smooth_mu;  /* smoothedL1 mu, > 0                                             */
  double max_vel;    /* MaxVelBox (config/planner.yaml:17)                             */
  double max_acc;    /* MaxAccBox (config/planner.yaml:19)                             */
  int32_t res;       /* ConstRes, samples per piece (config/planner.yaml:21)           */
  int32_t poly_rows; /* M: rows per polytope in `hpolys`; all-zero rows are padding    */
} anet_penalty;

/* Partial gradients of  [with_energy] int (p^(s))^2  +  [pen != NULL] J_pen  with respect to the
 * coefficients and durations, the coefficients being treated as independent variables
 * (MINCO getEnergyPartialGradByCoeffs / getEnergyPartialGradByTimes + the penalty functional).
 * hpolys: [N*M*4][ld], row r of polytope i at fields (i*M + r)*4 + {0,1,2,3} = a_x,a_y,a_z,b;
 * may be NULL (box rows only).  piece_cost (may be NULL): per-piece J_pen share, [N][ld].      */
int anet_minco_partial_grads_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int64_t ld,
                                 const double *coeffs, const double *T, const double *hpolys,
                                 const anet_penalty *pen, int with_energy,
                                 double *gdC,        /* [N*3*2s][ld] */
                                 double *gdT,        /* [N][ld]      */
                                 double *piece_cost, /* [N][ld]      */
                                 void *stream);

/* MINCO propogateGrad: total gradient of a scalar J(c(wps,T), T) w.r.t. the interior waypoints and
 * the durations from its partial gradients gdC, gdT; coeffs must be the MINCO coefficients for
 * (c, wps, T).  gradP: [(N-1)*3][ld] (waypoint-major, like wps), gradT: [N][ld].               */
int anet_minco_propagate_grad_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                  const double *T, const double *coeffs, const double *gdC,
                                  const double *gdT, double *gradP, double *gradT, void *stream);

/* Whole objective in one call: solve -> partial gradients -> propagate.
 * work: device scratch of anet_minco_cost_grad_workspace(s, N, ld) doubles.
 * coeffs_out may be NULL.  cost: [batch].                                                     */
int64_t anet_minco_cost_grad_workspace(int s, int n_pieces, int64_t ld);
/* Kernel launches anet_minco_cost_grad_dev makes for this shape (order s, boundary count c) on this context's device: 1
 * (k_minco_cost_grad_fused: batches of up to three rounds of one workgroup per compute unit, orders 3 and 4, res <= 64 -- six /
 * eight rounds for 8-piece snap / 16-piece jerk with c = 3 at res = 20, whose phase 2 runs on the FP64 matrix instructions) or 3
 * (k_minco_solve -> k_piece_grad[_mx] -> k_minco_propagate); negative = error.  For callers that label a measurement with the
 * kernel that ran (bench.py).                                                                                                  */
int anet_minco_cost_grad_launches(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const anet_penalty *pen);
/* The launch shape anet_minco_partial_grads_dev picks for this shape on this context's device: 0 one lane per (trajectory,
 * piece) (k_piece_grad, large batches); 1 two lanes per pair; 2 two lanes and the samples over a workgroup's four waves (the small
 * batches); 3 k_piece_grad_mx -- four lanes per pair, the contractions with the basis table on the FP64 matrix instructions
 * (large batches, orders 3 and 4, res = 20: the penalty functional of qp_solver.hpp:244-296's rows at planner.yaml:21's sampling);
 * negative = error.  For callers that label a measurement with the kernel that ran (bench.py).                                */
int anet_minco_piece_grad_shape(anet_ctx *ctx, int s, int n_pieces, int64_t batch, const anet_penalty *pen);
int anet_minco_cost_grad_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                             const double *head, const double *tail, const double *wps,
                             const double *T, const double *hpolys, const anet_penalty *pen,
                             double *work, double *cost, double *gradP, double *gradT,
                             double *coeffs_out, void *stream);
int anet_minco_cost_grad(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch,
                         const double *head, const double *tail, const double *wps, const double *T,
                         const double *hpolys, /* [batch][N][M][4] or NULL */
                         const anet_penalty *pen, double *cost, /* [batch] */
                         double *gradP,  /* [batch][N-1][3] */
                         double *gradT,  /* [batch][N]      */
                         double *coeffs_out /* [batch][N][3][2s] or NULL */);

/* ---- QP assembly (the reference's own formulation) ----------------------------------------- */
/* Replaces the assembly part of QPSolver::solve (planner/qp_solver.hpp:61-296: setOrder/zero_A_,
 * get_t_state, equality rows :139-177, objective :180-242, inequality rows :244-296) and its Python
 * twin MinTrajOpt.fill_eq_obj / fill_ineq (network/utils/min_traj_opt.py:300-697), batched.
 * Dense, trajectory-major outputs with the reference's shapes:
 *   n = 3*2s*N variables, m_e = 3*(6 + s*(N-1)) equalities, m_g = res*(sum_i rows_i + 12*N) inequalities
 *   Q [n][n], A [m_e][n], b [m_e], G [m_g][n], h [m_g]      (row-major, one set per trajectory)
 * row_order:  ANET_QP_ORDER_CPP    per piece, per sample: the polytope rows, then 12 box rows
 *                                  (qp_solver.hpp:258-294)
 *             ANET_QP_ORDER_PYTHON all corridor rows (G1,h1) first, then all box rows (G2,h2)
 *                                  (min_traj_opt.py:535-613; OsqpLayer stacks them, layers.py:66-70)
 * float_time != 0 reproduces the C++ planner's arithmetic: segment times arrive as float32 and the
 * time powers of the basis rows and of the cost block are formed in float (qp_solver.hpp:90-116 with
 * T = float, :183-236, :252-263); 0 = float64 throughout (the Python twin).
 * m34: (3,4) entry constant of the snap cost block, 1400.0 = reference, 1440.0 = true integral.
 * state: ini/fin PVA, [batch][2][3][3] = {ini,fin} x axis x (p,v,a) (learning_planning.cpp:150-151).
 * hpolys [batch][N][M][4] rows (a_x,a_y,a_z,b) meaning a.x <= b; rows[batch][N] = valid rows per
 * polytope (<= M).  All pointers are DEVICE pointers in the _dev variant.                        */
#define ANET_QP_ORDER_CPP 0
#define ANET_QP_ORDER_PYTHON 1
typedef struct anet_qp_dims { int64_t n, m_e, m_g; } anet_qp_dims;
/* Sizes for ONE trajectory given its polytope row counts (host array rows[N]). */
int anet_qp_dims_of(int s, int n_pieces, int res, const int32_t *rows, anet_qp_dims *out);
/* Every trajectory of a batch must have the same sum of polytope rows (so the dense outputs have one
 * shape); pad polytopes with all-zero rows (0.x <= 0, inert) to equalise, as the reference's own
 * tensor packing does (learning_planner.hpp:157-166, 50 x 4 x 5 zero-padded).                     */
int anet_qp_assemble_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M,
                         double max_vel, double max_acc, double m34, int float_time, int row_order,
                         const double *state, const double *T, const double *hpolys, const int32_t *rows,
                         double *Q, double *A, double *b, double *G, double *h, void *stream);
int anet_qp_assemble(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                     double max_acc, double m34, int float_time, int row_order, const double *state,
                     const double *T, const double *hpolys, const int32_t *rows, double *Q, double *A,
                     double *b, double *G, double *h);

/* ---- inequality-constrained QP solve (replaces OSQP) ------------------------------------- */
/* Replaces the solver part of QPSolver::solve (planner/qp_solver.hpp:299-358: OsqpEigen data/settings,
 * initSolver, solveProblem, getObjValue, getStatus, getSolution) and OsqpLayer's forward solve
 * (network/utils/learning/layers.py:66-81,167-181), batched: the SAME QP the reference assembles
 *      min 1/2 z'Qz   s.t.  A z = b,   G z <= h                      (anet_qp_assemble gives it densely)
 * OSQP itself is a third-party dependency absent from the reference tree: its iterates are not comparable, the
 * solution is (same convex problem) -- parity is checked through the KKT conditions.  Two methods:
 *   INTERIOR_POINT (default): the optimum to 1e-6 in 10-20 Newton steps.  On random corridor problems it returns
 *       `Solved` for every problem either method can solve (profiles/r02_qp_unsolved.json);
 *   ADMM: OSQP's own iteration (Stellato et al. 2020, Algorithm 1) with OSQP's default settings (below), its modified Ruiz
 *       equilibration (`scaling` = 10 passes on the reference's own matrices, cost scaling included), its stopping rule on the
 *       UNSCALED residuals and its rho estimate from the scaled ones.  At max_iter = 4000 it leaves 1-6 % of feasible 5- and
 *       8-piece snap problems unsolved (profiles/r05_qp_unsolved_admm.json), which QPSolver::solve's caller treats as a failed
 *       plan (qp_solver.hpp:334-352) -- hence not the default.  OSQP itself is not in the image: its iterates remain unpinned.
 * The solve never forms Q, A, G: see allocnet_amd/csrc/qp_ipm.h, qp_admm.h.                          */
typedef struct anet_qp_settings {
  double rho;        /* 0.1   OSQP default; equality rows use 1e3*rho like OSQP                 */
  double sigma;      /* 1e-6                                                                    */
  double alpha;      /* 1.6                                                                     */
  double eps_abs;    /* 1e-3  (the reference never changes it: qp_solver.hpp:301-302, layers.py:79) */
  double eps_rel;    /* 1e-3                                                                    */
  int32_t max_iter;  /* 4000                                                                    */
  int32_t check_termination; /* 25                                                              */
  int32_t adaptive_rho_interval; /* 100; 0 disables (OSQP picks its interval from wall-clock time) */
  int32_t scaled_termination;    /* 0: OSQP's rule -- residuals of the reference's own (unscaled) QP;
                                    1: residuals of the internally normalised QP (like OSQP's
                                    scaled_termination): ~2-3x fewer iterations, looser on stiff problems */
  int32_t method;                /* ANET_QP_METHOD_ADMM: OSQP's algorithm, settings above.
                                    ANET_QP_METHOD_INTERIOR_POINT (default): primal-dual interior point on the same QP in
                                    Hermite node coordinates (allocnet_amd/csrc/qp_ipm.h) -- the optimum to
                                    min(eps, 1e-6) in 10-20 Newton steps; uses only eps_abs/eps_rel and max_iter
                                    (capped at 200) of the fields above; infeasible problems are reported
                                    PRIMAL_INFEASIBLE when the iteration diverges, MAX_ITER_REACHED otherwise */
} anet_qp_settings;
#define ANET_QP_METHOD_ADMM 0
#define ANET_QP_METHOD_INTERIOR_POINT 1
void anet_qp_default_settings(anet_qp_settings *s);
#define ANET_QP_SOLVED 1          /* OSQP_SOLVED                 */
#define ANET_QP_MAX_ITER_REACHED (-2) /* OSQP_MAX_ITER_REACHED; the reference treats anything but Solved as failure (qp_solver.hpp:346-350) */
#define ANET_QP_PRIMAL_INFEASIBLE (-3) /* OSQP_PRIMAL_INFEASIBLE: OSQP's certificate test on y(k+1)-y(k), eps_prim_inf 1e-4 */
#define ANET_QP_UNSOLVED (-10)    /* OSQP_UNSOLVED: a problem no workgroup took (a launch_order that skips it); obj = NaN, iters = 0 */
/* hpolys [batch][N][M][4] rows a.x <= b with all-zero rows as inert padding.
 * coeffs [batch][N][3][2s] (the flatten order callModel unpacks, learning_planner.hpp:212,227),
 * obj [batch] = 1/2 z'Qz (QPSolver::getObjCost), status/iters [batch], residuals [batch][2] (primal, dual;
 * may be NULL).  All HOST pointers; the _dev variant takes DEVICE pointers (same trajectory-major
 * layout) plus a device workspace of anet_qp_solve_workspace() doubles (the slacks and multipliers of every row, and for the
 * two-launch form of large batches the parked iterates and the order of the second launch: always ask, never compute it). */
int anet_qp_solve(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                  double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                  const anet_qp_settings *settings, double *coeffs, double *obj, int32_t *status,
                  int32_t *iters, double *residuals);
int64_t anet_qp_solve_workspace(int s, int n_pieces, int64_t batch, int res, int M);
int anet_qp_solve_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                      double max_acc, double m34, const double *state, const double *T,
                      const double *hpolys, const anet_qp_settings *settings, double *work, double *coeffs,
                      double *obj, int32_t *status, int32_t *iters, double *residuals, void *stream);

/* The same with a launch order for the interior-point method (one workgroup per problem, 512 resident at a time: a batch ends
 * with whichever long problem started late -- 27 % above its balanced figure for 4096 problems): launch_order
 * (device, int32 [batch], a permutation of 0..batch-1) is the problem each successive workgroup takes; longest first from the
 * iters[] of a previous solve of the same or a similar batch (anet_launch_order_from_steps_dev) -- the re-solve of a receding-
 * horizon planner or a sampler.  Results are bit-identical for any order; NULL = as given; the ADMM method ignores it.
 * Entries are not checked on the host (device memory): an entry outside [0, batch) is skipped and a problem no entry names
 * reports status ANET_QP_UNSOLVED, iters 0 and obj NaN (its coeffs are left untouched).  A REPEATED entry is undefined
 * behaviour for the problem it names: two workgroups then run on the same per-problem slack / multiplier state and write
 * the same coeffs / status concurrently (a race, not a redundant solve) -- the order must be a permutation.
 * No reference counterpart (QPSolver::solve takes one problem: qp_solver.hpp:119).                                        */
int anet_qp_solve_ordered_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                              double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                              const anet_qp_settings *settings, const int32_t *launch_order, double *work, double *coeffs,
                              double *obj, int32_t *status, int32_t *iters, double *residuals, void *stream);

/* The same solve plus grad_T [batch][N] = d(obj)/dT_i, the derivative of the OPTIMAL cost 1/2 z*'Q z*
 * with respect to the segment durations -- the "time-allocation gradient" the reference's training loop
 * is after (network/layers.py:120-147 installs a -J^-1 grad KKT hook for it, a dense (n+m)^2 solve per
 * sample).  Because the loss IS the QP objective, no KKT solve is needed: by the
 * envelope theorem the derivative is dL/dT at the optimum, assembled inside the solve kernel (either method)
 * from the solution and its multipliers (allocnet_amd/csrc/qp_admm.h, qp_ipm.h).  Exact at the optimum of a problem with a
 * stable active set; its accuracy follows the solve tolerance.  Not what the reference's autograd
 * delivers today (its z is a detached leaf: that quantity is anet_traj_cost_grad_T) -- see DESIGN.md 8b. */
int anet_qp_solve_time_grad(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                            double max_acc, double m34, const double *state, const double *T,
                            const double *hpolys, const anet_qp_settings *settings, double *coeffs, double *obj,
                            int32_t *status, int32_t *iters, double *residuals, double *grad_T);
int anet_qp_solve_time_grad_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                                double max_acc, double m34, const double *state, const double *T,
                                const double *hpolys, const anet_qp_settings *settings, double *work,
                                double *coeffs, double *obj, int32_t *status, int32_t *iters, double *residuals,
                                double *grad_T, void *stream);

/* Backward pass through the QP (replaces the KKT hook of network/utils/learning/layers.py:129-141,
 * 230-243: grad <- -J^-1 grad with J = [[Q, G'diag(lambda), A'], [G, diag(Gz-h), 0], [A, 0, 0]], a dense (n+m)^2 solve
 * per sample whose result stops at the detached leaf z).  Solves the QP (interior-point method only) and returns,
 * for a caller-supplied gradient grad_z = d loss / d z* (layout of `coeffs`), grad_T [batch][N] = d loss / d T:
 * the adjoint is taken at the optimum with the method's own block-tridiagonal Newton matrix (Hermite coordinates:
 * the equality block is built in, the inequality rows enter through lambda / s) and contracted in closed form with
 * the dependence of the cost blocks, the continuity scalings, the pinned end states, the box bounds and the
 * coefficient scaling on the durations.  Any smooth loss of the optimal coefficients can be differentiated this way;
 * for loss = the QP objective it reproduces anet_qp_solve_time_grad (minus the explicit 1/2 z'(dQ/dT)z part, which
 * anet_traj_cost_grad_T gives).  settings as in anet_qp_solve (the tolerance is tightened to 1e-9).            */
int anet_qp_solve_vjp(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                      double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                      const anet_qp_settings *settings, const double *grad_z /* [batch][N][3][2s] */, double *coeffs,
                      double *obj, int32_t *status, int32_t *iters, double *residuals, double *grad_T /* [batch][N] */);
int anet_qp_solve_vjp_dev(anet_ctx *ctx, int s, int n_pieces, int64_t batch, int res, int M, double max_vel,
                          double max_acc, double m34, const double *state, const double *T, const double *hpolys,
                          const anet_qp_settings *settings, const double *grad_z, double *work, double *coeffs,
                          double *obj, int32_t *status, int32_t *iters, double *residuals, double *grad_T,
                          void *stream);

/* ---- polytope depth: geo_utils::findInterior / geo_utils::overlap, sfc_gen::shortCut --------------------------- */
/* Replaces the 4-variable linear programme of geo_utils::findInterior and geo_utils::overlap
 * (src/planner/include/gcopter/geo_utils.hpp:43-85, solved there by sdlp::linprog<4>), batched: for each polytope
 * (rows h: h0 x + h1 y + h2 z + h3 <= 0, all-zero rows are padding up to max_rows)
 *     depth = max t  s.t.  n.x + t <= -h3  for every row,   n = h[0:3] / |h[0:3]| (normalise = 1, findInterior)
 *                                                            or h[0:3]            (normalise = 0, overlap)
 * and the point x that attains it (point may be NULL).  findInterior(hPoly) is depth > 0; overlap(hPoly0, hPoly1, eps) is
 * depth > eps of the two polytopes' rows stacked (sfc_gen::shortCut, sfc_gen.hpp:188-226, calls it with eps = 0.1);
 * depth = -inf for a polytope of padding rows only, +inf for an unbounded one (the reference's tests read both as false).
 * Exact: an active-set ascent, one lane per polytope, whose result is certified (feasible, multipliers >= 0) before it is
 * returned; what it cannot certify falls back to the enumeration of the vertices (allocnet_amd/csrc/firi_kernels.h;
 * C(rows, 4) candidates, hence max_rows <= 256: ANET_ERR_UNSUPPORTED beyond).
 * Bounded polytopes (corridors always carry their bounding box).                                                       */
int anet_polytope_depth(anet_ctx *ctx, int64_t batch, int max_rows, const double *hpoly /* [batch][max_rows][4] */,
                        int normalise, double *depth /* [batch] */, double *point /* [batch][3] or NULL */);
int anet_polytope_depth_dev(anet_ctx *ctx, int64_t batch, int max_rows, const double *hpoly, int normalise,
                            double *depth, double *point, void *stream);

/* ---- batched L-BFGS ------------------------------------------------------------------------ */
/* lbfgs::lbfgs_parameter_t, same fields and defaults (gcopter/lbfgs.hpp:15-129). */
typedef struct anet_lbfgs_params {
  int32_t mem_size;       /* 8      */
  double g_epsilon;       /* 1e-5   */
  int32_t past;           /* 3      */
  double delta;           /* 1e-6   */
  int32_t max_iterations; /* 0 = until convergence */
  int32_t max_linesearch; /* 64     */
  double min_step;        /* 1e-20  */
  double max_step;        /* 1e+20  */
  double f_dec_coeff;     /* 1e-4   */
  double s_curv_coeff;    /* 0.9    */
  double cautious_factor; /* 1e-6   */
  double machine_prec;    /* 1e-16  */
} anet_lbfgs_params;
void anet_lbfgs_default_params(anet_lbfgs_params *p);
/* lbfgs_optimize's own parameter validation (lbfgs.hpp:449-495): 0 or the LBFGSERR_INVALID_* code. */
int anet_lbfgs_check_params(int n, const anet_lbfgs_params *p);
/* lbfgs::lbfgs_strerror (lbfgs.hpp:724-799). */
const char *anet_lbfgs_strerror(int code);

/* Per problem the control flow, line search (Lewis-Overton weak Wolfe), cautious update, stopping
 * tests and return codes are those of lbfgs::lbfgs_optimize / line_search_lewisoverton
 * (gcopter/lbfgs.hpp:276-384, 434-717); the batch advances one objective evaluation per step, each
 * problem in its own state.  status[b] is lbfgs_optimize's return value (0 convergence, 1 stop,
 * negative LBFGSERR_*), iters[b] its iteration counter k, evals[b] the number of objective
 * evaluations.  max_evals (> 0) bounds the evaluations per problem; problems still running then
 * report status ANET_LBFGS_RUNNING.                                                            */
#define ANET_LBFGS_RUNNING 2147483647

/* Objective = firi::costMVIE (gcopter/firi.hpp:86-157), the reference's only L-BFGS call site
 * (firi::maxVolInsEllipsoid, firi.hpp:207-227).  A: per problem the M x 3 matrix COLUMN-major, as the
 * reference packs optData (firi.hpp:186-200); x: 9 variables [p, sqrt-diag, off-diag], in/out.   */
int anet_lbfgs_mvie(anet_ctx *ctx, int64_t batch, int M, const double *A /* [batch][3*M] */,
                    double smooth_eps, double penalty_wt, double *x /* [batch][9] */,
                    double *f /* [batch] */, const anet_lbfgs_params *params, int max_evals,
                    int32_t *status, int32_t *iters, int32_t *evals);

/* lbfgs::lbfgs_optimize for an objective the CALLER evaluates on the device (lbfgs.hpp:434-440: x, f, proc_evaluate,
 * proc_stepbound, proc_progress, instance, param).  proc_evaluate (lbfgs.hpp:186-219) becomes a host function that ENQUEUES
 * the evaluation of the whole batch on `stream`: given x ([n][ld] device, batch-minor: variable i of problem b at x[i*ld + b])
 * it must leave f[b] and g[i*ld + b] for every problem b < batch (the values of problems that have stopped are ignored) and
 * return 0; anything else aborts the run with ANET_ERR_INVALID.  The call synchronises `stream` as it goes (completion polls) and at
 * its end: x, f, status, iters and evals are complete on return, on whatever stream the caller reads them.  It is called once per evaluation step of the batch (the
 * lockstep shape: every running problem consumes one evaluation per call; a completion flag is polled every eight steps, so
 * up to eight calls may follow the last problem's stop).  x is the start point in, the result out; f and g are the caller's
 * buffers the callback fills ([batch] and [n][ld]); on return f[b] is lbfgs_optimize's fx.  proc_stepbound: the one bound
 * with a use is built in -- bound_from < n keeps the variables i >= bound_from at or above bound_min within every line search
 * (lbfgs.hpp:557-565; bound_from >= n: none); proc_progress: anet_set_cancel_flag.  work: anet_lbfgs_workspace() doubles.  */
typedef int (*anet_lbfgs_evaluate_t)(void *instance, const double *x, double *f, double *g, int64_t batch, int64_t ld, int n,
                                     void *stream);
int64_t anet_lbfgs_workspace(int n, int64_t ld, const anet_lbfgs_params *params);
int anet_lbfgs_optimize_dev(anet_ctx *ctx, int n, int64_t batch, int64_t ld, double *x, double *f, double *g,
                            anet_lbfgs_evaluate_t proc_evaluate, void *instance, const anet_lbfgs_params *params,
                            int max_evals, int bound_from, double bound_min, double *work, int32_t *status,
                            int32_t *iters, int32_t *evals, void *stream);

/* lbfgs::lbfgs_optimize for an objective evaluated ON THE HOST, with the reference's three callbacks (lbfgs.hpp:186-246, called
 * as lbfgs.hpp:434-717 calls them): one problem, x [n] host memory (start point in, result out), *f = fx on return, *ret =
 * lbfgs_optimize's return value (LBFGS_CONVERGENCE / LBFGS_STOP / LBFGS_CANCELED / LBFGSERR_*, parameter errors included),
 * *iters = k, *evals = objective evaluations (iters / evals may be NULL).
 *   proc_evaluate (required): returns f(x) and fills g [n]                                    -- lbfgs_evaluate_t
 *   proc_stepbound (may be NULL): upper bound of the step along d from xp                    -- lbfgs_stepbound_t, lbfgs.hpp:557-565
 *   proc_progress (may be NULL): called after every successful line search with x, g, fx, step, k, ls;
 *                 a non-zero return ends the run with LBFGS_CANCELED                        -- lbfgs_progress_t, lbfgs.hpp:580-587
 * The optimiser's vectors and all its arithmetic (line-search bookkeeping, cautious update, two-loop recursion, stopping
 * tests) stay on the device -- the lockstep update kernel with a batch of one -- and the state machine parks where the
 * reference calls back: per evaluation x goes to the host and f, g come back; per line search xp and d go to the host and the
 * bound comes back.  There is no CPU optimiser in the library.  A PCIe round trip per evaluation: for objectives that can be
 * evaluated on the device use anet_lbfgs_optimize_dev.  The call returns ANET_OK when the run ended by lbfgs_optimize's own rules,
 * whatever *ret says. */
typedef double (*anet_lbfgs_host_evaluate_t)(void *instance, const double *x, double *g, int n);
typedef double (*anet_lbfgs_host_stepbound_t)(void *instance, const double *xp, const double *d, int n);
typedef int (*anet_lbfgs_host_progress_t)(void *instance, const double *x, const double *g, double fx, double step, int k,
                                          int ls, int n);
int anet_lbfgs_optimize_host(anet_ctx *ctx, int n, double *x, double *f, anet_lbfgs_host_evaluate_t proc_evaluate,
                             anet_lbfgs_host_stepbound_t proc_stepbound, anet_lbfgs_host_progress_t proc_progress,
                             void *instance, const anet_lbfgs_params *params, int32_t *ret, int32_t *iters, int32_t *evals);

/* Objective = the MINCO cost  int (p^(s))^2 + rho*sum(T) + J_pen  (anet_minco_cost_grad) over the
 * interior waypoints (opt_flags bit 0) and/or the durations (bit 1), the durations through the
 * smooth bijection T(tau) so the problem is unconstrained.  wps and T are in/out.
 * Accuracy: inside the loop the cost and its gradient come from the reduced (Hermite / block-tridiagonal) system, whose
 * error grows with the spread of the durations (1e-8 up to max T / min T = 100, 5e-4 on snap coefficients at 10^3:
 * DESIGN.md section 2) -- the optimiser is free to walk there when ANET_OPT_TIMES is set.  What is RETURNED does not
 * carry that envelope: coeffs_out of a problem whose optimised durations spread over more than 50 are re-solved by the
 * pivoted collocation solve (anet_minco_solve_wide_spread_dev); anet_minco_spread_flags_dev on the returned T tells
 * which problems those were.                                                                     */
#define ANET_OPT_WAYPOINTS 1
#define ANET_OPT_TIMES 2
/* Execution shape (same result up to rounding): by default problems that fit one wavefront (orders 3 and 4, at most
 * 64 variables, mem_size <= 8, past <= 64) run as ONE launch, one wave per problem, each until its
 * own stop (lbfgs.hpp:551-709 is one loop per problem); this bit forces the launch-per-evaluation kernels instead,
 * which advance the whole batch in lockstep (up to twice the throughput per evaluation step at batches of 10^5, so the
 * better shape for a small FIXED evaluation budget there; a run to convergence is faster in one launch at any batch).
 * Batches of >= 4096 problems with >= 36 variables take the wave-per-problem shape in TWO launches -- 1000 evaluations of
 * every problem, then the unfinished ones resumed longest-expected first -- with bit-identical results.   */
#define ANET_OPT_LOCKSTEP 4
int anet_lbfgs_minco(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                     const double *tail, double *wps, double *T, const double *hpolys,
                     const anet_penalty *pen, const anet_lbfgs_params *params, int opt_flags,
                     int max_evals, double *cost, double *coeffs_out, int32_t *status, int32_t *iters,
                     int32_t *evals);
/* Device variant: batch-minor device arrays, workspace from anet_lbfgs_minco_workspace() doubles.
 * status/iters/evals are device int32 arrays [batch].  The one-launch shape only enqueues kernels on `stream` (no
 * host-side test for completion, no allocation: it can be captured into a hipGraph); the lockstep shape synchronises
 * the stream every few evaluations to test for completion.                                         */
int64_t anet_lbfgs_minco_workspace(int s, int n_pieces, int64_t ld, const anet_lbfgs_params *params);
int anet_lbfgs_minco_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                         const double *head, const double *tail, double *wps, double *T,
                         const double *hpolys, const anet_penalty *pen,
                         const anet_lbfgs_params *params, int opt_flags, int max_evals, double *work,
                         double *cost, double *coeffs_out, int32_t *status, int32_t *iters,
                         int32_t *evals, void *stream);
/* The same with a launch order for the one-launch shape: launch_order (device, int32 [batch], a permutation of
 * 0..batch-1, or NULL) names the problem each successive workgroup takes.  Results do not depend on it (problems are
 * independent); the run time does: a batch larger than the 2048 waves the device holds ends with whichever problem
 * started late and runs long, so handing over the problems longest-first shortens the run (4096 x 16-segment jerk:
 * 0.19 s as given, 0.12 s by the true evaluation counts, 0.14 s by the counts of a previous solve of a perturbed copy
 * of the batch -- the re-solve case of a sampler or a receding-horizon planner: feed evals[] of the last call,
 * sorted descending).  Entries are not checked beyond their range: an out-of-range entry is skipped and a problem no
 * entry names is left unsolved (ANET_LBFGS_RUNNING, zero counters); a REPEATED entry is undefined behaviour for the
 * problem it names (two waves race on its state) -- the order must be a permutation.  Ignored by the lockstep shape.  */
int anet_lbfgs_minco_ordered_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                 const double *head, const double *tail, double *wps, double *T,
                                 const double *hpolys, const anet_penalty *pen,
                                 const anet_lbfgs_params *params, int opt_flags, int max_evals,
                                 const int32_t *launch_order, double *work, double *cost, double *coeffs_out,
                                 int32_t *status, int32_t *iters, int32_t *evals, void *stream);
/* The same with lbfgs_optimize's step bound (lbfgs.hpp:221-224 lbfgs_stepbound_t, applied as lbfgs.hpp:557-565:
 * step_max = min(proc_stepbound(xp, d), max_step); step = step < step_max ? step : step_max / 2; the line search then runs
 * with that step_max) -- a host callback cannot run inside the kernels, so the one bound with a use is built in: a MINIMUM
 * DURATION.  min_duration > 0 bounds every line search to the largest step along the search direction that keeps every
 * duration variable tau_i >= backward_T(min_duration), i.e. T_i >= min_duration: 1 / max_i(-d_i / (tau_i - tau_min)) over the
 * duration variables that move down.  min_duration = 0: no bound (identical to the calls above).  Both execution shapes
 * apply it (the one-launch kernel and, with ANET_OPT_LOCKSTEP or problems that do not fit a wave, the per-evaluation
 * update kernels).  Start durations below the minimum make the first bound 0 and the run end with
 * LBFGSERR_INVALIDPARAMETERS, as the reference's loop would.                                                          */
int anet_lbfgs_minco_bounded(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, const double *head,
                             const double *tail, double *wps, double *T, const double *hpolys,
                             const anet_penalty *pen, const anet_lbfgs_params *params, int opt_flags,
                             int max_evals, double min_duration, double *cost, double *coeffs_out, int32_t *status,
                             int32_t *iters, int32_t *evals);
int anet_lbfgs_minco_bounded_dev(anet_ctx *ctx, int s, int c, int n_pieces, int64_t batch, int64_t ld,
                                 const double *head, const double *tail, double *wps, double *T,
                                 const double *hpolys, const anet_penalty *pen, const anet_lbfgs_params *params,
                                 int opt_flags, int max_evals, double min_duration, const int32_t *launch_order, double *work,
                                 double *cost, double *coeffs_out, int32_t *status, int32_t *iters, int32_t *evals,
                                 void *stream);

/* lbfgs_optimize's progress callback (lbfgs.hpp:226-246 lbfgs_progress_t, called at lbfgs.hpp:580-587 after every
 * successful line search; a non-zero return ends the run with LBFGS_CANCELED) -- a host callback cannot run inside the
 * kernels, so its one effect is offered as a word the caller owns: `flag` (device-visible int32 -- device memory written from
 * another stream, or mapped pinned host memory; NULL: none) is read once per evaluation by every problem of the one-launch
 * MINCO L-BFGS calls (either execution shape) that follow on this context; while it is non-zero a problem stops after the iteration it is in, at
 * the point lbfgs.hpp:583 would, with status LBFGS_CANCELED (2), its iterate, cost and counters as they stand.  Problems
 * that stopped on their own keep their status.  Both shapes of the MINCO L-BFGS look at it (the lockstep update kernels
 * read it once per problem and evaluation); the MVIE objective does not.                                               */
int anet_set_cancel_flag(anet_ctx *ctx, const int32_t *flag);

/* launch_order for the call above from the evals[] of a previous solve of the same or a similar batch: longest first, in
 * buckets of 16 evaluations (device arrays; work: ANET_LAUNCH_ORDER_WORK_INTS int32 of scratch; asynchronous on `stream`). */
#define ANET_LAUNCH_ORDER_WORK_INTS 4096
int anet_launch_order_from_counts_dev(anet_ctx *ctx, int64_t batch, const int32_t *counts, int32_t *launch_order,
                                      int32_t *work, void *stream);
/* The same in buckets of ONE: for the Newton-step counts iters[] of anet_qp_solve_dev (-> anet_qp_solve_ordered_dev). */
int anet_launch_order_from_steps_dev(anet_ctx *ctx, int64_t batch, const int32_t *steps, int32_t *launch_order,
                                     int32_t *work, void *stream);

/* ---- corridor generation: batched FIRI ------------------------------------ */
/* firi::firi + firi::maxVolInsEllipsoid (gcopter/firi.hpp:159-416), the inner step of
 * sfc_gen::convexCover (gcopter/sfc_gen.hpp:116-186): for each corridor segment (a, b), obstacle points
 * pc and bounding half-spaces bd, alternate `iterations` times between the polytope that separates the
 * current ellipsoid from the obstacles and the maximum-volume ellipsoid inscribed in that polytope
 * (L-BFGS on costMVIE with the call-site parameters of firi.hpp:212-217).
 * bd [batch][n_bd][4] and hpoly [batch][max_rows][4] rows h: h0 x + h1 y + h2 z + h3 <= 0 (GCOPTER's raw
 * form; LearningPlanner normalises and negates it into a.x <= b, learning_planner.hpp:293-299);
 * pc [batch][max_points][3] with n_points[batch] valid points each; a, b [batch][3].
 * n_rows [batch] rows written per corridor (the rest of hpoly is zero);
 * ok [batch]: 1 = done, 2 = done, but an inner MVIE optimisation was cut off by mvie_max_evals (the corridor is still
 *            valid: every pass only shrinks towards the obstacles; the reference continues after a failed optimisation
 *            too, firi.hpp:229-232), 0 = a or b violates bd (firi returns false, firi.hpp:282-286),
 *            -1 = the polytope needs more than max_rows rows;   ellipsoid [batch][15] (optional) =
 * R row-major, p, r of the last inscribed ellipsoid.  All HOST pointers.
 * sdlp::linprog<4> and Eigen::JacobiSVD, third-party pieces of the reference, are replaced by exact
 * equivalents (allocnet_amd/csrc/firi_kernels.h).                                                   */
typedef struct anet_firi_params {
  int32_t iterations;     /* 4     firi.hpp:273 */
  double epsilon;         /* 1e-6  firi.hpp:274 */
  double smooth_eps;      /* 1e-2  firi.hpp:218 */
  double penalty_wt;      /* 1e3   firi.hpp:219 */
  int32_t mvie_max_evals; /* 20000 evaluation budget of one MVIE optimisation (the reference has none) */
} anet_firi_params;
void anet_firi_default_params(anet_firi_params *p);
int anet_firi(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
              const double *pc, const int32_t *n_points, const double *a, const double *b,
              const anet_firi_params *params, double *hpoly, int32_t *n_rows, int32_t *ok, double *ellipsoid);
/* The same with DEVICE pointers (same layouts; ok is required) and a device workspace of
 * anet_firi_workspace() doubles: corridor generation, the QP / MINCO solve and the evaluation chain on the
 * device without a PCIe round trip.  Asynchronous on `stream`.                                        */
int64_t anet_firi_workspace(int64_t batch, int max_points, int max_rows);
int anet_firi_dev(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
                  const double *pc, const int32_t *n_points, const double *a, const double *b,
                  const anet_firi_params *params, double *work, double *hpoly, int32_t *n_rows, int32_t *ok,
                  double *ellipsoid, void *stream);

/* The same with a pass count per corridor: iterations[b] (1 .. params->iterations; NULL: params->iterations for all) -- a
 * corridor keeps the polytope of its last pass and sits out the rest.  sfc_gen::convexCover (sfc_gen.hpp:163, 176) calls
 * firi::firi with 4 passes for a segment and with 1 for a gap polytope: with this entry point both kinds go in ONE batch.
 * The host variant rejects counts outside [1, params->iterations]; the device variant cannot look at them without a
 * synchronisation and clamps instead (below 1 -> one pass, above params->iterations -> params->iterations).
 * Consumers test ok >= 1: ok = 2 is a usable corridor whose inner MVIE optimisation ran into mvie_max_evals. */
int anet_firi_var(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
                  const double *pc, const int32_t *n_points, const double *a, const double *b,
                  const int32_t *iterations /* [batch] or NULL */, const anet_firi_params *params, double *hpoly,
                  int32_t *n_rows, int32_t *ok, double *ellipsoid);
int anet_firi_var_dev(anet_ctx *ctx, int64_t batch, int n_bd, int max_points, int max_rows, const double *bd,
                      const double *pc, const int32_t *n_points, const double *a, const double *b,
                      const int32_t *iterations, const anet_firi_params *params, double *work, double *hpoly,
                      int32_t *n_rows, int32_t *ok, double *ellipsoid, void *stream);

/* ---- multi-GPU: all-gather of the per-trajectory costs over RCCL / xGMI --------------------------- */
/* Trajectories are independent, so a batch shards contiguously across GPUs (one process and one
 * context per GPU) with no collective inside a solve; the only exchange the path has is this
 * all-gather of costs (8 B per trajectory: latency-bound).  RCCL is loaded at run time (dlopen of
 * librccl.so, override with ANET_RCCL_PATH) so single-GPU users carry no dependency.
 * Usage: rank 0 calls anet_comm_unique_id and ships the 128 bytes to the other ranks by any side
 * channel (file, socket, MPI, a torch.distributed store), then every rank calls anet_comm_init.       */
#define ANET_COMM_ID_BYTES 128
int anet_comm_unique_id(anet_ctx *ctx, unsigned char id[ANET_COMM_ID_BYTES]);
int anet_comm_init(anet_ctx *ctx, int nranks, int rank, const unsigned char id[ANET_COMM_ID_BYTES]);
/* recv[r*count + i] = rank r's send[i]; device pointers; asynchronous on `stream`. */
int anet_comm_allgather_costs_dev(anet_ctx *ctx, const double *send, double *recv, int64_t count,
                                  void *stream);
int anet_comm_destroy(anet_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ALLOCNET_AMD_H */
