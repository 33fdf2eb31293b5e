/*
 * ba_hip.h -- C ABI of libba_hip.so, the MI355X (gfx950) bundle-adjustment solve step.
 *
 * This is the boundary a maintainer of egirgin/bundle_adjustment binds (ctypes; see
 * INTEGRATION.md) to replace the solve step of src/bundle_adjuster.py.  Each entry point
 * names the reference interface it stands in for (paths relative to the reference root).
 *
 * Conventions: every function returns 0 on success and a negative ba_status on failure;
 * ba_last_error() then holds a message for the calling thread.  All pointers are host
 * pointers owned by the caller unless a function says "device"; the library copies in
 * and out.  One handle = one GPU = one host thread at a time.  No callbacks, no global
 * state besides the per-thread error string.
 *
 * Flat problem layout (what BundleAdjuster.run packs at src/bundle_adjuster.py:157-162,
 * plus the fixed keyframe as an ordinary camera whose index is `fixed_cam`):
 *   cams  double[Nc][6]   rvec(3) | tvec(3), world->camera (Xc = R(rvec) X + t)
 *   pts   double[Np][3]
 *   obs   cam_idx int32[Nobs], pt_idx int32[Nobs], uv double[Nobs][2]
 *         row 2i, 2i+1 of the residual vector <-> observation i (src/bundle_adjuster.py:52-70)
 *   K4    double[4]       fx, fy, cx, cy (the only entries of camera_matrix that
 *                          cv2.projectPoints reads, src/bundle_adjuster.py:67)
 */
#ifndef BA_HIP_H
#define BA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_handle ba_handle;

enum ba_status {
  BA_OK = 0,
  BA_ERR_INVALID = -1,   /* bad argument / shape / index out of range */
  BA_ERR_HIP = -2,       /* a HIP runtime call failed */
  BA_ERR_STATE = -3,     /* call order (e.g. solve before set_problem) */
  BA_ERR_NUMERIC = -4,   /* non-finite residuals or cost */
  BA_ERR_COMM = -5       /* RCCL load / init / collective failure */
};

enum ba_loss { BA_LOSS_LINEAR = 0, BA_LOSS_HUBER = 1 };
/* JACOBI: blocks of Hcc + lambda D.  SCHUR_JACOBI (default): the diagonal blocks of the reduced camera matrix S.
 * TWO_LEVEL: Schur-Jacobi plus an additive coarse correction P E^-1 P^T over aggregates of 16 consecutive cameras,
 * E = P^T S P, for band-structured problems (sequential captures such as BASELINE config 5, where block
 * preconditioners leave the drift modes along the chain): 2-5x fewer PCG iterations at equal damping; the coarse
 * matrix is rebuilt and inverted per damped system, which only pays off when a solve spends hundreds of PCG
 * iterations per LM iteration.  ba_solve returns BA_ERR_STATE when ba_set_problem found no band
 * structure (mean camera span of a track above Nc / 8) or the job has several ranks. */
enum ba_precond { BA_PRECOND_JACOBI = 0, BA_PRECOND_SCHUR_JACOBI = 1, BA_PRECOND_TWO_LEVEL = 2 };

/* Solver knobs.  The reference's literals at src/bundle_adjuster.py:170-174 are
 * loss='huber' (f_scale 1), xtol = ftol = 1e-5, max_nfev = 50. */
typedef struct ba_options {
  int32_t loss;            /* ba_loss */
  int32_t max_iters;       /* LM iterations (accepted + rejected) */
  double f_scale;          /* Huber threshold in pixels */
  double ftol;             /* stop when cost decrease <= ftol * cost (on an accepted step) */
  double xtol;             /* stop when |step| <= xtol * (xtol + |x|) */
  double gtol;             /* stop when max |gradient| <= gtol */
  double initial_lambda;   /* Marquardt damping at the first iteration */
  double pcg_tol;          /* PCG stops at sqrt(rz / rz0) <= pcg_tol */
  int32_t pcg_max_iters;
  int32_t pcg_min_iters;
  int32_t preconditioner;  /* ba_precond */
  int32_t jacobian_precision; /* 0 = f64 (default); 1 = f32 Jacobian blocks in the PCG passes, f64 accumulation / solve */
  int32_t reserved0;       /* must be 0 */
  int32_t profile;         /* 1 = bracket every kernel with HIP events (see ba_get_profile) */
  int32_t verbose;
  int32_t small_solver;    /* 0 = problems of at most 8 cameras and 6144 observations on one rank (the reference's sliding
                              window, src/pipeline.py:39 window_size 5) are solved by the single-launch window solver
                              (csrc/ba_small.hpp: whole LM loop in one kernel, reduced system formed by fp64 MFMA and
                              factorised exactly); 1 = always the multi-kernel LM / Schur / PCG path */
  double pcg_model_tol;    /* second PCG stopping test, on the quadratic model q(x) = 1/2 x^T S x - g^T x the iteration
                              minimises: stop after iteration i >= pcg_model_min_iters when i (q_{i-1} - q_i) <= pcg_model_tol |q_i|
                              (Nash & Sofer's truncated-Newton test; 0.5 is their value; 0 = off).  On ill-conditioned
                              reduced systems (long camera chains at small damping) the residual test of pcg_tol keeps
                              iterating long after the step has stopped improving the model: BASELINE config 5 on the
                              BAL camera at the reference's tolerances 1148 -> 557 PCG iterations, 39 -> 20 ms; on the
                              well-conditioned C3 it truncates useful iterations (a run to convergence needs 54 LM
                              iterations instead of 22), and on a chain driven to ftol = 1e-8 the truncated steps
                              stall the outer iteration.  Hence the default -1 = AUTOMATIC: 0.5 when ba_set_problem found
                              the problem band-structured (mean camera span of a track <= Nc / 8: sequential captures;
                              BA_STAT_BANDED; the statistic is summed over the landmark shards of a multi-rank job, so
                              every rank decides what a single rank would) AND the outer tolerance is loose (ftol >= 1e-6,
                              e.g. the reference's 1e-5); off otherwise. */
  int32_t pcg_model_min_iters; /* default 5 */
  int32_t precond_lag;     /* Schur-Jacobi only: how many consecutive damped systems may KEEP the preconditioner blocks
                              (M^-1 = blockdiag(S)^-1) built for an earlier one instead of rebuilding them (a preconditioner
                              need not be current: PCG solves the same system, only its iteration count can change).  A
                              kept system gets its right-hand side from the 6-sum camera pass instead of the 27-sum
                              one and skips the per-camera inversions.  The blocks are rebuilt anyway when the damping
                              has moved by more than 10x since they were built, when the last inner solve needed more
                              than 1.5x + 2 the iterations of the first solve after the build, or when the last accepted
                              step lowered the cost by more than 1 % (early, large steps: there a stale preconditioner
                              costs more PCG iterations than the pass it saves; measured at C3).  0 = rebuild for every
                              damped system.  Default 3 (ba_default_options).  Counted in BA_STAT_PRECOND_BUILDS / _REUSES. */
} ba_options;

typedef struct ba_summary {
  int32_t iterations;      /* LM iterations done */
  int32_t accepted;
  int32_t pcg_iterations;  /* total over all LM iterations */
  int32_t status;          /* 0 max_iters, 1 ftol, 2 xtol, 3 gtol, <0 ba_status */
  double initial_sse;      /* sum r^2 at entry == the reference's "Initial Cost" (:165) */
  double final_sse;        /* sum r^2 at exit  == "Final Cost" (:176) */
  double initial_cost;     /* 0.5 * sum rho(r^2) */
  double final_cost;
  double final_lambda;
  double seconds_total;    /* wall time of the solve loop */
  double seconds_linearize;
  double seconds_pcg;
  double seconds_update;
} ba_summary;

/* One record per LM iteration of the last ba_solve ("metrics / logging": the reference only
 * prints one line per run, src/bundle_adjuster.py:183-184; benchmarks and the drop-in's JSON metrics sink want the
 * trajectory).  Kept on the host, costs nothing on the device. */
typedef struct ba_iter_record {
  int32_t iteration;       /* 1-based */
  int32_t accepted;        /* 1 = step taken */
  int32_t pcg_iterations;  /* of this LM iteration */
  int32_t reserved;
  double cost;             /* 0.5 sum rho(r^2) before the step */
  double cost_trial;       /* ... at the trial point */
  double sse_trial;        /* sum r^2 at the trial point */
  double lambda;           /* damping the step was computed with */
  double gain_ratio;       /* actual / predicted decrease */
  double step_norm;        /* |dx| */
  double seconds;          /* wall time of this iteration on the host clock */
} ba_iter_record;

/* Per-kernel event timing collected when ba_options.profile = 1. */
#define BA_PROFILE_SLOTS 16
typedef struct ba_profile {
  int32_t launches[BA_PROFILE_SLOTS];          /* every launch */
  double total_ms[BA_PROFILE_SLOTS];
  int32_t working_launches[BA_PROFILE_SLOTS];  /* launches that did their work: PCG kernels exit at once */
  double working_ms[BA_PROFILE_SLOTS];         /* after convergence; those (< half the slot's longest) are left out */
} ba_profile;
/* slot ids */
enum ba_kernel_slot {
  BA_K_CAM_PREPARE = 0, BA_K_RESIDUAL = 1, BA_K_LINEARIZE_CAM = 2, BA_K_LINEARIZE_PT = 3,
  BA_K_POINT_INVERT = 4, BA_K_SCHUR_PT = 5, BA_K_SCHUR_CAM = 6, BA_K_PCG_UPDATE = 7,
  BA_K_PRECOND = 8, BA_K_BACKSUB = 9, BA_K_MISC = 10, BA_K_ALLREDUCE = 11,
  BA_K_SCHUR_PT_BACKSUB = 12   /* launches of the PCG point pass that found PCG finished and went on as the back substitution */
};

/* Event counters of a handle (ba_get_stat): which implementation served the window-sized solves, how often the
 * multi-workgroup window solver had to be replaced by the one-workgroup kernel (its workgroups were not resident together:
 * csrc/ba_small_mw.hpp), how often the multi-kernel loop built / kept the Schur-Jacobi preconditioner. */
enum ba_stat {
  BA_STAT_WINDOW_MW_LAUNCHES = 0,   /* launches of k_small_mw (several cooperating workgroups) */
  BA_STAT_WINDOW_LM_LAUNCHES = 1,   /* launches of k_small_lm (one workgroup) */
  BA_STAT_WINDOW_FALLBACKS = 2,     /* k_small_mw gave up at a barrier and the window was re-solved by k_small_lm */
  BA_STAT_PRECOND_BUILDS = 3,       /* damped systems whose Schur-Jacobi blocks were rebuilt */
  BA_STAT_PRECOND_REUSES = 4,       /* damped systems that kept the previous blocks (ba_options.precond_lag) */
  BA_STAT_BANDED = 5,               /* 1: ba_set_problem found the current problem band-structured (pcg_model_tol's automatic default) */
  BA_STAT_CAP_FLOOR_RAISES = 6,     /* LM iterations whose inner solve ran into pcg_max_iters and raised the damping floor */
  BA_STAT_IPC_EXCHANGES = 7,        /* PCG iterations whose reduced-system product was exchanged through IPC-mapped peer buffers (BA_IPC=1) */
  BA_STAT_PIXELS_F32 = 8,           /* 1: every pixel of the current problem is a float32 value (as cv2 keypoints are) and the
                                       multi-kernel path keeps its two pixel streams as float2, widened on load: same results,
                                       8 bytes per observation and pass less (BA_PIXELS=f64 switches it off) */
  BA_STAT_COUNT = 9
};

const char* ba_last_error(void);
const char* ba_kernel_name(int slot);
/* *value = counter `which` (ba_stat) of the handle since ba_create. */
int ba_get_stat(ba_handle* h, int32_t which, int64_t* value);
/* Test hook: occupies compute units from a SECOND stream of the handle -- n_workgroups workgroups of 256 threads with
 * lds_bytes of LDS each idle for `milliseconds` (at most 2000) of the device's wall clock, then leave.  Returns at once.
 * Lets a test hold the units the window solver's workgroups would need (tests/test_gpu_small.py). */
int ba_debug_occupy(ba_handle* h, int32_t n_workgroups, int32_t lds_bytes, double milliseconds);
/* Test hook: one array of the layout ba_set_problem built (the two observation orderings, their offsets, windows, grid
 * scalars), copied to out; `which` as listed in csrc/ba_hip.hip.  Lets a test compare the device build of large problems
 * (csrc/ba_setup.hpp) with the host build (BA_SETUP=host) element by element. */
int ba_debug_layout(ba_handle* h, int32_t which, void* out, int64_t capacity, int64_t* n);
/* Batched two-view triangulation + cheirality test: replaces VisualOdometryPipeline._triangulate_points,
 * src/pipeline.py:315-336 (cv2.triangulatePoints on P1 = K [I|0], P2 = K [R_rel|t_rel], division by (w + 1e-6),
 * z > 0 in both cameras).  K, R_rel row-major 3x3; pts1 / pts2 double[n][2] pixels in the two views; xyz double[n][3]
 * in the first camera's frame (EVERY point, kept or not); valid uint8[n] = the cheirality mask of :328-334.
 * Needs no ba_set_problem. */
int ba_triangulate(ba_handle* h, const double K[9], const double R_rel[9], const double t_rel[3], int64_t n,
                   const double* pts1, const double* pts2, double* xyz, uint8_t* valid);
/* Copies up to `capacity` records of the last ba_solve into out (may be NULL to ask for the count only);
 * *n = number of LM iterations recorded. */
int ba_get_trace(ba_handle* h, ba_iter_record* out, int32_t capacity, int32_t* n);

/* Device / handle -------------------------------------------------------------------- */
int ba_device_count(int* n);
int ba_create(int device_id, ba_handle** out);   /* replaces BundleAdjuster.__init__ state, :20-22 */
int ba_destroy(ba_handle* h);
int ba_synchronize(ba_handle* h);

/* Multi-GPU: one process per GPU; rank 0 makes the id, the host side ships the 128 bytes
 * to the other ranks (bench.py uses torch.distributed's store), every rank calls init.
 * With world == 1 nothing is loaded.  No reference counterpart. */
int ba_comm_unique_id(void* id128);
int ba_comm_init(ba_handle* h, int rank, int world, const void* id128);

/* Problem upload: replaces the dict/list gather of _gather_local_data (:195-218) and the
 * 0/1 jac_sparsity of _prepare_sparsity_matrix (:74-120) -- the library derives its own
 * camera-sorted and point-sorted orderings once.  In a multi-rank job each rank passes
 * ITS shard of points/observations (pt_idx local to the shard) and ALL cameras. */
int ba_set_problem(ba_handle* h, int32_t n_cams, int32_t n_pts, int64_t n_obs,
                   const int32_t* cam_idx, const int32_t* pt_idx, const double* uv,
                   const double K4[4], int32_t fixed_cam);
int ba_set_params(ba_handle* h, const double* cams, const double* pts);
int ba_get_params(ba_handle* h, double* cams, double* pts);
/* 3x3 rotation matrices of the current cameras, double[Nc][9] row-major: the
 * cv2.Rodrigues(rvec) of _update_map (:235-236). */
int ba_get_rotations(ba_handle* h, double* R);

/* Multi-rank write-back: after ba_solve every rank holds its own shard's points; this fills
 * pts_all double[n_total][3] with the points of ALL shards on every rank (the calling rank's
 * shard sits at [p_begin, p_begin + n_pts)); a collective, call it on every rank.  With one rank
 * it is ba_get_params' point copy.  Lets an SPMD BundleAdjuster.run finish _update_map (:239-240)
 * on every rank. */
int ba_allgather_points(ba_handle* h, int64_t p_begin, int64_t n_total, double* pts_all);

/* K1: residual vector in the caller's observation order == _cost_function (:24-72).
 * r may be NULL.  sse = sum r^2 (":165"), cost = 0.5 sum rho(r^2) for `loss`. */
int ba_residuals(ba_handle* h, int32_t loss, double f_scale, double* r, double* sse, double* cost);

/* K1 for the BAL 9-parameter camera [rvec | t | f k1 k2] (grail.cs.washington.edu/projects/bal; SURVEY.md 8f row 2;
 * the reference has no counterpart: its only camera is cv2.projectPoints(..., distCoeffs=None), :67): cameras (rvec, t)
 * and points as set by ba_set_params, intr double[Nc][3] = (f, k1, k2) per camera, pixels relative to the image
 * centre, the camera looking down -z.  Same row order and outputs as ba_residuals. */
int ba_residuals_bal(ba_handle* h, const double* intr, int32_t loss, double f_scale, double* r, double* sse, double* cost);

/* K2 for the BAL camera (parity hook, like ba_linearize): block normal equations at the current parameters with the 2x9
 * camera block [d/d rvec (additive) | d/d t | d/d f | d/d k1 | d/d k2].  Outputs (any may be NULL):
 *   Hcc double[Nc][45]  upper triangle of Jc^T w Jc, row-major (00 01 .. 08 11 .. 88);  bc double[Nc][9]  Jc^T w r
 *   Hpp double[Np][6], bp double[Np][3] as ba_linearize.  The fixed camera's blocks are zero.  In a multi-rank job Hcc | bc
 *   are all-reduced like ba_linearize's, Hpp | bp are the calling rank's shard. */
int ba_linearize_bal(ba_handle* h, const double* intr, int32_t loss, double f_scale, double* Hcc, double* bc, double* Hpp,
                     double* bp);

/* The solve step for the BAL 9-parameter camera: the SAME kernels and host loop as ba_solve, instantiated for the second
 * camera model of csrc/ba_models.hpp (BalCam; kernels in csrc/ba_kernels.hpp are templates over the model): LM + Schur
 * complement + matrix-free PCG with 9x9 camera blocks, device-side PCG / LM verdicts, speculated linearisation;
 * preconditioner BA_PRECOND_JACOBI (damped 9x9 camera blocks) or Schur-Jacobi (their Schur complements: the default;
 * BA_PRECOND_TWO_LEVEL means Schur-Jacobi here).  Same damping / gain-ratio / stopping rules, options, summary and trace as
 * ba_solve; jacobian_precision is honoured (1 = BASELINE config 5's "fp32 Jacobian + fp64 solve"); small_solver is ignored
 * (the window solver is built for the reference's pinhole only).  Multi-rank jobs are supported exactly as in ba_solve
 * (landmark shards, the same all-reduces; fold sizes follow the 9-parameter blocks).  Cameras (rvec, t) and points are the
 * handle's (ba_set_params before, ba_get_params after); intr double[Nc][3] = (f, k1, k2) per camera is read AND updated.
 * fixed_cam of ba_set_problem is honoured (-1: no camera held; the damping carries the gauge). */
int ba_solve_bal(ba_handle* h, double* intr, const ba_options* opts, ba_summary* sum);

/* K2/K3: linearise at the current parameters.  Outputs (any may be NULL):
 *   Hcc double[Nc][21]  upper triangle of Jc^T w Jc, row-major (00 01 .. 05 11 .. 55)
 *   bc  double[Nc][6]   Jc^T w r
 *   Hpp double[Np][6]   upper triangle of Jp^T w Jp (00 01 02 11 12 22)
 *   bp  double[Np][3]   Jp^T w r
 * The fixed camera's blocks are zero.  Replaces the finite-difference Jacobian scipy
 * builds from jac_sparsity (scipy/optimize/_numdiff.py:628-705). */
int ba_linearize(ba_handle* h, int32_t loss, double f_scale,
                 double* Hcc, double* bc, double* Hpp, double* bp);

/* K4 test hooks (need a prior ba_linearize): reduced camera system at damping lambda.
 *   ba_schur_rhs:   g = -(bc - W (Hpp+lam Dp)^-1 bp)          double[Nc][6]
 *   ba_schur_apply: out = S v, S = (Hcc+lam Dc) - W (Hpp+lam Dp)^-1 W^T
 * The fixed camera's row is identity / zero. */
int ba_schur_rhs(ba_handle* h, double lambda, double* g);
int ba_schur_apply(ba_handle* h, double lambda, const double* v, double* out);

/* K2-K7: the whole LM / Schur / PCG loop on the device; replaces the
 * scipy.optimize.least_squares call at src/bundle_adjuster.py:170-174. */
int ba_default_options(ba_options* opts);
int ba_solve(ba_handle* h, const ba_options* opts, ba_summary* summary);
int ba_get_profile(ba_handle* h, ba_profile* out);
int ba_reset_profile(ba_handle* h);

/* Bench hook: run one kernel `reps` times back to back on the solver stream between two
 * HIP events and return the mean duration in microseconds (state left as it was). */
int ba_time_kernel(ba_handle* h, int slot, int reps, double* mean_us);

#ifdef __cplusplus
}
#endif
#endif /* BA_HIP_H */
