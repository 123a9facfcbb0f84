/* liw_window.h — C ABI of the MI355X-native sliding-window estimator (libliw_window.so).
 *
 * Drop-in boundary for the hot path of LittleDang/2DLIW-SLAM: the library replaces what
 * `lvio_2d::solver` (reference src/factor/solver.h:28-79, src/factor/solver.cpp) and the factor
 * functors under src/factor/ compute, behind plain pointers and sizes.  The reference has no FFI of
 * its own (single C++ process); each entry point below names the C++ interface it stands in for.
 * INTEGRATION.md shows the `lvio_2d::solver` shim a maintainer adds on the reference side.
 *
 * Conventions
 *   - all floating point is IEEE fp64; every matrix crossing the ABI is ROW-MAJOR;
 *   - state of a frame = 15 doubles [p(3) q(3) v(3) ba(3) bw(3)], q = rotation vector world<-IMU
 *     (reference src/trajectory/trajectory_type.h:23-26, order src/factor/solver.cpp:332-342);
 *   - return 0 on success, negative LIW_E* otherwise; no exceptions cross the ABI; a ctx is not
 *     thread-safe (the reference's solver is driven by one dispatch thread, src/trajectory/dispatch.h:240);
 *   - there is NO CPU fallback: every compute entry point fails with LIW_ENODEV when no gfx950 device
 *     is usable.  (The host-side pre-integrators are sequential per-message code by design.)
 */
#ifndef LIW_WINDOW_H
#define LIW_WINDOW_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIW_OK 0
#define LIW_EINVAL (-22)
#define LIW_ENOMEM (-12)
#define LIW_ENODEV (-19)
#define LIW_EHIP (-5)
#define LIW_ESTATE (-1)

/* Topology of the residual blocks (who the laser blocks tie to, which states are held constant).
 *   LIW_MODE_INIT  = solver::do_init_solve  (solver.cpp:50-169): laser (frame 0, frame i), all 15n free
 *   LIW_MODE_TRACK = solver::solve          (solver.cpp:631-820): laser on the newest frame only against the
 *                    constant laser_match (p1,q1); p,q of frames 0..n-2 constant (bs too in fast_mode);
 *                    prior block on frame n-2 when one is stored and !fast_mode
 *   LIW_MODE_MARG  = solver::marginalization (solver.cpp:257-442): every frame's laser rows w.r.t. its own
 *                    pose only, prior rows on frame n-2, g = -J^T R
 * TRACK and MARG need n >= 2 (their prior block is tied to frame n-2); n = 1 is LIW_EINVAL there.      */
enum { LIW_MODE_INIT = 0, LIW_MODE_TRACK = 1, LIW_MODE_MARG = 2 };

/* The subset of param::manager the path reads (reference src/utilies/params.h; values config/office.yaml). */
typedef struct liw_params {
    double T_imu_to_wheel[16]; /* 4x4 row-major, as in the YAML */
    double T_imu_to_laser[16];
    double g;
    double line_to_line_sigma;
    double manifold_p_sigma, manifold_q_sigma;
    double imu_noise_acc_sigma[3], imu_bias_acc_sigma[3], imu_noise_gyro_sigma[3], imu_bias_gyro_sigma[3];
    double wheel_sigma[3];
    int fast_mode;
    int normalize_extrinsics; /* 1: re-orthonormalise through a quaternion like src/utilies/params.cpp:44-54 */
    int device;               /* HIP device ordinal */
} liw_params;

/* One window, host memory, array-of-frames flattening of std::deque<frame_info::ptr>
 * (src/trajectory/trajectory_type.h:9-75) and its laser_match (src/trajectory/laser_type.h:76-85). */
typedef struct liw_window {
    int n;                        /* frames */
    int L;                        /* laser_factor blocks (src/factor/laser_factor.h:26-100) */
    double* states;               /* [n][15]            in/out */
    const int* laser_frame;       /* [L] owning frame (frame_infos[i]->laser_match_ptr), ascending */
    const double* laser_pts;      /* [L][12] lines1[j].p1, lines1[j].p2, lines2[j].p1, lines2[j].p2 */
    double* match_pose;           /* [n][12] laser_match p1 q1 p2 q2   in/out */
    const unsigned char* has_match; /* [n] frame type == laser && laser_match_ptr != nullptr */
    const double* imu_X;          /* [n-1][15]  entry k = frame_infos[k+1]->imu_observation_reslut */
    const double* imu_J;          /* [n-1][225] */
    const double* imu_sqrtP;      /* [n-1][225] sqrt_inverse_P */
    const double* imu_Dt;         /* [n-1] */
    const double* wheel_T;        /* [n-1][12]  delta_Tij: R (9, row-major) then t (3) */
    const double* wheel_sqrtP;    /* [n-1][9]   */
    const double* wheel_Dt;       /* [n-1]      (carried, unused by the factors) */
} liw_window;

/* ceres::Solver::Summary subset (the reference discards it, solver.cpp:166-168; kept for tests/bench). */
typedef struct liw_summary {
    int iterations;        /* LM iterations executed (iteration 0 excluded) */
    int successful_steps;
    int termination;       /* 1 gradient tol, 2 function tol, 3 parameter tol, 4 max iterations, 5 min radius, 6 failure.
                            * 6 = Ceres' FAILURE: a non-finite residual / Jacobian entry at the initial point or at an accepted point (e.g. the
                            * NaN derivative of an exactly stationary wheel increment, wheel_factor.h:63), or 5 consecutive invalid steps; as in
                            * Ceres (solver.cc: IsSolutionUsable()) the states are then handed back as they were before the solve */
    double initial_cost, final_cost;
} liw_summary;

typedef struct liw_ctx liw_ctx;

/* ---- lifetime ------------------------------------------------------------------------------------ */
/* replaces: lvio_2d::solver::solver() (solver.cpp:43-48) + the PARAM()/noise singletons the factors read. */
liw_ctx* liw_create(const liw_params* prm);
void liw_destroy(liw_ctx* ctx);
const char* liw_last_error(const liw_ctx* ctx);
/* extrinsics actually used (after optional re-orthonormalisation), 4x4 row-major each */
int liw_get_extrinsics(const liw_ctx* ctx, double* T_imu_to_wheel16, double* T_imu_to_laser16);

/* ---- single window, host buffers (the lvio_2d::solver drop-in) ----------------------------------- */
/* Upload one window (batch of 1).  The liw_window arrays must stay valid until the next liw_set_window:
 * liw_solve / liw_marginalize scatter results back into `states` and `match_pose` in place, as the
 * reference mutates frame_info / laser_match in place.  The arrays are staged into one page-locked image and copied with ONE
 * asynchronous host-to-device copy, ordered before the next solve on the ctx stream (the call does not wait for it).  A window whose
 * bytes equal what the device already holds (the previous window with the solved states folded in — what lvio_2d::solver passes to
 * marginalization() right after solve()) is recognised: nothing is uploaded and the results of the last liw_solve stay attached. */
int liw_set_window(liw_ctx* ctx, const liw_window* w);
/* Forget the uploaded window (the ctx keeps no host pointers afterwards); window-level calls then return LIW_ESTATE
 * until the next liw_set_window.  The stored prior (solver.h:31-37) and the device copy of the window are kept. */
int liw_clear_window(liw_ctx* ctx);
/* replaces: solver::init_solve (mode INIT, incl. the laser_match fix-up solver.cpp:176-190) and
 * solver::solve (mode TRACK, incl. the p2,q2 write-back solver.cpp:804-814).  max_iters <= 0 -> Ceres
 * default 50 (10 in fast_mode for TRACK, solver.cpp:800-801).
 * One submission and one synchronisation per solve that converges within the first chunk of 4 LM iterations (tracking windows do):
 * the launches of the chunk, the write-backs and ONE packed read-back record (summary, states, laser_match poses).  A TRACK solve
 * also enqueues — behind the solve, gated on the device by the window's termination flag — the marginalisation the reference runs
 * next (trajectory.cpp:548-559); its outputs ride in the same record and its prior goes to a second set of device buffers, so the
 * live prior is untouched until liw_marginalize asks for the result (a new window, liw_set_prior or another solve drop it).
 * Environment: LIW_NO_SPEC_MARG=1 (read at liw_create) turns that off. */
int liw_solve(liw_ctx* ctx, int mode, int max_iters, liw_summary* summary);
/* Per-iteration free-state history of the last liw_solve: x[(iters+1)][n][15] (all states, constant ones
 * included); returns the number of records written (tests / parity gate, BASELINE.md "equality gate"), or
 * LIW_ESTATE when no liw_solve has completed since the last liw_set_window. */
int liw_get_history(liw_ctx* ctx, double* x, int max_records);
/* Normal equations at the current states, dense over the full ordering [p q v ba bw] x n:
 * INIT/TRACK: tangent-space H = J^T J, g = J^T r (constant blocks -> zero rows/cols), cost = 1/2|r|^2
 * (what ceres evaluates); MARG: H = J^T J, g = -J^T R as solver.cpp:12-13.  H is (15n)^2 row-major. */
int liw_linearize(liw_ctx* ctx, int mode, double* H, double* g, double* cost);
/* Per-factor residuals and ambient Jacobians (auto_diff::compute_res_and_jacobi, src/utilies/common.h:201-217).
 * Any pointer may be NULL.  laser_jac [L][2][12] cols = [p_a q_a p_b q_b]; imu_jac [n-1][15][30] cols =
 * [x_i(15) x_j(15)]; wheel_jac [n-1][3][12]; ground_res [n][2] (p then q), ground_jac [n][2][6]. */
int liw_eval_factors(liw_ctx* ctx, int mode, double* laser_res, double* laser_jac, double* imu_res, double* imu_jac,
                     double* wheel_res, double* wheel_jac, double* ground_res, double* ground_jac);
/* replaces: solver::marginalization (solver.cpp:257-442): updates the ctx-owned prior
 * (linearized_X / linearized_jacobians / linearized_residuals) and returns frame_infos.back()->sqrt_H (6x6).
 * No-op returning 0 in fast_mode (solver.cpp:259-260).  Delta_H (15x15) / Delta_g (15) optional outputs.
 * Right after a TRACK liw_solve of the same window this hands over the result computed behind that solve (no launch, no
 * synchronisation: the new prior buffers are swapped in); otherwise it linearises and eliminates now. */
int liw_marginalize(liw_ctx* ctx, double* sqrt_H36, double* Delta_H225, double* Delta_g15);
/* the solver's persistent private state (solver.h:31-37); returns 1/0 = has_linearized_block */
int liw_get_prior(liw_ctx* ctx, double* X15, double* J225, double* R15);
int liw_set_prior(liw_ctx* ctx, int has_prior, const double* X15, const double* J225, const double* R15);

/* ---- batched windows, device-resident buffers (throughput path; caller owns all device memory) ---- */
typedef struct liw_batch {
    int B;                     /* windows */
    int n;                     /* frames per window (uniform) */
    int Ltot;                  /* total laser blocks in the batch */
    double* x;                 /* [B][n][15]   states, updated in place by liw_batch_solve */
    const int* laser_off;      /* [B+1] window b owns blocks laser_off[b] .. laser_off[b+1]-1 */
    const int* laser_frame;    /* [Ltot] owning frame inside its window, ascending per window */
    const double* laser_pts;   /* [12][Ltot] component-major (SoA): c = 3*point + axis, points l1_p1,l1_p2,l2_p1,l2_p2 */
    double* match_pose;        /* [B][n][12] */
    const unsigned char* has_match; /* [B][n] */
    const double* imu_X;       /* [B][n-1][15] */
    const double* imu_J;       /* [B][n-1][225] */
    const double* imu_sqrtP;   /* [B][n-1][225] */
    const double* imu_Dt;      /* [B][n-1] */
    const double* wheel_T;     /* [B][n-1][12] */
    const double* wheel_sqrtP; /* [B][n-1][9] */
    double* prior_X;           /* [B][15]   in/out (liw_batch_marginalize writes it) */
    double* prior_J;           /* [B][225]  */
    double* prior_R;           /* [B][15]   */
    int* has_prior;            /* [B] */
    /* factor sharding (multi-GPU): this rank evaluates laser blocks only; small factors are evaluated when
     * eval_small != 0 (every rank evaluates them redundantly, only laser partial sums are all-reduced). */
    int eval_small;
    /* > 0: liw_batch_lm_step records the states after every LM iteration into the workspace (tests) */
    int history_records;
} liw_batch;

/* Workspace: one caller-allocated device buffer; layout queried here so the host (torch.distributed) can
 * all-reduce the laser partial-sum region between liw_batch_lm_linearize and liw_batch_lm_step. */
typedef struct liw_ws_layout {
    size_t bytes;                /* total */
    size_t laser_partial_off[2]; /* byte offsets of the two laser partial-sum buffers (a window's current linearisation lives in the
                                  * buffer its LM state selects, the candidate in the other: exchange them with liw_batch_exchange_*) */
    size_t laser_partial_bytes;  /* B*n*LIW_LASER_PARTIAL doubles */
    size_t info_off;             /* liw_summary[B] (device) */
    size_t history_off;          /* optional x history, 0 if not requested */
} liw_ws_layout;
#define LIW_LASER_PARTIAL 128
int liw_batch_ws_layout(int B, int n, int history_records, liw_ws_layout* out);

/* LM driver pieces (what ceres::Solve iterates, solver.cpp:168,802).  All launches go to `stream`
 * (a hipStream_t passed as void*); nothing synchronises.  Sequence for a solve:
 *   liw_batch_lm_begin -> liw_batch_lm_linearize(buf 0 at x) -> [ liw_batch_lm_step ; liw_batch_lm_linearize(candidate) ] x K
 *   -> liw_batch_lm_finish.   liw_batch_solve runs exactly that (optionally as one hipGraph).
 * The factor arrays of `b` are constants of a solve (as the frame_infos are for ceres::Solve): liw_batch_lm_begin reads them once — it
 * packs the IMU block inputs the factor uses (imu_factor.h:40-85: observation, Dt, the bias blocks of the pre-integration Jacobian, the
 * upper triangle of sqrt_inverse_P, imu_preintegraption.h:149) into the workspace and notes whether any laser end point has a z
 * component (2-D scans have none, src/utilies/common.cpp:22-24) — and the linearisations up to liw_batch_lm_finish read those.  Arrays
 * that do not fit the assumptions (a sqrt_inverse_P with entries below its diagonal, end points off the scan plane) are detected on the
 * device and evaluated in full from the caller's arrays: same results either way. */
int liw_batch_set_max_iters(liw_ctx* ctx, int mode, int max_iters); /* cap enforced by liw_batch_lm_step; returns it */
/* stand-alone linearisation at b->x (partials "current", no LM state): what liw_linearize and the bench kernel
 * timing use */
int liw_batch_linearize(liw_ctx* ctx, const liw_batch* b, int mode, void* ws, void* stream);
/* Profiling aid (bench.py `kernel_times`; no reference counterpart): average device time in ms — HIP events on `stream`, `reps` repeats —
 * of the kernels of ONE LM iteration of the batch as it stands (every window active), each launched ALONE back to back, and of the
 * marginalisation: out_ms[0] laser role, [1] IMU role, [2] wheel + ground role, [3] the LM step (not the first of the solve, which also
 * builds the Jacobi scaling), [4] chain Schur complement + eigen square root (k_marg_schur4 / k_marg_schur_chain + k_marg_schur_eigq), [5] laser role of the marginalisation
 * topology.  Opens a solve (liw_batch_lm_begin) and moves b->x along `reps` + 1 LM steps; the caller's prior is left alone. */
int liw_batch_time_kernels(liw_ctx* ctx, const liw_batch* b, int mode, void* ws, void* stream, int reps, double* out_ms);
/* NOT purely stream-asynchronous for large batches: when the batch qualifies for the lane-per-group laser kernel (INIT topology, 2-D
 * scans, >= 2 048 (slab of 64 windows, frame) pairs) liw_batch_lm_begin BLOCKS once on `stream` — a 16-byte read-back sizes the
 * ctx-owned re-packed copy of the laser end points, which may be (re)allocated with hipMalloc / hipFree.  Consequences: two solves
 * on different streams of one process serialise at this point, and it must not run while ANOTHER stream of the process is being
 * captured in global capture mode (capture on `stream` itself is detected: the re-pack is skipped and the lane-per-block kernel
 * used).  LIW_NO_LASER_SLAB=1 removes the blocking step (and the kernel).  liw_batch_solve / _solve_sharded inherit this. */
int liw_batch_lm_begin(liw_ctx* ctx, const liw_batch* b, int mode, int max_iters, void* ws, void* stream);
/* Which kernels the solve opened by the last liw_batch_lm_begin(ctx, b, ..., ws, ...) runs (tests pin the benchmarked launch shape with
 * it; no reference counterpart): *flags bit 0 = large-batch record format (per-frame IMU records + compact cost array: k_lin_imu_chain,
 * k_lm_step_quad), bit 1 = the lane-per-group laser kernel is armed for this (batch, workspace) (k_lin_laser_slab; k_lin_laser_slab1 for TRACK and
 * for the marginalisation that follows on the same arrays). */
int liw_batch_launch_paths(liw_ctx* ctx, const liw_batch* b, const void* ws, int* flags);
/* Size of the re-packed laser rows of that solve (measurement aid, no reference counterpart): *rows = 64-block rows packed (0: the
 * lane-per-group kernel is not armed for this batch), *blocks = laser blocks of the batch; 64 * rows / blocks = the padding ratio — 1.0
 * means every lane of every row carries a block.  Since round 6 the windows are taken in a per-frame order by group length, so ragged
 * batches stay near 1 (they packed 4x the data in batch order). */
int liw_batch_packed_rows(liw_ctx* ctx, const liw_batch* b, const void* ws, long long* rows, long long* blocks);
int liw_batch_lm_linearize(liw_ctx* ctx, const liw_batch* b, int mode, int candidate, void* ws, void* stream);
int liw_batch_lm_step(liw_ctx* ctx, const liw_batch* b, int mode, void* ws, void* stream);
int liw_batch_lm_finish(liw_ctx* ctx, const liw_batch* b, int mode, void* ws, void* stream);
/* Factor-sharded (multi-GPU) exchange of the laser partial sums, SURVEY.md 8(e).  A laser group record (LIW_LASER_PARTIAL slots)
 * is a signed expansion of 45 unique pair totals (INIT: both poses free) or 21 (TRACK / MARG), so ranks exchange
 * liw_batch_exchange_doubles() = B*n*{45|21} + 1 doubles instead of B*n*128; the trailing double is the number of windows that
 * are still iterating (identical on every rank after a sum, used for the early exit of the sharded loop).
 *   liw_batch_lm_linearize_async : like liw_batch_lm_linearize, but the laser role alone is ordered on `stream`; the IMU / wheel /
 *                                  ground roles keep running on the ctx's side streams until liw_batch_lm_join(stream)
 *   liw_batch_exchange_pack      : records of buffer `candidate` -> buf (device)
 *   [ host: all-reduce(buf) over RCCL, or all-gather into `copies` consecutive images (one-shot exchange) ]
 *   liw_batch_exchange_unpack    : sum of the `copies` images in rank order -> records of buffer `candidate`
 *   liw_batch_lm_join            : `stream` waits for the side roles; then liw_batch_lm_step */
int liw_batch_lm_linearize_async(liw_ctx* ctx, const liw_batch* b, int mode, int candidate, void* ws, void* stream);
int liw_batch_lm_join(liw_ctx* ctx, void* stream);
/* The whole factor-sharded solve as ONE call (round 3; the loop lived in 2dliw-slam_amd/batch.py before): lm_begin, then per LM iteration
 *   step -> linearise (laser role on `stream`, small roles on the ctx's side streams) -> pack -> exchange -> unpack -> join
 * in growing chunks (4, 8, 16 ...) with the early exit read from the exchanged active-window count, then lm_finish.  The caller supplies
 * the collective: `exchange(user, buf, all, doubles, stream)` is called once per linearisation, in `stream` order, with this rank's
 * packed record in `buf` (device, `doubles` = liw_batch_exchange_doubles()); it must enqueue on `stream` either the elementwise SUM over
 * all ranks into `buf` and return 1 (e.g. ncclAllReduce(buf, buf, doubles, ncclDouble, ncclSum, comm, stream)), or the `world` images in
 * rank order into `all` (device, world * doubles; e.g. ncclAllGather) and return `world` — the one-shot exchange, added up in rank order
 * by liw_batch_exchange_unpack so that every rank forms identical bits.  Anything else is LIW_EINVAL.  `all` may be NULL for a
 * sum-only callback.  Replaces, on the reference side, nothing: the reference is single-process (INTEGRATION.md 5 shows the driver). */
typedef int (*liw_exchange_fn)(void* user, double* buf, double* all, size_t doubles, void* stream);
int liw_batch_solve_sharded(liw_ctx* ctx, const liw_batch* b, int mode, int max_iters, void* ws, void* stream,
                            double* xbuf, double* xall, int world, liw_exchange_fn exchange, void* user);
/* Native one-shot exchange (SURVEY 5 last row; OFF unless set up): instead of a collective, every rank WRITES its packed record into its
 * slot of every peer's receive area (P-1 pushes over the P-1 dedicated xGMI links, one hop of latency), raises a flag there and waits for
 * the P flags of its own area; liw_batch_exchange_unpack then adds the P images in rank order (identical bits on every rank).
 *   receive area of a rank : liw_batch_p2p_area_doubles() = 2 (parities) x world x liw_batch_exchange_doubles() doubles, and `world`
 *                            unsigned 64-bit flags, zero-initialised once; device memory other devices can write: hipMalloc + hipIpc handles
 *                            between processes (fine-grained / uncached allocation recommended), plain pointers inside one process
 *   liw_batch_p2p_setup    : areas[r] / flags[r] = rank r's area / flags AS MAPPED INTO THIS PROCESS (r = 0 .. world-1, own included);
 *                            world = 0 switches the path off again
 *   liw_batch_solve_sharded(..., exchange = NULL, user = NULL) then uses it (xall unused); a peer whose flag does not arrive within ~4 s
 *   makes the solve return LIW_EHIP instead of hanging the device (liw_batch_p2p_status names the rank).
 * Exercised on one device only (tests/test_gpu_p2p_exchange.py: one rank onto itself, two rank objects of one process on two streams);
 * the cross-device visibility rules (system-scope stores / loads, release / acquire on the flags) are written per the ISA guide but have
 * not run across xGMI. */
size_t liw_batch_p2p_area_doubles(int B, int n, int mode, int world);
int liw_batch_p2p_setup(liw_ctx* ctx, int rank, int world, double* const* areas, unsigned long long* const* flags);
int liw_batch_p2p_status(liw_ctx* ctx, int* timed_out_rank_plus_1);
/* enable != 0: time every exchange (pack + collective + unpack, HIP events on `stream`); avg_ms / count (either may be NULL) return the
 * figures gathered since the last call */
int liw_batch_exchange_timing(liw_ctx* ctx, int enable, double* avg_ms, int* count);
int liw_batch_exchange_doubles(int B, int n, int mode);
int liw_batch_exchange_pack(liw_ctx* ctx, const liw_batch* b, int mode, int candidate, void* ws, double* buf, void* stream);
int liw_batch_exchange_unpack(liw_ctx* ctx, const liw_batch* b, int mode, int candidate, void* ws, const double* buf, int copies, void* stream);
/* The whole LM solve of the batch (reference: solver::init_solve / solver::solve for every window, src/factor/solver.cpp:50-195, :631-820):
 * lm_begin, linearise, max_iters x [step, linearise(candidate)], step, finish on `stream`.
 * Blocking points (round 6): besides the one of liw_batch_lm_begin, batches of 512 windows and more read back the number of windows still
 * iterating (4 bytes + a status word, hipStreamSynchronize on `stream`) after LM iteration 3 and then every 2 .. 8 iterations (the fewer
 * windows are left, the more often), and stop launching once it is 0 — the launches of an iteration in which no window iterates cost ~0.2 ms per
 * 49 152 windows, two thirds of a batched tracking frame.  Results are bit-identical to the full-length loop (a finished window's kernels
 * return at once).  Not under stream capture (use_graph != 0, or `stream` being captured by the caller): the captured sequence runs every
 * iteration.  LIW_NO_EARLY_EXIT=1 restores the fixed-length loop. */
int liw_batch_solve(liw_ctx* ctx, const liw_batch* b, int mode, int max_iters, void* ws, void* stream, int use_graph);
/* marginalisation of every window of the batch (linearise in MARG topology + chain Schur + eigen sqrt);
 * sqrt_H [B][36], Delta_H [B][225], Delta_g [B][15] optional device outputs.  Batches above 256 windows run two kernels on `stream`
 * (the chain, then the eigen square root of four windows per wave) and hand Delta_H | Delta_g over in the windows' factorisation scratch
 * inside `ws` — like every other region of `ws`, not to be touched by another stream while the call is in flight.
 * Packed laser rows (round 6): a solve of a large 2-D batch (liw_batch_solve / liw_batch_lm_begin) re-packs the batch's laser blocks once into
 * ctx-owned memory; liw_batch_marg_linearize on the SAME arrays (same laser_pts / laser_frame allocations, B, n, Ltot, ws) reads those rows
 * instead of the arrays — the reference marginalises the frames it has just solved (trajectory.cpp:446-479, :534-544).  A caller that
 * REWRITES laser_pts / laser_off / laser_frame in place between the solve and the marginalisation must start a new solve first (or hand
 * over other allocations): the rows are keyed by address and size, not by content. */
int liw_batch_marg_linearize(liw_ctx* ctx, const liw_batch* b, void* ws, void* stream);
int liw_batch_marg_schur(liw_ctx* ctx, const liw_batch* b, void* ws, double* sqrt_H, double* Delta_H, double* Delta_g, void* stream);
/* dense export of the assembled normal equations of buffer `buf` (tests / liw_linearize) */
int liw_batch_export_dense(liw_ctx* ctx, const liw_batch* b, int mode, int buf, void* ws, double* H, double* g, double* cost, void* stream);
/* timing hook for bench.py: average device time (ms) of the last `liw_batch_solve`'s linearise launches,
 * measured with hipEvents on `stream` (0 if timing was not enabled) */
int liw_set_timing(liw_ctx* ctx, int enable);
int liw_get_timing(liw_ctx* ctx, double* linearize_ms_avg, int* linearize_launches, double* step_ms_avg, int* step_launches);

/* ---- batched pre-integration on the device ("batch replay" form of the accumulators below): M independent
 *      intervals, interval m owns samples [sample_off[m], sample_off[m+1]).  Sample 0 of an interval only seeds the
 *      "previous measurement"; the accumulator is reset at t_start[m] and integrated to t_end[m], i.e. the same
 *      call sequence as reference src/trajectory/trajectory.cpp:176-184 followed by get_preintegraption_result().
 *      All pointers are DEVICE pointers.  imu samples [S][7] = t, acc(3), gyro(3); bias6 [M][6] = ba, bw;
 *      outputs X [M][15], J [M][225], sqrt_inverse_P [M][225], Dt [M]; P_scratch [M][225] receives the raw
 *      covariance.  wheel samples [S][13] = t, R(9 row-major), t(3); outputs delta_Tij [M][12], sqrt_inverse_P
 *      [M][9], Dt [M].  Replaces imu_preintegraption::{add_imu_measure, update, get_preintegraption_result}
 *      (src/factor/imu_preintegraption.h:124-208) and wheel_odom_preintegration::{add_wheel_odom_measure,
 *      update_by_v, get_preintegraption_result} (src/factor/wheel_odom_preintegration.h:62-152). */
int liw_batch_imu_preint(liw_ctx* ctx, int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end,
                         const double* bias6, double* X, double* J, double* P_scratch, double* sqrt_inverse_P, double* Dt, void* stream);
int liw_batch_wheel_preint(liw_ctx* ctx, int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end,
                           double* delta_Tij, double* sqrt_inverse_P, double* Dt, void* stream);

/* ---- host pre-integrators (sequential per message; replace imu_preintegraption / wheel_odom_preintegration,
 *      src/factor/imu_preintegraption.h:105-208, src/factor/wheel_odom_preintegration.h:44-152) -------- */
typedef struct liw_imu_preint liw_imu_preint;
liw_imu_preint* liw_imu_preint_create(const liw_params* prm);
void liw_imu_preint_destroy(liw_imu_preint* p);
void liw_imu_preint_reset(liw_imu_preint* p, double time, const double* acc_bias3, const double* gyr_bias3);
int liw_imu_preint_add(liw_imu_preint* p, double time_stamp, const double* acc3, const double* gyro3); /* 1 if integrated */
void liw_imu_preint_update_only_t(liw_imu_preint* p, double time);
double liw_imu_preint_Dt(const liw_imu_preint* p);
void liw_imu_preint_result(const liw_imu_preint* p, double* X15, double* J225, double* sqrt_inverse_P225, double* Dt);

typedef struct liw_wheel_preint liw_wheel_preint;
liw_wheel_preint* liw_wheel_preint_create(const liw_params* prm);
void liw_wheel_preint_destroy(liw_wheel_preint* p);
void liw_wheel_preint_reset(liw_wheel_preint* p, double time);
int liw_wheel_preint_add(liw_wheel_preint* p, double time_stamp, const double* pose_R9, const double* pose_t3);
void liw_wheel_preint_update_only_t(liw_wheel_preint* p, double time);
void liw_wheel_preint_result(const liw_wheel_preint* p, double* T12, double* sqrt_inverse_P9, double* Dt);

#ifdef __cplusplus
}
#endif
#endif /* LIW_WINDOW_H */
