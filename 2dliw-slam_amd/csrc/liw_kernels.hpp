// liw_kernels.hpp — shared device-side declarations of libliw_window.so (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/liw_window.h"
#include "liw_dual.hpp"

namespace liw {

// Device copy of the parameters the factors read.
struct DevParams {
    double Riw[9], tiw[3];  // T_imu_to_wheel (wheel frame in IMU frame)
    double Ril[9], til[3];  // T_imu_to_laser
    double g;
    double laser_sqrt_info;    // 1/line_to_line_sigma   (laser_noise, laser_factor.h:19-24)
    double ground_p_info;      // 1/manifold_p_sigma     (ground_noise, ground_factor.h:18-22)
    double ground_q_info;      // 1/manifold_q_sigma
    int fast_mode;
};

// Partial-sum slots written by k_linearize, per buffer (doubles):
//   PL[B][n][LP]   laser group (window, owning frame): Haa(36) Hbb(36) Hab(36) ga(6) gb(6) sum r^2 (1), pad -> 128
//   PI[B][n-1][PIS] IMU block k (frames k,k+1): G = Y^T Y, Y = [J(15x30) | r]: blocks ii, jj (packed upper triangles), ij (15x15), g(30), sum r^2 -> 496
//   PW[B][n-1][PWS] wheel block k: G = Y^T Y, Y = [J(3x12) | r]: blocks ii, jj as packed upper triangles (21 each), ij (6x6 row-major; the ji
//                   block is ij^T and not stored), gradient (12), sum r^2 -> 91, pad -> 92   (a full 13x13 = 172 until round 3, square ii / jj = 122 until round 4)
//   PG[B][n][PGS]   ground of frame i: n * G 7x7 (Y = [J(2x6) | r]) as its packed upper triangle: 28   (the full 7x7 padded to 52 until round 4)
constexpr int LP = LIW_LASER_PARTIAL;
constexpr int PIS = 496;
// compact IMU partial: G = Y^T Y restricted to what the assembly reads; the symmetric blocks ii and jj as packed upper triangles
// (k_lin_imu is HBM-bound on this record: 708 -> 496 doubles per block)
constexpr int PI_II = 0, PI_IJ = 120, PI_JJ = 345, PI_G = 465, PI_C = 495;   // ii(120) ij(15x15) jj(120) g(30) sum r^2
__host__ __device__ inline int pi_tri(int r, int c) {   // offset of entry (r, c) = (c, r) inside a packed upper triangle of order 15
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    return lo * 15 - (lo * (lo - 1)) / 2 + (hi - lo);
}
// Large batches (pi_frame_format): the IMU partial PER FRAME instead of per block — PI[B][n][PIFS]: frame f's record holds the complete IMU
// diagonal tile of the frame, ii of block (f, f+1) + jj of block (f-1, f) (the same 15x15 index space; the role chains the jj accumulator
// of a block into the next block's ii product as its MFMA C operand), the coupling ij of block (f-1, f), both gradient parts and the
// cost of block (f-1, f): 376 instead of 496 doubles per frame, and no consumer has to add a neighbour block's share to its diagonal.
constexpr int PIF_D = 0, PIF_IJ = 120, PIF_GJ = 345, PIF_GI = 360, PIF_C = 375, PIFS = 376;
// the regime of k_lm_step_quad (which reads per-frame records only): batches of QUAD_MIN_BATCH windows and more, or the kernel forced by LIW_STEP_VARIANT=3
// (tools/step_variant_sweep.py, end of round 4, C2 windows, 12 iterations: 1 024 windows 185 k solves/s with the quad kernel against 173 k with
//  the one-wave kernels, 2 048: 280 k against 255 k; 512: 111 k against 122 k for the four-wave kernel.  The threshold was 2 049 until then.)
constexpr int QUAD_MIN_BATCH = 1024;
inline bool pi_frame_format(int B) {
    const char* env = getenv("LIW_STEP_VARIANT");
    return env ? env[0] == '3' : B >= QUAD_MIN_BATCH;
}
__host__ __device__ inline size_t pi_doubles_per_window(int n) {               // room for either format
    const size_t a = (size_t)(n > 1 ? n - 1 : 1) * PIS, b = (size_t)n * PIFS;
    return a > b ? a : b;
}
// offset of entry (r, c) = (c, r) inside a packed upper triangle of order N (row-major over r <= c)
template <int N> __host__ __device__ constexpr int tri_sym(int r, int c) {
    return r <= c ? r * N - (r * (r - 1)) / 2 + (c - r) : c * N - (c * (c - 1)) / 2 + (r - c);
}
// Compact cost array of the large-batch format, CS[B][4][n] per partial buffer: the sum r^2 of (window b, role, record) — laser group /
// ground record of frame f, IMU / wheel record of block k — stored a second time by the role that writes the record.  The prologue of
// k_lm_step_quad needs nothing but these 4 n - 2 numbers of a window's candidate linearisation; in the records they sit in 4 n - 2
// different 128-byte lines (15 kB of HBM traffic per window and LM iteration for 1 kB of values).
constexpr int CS_LASER = 0, CS_GROUND = 1, CS_IMU = 2, CS_WHEEL = 3;
__host__ __device__ inline size_t cs_index(int n, int b, int role, int rec) { return ((size_t)b * 4 + role) * n + rec; }
constexpr int PWS = 92;
// entries of a wheel record: r, c < 6 index the pose entries of frame k (ii), of frame k+1 (jj), or one of each (ij: row = frame k).
// The step kernels stream these records once per LM iteration and are HBM-bound: the symmetric blocks are stored once (122 -> 92 doubles).
__host__ __device__ constexpr int PW_II(int r, int c) { return tri_sym<6>(r, c); }
__host__ __device__ constexpr int PW_IJ(int r, int c) { return 21 + r * 6 + c; }
__host__ __device__ constexpr int PW_JJ(int r, int c) { return 57 + tri_sym<6>(r, c); }
__host__ __device__ constexpr int PW_G(int e) { return 78 + e; }   // e < 12: frame k's pose entries, then frame k+1's
constexpr int PW_C = 90;
// entries of a ground record: H (6x6, symmetric), gradient (6), sum r^2 = the packed upper triangle of the 7x7 G (52 -> 28 doubles)
constexpr int PGS = 28;
__host__ __device__ constexpr int PG_H(int r, int c) { return tri_sym<7>(r, c); }
__host__ __device__ constexpr int PG_G(int r) { return tri_sym<7>(r, 6); }
constexpr int PG_C = 27;
constexpr int FTF = 32;   // frame transform record (k_frame_tf)

// per-window LM state kept on the device across the launches of one solve
struct LmState {
    double radius, decrease_factor, x_cost, x_norm, minimum_cost;
    double cand_step_norm, model_cost_change;
    int reuse_diagonal, iteration, done, termination, successful, cur, invalid_steps, have_candidate;
    int max_iters, pad_;        // iteration cap of this solve (set by k_lm_begin: step / finish never depend on host state)
    double initial_cost;
    double scale[15 * 64];      // Jacobi scaling, up to n = 64 frames
    double diagonal[15 * 64];
    double x0[15 * 64];         // the states the solve started from: a FAILURE termination hands them back (Ceres solver.cc Minimize():
                                // StateVectorToParameterBlocks(IsSolutionUsable() ? reduced : original_reduced_parameters))
};

// state of a window at the start of a solve (k_lm_begin; the first k_lin_all of a single-window solve)
__device__ __forceinline__ void lm_reset(LmState& s, int max_iters) {
    s.radius = 1e4; s.decrease_factor = 2.0; s.x_cost = 0.0; s.x_norm = 0.0; s.minimum_cost = 0.0;
    s.cand_step_norm = 0.0; s.model_cost_change = 0.0; s.reuse_diagonal = 0; s.iteration = 0; s.done = 0; s.termination = 0;
    s.successful = 0; s.cur = 0; s.invalid_steps = 0; s.have_candidate = 0; s.initial_cost = 0.0; s.max_iters = max_iters; s.pad_ = 0;
}

struct WsView {
    // all device pointers into the caller's workspace
    // every role has two partial buffers: the current one of window b is lm[b].cur, its candidate linearisation goes to the other
    // (an accepted step just flips lm[b].cur; round 1 copied the laser partials on accept).  A factor-sharded run exchanges the laser
    // partials through liw_batch_exchange_pack / _unpack, which follow the same per-window selection.
    double* PL[2]; double* PI[2]; double* PW[2]; double* PG[2];
    double* x_cand;       // [B][n][15]
    int* group_off;       // [B][n+1] laser block range of each (window, frame)
    LmState* lm;          // [B]
    double* solve_ws;     // [B][n][SOLVE_WS] factorisation scratch
    liw_summary* info;    // [B]
    double* history;      // [(records)][B][n][15] or null
    int history_records;
    int* active;          // [1 + B] compacted list of the windows still iterating (IMU / wheel / ground roles index their blocks over it);
                          //    behind it the ticket word and the per-group publication words of k_compact_active (zeroed by lm_begin)
    int pi_frame;         // 1: PI holds per-frame records (PIF_*, pi_frame_format(B)), 0: per-block records (PI_*)
    double* CS[2];        // [B][4][n] per buffer: the cost slot (sum r^2) of every record once more, compact (cs_index; written in the per-frame format only)
    double* imu_pk;       // [B][n-1][IMU_PK] packed IMU block records of the solve in progress (launch_imu_pack, from liw_batch_lm_begin)
    int* imu_pk_bad;      // [0] != 0: some sqrt_inverse_P is not upper triangular -> the IMU role reads the caller's arrays;
                          // [1] != 0: some laser end point has a z component (launch_laser_z_scan)
};
// Packed IMU block record (built once per solve: the block's inputs are constant over its LM iterations, and only 190 of their 466
// doubles are ever used): observation X (15) | Dt | rows 0..8 x columns 9..14 of the pre-integration Jacobian (the bias blocks the
// factor reads, imu_factor.h:40-85) | upper triangle of sqrt_inverse_P = LLT(P^-1).matrixL().transpose() (imu_preintegraption.h:149)
constexpr int IMU_PK = 192, IPK_DT = 15, IPK_J = 16, IPK_S = 70;
constexpr int REC_LD = 22;                 // rec[15][22]: back-substitution operators Yo (15), Yr (6), yz columns of a frame
constexpr int REC_GS = 15 * REC_LD;       // + the scaled gradient (model decrease)
constexpr int SOLVE_WS = REC_GS + 16;

struct LinArgs {
    int B, n, mode, eval_small;
    const double* x;            // states to linearise at [B][n][15]
    const int* group_off;
    const int* laser_off;
    const double* laser_pts; int Ltot;
    const double* match_pose;
    const unsigned char* has_match;
    const double* imu_X; const double* imu_J; const double* imu_sqrtP; const double* imu_Dt;
    const double* wheel_T; const double* wheel_sqrtP;
    double* PL[2]; double* PI[2]; double* PW[2]; double* PG[2];   // partial buffers: window b writes buffer lm[b].cur (candidate: the other one)
    const LmState* lm;          // null (buffer 0, no skipping), or per-window state: done windows skip
    const LmState* gate;        // non-null: linearise window b only if gate[b].done (a marginalisation enqueued speculatively behind a solve)
    LmState* reset_lm;          // non-null (single-window liw_solve, first linearisation, lm == null): one more work-group of k_lin_all resets
    int reset_iters;            //    the LM state of every window (iteration cap reset_iters) — k_lm_begin without a launch of its own
    int marg_older;             // TRACK only, 1: the laser role also evaluates the OLDER frames' groups (own pose free against the constant
                                //    laser_match pose = their records in the marginalisation topology).  Their poses are constants of a tracking
                                //    solve, so the step masks those entries and leaves their cost out; the marginalisation enqueued behind the
                                //    solve then finds every record it needs in the window's current buffer (no launch of its own)
    int candidate;              // 1: write the small-factor partials of window b into buffer 1 - lm[b].cur
    int small_per_wave;         // wheel blocks per wave (set by launch_linearize)
    int imu_per_wave;           // IMU blocks per wave (set by launch_linearize)
    int small_nd;               // derivative directions per lane of the IMU / wheel roles: 3 (batches) or 1 (k_lin_all on a few windows)
    int* active;                // [1 + B]: number of windows still iterating, then their ids (built per linearisation when lm != null)
    int pi_frame;               // 1: write per-frame IMU records (WsView::pi_frame)
    double* CS[2];              // non-null (per-frame format): every role also stores its record's cost into the compact cost array (cs_index)
    const double* imu_pk;       // packed IMU block records (WsView::imu_pk) or null
    const int* imu_pk_bad;      //   ... usable iff *imu_pk_bad == 0
    const int* laser_hz;        // null, or -> 0 when no laser end point of the batch has a z component (2-D scans): the z planes are skipped
    const double* laser_pk;     // non-null: the batch's laser blocks re-packed per (slab of 64 windows, frame, block) rows (k_laser_slab.hip) ...
    const long long* laser_slab_off;   // ... and the first row of every (slab, frame): the laser role runs lane-per-group
    const int* laser_perm;             // ... with lane l of (slab s, frame f) = window laser_perm[f][64 s + l] (per-frame order by group length; -1: no window)
    int role_mask;              // 0 = every role; else bit 0 laser, bit 1 IMU, bit 2 wheel + ground (liw_batch_time_kernels: one role kernel alone)
    // optional per-factor outputs (liw_eval_factors)
    double* dbg_laser_res; double* dbg_laser_jac; double* dbg_imu_res; double* dbg_imu_jac;
    double* dbg_wheel_res; double* dbg_wheel_jac; double* dbg_ground_res; double* dbg_ground_jac;
};

// Slot map of a laser group record (LP slots): which pair total (bits 0..5) of the group's NP unique pair products a slot holds,
// bit 6 = negated, -1 = structural zero.  Unique columns of a block's two Jacobian rows: BOTH poses free
// [a_x a_y a_th0..2 b_th0..2 r] (9 -> 45 pairs; b_x = -a_x, b_y = -a_y), one free pose [b_x b_y b_th0..2 r] (6 -> 21 pairs).
// Record: Haa (0..35) Hbb (36..71) Hab (72..107) ga (108..113) gb (114..119) sum r^2 (120).  The laser kernel builds the same
// table in LDS (k_lin_laser_body.inc); the factor-sharded exchange packs / unpacks records with it.
template <bool BOTH>
__host__ __device__ inline int laser_slot_code(int s) {
    constexpr int NC = BOTH ? 9 : 6, RC = NC - 1;
    auto pairidx = [](int c1, int c2) { if (c1 > c2) { const int t = c1; c1 = c2; c2 = t; } return c1 * NC - c1 * (c1 - 1) / 2 + (c2 - c1); };
    auto col_a = [](int idx) { return idx < 2 ? idx : (idx == 2 ? -1 : idx - 1); };
    auto col_b = [](int idx) { return BOTH ? (idx < 2 ? idx : (idx == 2 ? -1 : idx + 2)) : (idx == 2 ? -1 : (idx < 2 ? idx : idx - 1)); };
    auto neg_b = [](int idx) { return BOTH && idx < 2; };
    int src = -1;
    bool neg = false;
    if (s < 36) {
        const int ca = col_a(s / 6), cb = col_a(s % 6);
        if (BOTH && ca >= 0 && cb >= 0) src = pairidx(ca, cb);
    } else if (s < 72) {
        const int ia = (s - 36) / 6, ib = (s - 36) % 6, ca = col_b(ia), cb = col_b(ib);
        if (ca >= 0 && cb >= 0) { src = pairidx(ca, cb); neg = neg_b(ia) != neg_b(ib); }
    } else if (s < 108) {
        const int ia = (s - 72) / 6, ib = (s - 72) % 6, ca = col_a(ia), cb = col_b(ib);
        if (BOTH && ca >= 0 && cb >= 0) { src = pairidx(ca, cb); neg = neg_b(ib); }
    } else if (s < 114) {
        const int ca = col_a(s - 108);
        if (BOTH && ca >= 0) src = pairidx(ca, RC);
    } else if (s < 120) {
        const int cb = col_b(s - 114);
        if (cb >= 0) { src = pairidx(cb, RC); neg = neg_b(s - 114); }
    } else if (s == 120) {
        src = pairidx(RC, RC);
    }
    return src < 0 ? -1 : (src | (neg ? 64 : 0));
}
struct LaserPackTable { short slot[45]; unsigned char neg[45]; };   // representative record slot (and sign) of every pair total

// Work-groups of the role / step kernels are ONE wavefront .  DS
// instructions of a wave execute in order, so a wave's LDS write is visible to its later LDS reads from any lane without s_barrier /
// vmcnt drains; only compiler reordering has to be fenced.
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the same for GLOBAL memory handed from one lane of a wave to another (drains the wave's outstanding stores / loads first)
__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// A/B aid: nontemporal hints on the streams a launch touches once (bits: 4 IMU frame records stored, 8 wheel / ground records stored,
// 16 first-sweep LDS-DMA pieces of the quad step kernel, 32 its back-substitution record stored, 64 its second-sweep pieces)
#ifndef LIW_NT_MASK
#define LIW_NT_MASK 4       // measured per 49 152 C2 windows: 4 -> k_lin_imu_chain 2.19 -> 2.12 ms; 8 / 16 / 32 / 64: no change (left off)
#endif
template <int BIT, typename T> __device__ __forceinline__ void nt_store(T* p, T v) {
    if constexpr ((LIW_NT_MASK & BIT) != 0) __builtin_nontemporal_store(v, p); else *p = v;
}

// reciprocal square root / reciprocal from the hardware estimate + two Newton steps (7 / 5 instructions; the library rsqrt() and
// an IEEE division expand to 3-4 times that, on the dependent chain of single-wave code)
__device__ __forceinline__ double fast_rsqrt(double x) {   // v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx, y * y, 1.5);
    y = y * __builtin_fma(-hx, y * y, 1.5);
    return y;
}
__device__ __forceinline__ double fast_rcp(double x) {     // v_rcp_f64 + two Newton steps
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}

// The compacted list of the windows still iterating (WsView::active) is usable when the status word behind it says so: 1 = complete;
// 0 = no compaction has run in this workspace since liw_batch_lm_begin cleared it (a step driven by hand behind a stand-alone
// linearisation, a fresh workspace); bit 1 = a group of k_compact_active gave up waiting for a predecessor (offsets wrong).  Consumers
// then index by window — every launch is sized for that — so neither case reads an unfinished or uninitialised list.
__device__ __forceinline__ const int* usable_active_list(const int* active, int B) {
    if (!active) return nullptr;
    const int st = __builtin_amdgcn_readfirstlane(active[B + 2]);
    const int cnt = __builtin_amdgcn_readfirstlane(active[0]);
    return (st == 1 && cnt >= 0 && cnt <= B) ? active : nullptr;
}
// one load per 128-byte line of [p, p + bytes), bytes <= 8 kB: lane l takes line l (lanes past the end take the last line again, so the
// load is unconditional: no branch, no wait in between).  The value only keeps the load alive: what counts is that the lines are in this
// CU's vector L1 when the dependent loads of the step ask for them.
__device__ __forceinline__ double touch_lines(const void* p, size_t bytes) {
    const int lane = threadIdx.x & 63;
    const size_t a0 = (size_t)p & ~(size_t)127, last = ((size_t)p + bytes - 8) & ~(size_t)127;
    const size_t q = a0 + 128 * (size_t)lane;
    return *reinterpret_cast<const double*>(q < last ? q : last);
}

// is window b linearised by this launch?  (with a compacted `active` list the roles index live windows only and skip this test)
__device__ __forceinline__ bool window_live(const LinArgs& A, int b) {
    if (A.gate) return A.gate[b].done != 0;
    if (A.lm) return A.lm[b].done == 0;
    return true;
}

// side streams + events used to run the independent role kernels of one linearisation concurrently
struct LinFork {
    hipStream_t side[2];
    hipEvent_t ev_fork, ev_join[2], ev_compact;
};


// ---- arguments of the LM / marginalisation kernels (k_lm.hip), shared with the host side (liw_capi.hip)
struct StepArgs {
    int B, n, mode, max_iters, fast_mode;
    double* x;                   // [B][n][15] live states
    double* match_pose;
    const unsigned char* has_match;
    const double* prior_X; const double* prior_J; const int* has_prior;
    WsView w;
    int only_slow;               // 1: k_lm_step handles only the windows k_lm_step_quad leaves out in this launch (|theta| > pi somewhere)
    int use_active;              // 1: w.active holds the windows still iterating (built by the linearisation in front of this step)
};
struct ExportArgs {
    int B, n, mode, fast_mode, buf;
    const double* x; const double* prior_X; const double* prior_J; const int* has_prior;
    WsView w;
    double* H; double* g; double* cost;
};
struct MargArgs {
    int B, n;
    const double* x; double* prior_X; double* prior_J; double* prior_R; int* has_prior;
    WsView w;
    double* sqrt_H; double* Delta_H; double* Delta_g; int* status;
    // the new prior goes to out_* when set (a marginalisation enqueued speculatively behind liw_solve must not replace the live prior
    // before the caller asks for it), else in place; gate: run window b only if its solve has terminated, status 2 otherwise
    double* out_X; double* out_J; double* out_R; int* out_has;
    const LmState* gate;
    int use_cur;                  // 1: read the partial sums of window b from its current LM buffer (lm[b].cur) instead of buffer 0
};
constexpr int LIW_RESULT_HDR = 8;   // doubles: 4 ints, then liw_summary (32 bytes), padded
struct PackArgs {
    int n, mode;
    const LmState* lm; liw_summary* info; const double* x; double* match_pose; const unsigned char* has_match;
    const double* marg; const int* marg_status;   // null without a speculative marginalisation
    double* out;
    int seq;                                      // written to header word 3 last (a polling host's completion signal)
};
void launch_linearize(const LinArgs& A, const DevParams& P, hipStream_t s, const LinFork* fk, bool defer_join = false);
void launch_linearize_join(hipStream_t s, const LinFork* fk);
void launch_exchange_pack(int B, int n, bool both, const double* PL0, const double* PL1, int candidate, const LmState* lm, double* buf, hipStream_t s);
void launch_exchange_unpack(int B, int n, bool both, int world, size_t stride, const double* buf, double* PL0, double* PL1, double* CS0, double* CS1, int candidate, const LmState* lm, hipStream_t s, bool sysload = false);   // CS0 / CS1: the compact cost arrays of the large-batch format (or null)
constexpr int P2P_MAX = 16;
struct P2pPeers { double* area[P2P_MAX]; unsigned long long* flags[P2P_MAX]; };   // device pointers to every rank's receive area / flags, as mapped here
void launch_p2p_exchange(size_t nd, const double* buf, const P2pPeers& peers, int rank, int world, unsigned long long epoch, int* err, hipStream_t s);
void launch_group_offsets(int B, int n, const int* laser_off, const int* laser_frame, int* group_off, hipStream_t s);
void launch_laser_z_scan(long Ltot, const double* laser_pts, int* flag, hipStream_t s);
// k_laser_slab.hip: lane-per-(window, frame) laser role of large 2-D batches
__host__ __device__ inline int laser_slab_count(int B) { return (B + 63) / 64; }   // slabs of 64 windows
void launch_laser_slab_prepare(int B, int n, const int* group_off, int* perm, int* mx, long long* off, const int* hz, hipStream_t s);
#ifndef LIW_SLAB_WPLANE
#define LIW_SLAB_WPLANE 0     // 1: the re-pack appends the block's weight sqrt(min(len1, len2) / 0.04) (laser_factor.h:38-42, constant over the LM
                              // iterations) as a ninth plane: -30 of a block's 274 VALU instructions for +12.5 % of row bytes.  Measured twice (round 5): the
                              // kernel alone 1.51 -> 1.56 ms (it is bandwidth-bound), the linearise bracket 3.907 -> 3.897 ms, 130.2 k -> 130.7 k solves/s: off
#endif
constexpr int LASER_SLAB_ROWD = (8 + LIW_SLAB_WPLANE) * 64;      // doubles per packed row of k_laser_slab.hip
void launch_laser_slab_pack(int B, int n, long Ltot, const int* group_off, const int* perm, const double* pts, const long long* off, const int* mx, double* pk, hipStream_t s);
void launch_lin_laser_slab(const LinArgs& A, const DevParams& P, hipStream_t s);
void launch_imu_pack(int B, int n, const double* imu_X, const double* imu_J, const double* imu_sqrtP, const double* imu_Dt, double* pk, int* bad, hipStream_t s);
void launch_pack_result(const PackArgs& a, hipStream_t s);
void launch_lm_begin(int B, int n, LmState* lm, int max_iters, hipStream_t s);
void launch_lm_step(const StepArgs& a, hipStream_t s);
void launch_lm_step_quad(const StepArgs& a, hipStream_t s);   // k_lm_quad.hip: four windows per wave (INIT topology, large batches)
bool lm_step_quad_fits(const StepArgs& a);
size_t compact_list_bytes_host(int B);                // bytes of WsView::active (list + ticket + publication words)
bool lin_builds_active_list(int B, int eval_small);   // k_linearize.hip: does launch_linearize (with LM state) compact the active windows?
void launch_lm_finish(const StepArgs& a, hipStream_t s);
void launch_export_dense(const ExportArgs& a, hipStream_t s);
void launch_marg_schur(const MargArgs& a, hipStream_t s);

// batched pre-integration (k_preint.hip)
struct PreintNoise { double q_na[3], q_nw[3], q_nba[3], q_nbw[3], wheel_cov[3]; };
void launch_preint_imu(int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end, const double* bias6,
                       const PreintNoise& N, double* X, double* J, double* Pscratch, double* sqrtP, double* Dt, hipStream_t s);
void launch_preint_wheel(int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end,
                         const PreintNoise& N, double* T12, double* sq9, double* Dt, hipStream_t s);

}  // namespace liw
