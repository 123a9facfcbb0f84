// k_lm_quad.hip — the LM step for LARGE batches: FOUR windows per wavefront, one 16-lane DPP row per window (gfx950, fp64).
//
// k_lm_step (k_lm.hip) gives a whole wave to one window and broadcasts every L[r][k] of the 15x15 eliminations with two
// v_readlane + one FMA: at 12 288 windows it is bound by the NUMBER of instructions it issues (profiles/r02_v9: 54 k per window and
// iteration, 91 % of the SIMD's issue cycles, 11 % MFMA-busy).  Here a window lives in ONE ROW of 16 lanes: lane j < 15 owns column j
// of the damped diagonal tile D, of O^T (coupling to frame i-1) and, j < 6, of R^T (arrow to frame 0's pose) — three register sets
// in the same lanes — and lane 15 rides along in the O^T set with the gradient.  The broadcast of L[r][k] inside a row is the
// DP-ALU's DPP operand (row_newbcast:r, the only DPP control 64-bit instructions take on gfx90a+):
//     v_fmac_f64_dpp  a[r], -wk (row_newbcast:r), wk        ==   a[r] -= L[r][k] * wk     for 4 windows x 16 lanes in ONE instruction
// at the plain FMA's rate when consecutive instructions read different sources (tools/ubench/dpp64_rate.hip; the 8.2-cycle figure of
// tools/ubench/dpp64.hip is the pattern where every instruction reads the same source register), i.e. one cycle per window and row
// update instead of 12.  Rank-1
// updates, the Schur products W^T W (next frame's tiles come out directly in the lane layout: no LDS, no MFMA operand tiles) and the
// back-substitution operators L^-T W all take that form; nothing leaves the registers between assembly and the factor record.
//
// Same algorithm and the same per-launch protocol as k_lm_step (candidate test -> accept / reject -> eliminate frames n-1 .. 0 ->
// back substitution -> candidate states; Ceres' TrustRegionMinimizer restated, see k_lm.hip), so the two kernels can serve different
// windows of one batch: windows whose rotation vector left the |theta| <= pi ball (so3 Plus Jacobian != I, rare: Plus normalises) stay
// on k_lm_step.  INIT and TRACK topologies.  Reference call sites: src/factor/solver.cpp:161-168 (init), :795-802 (tracking).
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "liw_kernels.hpp"
#include "k_lm_common.hpp"

namespace liw {

template <int B_, int E_, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B_ < E_) { f(std::integral_constant<int, B_>{}); sfor<B_ + 1, E_>(f); }
}
#define KI(K) (std::remove_reference_t<decltype(K)>::value)
template <int V> using KC = std::integral_constant<int, V>;
__device__ __forceinline__ double bits_and(double x, unsigned long long m) { return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(x) & m)); }

// ---- DP-ALU DPP primitives.  volatile: they must execute with all 64 lanes enabled (a DPP read of a disabled lane is not a read), so
// the compiler may neither sink them into divergent code nor reorder them against each other; program order below is written for ILP
// (consecutive instructions hit different accumulators).  A VGPR written by a VALU instruction needs 2 wait states before a DPP
// instruction reads it (the hazard recogniser does not look inside inline asm): fresh DPP sources are produced by mul_nop().
template <int L> __device__ __forceinline__ double bc(double v) {            // value of lane L of this lane's row
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
template <int L> __device__ __forceinline__ void fnma_bc(double& acc, double src, double mul) {   // acc -= src@L * mul
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}
template <int L> __device__ __forceinline__ double mul_bc(double src, double mul) {              // src@L * mul
    double r;
    asm volatile("v_mul_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "v"(mul), "n"(L));
    return r;
}
__device__ __forceinline__ double mul_nop(double x, double y) {              // x * y, safe as a DPP source right away
    double r;
    asm volatile("v_mul_f64 %0, %1, %2\n s_nop 1" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ void dpp_fence() { asm volatile("s_nop 1"); }    // before the first DPP read of values written by plain code

__device__ __forceinline__ double row_max(double v) {
    v = fmax(v, dpp64<0xB1>(v, v)); v = fmax(v, dpp64<0x4E>(v, v)); v = fmax(v, dpp64<0x141>(v, v)); v = fmax(v, dpp64<0x140>(v, v));
    return v;
}

// phase stamps (tools/clk_probe_quad.py): s_memtime of one wave at one frame of one LM iteration; dormant unless switched on
__device__ long long g_qclk[64];
__device__ int g_qclk_on[4];   // on, block, frame, iteration
__device__ int g_qprobe;       // LIW_QUAD_PROBE (diagnosis only, results are wrong): bit 0 = every row reads window 0's partial records (cache-resident),
                               // bit 1 = every row's back-substitution record is window 0's
#define QSTAMP(id) do { if (clk_on && i == clk_frame) { if (lane == 0) g_qclk[(id)] = clock64(); } } while (0)

#ifdef LIW_QUAD_TILE_ALIAS   // occupancy experiment only (WRONG results): the tile aliases the IMU record's ii block, LDS = the records alone
constexpr int QTR = 0;
#else
constexpr int QNT1 = (LIW_NT_MASK & 16) ? 2 : 0, QNT2 = (LIW_NT_MASK & 64) ? 2 : 0;   // cache policy of the LDS-DMA pieces (2 = nt)
constexpr int QTR = 4 * 15 * 6;                           // transposition tile of the carried arrow block (below)
#endif
// Gather table of the assembly phase: one 16-bit LDS byte offset per (read, lane) — see the kernel.  Reads: K_A laser Hbb | gb, K_B wheel
// jj | g_j, K_C ground H | g, K_HA laser Haa | ga, K_D IMU diagonal tile, K_CW wheel ii | g_i, K_GSH wheel g_i lane per entry.
constexpr int K_A = 0, K_B = 6, K_C = 12, K_HA = 18, K_D = 24, K_CW = 39, K_GSH = 45, NRD = 46;
constexpr int QTAB = NRD * 16;                            // doubles: NRD x 64 unsigned shorts
constexpr int QZB = 86;                                  // a block of zeros: lanes that take no part in a strided read
constexpr int QTOT_1 = 4 * (PIFS + LP + PWS + PGS) + 32 + QTR + QZB + QTAB;   // first sweep
#ifndef LIW_QUAD_BSD
#define LIW_QUAD_BSD 3   // frames of second-sweep records in flight (2: 30 kB of LDS per wave instead of 38 — co-residency experiments)
#endif
constexpr int BSD2 = LIW_QUAD_BSD;
constexpr int QTOT_2 = BSD2 * (((4 * REC_GS + 127) / 128) * 128 + 6 * 32);             // second sweep: three frames of records + state / scale / diagonal entries
constexpr int QTOT = QTOT_1 > QTOT_2 ? QTOT_1 : QTOT_2;   // LDS doubles per wave: the prefetched partial records of its four rows, the tile, a zero word, the gather table (35.6 kB: four waves per CU)

// offset of entry (r, j) inside a packed upper triangle of order 15, r a compile-time constant
template <int R> __device__ __forceinline__ int tri_rc(int j, int cj) {   // cj = 14 j - j (j - 1) / 2
    constexpr int cR = 14 * R - (R * (R - 1)) / 2;
    return R <= j ? cR + j : cj + R;
}

#ifndef LIW_QUAD_OCC
#define LIW_QUAD_OCC 1
#endif
__global__ __launch_bounds__(64, LIW_QUAD_OCC) void k_lm_step_quad(StepArgs a) {
    __shared__ double S[QTOT];
    const int lane = threadIdx.x & 63, j = lane & 15, w = lane >> 4;
    const int n = a.n;
    // TRACK topology (solver.cpp:631-820): p, q of every frame but the newest are constant (fast mode: its biases too), laser blocks tie
    // the newest frame to constant laser_match poses (no arrow), the prior block sits on frame n-2.  Uniform per launch.
    const bool track = a.mode == LIW_MODE_TRACK;
    const int fast = a.fast_mode;
    auto is_const = [&](int i, int v) { return track && i < n - 1 && (v < 6 || (fast && v >= 9)); };
    // row -> window: over the compacted list of the windows still iterating when the last linearisation built one (k_compact_active:
    // exactly the windows this step has to take), so that finished windows do not leave rows of a wave idle; else by index
    int b = (int)blockIdx.x * 4 + w;
    bool act;
    const int* const alist = a.use_active ? usable_active_list(a.w.active, a.B) : nullptr;   // null: no complete list in this workspace
    if (alist) {
        act = b < alist[0];
        b = act ? alist[1 + b] : 0;
        if ((unsigned)b >= (unsigned)a.B) { act = false; b = 0; }
    } else {
        act = b < a.B;
        b = act ? b : a.B - 1;
    }
    // Row-dependent addresses are wave-uniform bases (kernel arguments, SGPR pairs) + unsigned 32-bit element offsets (one VGPR each);
    // launch_lm_step_quad checks that every offset fits.
    double* const X = a.x;
    double* const XC = a.w.x_cand;
    double* const LMD = reinterpret_cast<double*>(a.w.lm);                // LmState as doubles
    constexpr unsigned LMS = sizeof(LmState) / 8, LM_SCALE = offsetof(LmState, scale) / 8, LM_DIAG = offsetof(LmState, diagonal) / 8,
                       LM_X0 = offsetof(LmState, x0) / 8;
    const unsigned oX = (unsigned)b * (unsigned)(n * 15), oLM = (unsigned)b * LMS;
    LmState& st = a.w.lm[b];
    act = act && !st.done;
    const int have_cand = st.have_candidate;
    {
        const double sl = quad_slow_lane(X + oX, XC + oX, have_cand, n, j, 16) ? 1.0 : 0.0;
        act = act && !(row_max(sl) > 0.0);
    }
    // windows this launch steps are marked: the one-wave kernel launched behind it (only_slow) takes exactly the unmarked ones — it cannot
    // repeat the test above, because by then this kernel has written new candidates
    if (act && j == 0) st.pad_ = 1;
    if (!__any(act)) return;

    int reuse = st.reuse_diagonal, cur = st.cur;
    const int iteration = st.iteration;
    bool proceed = act, last_successful = true;
    const bool fresh = iteration == 0 && !have_cand;
    const bool clk_k = g_qclk_on[0] && (int)blockIdx.x == g_qclk_on[1] && iteration == g_qclk_on[3] && lane == 0;
    if (clk_k) g_qclk[12] = clock64();
    double inv_radius;
    // ---------------------------------------------------------------- prologue: cost of the initial point / the pending candidate,
    //      accept / reject, Ceres' termination tests (k_lm_step's prologue, statement for statement, per row)
    {
        double radius = st.radius, dec = st.decrease_factor, x_cost = st.x_cost, x_norm = st.x_norm;
        const int max_iters = st.max_iters, successful0 = st.successful;
        int successful = successful0;
        const double model0 = st.model_cost_change, cand_step_norm = st.cand_step_norm, initial_cost0 = st.initial_cost;
        double minimum_cost = st.minimum_cost, initial_cost = initial_cost0;
        const int cb = (fresh || !have_cand) ? cur : 1 - cur;
        double cost;
        {
            // the records' cost slots, from the compact copy the roles keep (liw_kernels.hpp, cs_index): 8 lines per window instead of 4 n - 2
            const double* CSb = (cb ? a.w.CS[1] : a.w.CS[0]) + cs_index(n, b, 0, 0);
            double s = 0.0;
            for (int i = j; i < n; i += 16) {   // (blocks whose parameters are all constant are not in Ceres' problem)
                s += CSb[CS_LASER * n + i];
                if (!(track && i < n - 1)) s += CSb[CS_GROUND * n + i];
            }
            for (int k = j; k < n - 1; k += 16) {
                s += CSb[CS_IMU * n + k];
                if (!(track && k < n - 2)) s += CSb[CS_WHEEL * n + k];
            }
            if (track && !fast && a.has_prior[b] && j < 15) {   // marginalization_factor: r = linearized_J (x_{n-2} - linearized_X)  (:22-53)
                const double* xs = ((fresh || !have_cand) ? X : XC) + oX + (unsigned)((n - 2) * 15);
                const double* pJ = a.prior_J + (size_t)b * 225 + j * 15;
                const double* pX = a.prior_X + (size_t)b * 15;
                double r = 0.0;
#pragma unroll
                for (int q = 0; q < 15; ++q) r += pJ[q] * (xs[q] - pX[q]);
                s += r * r;
            }
            cost = 0.5 * row_sum(s);
        }
        int term = 0;
        bool accept = false, fail_eval = false;
        if (fresh) {
            x_cost = cost; initial_cost = cost; minimum_cost = cost;
            fail_eval = !isfinite(cost);          // IterationZero: non-finite residual -> FAILURE, nothing applied (Jacobians: after the sweep below)
        } else if (have_cand) {
            const double cand_cost = isfinite(cost) ? cost : 1.7976931348623157e308;
            if (cand_step_norm <= kParamTol * (x_norm + kParamTol)) term = 3;
            else if (fabs(x_cost - cand_cost) <= kFuncTol * x_cost) term = 2;
            if (!term) {
                const double rho = (x_cost - cand_cost) / model0;
                accept = rho > kMinRelDec;
                if (accept) {
                    cur = cb; x_cost = cand_cost;
                    const double t3 = 2.0 * rho - 1.0;
                    radius = fmin(kMaxRadius, radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3));
                    dec = 2.0; reuse = 0; successful += 1;
                    if (x_cost < minimum_cost) minimum_cost = x_cost;
                } else {
                    radius = radius / dec; dec *= 2.0; reuse = 1; last_successful = false;
                }
            }
        } else {
            last_successful = false;   // previous step was invalid
        }
        // states: accepted candidate -> live states; a failed evaluation hands back the states the solve started from; x0 / x_norm
        {
            double s2 = 0.0;
            const bool restore = act && fail_eval && !fresh && successful0 > 0;
            const bool hist = act && a.w.history && iteration < a.w.history_records && !fail_eval;
            for (int e0 = j; e0 < n * 15; e0 += 16 * 8) {      // eight entries per lane at a time: their loads are in flight together
                double xv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = e0 + 16 * q;
                    const unsigned ee = (unsigned)(e < n * 15 ? e : j);
                    xv[q] = restore ? LMD[oLM + LM_X0 + ee] : (accept ? XC[oX + ee] : X[oX + ee]);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = e0 + 16 * q;
                    if (e < n * 15) {
                        if (act && (accept || restore)) X[oX + (unsigned)e] = xv[q];
                        if (act && fresh) LMD[oLM + LM_X0 + (unsigned)e] = xv[q];
                        if (!is_const(e / 15, e % 15)) s2 += xv[q] * xv[q];
                        if (hist) a.w.history[((size_t)iteration * a.B + b) * (size_t)(n * 15) + e] = xv[q];
                    }
                }
            }
            s2 = row_sum(s2);
            if (fresh || accept) x_norm = sqrt(s2);
        }
        if (act && (fail_eval || term)) {
            if (j == 0) {
                st.done = 1; st.termination = fail_eval ? 6 : term;
                if (fail_eval) { st.x_cost = fresh ? cost : initial_cost0; st.have_candidate = 0; if (!fresh) st.iteration = iteration - 1; }
                if (fresh) { st.initial_cost = initial_cost; st.minimum_cost = minimum_cost; }
            }
            proceed = false;
        }
        const bool cap = proceed && iteration >= max_iters && !fresh;
        // the LM state as k_lm_step leaves it when a launch ends after the prologue; the tail below re-reads radius / decrease_factor
        if (proceed && j == 0) {
            st.radius = radius; st.decrease_factor = dec; st.x_cost = x_cost; st.x_norm = x_norm; st.cur = cur;
            st.successful = successful; st.minimum_cost = minimum_cost;
            if (fresh) st.initial_cost = initial_cost;
            if (cap) { st.done = 1; st.termination = 4; st.reuse_diagonal = reuse; st.have_candidate = 0; }
        }
        if (cap) proceed = false;
        inv_radius = 1.0 / radius;
    }
    if (!__any(proceed)) return;

    // current linearisation of this row's window
    const double* const PL0 = a.w.PL[0];
    const double* const PI0 = a.w.PI[0];
    const double* const PW0 = a.w.PW[0];
    const double* const PG0 = a.w.PG[0];
    const unsigned bp = (g_qprobe & 1) ? 0u : (unsigned)b;
    const unsigned oPL = bp * (unsigned)(n * LP) + (cur ? (unsigned)(a.w.PL[1] - a.w.PL[0]) : 0u);
    const unsigned oPI = bp * (unsigned)(n * PIFS) + (cur ? (unsigned)(a.w.PI[1] - a.w.PI[0]) : 0u);
    const unsigned oPW = bp * (unsigned)((n - 1) * PWS) + (cur ? (unsigned)(a.w.PW[1] - a.w.PW[0]) : 0u);
    const unsigned oPG = bp * (unsigned)(n * PGS) + (cur ? (unsigned)(a.w.PG[1] - a.w.PG[0]) : 0u);
    const int qprobe = g_qprobe;
    const unsigned oWS = (qprobe & 2) ? 0u : (unsigned)b * (unsigned)(n * SOLVE_WS);
    double* const WS = a.w.solve_ws;
    const unsigned oSC = oLM + LM_SCALE, oDG = oLM + LM_DIAG;
    const int jc = j < 15 ? j : 0;            // clamped column for addresses
    const bool l6 = j < 6, lm = j < 15;

    const bool prior_row = track && !fast && a.has_prior[b] != 0;   // this row's window carries the prior block (frame n-2)
    if (__any(proceed && fresh)) {   // Jacobi scaling 1 / (1 + sqrt(H_jj)), once per solve (same sums, same order as k_lm_step's frame_diag)
        // Round 6: the loads of FOUR frames are issued before the first of their sums is stored (as one frame per trip the store of frame i
        // fenced the loads of frame i + 1: six dependent memory round trips per frame on a wave that is alone on its SIMD — the first step
        // of 49 152 two-frame tracking windows took 0.47 ms against 0.17 ms for the others).  Unconditional loads at clamped offsets, the
        // terms that do not apply are dropped by selects; every sum keeps its terms and their order.
        double sp = 0.0;
        if (prior_row) for (int k = 0; k < 15; ++k) { const double v = a.prior_J[(size_t)b * 225 + k * 15 + jc]; sp += v * v; }
        for (int i0 = 0; i0 < n; i0 += 4) {
            double vbb[4], vjj[4], vii[4], vg[4], vi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u, n - 1);
                const int j6 = l6 ? jc : 0;          // pose entry of the 6 x 6 / 7 x 7 records (the loads of the other lanes are dropped below: any valid offset)
                vbb[u] = PL0[oPL + (unsigned)(i * LP + 36 + j6 * 7)];
                vjj[u] = n > 1 ? PW0[oPW + (unsigned)(max(i - 1, 0) * PWS + PW_JJ(j6, j6))] : 0.0;         // (n == 1: no wheel block at all)
                vii[u] = n > 1 ? PW0[oPW + (unsigned)(min(i, n - 2) * PWS + PW_II(j6, j6))] : 0.0;
                vg[u] = PG0[oPG + (unsigned)(i * PGS + PG_H(j6, j6))];
                vi[u] = PI0[oPI + (unsigned)(i * PIFS + PIF_D + pi_tri(jc, jc))];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i >= n) break;
                double dd = 0.0;
                if (l6) {
                    dd += vbb[u];
                    if (i == 0) for (int f = 0; f < n; ++f) dd += PL0[oPL + (unsigned)(f * LP + jc * 7)];   // frame 0's pose: the H_aa diagonals of every laser group, in group order
                    if (i >= 1) dd += vjj[u];
                    if (i <= n - 2) dd += vii[u];
                    dd += vg[u];
                }
                if (n > 1) dd += vi[u];                              // (the frame's complete IMU diagonal)
                if (prior_row && i == n - 2) dd += sp;
                if (proceed && fresh && lm) LMD[oSC + (unsigned)(i * 15 + j)] = is_const(i, j) ? 1.0 : 1.0 / (1.0 + sqrt(dd));
            }
        }
    }
    const double sc0 = l6 ? LMD[oSC + (unsigned)jc] : 0.0;           // scale of frame 0's pose entry j (columns of the arrow block)

    // Everything frame f needs from HBM is fetched ONE FRAME AHEAD, behind the elimination of frame f+1: the partial records by LDS-DMA
    // (global_load_lds: no VGPRs, no waits; LDS destination = uniform base + lane * 16 bytes, so every instruction fills a lane-linear
    // piece of one row's area), the Jacobi scale / LM diagonal / state entry of this lane in three registers.
    //   S_IMU[w][376]  per-frame IMU record of frame f           3 pieces per row
    //   S_PL[w][128]   laser group record of frame f             1 piece per row (the whole record: Haa / ga feed the hub accumulators)
    //   S_PW[w][92]    wheel partial of block (f-1, f)           1 piece per row (46 lanes)
    //   S_PG[w][28]    ground partial of frame f                 1 piece for the four rows (14 lanes each)
    // Round 4 measured what the staging costs and what does NOT change it (tools/quad_occ_probe2.sh, tools/quad_probe.py,
    // tools/clk_probe_quad.py): issuing a frame's 26 pieces (21 since the packed wheel / ground records) stalls the wave for ~4 k of its ~18.6 k cycles; with two waves per SIMD (a
    // <= 256-register build, six waves per CU) the SAME phase takes 5 - 11 k per wave and the kernel is no faster — the other phases keep
    // their length, so the ALUs are not what the waves share.  Plain global_load_dwordx4 into registers + ds_write_b128 (25 pieces, in three
    // batches behind the elimination phases, or all at once held in AGPRs) stalls just as long at issue (~100 - 170 cycles per 1-KiB
    // instruction): the CU's memory pipeline hands out ~8 B per clock — its share of what HBM delivers to this read / write mix
    // (4.9 TB/s chip-wide during the sweep) — and whoever issues next waits for a slot.  Probes with the records aliased to one window
    // (cache-resident) bound the memory share of the kernel at 21 %; the rest is the instruction stream.  The DMA form stays: it needs
    // no registers.
    constexpr int S_IMU = 0, S_PL = 4 * PIFS, S_PW = S_PL + 4 * LP, S_PG = S_PW + 4 * PWS, S_TR = S_PG + 4 * PGS + 32;   // (32 doubles of pad: the over-read of the last piece)
    constexpr int S_ZERO = S_TR + QTR, S_TAB = S_ZERO + QZB;
    static_assert(S_TAB + QTAB <= QTOT && 8 * QTOT < 65536, "LDS layout");
    const unsigned rPL[4] = {(unsigned)__builtin_amdgcn_readlane(oPL, 0), (unsigned)__builtin_amdgcn_readlane(oPL, 16), (unsigned)__builtin_amdgcn_readlane(oPL, 32), (unsigned)__builtin_amdgcn_readlane(oPL, 48)};
    const unsigned rPI[4] = {(unsigned)__builtin_amdgcn_readlane(oPI, 0), (unsigned)__builtin_amdgcn_readlane(oPI, 16), (unsigned)__builtin_amdgcn_readlane(oPI, 32), (unsigned)__builtin_amdgcn_readlane(oPI, 48)};
    const unsigned rPW[4] = {(unsigned)__builtin_amdgcn_readlane(oPW, 0), (unsigned)__builtin_amdgcn_readlane(oPW, 16), (unsigned)__builtin_amdgcn_readlane(oPW, 32), (unsigned)__builtin_amdgcn_readlane(oPW, 48)};
    const unsigned oPGw = oPG + (unsigned)(PG0 - PW0);   // ground records relative to the wheel buffer (launch_lm_step_quad checks the span): one base pointer per piece
    const unsigned rPGw[4] = {(unsigned)__builtin_amdgcn_readlane(oPGw, 0), (unsigned)__builtin_amdgcn_readlane(oPGw, 16), (unsigned)__builtin_amdgcn_readlane(oPGw, 32), (unsigned)__builtin_amdgcn_readlane(oPGw, 48)};
    typedef __attribute__((address_space(3))) void* lds_t;
    const int j_ = j;
    double scm_n = 1.0, dg_n = 0.0, xq_n = 0.0;     // prefetched: scale of frame f-1, LM diagonal and state entry of frame f (lane j)
    auto prefetch_regs = [&](int f) {
        const int jq = j_ < 15 ? j_ : 0;
        scm_n = (j_ < 15 && f >= 1) ? LMD[oSC + (unsigned)((f - 1) * 15 + jq)] : 1.0;
        dg_n = LMD[oDG + (unsigned)(f * 15 + jq)];
        xq_n = X[oX + (unsigned)(f * 15 + jq)];
    };
    // The 26 pieces of a frame as STRAIGHT-LINE code.  Round 3 issued each piece under its own lane mask (the last piece of a record does
    // not fill 64 lanes) and its own `block exists` test: 26 basic blocks of ~22 instructions each — a branch, the 64-bit scalar address
    // rebuilt from spilled SGPRs (v_readlane / v_writelane), m0 — ~580 instructions and ~4 k of a frame's ~18.6 k cycles with every record
    // cache-resident (tools/clk_probe_quad.py under LIW_QUAD_PROBE=3: the phase is instruction-bound, not memory-bound).  Now every piece
    // runs with all 64 lanes: lanes past the end of a record read on into whatever follows it in the workspace (always inside it) and their
    // 16 bytes land in the LDS words right behind the record's area — the start of the NEXT area in the layout, whose own piece is issued
    // later and overwrites them (loads of a wave return in order); behind the last area (ground) sits a 24-double pad.  The only test
    // left is the uniform `this frame has a block towards the frame before` around the IMU and wheel pieces.
    auto prefetch = [&](int f) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int k = f - 1;                        // frame f's block towards the frame before
        int lane2 = lane * 2;                       // the lane's 16-byte slot of a piece, in doubles: the ONLY lane-dependent part of an address
        asm volatile("" : "+v"(lane2));             // (laundered: per-lane pointers are formed here, never hoisted out of the frame loop)
        if (n > 1) {
            sfor<0, 4>([&](auto W) {                // per-frame IMU record: 3 pieces per row (the immediate offset moves the global AND the LDS address)
                constexpr int ws = KI(W);
                const double* g = PI0 + rPI[ws] + (unsigned)(f * PIFS) + lane2;
                sfor<0, 3>([&](auto Q) { __builtin_amdgcn_global_load_lds(g, (lds_t)(S + S_IMU + ws * PIFS), 16, KI(Q) * 1024, QNT1); });
            });
        }
        sfor<0, 4>([&](auto W) {                    // laser group record: exactly one piece
            constexpr int ws = KI(W);
            static_assert(LP == 128, "one piece");
            __builtin_amdgcn_global_load_lds(PL0 + rPL[ws] + (unsigned)(f * LP) + lane2, (lds_t)(S + S_PL + ws * LP), 16, 0, QNT1);
        });
        // wheel partials (block f-1; frame 0 re-reads block 0, unused) and ground partials of the four rows: the 4 x 92 + 4 x 28 doubles that lie
        // back to back in LDS are cut into 128-double pieces wherever the cuts fall (4 pieces; a piece per wheel record + one for the ground
        // records was 5): a lane's global address = the base of the record its 16 bytes belong to + its offset inside it; the lanes behind
        // the last record read on (over-read into the pad)
        {
            const unsigned fw = (unsigned)((k >= 0 ? k : 0) * PWS), fg = (unsigned)(f * PGS);
            sfor<0, 4>([&](auto P) {
                constexpr int q0 = KI(P) * 128;                     // first stream double of this piece (stream = the S_PW .. S_PG areas)
                unsigned off = 0;
                sfor<0, 8>([&](auto G) {
                    constexpr int g = KI(G);
                    constexpr int st = g < 4 ? g * PWS : 4 * PWS + (g - 4) * PGS, en = g == 7 ? (1 << 20) : st + (g < 4 ? PWS : PGS);
                    if constexpr (st < q0 + 128 && en > q0) {
                        const unsigned v = (g < 4 ? rPW[g & 3] + fw : rPGw[g & 3] + fg) + (unsigned)(lane2 - (st - q0));
                        if constexpr (st <= q0) off = v; else off = lane2 >= st - q0 ? v : off;
                    }
                });
                __builtin_amdgcn_global_load_lds(PW0 + off, (lds_t)(S + S_PW + q0), 16, 0, QNT1);
            });
        }
        asm volatile("" ::: "memory");
        prefetch_regs(f);
    };
    const double* SI = S + S_IMU + w * PIFS;
    // Gather table.  The assembly of a frame reads 88 values per lane out of the staged records, at addresses that depend on the lane's role
    // (its column j of the 6x6 pose blocks / of the 15x15 IMU tiles, lane 15 = the gradient column, lanes that take no part).  Computed per
    // frame those addresses were ~900 issued instructions (integer selects and multiplies, the nested lane tests as divergent branches):
    // 5.0 k of a frame's 18.9 k cycles (tools/clk_probe_quad.py).  They do not depend on the frame.  42 of the reads are lane base +
    // compile-time offset (the assembly forms four bases per frame); the other 46 — packed triangles, a block and its gradient with
    // different strides — take their byte offset from LDS, TAB[k][lane], built here once per launch: ds_read_u16 + ds_read_b64, no
    // vector ALU.  A lane that takes no part in a read is pointed at a block of zeros, so most lane masks of the assembly vanish as well.
    unsigned short* const TAB = reinterpret_cast<unsigned short*>(S + S_TAB);
    {
        const int jt = j < 15 ? j : 0, cjt = 14 * jt - (jt * (jt - 1)) / 2, j6t = j < 6 ? j : 0;
        const bool t6 = j < 6, t15 = j == 15, tm = j < 15;
        const int oL = S_PL + w * LP, oW_ = S_PW + w * PWS, oG = S_PG + w * PGS, oI = S_IMU + w * PIFS;
        auto put = [&](int k, bool on, int idx) { TAB[k * 64 + lane] = (unsigned short)(8 * (on ? idx : S_ZERO)); };
        for (int e = lane; e < QZB; e += 64) S[S_ZERO + e] = 0.0;
        sfor<0, 6>([&](auto R) {
            constexpr int r = KI(R);
            put(K_A + r, t6 || t15, oL + (t15 ? 114 + r : 36 + r * 6 + j6t));
            put(K_B + r, t6 || t15, oW_ + (t15 ? PW_G(6 + r) : PW_JJ(r, j6t)));
            put(K_C + r, t6 || t15, oG + (t15 ? PG_G(r) : PG_H(r, j6t)));
            put(K_HA + r, t6 || t15, oL + (t15 ? 108 + r : r * 6 + j6t));
            put(K_CW + r, t6 || t15, oW_ + (t15 ? PW_G(r) : PW_II(r, j6t)));
        });
        put(K_GSH, t6, oW_ + PW_G(j6t));
        sfor<0, 15>([&](auto R) {
            constexpr int r = KI(R);
            put(K_D + r, tm, oI + PIF_D + tri_rc<r>(jt, cjt));
        });
    }
    const int lane_tab = lane * 2;                  // byte offset of the lane's entry inside a table row
    auto RD = [&](auto K) -> double {               // read K of the gather table
        constexpr int k = KI(K);
        const unsigned off = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(TAB) + k * 128 + lane_tab);
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(S) + off);
    };
#ifdef LIW_QUAD_TILE_ALIAS
    double* ST = S + S_IMU + w * PIFS + PIF_IJ + 100;
#else
    double* ST = S + S_TR + w * 90;               // ST[c * 6 + q]: carried arrow block of the frame in front, row c (lane c), hub variable q
#endif

    // The arrow block a frame hands to the frame in front of it (R' = -Wo^T Wr: 15 rows x 6 hub variables) is accumulated TRANSPOSED: lane c
    // = its row c, register q = hub variable q — six registers and 90 DPP FMAs per frame instead of fifteen and 225 with the six busy
    // lanes of the column layout — and turned into the column layout the elimination wants by one pass through LDS (6 writes, 15 reads).
    // gsh (lane c < 15): the unscaled gradient share of block (i, i+1) that belongs to frame i (gradient max-norm, checksum).
    double d[15], o[15], rr[15], cd[15], crt[6], D0[6], hA[6];
    double g0 = 0.0, gm = 0.0, ytg = 0.0, pdiag = 0.0, gsum = 0.0, gsh = 0.0;
    bool solved = true;
    sfor<0, 15>([&](auto R) { constexpr int r = KI(R); cd[r] = 0.0; });
    sfor<0, 6>([&](auto R) { D0[KI(R)] = 0.0; crt[KI(R)] = 0.0; hA[KI(R)] = 0.0; });
    const bool clk_on = g_qclk_on[0] && (int)blockIdx.x == g_qclk_on[1] && iteration == g_qclk_on[3];
    const int clk_frame = g_qclk_on[2];
    if (clk_k) g_qclk[13] = clock64();
    prefetch(n - 1);
    double sci_carry = (j_ < 15) ? LMD[oSC + (unsigned)((n - 1) * 15 + (j_ < 15 ? j_ : 0))] : 1.0;   // scale of frame n-1; later frames reuse scm
    for (int i = n - 1; i >= 0; --i) {
        QSTAMP(0);
        // the lane id is laundered once per frame: lane-derived addresses and masks are recomputed (a few integer ops) instead of being
        // hoisted out of the loop into registers that then spill (the same trick as in k_lm_step)
        int j = j_;
        asm volatile("" : "+v"(j));
        const int jc = j < 15 ? j : 0, j6 = j < 6 ? j : 0;
        const bool l6 = j < 6, l15 = j == 15, lm = j < 15;
        const bool hasm = i >= 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this frame's records have landed in LDS (and the row's earlier stores are done)
        QSTAMP(1);
        const double sci = sci_carry, scm = scm_n, dg_old = dg_n, xq = xq_n;
        sci_carry = scm;
        // ---- the frame's blocks out of the staged records, in TWO LDS round trips: (1) the table offsets of the reads whose addresses are not
        //      affine in the lane (packed triangles, block | gradient with different strides) together with the values of those that are
        //      (lane base + compile-time offset: the IMU coupling ij | g_j, g_i, the wheel ij and laser Hab blocks), (2) the values behind the offsets.
        //      Pose blocks: lanes j < 6 the 6x6 blocks, lane 15 the gradient slots, zero in the lanes between; IMU: lane j < 15 column j of
        //      the complete diagonal tile -> D, row j of ij of block (i-1, i) -> O^T, both gradient parts -> lane 15.
        // (lane and frame conditions as bit masks: written as selects, `x + (c ? y : 0.0)` came back from the compiler as nests of divergent
        // branches around the additions — ~40 instructions per entry)
        const unsigned long long m15 = l15 ? ~0ull : 0ull, mh = hasm ? ~0ull : 0ull;
        double tS[6], oW[6], rL[6];
        {
            constexpr int NT = K_D - K_A;                // K_A, K_B, K_C, K_HA; K_D behind them
            unsigned of[NT + 15];
            auto tab_at = [&](int k) { return (unsigned)*reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(TAB) + k * 128 + lane_tab); };
            sfor<0, NT>([&](auto Q) { constexpr int q = KI(Q); of[q] = tab_at(K_A + q); });
            if (n > 1) sfor<0, 15>([&](auto Q) { constexpr int q = KI(Q); of[NT + q] = tab_at(K_D + q); });
            const char* const Sb = reinterpret_cast<const char*>(S);
            const int w8 = (lane >> 4) * 8;
            const char* const bOV = Sb + (l15 ? 8 * (S_IMU + PIF_GJ) : 8 * (S_IMU + PIF_IJ) + jc * 120) + w8 * PIFS;
            const char* const bGI = Sb + (l15 ? 8 * (S_IMU + PIF_GI) + w8 * PIFS : 8 * S_ZERO);
            const char* const bRL = Sb + (l6 ? 8 * (S_PL + 72) + w8 * LP + j6 * 48 : 8 * S_ZERO);
            const char* const bOW = Sb + (l6 ? 8 * (S_PW + PW_IJ(0, 0)) + w8 * PWS + j6 * 48 : 8 * S_ZERO);
            double gi[15];
            if (n > 1) {
                sfor<0, 15>([&](auto R) { constexpr int r = KI(R); o[r] = *reinterpret_cast<const double*>(bOV + 8 * r); gi[r] = *reinterpret_cast<const double*>(bGI + 8 * r); });
            } else {
                sfor<0, 15>([&](auto R) { constexpr int r = KI(R); d[r] = 0.0; o[r] = 0.0; gi[r] = 0.0; });
            }
            sfor<0, 6>([&](auto R) { constexpr int r = KI(R); rL[r] = *reinterpret_cast<const double*>(bRL + 8 * r); oW[r] = *reinterpret_cast<const double*>(bOW + 8 * r); });
            asm volatile("" ::: "memory");
            double pb[NT];
            sfor<0, NT>([&](auto Q) { constexpr int q = KI(Q); pb[q] = *reinterpret_cast<const double*>(Sb + of[q]); });
            if (n > 1) sfor<0, 15>([&](auto Q) { constexpr int q = KI(Q); d[q] = *reinterpret_cast<const double*>(Sb + of[NT + q]); });
            asm volatile("" ::: "memory");
            sfor<0, 6>([&](auto R) { constexpr int r = KI(R); tS[r] = pb[K_A + r] + bits_and(pb[K_B + r], mh) + pb[K_C + r]; });
            if (!track) {   // every laser frame's Haa / ga lands on frame 0's pose (init topology): summed as the frames stream by (until round 4
                            // frame 0 re-read the n records from HBM)
                sfor<0, 6>([&](auto R) { constexpr int r = KI(R); hA[r] += pb[K_HA + r]; });
                if (i == 0) sfor<0, 6>([&](auto R) { tS[KI(R)] += hA[KI(R)]; });
            }
            sfor<0, 15>([&](auto R) { constexpr int r = KI(R); o[r] = bits_and(o[r], mh) + gi[r]; });   // (frame 0's record has no block in front of it)
        }
        sfor<0, 15>([&](auto R) { rr[KI(R)] = 0.0; });
        sfor<0, 6>([&](auto R) {
            constexpr int r = KI(R);
            d[r] += bits_and(tS[r], ~m15);
            o[r] += bits_and(tS[r], m15) + bits_and(oW[r], mh);   // (oW, rL: zero outside lanes 0 .. 5)
            rr[r] = bits_and(rL[r], mh);
        });
        if (track && i == n - 2 && __any(prior_row)) {
            // marginalization_factor (marginalization_factor.h:22-53, linearized_R omitted there): r = J (x - X), H += J^T J, g += J^T r.
            // Lane j holds column j of linearized_J; (J^T J)[r][j] = sum_k J[k][r] J[k][j] is one DPP FMA per (r, k).
            double Jc[15];
            const double dxl = (lm && prior_row) ? X[oX + (unsigned)((n - 2) * 15 + jc)] - a.prior_X[(size_t)b * 15 + jc] : 0.0;
            sfor<0, 15>([&](auto K) { constexpr int k = KI(K); Jc[k] = (lm && prior_row) ? a.prior_J[(size_t)b * 225 + k * 15 + jc] : 0.0; });
            double gp = 0.0;
            sfor<0, 15>([&](auto K) { constexpr int k = KI(K); const double rp = row_sum(Jc[k] * dxl); gp = __builtin_fma(Jc[k], rp, gp); });
            dpp_fence();
            sfor<0, 15>([&](auto R) {
                constexpr int r = KI(R);
                sfor<0, 15>([&](auto K) { constexpr int k = KI(K); fnma_bc<r>(d[r], Jc[k], -Jc[k]); });
                const double gr = bc<r>(gp);
                o[r] += l15 ? gr : 0.0;
            });
        }
        QSTAMP(2);
        // ---- gradient max-norm of this frame, |x - Plus(x, -g)|, lane per entry: lane r takes entry r of the unscaled tangent gradient = the
        //      share of block (i-1, i) + the pose blocks (lane 15's o registers, turned into a lane-per-entry vector through the LDS words of
        //      the IMU record's ij block, which the assembly above has consumed) + the share of block (i, i+1) (gsh)
        QSTAMP(3);
        {
            double* GT = const_cast<double*>(SI) + PIF_IJ;
            if (l15) sfor<0, 15>([&](auto R) { constexpr int r = KI(R); GT[r] = o[r]; });
            const double gt = gsh + (lm ? GT[jc] : 0.0);
            double m = fabs(gt);                                   // |q - (q - g)| when Plus does not wrap (rotation entries), |g| elsewhere
            {
                const double av = xq - gt;
                const double a2 = mul_nop(av, av);
                const double s2 = bc<3>(a2) + bc<4>(a2) + bc<5>(a2);
                if (__any(proceed && !(s2 < 9.8))) {
                    dpp_fence();
                    const V3<double> nq = normalize_so3(V3<double>(bc<3>(av), bc<4>(av), bc<5>(av)));
                    if (j == 3) m = fabs(xq - nq.x);
                    if (j == 4) m = fabs(xq - nq.y);
                    if (j == 5) m = fabs(xq - nq.z);
                }
            }
            if (!lm || is_const(i, j)) m = 0.0;                    // (constant blocks are not parameters of the problem)
            gm = fmax(gm, m);
            // checksum of the assembled gradient: a non-finite residual or Jacobian entry anywhere in the evaluation makes it non-finite
            gsum += lm ? gt : 0.0;
        }
        // ---- LM diagonal (LevenbergMarquardtStrategy::ComputeStep): clamp(S H S, 1e-6, 1e32) at the last accepted point
        double dmp;
        {
            double mjj = d[0];
            sfor<1, 15>([&](auto R) { constexpr int r = KI(R); mjj = (j == r) ? d[r] : mjj; });
            double dgv = dg_old;
            if (!reuse) {
                dgv = fmin(fmax(__builtin_fma(mjj, sci * sci, pdiag), kMinDiag), kMaxDiag);   // (pdiag: the diagonal of block (i, i+1)'s share, already scaled)
                if (proceed && lm) LMD[oDG + (unsigned)(i * 15 + j)] = dgv;
            }
            dmp = dgv * inv_radius;
        }
        // ---- Jacobi scaling, carried Schur terms, damping
        QSTAMP(4);
        dpp_fence();
        const double* const stp = (l6 && i < n - 1) ? ST + j6 : S + S_ZERO;   // (a zero block needs 14 * 6 + 1 doubles behind it: QZB)
        sfor<0, 15>([&](auto R) {
            constexpr int r = KI(R);
            const double rs = bc<r>(sci);
            d[r] = __builtin_fma(d[r], rs * sci, cd[r]) + ((j == r) ? dmp : 0.0);
            o[r] = __builtin_fma(o[r], rs * scm, bits_and(cd[r], m15));
            rr[r] = __builtin_fma(rr[r], rs * sc0, stp[r * 6]);   // + the carried arrow block, column layout (zeros outside lanes 0 .. 5 and at frame n-1)
        });
        if (i == 1) {   // frame 0 is both the chain neighbour and the arrow target: fold R^T into O^T
            sfor<0, 15>([&](auto R) { constexpr int r = KI(R); o[r] += rr[r]; rr[r] = 0.0; });
        }
        if (i == 0) {   // the hub: Schur terms every laser frame left on frame 0's pose
            dpp_fence();
            sfor<0, 6>([&](auto R) {
                constexpr int r = KI(R);
                const double gr = bc<r>(g0);
                d[r] += D0[r];
                o[r] += l15 ? gr : 0.0;
            });
        }
        // ---- constant parameter blocks (solver.cpp:787-794): their rows / columns leave the system — unit pivot, nothing coupled
        if (track) {
            const bool colc = lm && is_const(i, j);                   // this lane's column of D
            const bool nbc = lm && hasm && is_const(i - 1, j);        // this lane's column of O^T (a variable of frame i-1)
            sfor<0, 15>([&](auto R) {
                constexpr int r = KI(R);
                const bool rowc = is_const(i, r);                     // (uniform)
                if (rowc || colc) d[r] = (j == r) ? 1.0 : 0.0;
                if (rowc || nbc) o[r] = 0.0;                          // (lane 15: the gradient entry of a constant row)
                if (rowc) rr[r] = 0.0;
            });
        }
        // ---- frame i-1's share of block (i-1, i), in scaled space, starts its carried terms
        QSTAMP(5);
        if (hasm) {
            dpp_fence();
            // (the IMU share of frame i-1 arrives with that frame's own record: only the wheel block's pose share is folded here)
            double cw[6];
            sfor<0, 6>([&](auto R) { constexpr int r = KI(R); cw[r] = RD(KC<K_CW + r>{}); });   // (zero outside lanes 0..5, 15)
            gsh = RD(KC<K_GSH>{});                          // frame i-1's unscaled wheel gradient share, lane per entry
            __builtin_amdgcn_sched_barrier(0);
            sfor<0, 15>([&](auto R) {
                constexpr int r = KI(R);
                double fi = 0.0;
                if constexpr (r < 6) {
                    const double rs = bc<r>(scm);
                    fi = cw[r] * (rs * scm);
                }
                cd[r] = fi;
                pdiag = (j == r) ? cd[r] : pdiag;  // its diagonal (LM diagonal of frame i-1)
            });
        } else {
            gsh = 0.0;
        }
        sfor<0, 6>([&](auto Q) { crt[KI(Q)] = 0.0; });
        // this frame is in registers: the next one streams in behind the elimination.  (Spreading the 30 pieces over the pivots of the
        // elimination was measured: each piece still costs ~130 ticks of issue there, no gain over the burst.)
        if (i >= 1) prefetch(i - 1);
        QSTAMP(6);
        // ---- right-looking Cholesky of D fused with the forward substitution of O^T | g and R^T: pivot k broadcasts L_kk, every lane
        //      forms w_k = a[k] / L_kk of its three columns and updates a[r] -= L[r][k] w_k with L[r][k] = w_k of lane r (DPP operand).
        //      Lane 15 keeps 1 / L_kk in its (otherwise unused) D registers for the back substitution.
        sfor<0, 15>([&](auto K) {
            constexpr int k = KI(K);
            const double piv = bc<k>(d[k]);
            if (k == 14) solved = solved && (piv > 0.0) && isfinite(piv);   // a bad pivot poisons every later one: testing the last tests all
            const double inv = fast_rsqrt(piv);
            const double wkd = mul_nop(d[k], inv);
            const double wko = o[k] * inv, wkr = rr[k] * inv;
            d[k] = l15 ? inv : wkd;
            o[k] = wko; rr[k] = wkr;
            sfor<k + 1, 15>([&](auto R) {
                constexpr int r = KI(R);
                fnma_bc<r>(d[r], wkd, wkd);
                fnma_bc<r>(o[r], wkd, wko);
                fnma_bc<r>(rr[r], wkd, wkr);
            });
        });
        QSTAMP(7);
        // y' g_s of the model decrease: (A + D^2) y = g_s eliminated block by block is the Cholesky of the whole system, so
        // y' g_s = |L^-1 g_s|^2 = the sum over the frames of |z|^2, z = this frame's forward-substituted gradient (lane 15)
        sfor<0, 15>([&](auto K) { constexpr int k = KI(K); ytg = __builtin_fma(o[k], o[k], ytg); });
        // ---- Schur terms [Wo|z]^T [Wo|z], Wr^T [Wo|z], Wr^T Wr straight into the lane layout of frame i-1 / the hub accumulators
        if (hasm) {
            dpp_fence();
            sfor<0, 15>([&](auto K) {
                constexpr int k = KI(K);
                sfor<0, 15>([&](auto C) { constexpr int c = KI(C); fnma_bc<c>(cd[c], o[k], o[k]); });
            });
            if (i >= 2) {
                sfor<0, 15>([&](auto K) {
                    constexpr int k = KI(K);
                    sfor<0, 6>([&](auto Q) { constexpr int q = KI(Q); fnma_bc<q>(crt[q], rr[k], o[k]); });   // R'[c][q] -= Wr[k][q] Wo[k][c], lane c
                    sfor<0, 6>([&](auto C) { constexpr int c = KI(C); fnma_bc<c>(D0[c], rr[k], rr[k]); });
                    fnma_bc<15>(g0, o[k], rr[k]);
                });
            }
            // the arrow block for frame i-1 goes through LDS: lane c < 15 leaves its row, lanes q < 6 pick their column up at the next frame's
            // scaling step (DS operations of a wave execute in order: no barrier)
            if (lm) sfor<0, 6>([&](auto Q) { constexpr int q = KI(Q); ST[jc * 6 + q] = crt[q]; });
        }
        QSTAMP(8);
        // ---- back-substitution operators [Yo | yz] = L^-T [Wo | z], Yr = L^-T Wr in place (right-looking: once row r is final, every
        //      earlier row k < r takes its L[r][k] y[r], L[r][k] = d[k] of lane r), then the 22-column record
        dpp_fence();
        sfor<0, 15>([&](auto T_) {
            constexpr int r = 14 - KI(T_);
            const double ir = bc<15>(d[r]);
            o[r] *= ir; rr[r] *= ir;
            sfor<0, r>([&](auto K) {
                constexpr int k = KI(K);
                fnma_bc<r>(o[k], d[k], o[r]);
                fnma_bc<r>(rr[k], d[k], rr[r]);
            });
        });
        QSTAMP(9);
        if (proceed) {
            const unsigned of = oWS + (unsigned)(i * SOLVE_WS);
            sfor<0, 15>([&](auto K) {
                constexpr int k = KI(K);
                nt_store<32>(&WS[of + (unsigned)(k * REC_LD + (l15 ? 21 : j))], o[k]);
                if (l6) nt_store<32>(&WS[of + (unsigned)(k * REC_LD + 15 + j)], rr[k]);
            });
        }
        QSTAMP(10);
    }
    if (clk_k) g_qclk[14] = clock64();
    dpp_fence();
    const double gmax = row_max(gm);
    ytg = bc<15>(ytg);
    // ---- was the evaluation this linearisation came from valid?  (see k_lm_step: Ceres' IsEvaluationValid on the fused partial sums)
    {
        const bool bad = !isfinite(row_sum(gsum));
        if (proceed && bad) {   // IterationZero / HandleSuccessfulStep: evaluation failed -> FAILURE, the states the solve started from are handed back
            if (!fresh) for (int e = j; e < n * 15; e += 16) X[oX + (unsigned)e] = LMD[oLM + LM_X0 + (unsigned)e];
            if (j == 0) {
                st.done = 1; st.termination = 6; st.have_candidate = 0;
                if (!fresh) { st.iteration = iteration - 1; st.x_cost = st.initial_cost; }
            }
            proceed = false;
        }
    }
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue, part 2
    {
        int t2 = 0;
        if (fresh) { if (gmax <= kGradTol) t2 = 1; }
        else if (last_successful && gmax <= kGradTol) t2 = 1;
        if (!t2 && !(1.0 > kMinRadius * inv_radius)) t2 = 5;     // radius > min_trust_region_radius
        if (proceed && t2) {
            if (j == 0) { st.done = 1; st.termination = t2; st.reuse_diagonal = reuse; st.have_candidate = 0; }
            proceed = false;
        }
    }
    if (!__any(proceed)) return;

    // ---------------------------------------------------------------- back substitution, frame 0 first; lane r owns unknown r
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the records written above are read by other lanes of the row
    double sn2 = 0.0, dsum = 0.0, yprev = 0.0, y0v = 0.0;
    // The sweep is a chain of small matrix-vector products (21 DPP FMAs per frame) fed by 2.6 kB of record per frame and window.  Until round 4
    // every lane loaded its own 176-byte record row with eleven 16-byte loads — 64 separate lines per instruction, ~700 line requests per
    // frame and wave, and the XC stores of the sweep in the same counter (loads and stores may complete out of order, so every use of a loaded
    // value waited for ALL outstanding operations): 7 k cycles per frame, 30 % of the kernel.  Now the records of a frame's four rows are
    // staged like the partial records of the first sweep: 11 lane-linear LDS-DMA pieces (the 4 x 330 doubles cut every 128, a lane's
    // address = its record's base + its offset inside it), the state / Jacobi scale / LM diagonal entries as 6 four-byte pieces, three
    // frames deep into the (now free) LDS of the wave; no register loads are left, so the only waits are the explicit ones below —
    // s_waitcnt vmcnt(N) with N = the LOADS issued behind the frame's own (loads return in order; a store that completes early only
    // makes the wait stricter).
    constexpr int RECP = (4 * REC_GS + 127) / 128, XP = (4 * 30 + 63) / 64, LP_ = (8 * 30 + 63) / 64;   // 16-byte pieces of the records; 4-byte pieces of X; of scale + diagonal
    constexpr int NLD = RECP + XP + LP_, BUFD = RECP * 128 + (XP + LP_) * 32;
    static_assert(BSD2 * BUFD <= QTOT && 2 * NLD <= 63 && (BSD2 == 2 || BSD2 == 3), "second-sweep staging");
    const unsigned rWS[4] = {(unsigned)__builtin_amdgcn_readlane(oWS, 0), (unsigned)__builtin_amdgcn_readlane(oWS, 16), (unsigned)__builtin_amdgcn_readlane(oWS, 32), (unsigned)__builtin_amdgcn_readlane(oWS, 48)};
    const unsigned rX[4] = {(unsigned)__builtin_amdgcn_readlane(oX, 0), (unsigned)__builtin_amdgcn_readlane(oX, 16), (unsigned)__builtin_amdgcn_readlane(oX, 32), (unsigned)__builtin_amdgcn_readlane(oX, 48)};
    const unsigned rLM[4] = {(unsigned)__builtin_amdgcn_readlane(oLM, 0), (unsigned)__builtin_amdgcn_readlane(oLM, 16), (unsigned)__builtin_amdgcn_readlane(oLM, 32), (unsigned)__builtin_amdgcn_readlane(oLM, 48)};
    auto stage2 = [&](int f, int buf) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer's previous contents have been read
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lane2 = ln * 2;
        double* const dst = S + buf * BUFD;
        const unsigned fo = (unsigned)(f * SOLVE_WS), fx = (unsigned)(f * 15);
        sfor<0, RECP>([&](auto P) {
            constexpr int q0 = KI(P) * 128;
            unsigned off = 0;
            sfor<0, 4>([&](auto G) {
                constexpr int g = KI(G), st = g * REC_GS, en = g == 3 ? (1 << 20) : st + REC_GS;
                if constexpr (st < q0 + 128 && en > q0) {
                    const unsigned v = rWS[g] + fo + (unsigned)(lane2 - (st - q0));
                    if constexpr (st <= q0) off = v; else off = lane2 >= st - q0 ? v : off;
                }
            });
            __builtin_amdgcn_global_load_lds(WS + off, (lds_t)(dst + q0), 16, 0, QNT2);
        });
        const unsigned* const X32 = reinterpret_cast<const unsigned*>(X);
        const unsigned* const L32 = reinterpret_cast<const unsigned*>(LMD);
        sfor<0, XP>([&](auto P) {                                // states: 30 dwords per row
            constexpr int q0 = KI(P) * 64;
            unsigned off = 0;
            sfor<0, 4>([&](auto G) {
                constexpr int g = KI(G), st = g * 30, en = g == 3 ? (1 << 20) : st + 30;
                if constexpr (st < q0 + 64 && en > q0) {
                    const unsigned v = 2u * (rX[g] + fx) + (unsigned)(ln - (st - q0));
                    if constexpr (st <= q0) off = v; else off = ln >= st - q0 ? v : off;
                }
            });
            // (X is the CALLER's array: the lanes behind the last row's 30 dwords re-read its first ones instead of reading on past the
            // end of the allocation; everything else staged here lies inside the workspace, where reading on is harmless)
            if constexpr (q0 + 64 > 120) off = ln >= 120 - q0 ? 2u * (rX[3] + fx) + (unsigned)(ln - (120 - q0)) : off;
            __builtin_amdgcn_global_load_lds(X32 + off, (lds_t)(dst + RECP * 128 + KI(P) * 32), 4, 0, QNT2);
        });
        sfor<0, LP_>([&](auto P) {                               // Jacobi scale (rows 0 .. 3), then LM diagonal (rows 0 .. 3)
            constexpr int q0 = KI(P) * 64;
            unsigned off = 0;
            sfor<0, 8>([&](auto G) {
                constexpr int g = KI(G), st = g * 30, en = g == 7 ? (1 << 20) : st + 30;
                if constexpr (st < q0 + 64 && en > q0) {
                    const unsigned v = 2u * (rLM[g & 3] + (g < 4 ? LM_SCALE : LM_DIAG) + fx) + (unsigned)(ln - (st - q0));
                    if constexpr (st <= q0) off = v; else off = ln >= st - q0 ? v : off;
                }
            });
            __builtin_amdgcn_global_load_lds(L32 + off, (lds_t)(dst + RECP * 128 + XP * 32 + KI(P) * 32), 4, 0, QNT2);
        });
        asm volatile("" ::: "memory");
    };
    auto solve_frame = [&](int buf, int i) {
        const double* const B_ = S + buf * BUFD;
        const double* const rowp = B_ + w * REC_GS + jc * REC_LD;
        double row[22];
        sfor<0, 11>([&](auto K) { constexpr int k = KI(K); const double2 v = reinterpret_cast<const double2*>(rowp)[k]; row[2 * k] = v.x; row[2 * k + 1] = v.y; });
        const double xold = B_[RECP * 128 + w * 15 + jc], scv = B_[RECP * 128 + XP * 32 + w * 15 + jc], dgv = B_[RECP * 128 + XP * 32 + 60 + w * 15 + jc];
        double t = row[21];
        dpp_fence();
        if (i >= 1) sfor<0, 15>([&](auto K) { constexpr int k = KI(K); fnma_bc<k>(t, yprev, row[k]); });
        if (i >= 2) sfor<0, 6>([&](auto K) { constexpr int k = KI(K); fnma_bc<k>(t, y0v, row[15 + k]); });
        yprev = t;
        if (i == 0) y0v = t;
        const bool cst = is_const(i, j);
        const double del = (lm && !cst) ? -t * scv : 0.0;
        double xnew = xold + del;
        {
            dpp_fence();
            const double a0 = bc<3>(xnew), a1 = bc<4>(xnew), a2 = bc<5>(xnew);
            if (!is_const(i, 3) && __any(proceed && !(a0 * a0 + a1 * a1 + a2 * a2 < 9.8))) {   // so3 Plus = normalize_so3(x + d) (factor_common.h:41-53)
                const V3<double> nq = normalize_so3(V3<double>(a0, a1, a2));
                if (j == 3) xnew = nq.x;
                if (j == 4) xnew = nq.y;
                if (j == 5) xnew = nq.z;
            }
        }
        if (lm) {
            if (proceed) XC[oX + (unsigned)(i * 15 + j)] = xnew;
            if (!cst) {
                sn2 += (xold - xnew) * (xold - xnew);
                dsum += dgv * inv_radius * t * t;
            }
        }
    };
    sfor<0, BSD2>([&](auto Q) { constexpr int q = KI(Q); if (q < n) stage2(q, q); });
    for (int i0 = 0; i0 < n; i0 += BSD2) {
        sfor<0, BSD2>([&](auto Q) {
            constexpr int q = KI(Q);
            __builtin_amdgcn_sched_barrier(0);
            const int i = i0 + q;
            if (i < n) {
                const int later = min(n - 1 - i, BSD2 - 1);      // frames staged behind frame i at this point
                if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                solve_frame(q, i);
                if (i + BSD2 < n) stage2(i + BSD2, q);
            }
        });
    }
    if (clk_k) g_qclk[15] = clock64();
    const double step_norm = sqrt(row_sum(sn2));
    // model cost change -(s'g_s + s'A s/2) with s = -y and (A + D^2) y = g_s  ==  (y'g_s + y'D^2 y)/2
    const double model_cost_change = 0.5 * (ytg + row_sum(dsum));
    const bool valid = solved && model_cost_change > 0.0 && isfinite(model_cost_change);
    const int invalid0 = st.invalid_steps, successful = st.successful;
    const bool fail5 = !valid && invalid0 + 1 >= 5;
    if (proceed && fail5 && successful > 0) {   // max_num_consecutive_invalid_steps: FAILURE hands back the states the solve started from
        for (int e = j; e < n * 15; e += 16) X[oX + (unsigned)e] = LMD[oLM + LM_X0 + (unsigned)e];
    }
    if (proceed && j == 0) {
        const double radius = st.radius, dec = st.decrease_factor;
        st.iteration = iteration + 1;
        if (fail5 && successful > 0) st.x_cost = st.initial_cost;
        if (valid) {
            st.reuse_diagonal = 1;
            st.model_cost_change = model_cost_change; st.cand_step_norm = step_norm; st.have_candidate = 1; st.invalid_steps = 0;
        } else {
            st.invalid_steps = invalid0 + 1;
            st.have_candidate = 0;
            if (fail5) { st.done = 1; st.termination = 6; st.iteration = iteration; }
            st.radius = radius / dec; st.decrease_factor = dec * 2.0; st.reuse_diagonal = 1;
        }
    }
}

// every row-dependent address of the kernel is a base + unsigned 32-bit element offset
bool lm_step_quad_fits(const StepArgs& a) {
    const unsigned long long lim = 0xFFFFFFFFull - 4096;
    auto span = [&](const double* p0, const double* p1, unsigned long long per) {
        const unsigned long long d = p1 >= p0 ? (unsigned long long)(p1 - p0) : ~0ull;
        return d <= lim && d + (unsigned long long)a.B * per <= lim;
    };
    const int nm = a.n > 1 ? a.n - 1 : 1;
    return a.n >= 1 && (a.mode == LIW_MODE_INIT || (a.mode == LIW_MODE_TRACK && a.n >= 2)) && span(a.w.PL[0], a.w.PL[1], (unsigned long long)a.n * LP) && a.w.pi_frame && a.w.CS[0] && a.w.CS[1] && span(a.w.PI[0], a.w.PI[1], (unsigned long long)a.n * PIFS) &&
           span(a.w.PW[0], a.w.PW[1], (unsigned long long)nm * PWS) && span(a.w.PG[0], a.w.PG[1], (unsigned long long)a.n * PGS) &&
           a.w.PG[0] >= a.w.PW[0] && span(a.w.PW[0], a.w.PG[1], (unsigned long long)a.n * PGS) &&   // (ground records are addressed from the wheel buffer)
           (unsigned long long)a.B * (sizeof(LmState) / 8) <= lim && (unsigned long long)a.B * a.n * SOLVE_WS <= lim;
}
extern "C" void liw_debug_quad_clk(int on, int block, int frame, int iteration, long long* out) {
    const int v[4] = {on, block, frame, iteration};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qclk_on), v, sizeof(v));
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qclk), sizeof(long long) * 64);
}
void launch_lm_step_quad(const StepArgs& a, hipStream_t s) {
    static const int probe = [] { const char* e = getenv("LIW_QUAD_PROBE"); const int v = e ? atoi(e) : 0; if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qprobe), &v, sizeof(v)); return v; }();
    (void)probe;
    hipLaunchKernelGGL(k_lm_step_quad, dim3((a.B + 3) / 4), dim3(64), 0, s, a);
}

}  // namespace liw
