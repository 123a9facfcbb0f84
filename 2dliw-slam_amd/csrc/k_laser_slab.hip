// k_laser_slab.hip — the laser role for LARGE batches of 2-D scans, init topology: a LANE is one (window, owning frame) group (gfx950, fp64).
//
// k_lin_laser (k_linearize.hip) makes a lane one laser_factor block and reduces every group's 45 pair totals across the wave: ~1 000 issued
// instructions per 64 blocks, of which ~290 are the block's own arithmetic — the rest is the per-group wave reduction (~260 per group end),
// the second masked round of pair products where a 64-block chunk straddles two groups, the transform records re-read from LDS, masks.
// With thousands of windows in a batch there is a mapping without any of that: lane l of a wave is window 64 s + l of slab s, the wave is
// one owning frame f of that slab, and every lane walks the blocks of ITS (window, frame) group one after the other — pair products
// accumulate in the lane's own registers in block order (the order of the reference's sequential loop over residual blocks,
// src/factor/solver.cpp:93-106), no cross-lane reduction, no group boundaries, the transform record of the lane's own pose in registers.
// What that needs is block j of 64 DIFFERENT windows on consecutive addresses: liw_batch_lm_begin re-packs the caller's component-major
// end-point planes once per solve (k_laser_slab_pack, the inputs are constant over the LM iterations) into rows
//     row (s, f, j) = [8 planes: l1p1.x l1p1.y l1p2.x l1p2.y l2p1.x l2p1.y l2p2.x l2p2.y][64 lanes]          (4 KiB, lane-linear)
// so that a row is eight coalesced 512-byte loads.  Rows of a (slab, frame) are padded to the slab's longest group.
// Same per-block arithmetic as k_lin_laser_body.inc (reference src/factor/laser_factor.h:45-89, src/utilies/common.h:86-95 incl. the NaN
// of a point exactly on the line); the sums differ from k_lin_laser's tree order by round-off only (tests/test_gpu_laser_slab.py).
#include "liw_kernels.hpp"
#include <algorithm>
#include <cstdlib>

namespace liw {

#ifdef LIW_CLK     // s_memtime stamps of the middle work-group's wave (tools/clk_probe_slab.py)
__device__ long long g_clk_slab[16];
#define SSTAMP(id) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (item == slab_items_probe && (threadIdx.x & 63) == 0) g_clk_slab[(id)] = clock64(); } while (0)
__device__ int slab_items_probe = 11000;
#else
#define SSTAMP(id) do { } while (0)
#endif

namespace {

// compact frame transform record of a 2-D scan: rows 0,1 x columns 0,1 of make_tf(p, theta) * T_imu_to_laser and their three d/d theta_k
//   [0..3] M[r][c]   [4..5] t[r]   [6 + 4k + 2r + c] dM_k[r][c]   [18 + 2k + r] dt_k[r]
constexpr int TF2 = 24;
__device__ __forceinline__ void frame_tf2(const DevParams& P, const double* pose6, double* o) {
    typedef LJN<3> J3;
    const V3<J3> p = cast_v3<J3>(pose6);
    const V3<J3> th(seed<3>(pose6[3], 0, true), seed<3>(pose6[4], 1, true), seed<3>(pose6[5], 2, true));
    const Iso<J3> Twl = mul(make_tf(p, th), cast_iso<J3>(P.Ril, P.til));
    const J3 tt[2] = {Twl.t.x, Twl.t.y};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            o[r * 2 + c] = Twl.R(r, c).v;
#pragma unroll
            for (int k = 0; k < 3; ++k) o[6 + 4 * k + 2 * r + c] = Twl.R(r, c).d[k];
        }
        o[4 + r] = tt[r].v;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[18 + 2 * k + r] = tt[r].d[k];
    }
}

// slot s of the 128-slot group record (liw_kernels.hpp) as (pair total | 64 if negated), -1 = structural zero; both poses free.
// The constexpr twin of laser_slot_code<true>: a lane writes its own record, so the expansion is resolved at compile time.
constexpr int slab_pairidx(int c1, int c2) { return c1 > c2 ? slab_pairidx(c2, c1) : c1 * 9 - c1 * (c1 - 1) / 2 + (c2 - c1); }
constexpr int slab_col_a(int idx) { return idx < 2 ? idx : (idx == 2 ? -1 : idx - 1); }
constexpr int slab_col_b(int idx) { return idx < 2 ? idx : (idx == 2 ? -1 : idx + 2); }
constexpr int slab_slot_code(int s) {
    if (s < 36) { const int ca = slab_col_a(s / 6), cb = slab_col_a(s % 6); return (ca >= 0 && cb >= 0) ? slab_pairidx(ca, cb) : -1; }
    if (s < 72) {
        const int ia = (s - 36) / 6, ib = (s - 36) % 6, ca = slab_col_b(ia), cb = slab_col_b(ib);
        return (ca >= 0 && cb >= 0) ? (slab_pairidx(ca, cb) | (((ia < 2) != (ib < 2)) ? 64 : 0)) : -1;
    }
    if (s < 108) {
        const int ia = (s - 72) / 6, ib = (s - 72) % 6, ca = slab_col_a(ia), cb = slab_col_b(ib);
        return (ca >= 0 && cb >= 0) ? (slab_pairidx(ca, cb) | ((ib < 2) ? 64 : 0)) : -1;
    }
    if (s < 114) { const int ca = slab_col_a(s - 108); return ca >= 0 ? slab_pairidx(ca, 8) : -1; }
    if (s < 120) { const int cb = slab_col_b(s - 114); return cb >= 0 ? (slab_pairidx(cb, 8) | ((s - 114 < 2) ? 64 : 0)) : -1; }
    if (s == 120) return slab_pairidx(8, 8);
    return -1;
}
// the same for the ONE-free-pose record (MARG / TRACK topology: the block's a-pose is the constant laser_match pose, solver.cpp:471-472,
// :669-698): unique columns px py th0 th1 th2 of the owning frame + the residual = 21 pair totals, H_bb / g_b / cost slots only, no sign
// (the constexpr twin of laser_slot_code<false>)
constexpr int slab_pairidx1(int c1, int c2) { return c1 > c2 ? slab_pairidx1(c2, c1) : c1 * 6 - c1 * (c1 - 1) / 2 + (c2 - c1); }
constexpr int slab_col_b1(int idx) { return idx == 2 ? -1 : (idx < 2 ? idx : idx - 1); }
constexpr int slab_slot_code1(int s) {
    if (s >= 36 && s < 72) { const int ca = slab_col_b1((s - 36) / 6), cb = slab_col_b1((s - 36) % 6); return (ca >= 0 && cb >= 0) ? slab_pairidx1(ca, cb) : -1; }
    if (s >= 114 && s < 120) { const int cb = slab_col_b1(s - 114); return cb >= 0 ? slab_pairidx1(cb, 5) : -1; }
    if (s == 120) return slab_pairidx1(5, 5);
    return -1;
}
template <bool BOTH, int S_> __device__ __forceinline__ double slab_slot(const double* acc) {
    constexpr int code = BOTH ? slab_slot_code(S_) : slab_slot_code1(S_);
    if constexpr (code < 0) return 0.0;
    else if constexpr ((code & 64) != 0) return -acc[code & 63];
    else return acc[code & 63];
}
template <bool BOTH, int S_> __device__ __forceinline__ void slab_store(double* out, const double* acc) {
    if constexpr (S_ < LP) {
        typedef double __attribute__((ext_vector_type(2))) dbl2;
        dbl2 v;
        v.x = slab_slot<BOTH, S_>(acc); v.y = slab_slot<BOTH, S_ + 1>(acc);
        *reinterpret_cast<dbl2*>(out + S_) = v;
        slab_store<BOTH, S_ + 2>(out, acc);
    }
}

// Epilogue (round 5, late): the 64 records of a wave lie 1 KiB x n apart, so a lane that stores its own record writes 16 bytes into 64
// different cache lines per instruction -- 0.47 of the kernel's 1.79 ms per 49 152 C2 windows went into those partial-line writes
// (probe build without the stores).  The records are staged through LDS instead, 32 slots of every lane at a time ([lane][32 + 2 pad]),
// and leave as 256-byte runs: a store instruction covers four windows x 16 lanes x 16 bytes = eight full lines.
#ifndef LIW_SLAB_NT
#define LIW_SLAB_NT 3                 // bit 0: the packed rows are loaded, bit 1: the records stored with the nontemporal hint (both are touched once per
#endif                                // launch; 1.67 -> 1.53 ms per 49 152 C2 windows on one box, 1.64 / 1.58 with one of the two)
constexpr int STG = 34;               // doubles per lane of a staged chunk (32 slots + 2 of padding: conflict-free 16-byte accesses)
template <bool BOTH, int C_, int I_ = 0> __device__ __forceinline__ void slab_stage(double* st, const double* acc) {
    if constexpr (I_ < 16) {
        typedef double __attribute__((ext_vector_type(2))) dbl2;
        dbl2 v;
        v.x = slab_slot<BOTH, 32 * C_ + 2 * I_>(acc); v.y = slab_slot<BOTH, 32 * C_ + 2 * I_ + 1>(acc);
        *reinterpret_cast<dbl2*>(st + 2 * I_) = v;
        slab_stage<BOTH, C_, I_ + 1>(st, acc);
    }
}
template <bool BOTH, int C_> __device__ __forceinline__ void slab_flush(double* lds, const unsigned long long* pw, const double* acc, int lane) {
    if constexpr (C_ < 4) {
        typedef double __attribute__((ext_vector_type(2))) dbl2;
        lds_sync();
        slab_stage<BOTH, C_>(lds + lane * STG, acc);
        lds_sync();
        const int wq = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int w = 4 * t + wq;
            const unsigned long long p = pw[w];
            const dbl2 v = *reinterpret_cast<const dbl2*>(lds + w * STG + 2 * pc);
#if LIW_SLAB_NT & 2
            if (p) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(reinterpret_cast<double*>(p) + 32 * C_ + 2 * pc));
#else
            if (p) *reinterpret_cast<dbl2*>(reinterpret_cast<double*>(p) + 32 * C_ + 2 * pc) = v;
#endif
        }
        slab_flush<BOTH, C_ + 1>(lds, pw, acc, lane);
    }
}

// sqrt(min(len1, len2) / 0.04) of a block (laser_factor.h:38-42) from the two segments' difference vectors
__device__ __forceinline__ double slab_weight(double d1x, double d1y, double d2x, double d2y) {
    const double l2 = fmin(d1x * d1x + d1y * d1y, d2x * d2x + d2y * d2y);
    const double lmin25 = l2 > 0.0 ? 25.0 * (l2 * fast_rsqrt(l2)) : 0.0;
    return lmin25 > 0.0 ? lmin25 * fast_rsqrt(lmin25) : 0.0;
}

constexpr int SLAB = 64;
constexpr int ROWD = LASER_SLAB_ROWD;  // doubles per packed row
constexpr int NPL = ROWD / SLAB;       // planes per row: 8 end-point components (+ the block weight)

}  // namespace

// TWO waves per SIMD, eight per CU (round 5; one wave with the whole 512-entry register file until then: 256 + 74 registers, 24.5 kB of LDS).
// What made the <= 256-register build possible is not fewer values but shorter live ranges: left alone, the scheduler interleaves the
// phases of a block (transforms, line direction, the two residual rows, 90 pair products) and the blocks of the unrolled loop for
// instruction-level parallelism and needs ~370 registers (113 spilled at a 256 budget: 1.56 ms per 24 576 windows in round 4);
// __builtin_amdgcn_sched_barrier(0) between the phases and between the blocks keeps the order written here: 250 registers, no scratch, with
// 45 accumulators and TWO rows of end points in flight (the second wave of the SIMD covers the load latency that five rows in flight
// covered before; three rows: 6 spills, same time; four: 29 spills, 1.5x slower).  LDS: 20 of a transform record's 24 entries per pose,
// [entry][lane], re-read per block (40 ds_read_b64); the 2 x 2 matrices M of both poses stay in registers: 20 kB per wave.
// Measured per 49 152 C2 windows (tools/ktimes.py): 2.26 -> 1.94 ms (1.87 on a faster box of the pool); a block is ~360 fp64 instructions
// (ISA count; 289 by hand), i.e. the two waves keep the SIMD's fp64 pipe ~55 % busy.
// What did not help (each built and timed in round 4): blocks taken in pairs or as branch-free straight-line code for more instruction-level
// parallelism (register pressure: 1.23 ms / spills).
// BOTH = false (round 6): the one-free-pose topologies — MARG (every frame's blocks against the constant laser_match pose of the frame,
// solver.cpp:453-476) and TRACK (the newest frame's only, :669-698; the older frames' too when LinArgs::marg_older) — over the SAME packed
// rows: pose a is match_pose[window][frame][0:6], the theta_a columns and their ~40 % of a block's arithmetic drop out, 21 accumulators.
template <bool BOTH>
__device__ __forceinline__ void slab_item(const LinArgs& A, const DevParams& P, const int item) {
    const int lane = threadIdx.x & 63, n = A.n;
    const int s = item / n, f = item % n;
    SSTAMP(0);
    // lane -> window through the per-frame order of the windows by group length (k_laser_slab_order): the 64 lanes of a wave walk groups of
    // (nearly) equal length, whatever the spread of the batch
    const int b = A.laser_perm[(size_t)f * ((size_t)laser_slab_count(A.B) * SLAB) + (size_t)s * SLAB + lane];
    bool in = b >= 0;
    const int bb = in ? b : 0;
    // every index / state load of the prologue in ONE batch, unconditionally (bb is a valid window): as written until late round 5 — live
    // test, then the group range, then the partial-buffer selector, then the poses — a wave spent 13 k cycles on four dependent round trips
    const int g0 = A.group_off[bb * (n + 1) + f], g1 = A.group_off[bb * (n + 1) + f + 1];
    const int hm = A.has_match[bb * n + f];
    const int curv = A.lm ? A.lm[bb].cur : 0;
    double pa6[6], pb6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        pa6[k] = BOTH ? A.x[(size_t)bb * n * 15 + k] : A.match_pose[((size_t)bb * n + f) * 12 + k];
        pb6[k] = A.x[((size_t)bb * n + f) * 15 + k];
    }
    if (in) in = window_live(A, bb);
    if (!__any(in)) return;
    // (tracking: only the newest frame's blocks are part of the problem; the older frames' records are written as zeros — or, marg_older,
    //  as their marginalisation-topology sums — exactly as k_lin_laser<false> leaves them)
    const bool fon = in && hm != 0 && (BOTH || A.mode != LIW_MODE_TRACK || f == n - 1 || A.marg_older);
    const int cnt = fon ? g1 - g0 : 0;
    int maxc = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o, 64));
    maxc = __builtin_amdgcn_readfirstlane(maxc);
    const int psel = (in && A.lm) ? (A.candidate ? 1 - curv : curv) : 0;
    // transform records of pose a (frame 0 of the lane's window) and pose b (frame f): [entry][lane] in LDS, re-read per block
    constexpr int TFR = 4;                       // the first TFR entries of a record (M) stay in registers: 2 x 20 x 64 doubles of LDS = 20 kB per wave, eight waves per CU
    __shared__ double lds[2 * (TF2 - TFR) * SLAB];      // (the epilogue stages the records through the same 20 kB)
    double* const lta = lds;
    double* const ltb = lds + (TF2 - TFR) * SLAB;
    double ra[TFR], rb[TFR];
    SSTAMP(1);
    {
        // both records under ONE condition: the two dual-number chains are independent and interleave (as two conditionals they ran one
        // after the other: 18 k cycles of a wave's 260 k)
        double ta2[TF2], tb2[TF2];
        if (in && maxc > 0) { frame_tf2(P, pa6, ta2); frame_tf2(P, pb6, tb2); }
        else {
#pragma unroll
            for (int k = 0; k < TF2; ++k) { ta2[k] = 0.0; tb2[k] = 0.0; }
        }
#pragma unroll
        for (int k = 0; k < TF2; ++k) {
            if (k < TFR) { ra[k] = ta2[k]; rb[k] = tb2[k]; }
            else { lta[(k - TFR) * SLAB + lane] = ta2[k]; ltb[(k - TFR) * SLAB + lane] = tb2[k]; }
        }
    }
    lds_sync();
    SSTAMP(2);
    constexpr int NACC = BOTH ? 45 : 21;
    double acc[NACC];
#pragma unroll
    for (int e = 0; e < NACC; ++e) acc[e] = 0.0;
    const double* row0 = A.laser_pk + (size_t)A.laser_slab_off[(size_t)s * n + f] * ROWD + lane;
    auto load_row = [&](double* q, int j) {
#if defined(LIW_SLAB_PROBE) && LIW_SLAB_PROBE == 1
        const double* r = row0 + (size_t)(j & 1) * ROWD;       // probe: every load hits the slab's first two rows (cache-resident): compute-only time
#else
        const double* r = row0 + (size_t)j * ROWD;
#endif
#pragma unroll
#if LIW_SLAB_NT & 1
        for (int c = 0; c < NPL; ++c) q[c] = __builtin_nontemporal_load(r + c * SLAB);
#else
        for (int c = 0; c < NPL; ++c) q[c] = r[c * SLAB];
#endif
    };
    const double w0 = P.laser_sqrt_info;
#define TA(k) ((k) < TFR ? ra[(k) < TFR ? (k) : 0] : lta[((k) < TFR ? 0 : (k) - TFR) * SLAB + lane])
#define TB(k) ((k) < TFR ? rb[(k) < TFR ? (k) : 0] : ltb[((k) < TFR ? 0 : (k) - TFR) * SLAB + lane])
    auto block = [&](const double* p) {     // one laser_factor block (k_lin_laser_body.inc, 2-D, both poses free): rows + pair products
#if defined(LIW_SLAB_PROBE) && LIW_SLAB_PROBE == 2
        // probe: no arithmetic, the loads only (memory-only time)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += p[c];
        return;
#endif
        const double d1x = p[0] - p[2], d1y = p[1] - p[3];
        const double d2x = p[4] - p[6], d2y = p[5] - p[7];
        const double l2 = fmin(d1x * d1x + d1y * d1y, d2x * d2x + d2y * d2y);
        const double lmin25 = l2 > 0.0 ? 25.0 * (l2 * fast_rsqrt(l2)) : 0.0;
        const double sum = lmin25 > 0.0 ? lmin25 * fast_rsqrt(lmin25) : 0.0;
        double Ap[2], Bp[2], C[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            Ap[r] = TA(4 + r) + TA(r * 2) * p[0] + TA(r * 2 + 1) * p[1];
            Bp[r] = TA(4 + r) + TA(r * 2) * p[2] + TA(r * 2 + 1) * p[3];
            C[0][r] = TB(4 + r) + TB(r * 2) * p[4] + TB(r * 2 + 1) * p[5];
            C[1][r] = TB(4 + r) + TB(r * 2) * p[6] + TB(r * 2 + 1) * p[7];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double ux = Bp[0] - Ap[0], uy = Bp[1] - Ap[1];
        const double zz = ux * ux + uy * uy;
        const bool regular = zz > 0.0;
        const double rlen = regular ? fast_rsqrt(zz) : 1.0;
        const double lx = ux * rlen, ly = uy * rlen;
        double dBx[3], dBy[3], dlx[3], dly[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double m0 = TA(6 + 4 * k), m1 = TA(6 + 4 * k + 1), m3 = TA(6 + 4 * k + 2), m4 = TA(6 + 4 * k + 3);
            const double t0 = TA(18 + 2 * k), t1 = TA(18 + 2 * k + 1);
            const double dAx = t0 + m0 * p[0] + m1 * p[1];
            const double dAy = t1 + m3 * p[0] + m4 * p[1];
            dBx[k] = t0 + m0 * p[2] + m1 * p[3];
            dBy[k] = t1 + m3 * p[2] + m4 * p[3];
            const double dux = dBx[k] - dAx, duy = dBy[k] - dAy;
            const double pr = lx * dux + ly * duy;
            dlx[k] = (dux - lx * pr) * rlen;
            dly[k] = (duy - ly * pr) * rlen;
        }
        const double w = sum * w0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            const double pt0 = p[4 + 2 * k], pt1 = p[5 + 2 * k];
            const double ex = C[k][0] - Bp[0], ey = C[k][1] - Bp[1];
            double dCx[3], dCy[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                dCx[m] = TB(18 + 2 * m) + TB(6 + 4 * m) * pt0 + TB(6 + 4 * m + 1) * pt1;
                dCy[m] = TB(18 + 2 * m + 1) + TB(6 + 4 * m + 2) * pt0 + TB(6 + 4 * m + 3) * pt1;
            }
            double dist, jc[8], ws = w;
            if (regular) {
                const double sg = lx * ey - ly * ex;
                ws = sg < 0.0 ? -w : w;
                dist = fabs(sg);
                {   // a point exactly on the line: norm() of a zero Jet in the reference (common.h:94) -> NaN derivative (k_lin_laser_body.inc)
                    const double prj = lx * ex + ly * ey;
                    const double vx = ex - prj * lx, vy = ey - prj * ly;
                    if (vx * vx + vy * vy == 0.0) ws = __builtin_nan("");
                }
                jc[0] = ly; jc[1] = -lx;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    jc[2 + m] = dlx[m] * ey - dly[m] * ex - lx * dBy[m] + ly * dBx[m];
                    jc[5 + m] = lx * dCy[m] - ly * dCx[m];
                }
            } else {   // zero-length reference segment: distance to the point B (Jet semantics of normalized(0))
                dist = sqrt(ex * ex + ey * ey);
                const double nx = ex / dist, ny = ey / dist;
                jc[0] = -nx; jc[1] = -ny;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    jc[2 + m] = -(nx * dBx[m] + ny * dBy[m]);
                    jc[5 + m] = nx * dCx[m] + ny * dCy[m];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            double rw[9];
#pragma unroll
            for (int c = 0; c < 8; ++c) rw[c] = ws * jc[c];
            rw[8] = sum * (w0 * dist);
            if constexpr (BOTH) {
#pragma unroll
                for (int c1 = 0; c1 < 9; ++c1)
#pragma unroll
                    for (int c2 = c1; c2 < 9; ++c2) acc[c1 * 9 - c1 * (c1 - 1) / 2 + (c2 - c1)] += rw[c1] * rw[c2];
            } else {   // own-pose columns: d/dp_b = -d/dp_a, theta_b, residual
                const double r1[6] = {-rw[0], -rw[1], rw[5], rw[6], rw[7], rw[8]};
#pragma unroll
                for (int c1 = 0; c1 < 6; ++c1)
#pragma unroll
                    for (int c2 = c1; c2 < 6; ++c2) acc[c1 * 6 - c1 * (c1 - 1) / 2 + (c2 - c1)] += r1[c1] * r1[c2];
            }
        }
    };
#ifndef LIW_SLAB_ALG
#define LIW_SLAB_ALG 1      // 0: every block through the general form (A/B aid)
#endif
#if LIW_SLAB_ALG == 1
    // The same block with the algebra folded (round 5, late): only DIFFERENCES of mapped points enter the rows, so the reference segment's
    // direction is M_a (B - A) (no translation, no mapped A), the whole row is scaled by the block's weight ONCE through the line
    // direction (l_w = w l: the sign of the distance then needs no select — the pair products of a row are even in it, and the products
    // with the residual carry w^2 J sg), the theta_b columns are affine in the point with per-block coefficients
    // (lx dCy - ly dCx = alpha + beta x + gamma y), the theta_a columns share kappa_m = l_w x dB_m between the two points, and the
    // reference's norm()-of-a-zero-Jet test (common.h:94) is only evaluated where |sg| is at round-off level.
    bool irregular = false;
    auto block_fast = [&](const double* p) {
        const double d1x = p[0] - p[2], d1y = p[1] - p[3];
#if LIW_SLAB_WPLANE
        const double w = p[8] * w0;                 // the block's weight from the ninth plane of the row (slab_weight, computed once per solve by the re-pack)
#else
        const double d2x = p[4] - p[6], d2y = p[5] - p[7];
        const double w = slab_weight(d1x, d1y, d2x, d2y) * w0;
#endif
        const double ux = -(ra[0] * d1x + ra[1] * d1y), uy = -(ra[2] * d1x + ra[3] * d1y);
        const double zz = ux * ux + uy * uy;
        if (__builtin_expect(!(zz > 0.0), 0)) { irregular = true; return; }      // zero-length reference segment (or NaN input): the general form, in a pass of its own behind the loop
        const double Bx = TA(4) + ra[0] * p[2] + ra[1] * p[3], By = TA(5) + ra[2] * p[2] + ra[3] * p[3];
        const double tbx = TB(4), tby = TB(5);
        double ex[2], ey[2];      // C - B with both mapped points formed as the reference forms them: a point that coincides with B gives an exact zero
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            ex[k] = (tbx + rb[0] * p[4 + 2 * k] + rb[1] * p[5 + 2 * k]) - Bx;
            ey[k] = (tby + rb[2] * p[4 + 2 * k] + rb[3] * p[5 + 2 * k]) - By;
        }
        __builtin_amdgcn_sched_barrier(0);
        const double rlen = fast_rsqrt(zz);
        const double lx = ux * rlen, ly = uy * rlen;
        const double rlw = rlen * w;
        const double lxw = ux * rlw, lyw = uy * rlw;
        double dlxw[3], dlyw[3], kap[3];
        if constexpr (BOTH) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double m0 = TA(6 + 4 * k), m1 = TA(6 + 4 * k + 1), m3 = TA(6 + 4 * k + 2), m4 = TA(6 + 4 * k + 3);
                const double dux = -(m0 * d1x + m1 * d1y), duy = -(m3 * d1x + m4 * d1y);
                const double pr = lx * dux + ly * duy;
                dlxw[k] = (dux - lx * pr) * rlw;
                dlyw[k] = (duy - ly * pr) * rlw;
                const double dBx = TA(18 + 2 * k) + m0 * p[2] + m1 * p[3];
                const double dBy = TA(18 + 2 * k + 1) + m3 * p[2] + m4 * p[3];
                kap[k] = lyw * dBx - lxw * dBy;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        double al[3], be[3], ga[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            al[m] = lxw * TB(18 + 2 * m + 1) - lyw * TB(18 + 2 * m);
            be[m] = lxw * TB(6 + 4 * m + 2) - lyw * TB(6 + 4 * m);
            ga[m] = lxw * TB(6 + 4 * m + 3) - lyw * TB(6 + 4 * m + 1);
        }
        const double tiny = 1e-13 * w;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            const double pt0 = p[4 + 2 * k], pt1 = p[5 + 2 * k];
            if constexpr (BOTH) {
                double rw[9];
                rw[0] = lyw; rw[1] = -lxw;
                rw[8] = lxw * ey[k] - lyw * ex[k];               // w sg, signed
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    rw[2 + m] = dlxw[m] * ey[k] - dlyw[m] * ex[k] + kap[m];
                    rw[5 + m] = al[m] + be[m] * pt0 + ga[m] * pt1;
                }
                if (__builtin_expect(fabs(rw[8]) <= tiny * (fabs(ex[k]) + fabs(ey[k])), 0)) {
                    // a point exactly on the line: norm() of a zero Jet in the reference (common.h:94) -> NaN derivative (k_lin_laser_body.inc)
                    const double prj = lx * ex[k] + ly * ey[k];
                    const double vx = ex[k] - prj * lx, vy = ey[k] - prj * ly;
                    if (vx * vx + vy * vy == 0.0) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) rw[c] = __builtin_nan("");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c1 = 0; c1 < 9; ++c1)
#pragma unroll
                    for (int c2 = c1; c2 < 9; ++c2) acc[c1 * 9 - c1 * (c1 - 1) / 2 + (c2 - c1)] += rw[c1] * rw[c2];
            } else {
                double rw[6];                                    // own-pose columns px py th0 th1 th2 | residual (d/dp_b = -d/dp_a)
                rw[0] = -lyw; rw[1] = lxw;
                rw[5] = lxw * ey[k] - lyw * ex[k];
#pragma unroll
                for (int m = 0; m < 3; ++m) rw[2 + m] = al[m] + be[m] * pt0 + ga[m] * pt1;
                if (__builtin_expect(fabs(rw[5]) <= tiny * (fabs(ex[k]) + fabs(ey[k])), 0)) {
                    const double prj = lx * ex[k] + ly * ey[k];
                    const double vx = ex[k] - prj * lx, vy = ey[k] - prj * ly;
                    if (vx * vx + vy * vy == 0.0) {
#pragma unroll
                        for (int c = 0; c < 5; ++c) rw[c] = __builtin_nan("");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c1 = 0; c1 < 6; ++c1)
#pragma unroll
                    for (int c2 = c1; c2 < 6; ++c2) acc[c1 * 6 - c1 * (c1 - 1) / 2 + (c2 - c1)] += rw[c1] * rw[c2];
            }
        }
    };
#define LIW_SLAB_BLOCK block_fast
#else
#define LIW_SLAB_BLOCK block
#endif
    // rows in flight: LIW_SLAB_ROWS register sets in rotation.  The loads are UNCONDITIONAL (row index clamped to the slab's last row):
    // behind a branch the compiler can no longer count them and waits for every outstanding load before each block
#ifndef LIW_SLAB_ROWS
#define LIW_SLAB_ROWS 2
#endif
    const int last = maxc - 1;
#if LIW_SLAB_ROWS == 2
    double q0[NPL], q1[NPL];
    if (maxc > 0) {
        load_row(q0, 0);
        for (int j = 0; j < maxc; j += 2) {
            asm volatile("" ::: "memory");
            load_row(q1, min(j + 1, last));
            if (j < cnt) LIW_SLAB_BLOCK(q0);
            __builtin_amdgcn_sched_barrier(0);
            load_row(q0, min(j + 2, last));
            if (j + 1 < cnt) LIW_SLAB_BLOCK(q1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#elif LIW_SLAB_ROWS == 3
    double q0[NPL], q1[NPL], q2[NPL];
    if (maxc > 0) {
        load_row(q0, 0); load_row(q1, min(1, last));
        for (int j = 0; j < maxc; j += 3) {
            asm volatile("" ::: "memory");
            load_row(q2, min(j + 2, last));
            if (j < cnt) LIW_SLAB_BLOCK(q0);
            __builtin_amdgcn_sched_barrier(0);
            load_row(q0, min(j + 3, last));
            if (j + 1 < cnt) LIW_SLAB_BLOCK(q1);
            __builtin_amdgcn_sched_barrier(0);
            load_row(q1, min(j + 4, last));
            if (j + 2 < cnt) LIW_SLAB_BLOCK(q2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#else
    double q0[NPL], q1[NPL], q2[NPL], q3[NPL];
    if (maxc > 0) {
        load_row(q0, 0); load_row(q1, min(1, last)); load_row(q2, min(2, last));
        for (int j = 0; j < maxc; j += 4) {
            asm volatile("" ::: "memory");
            load_row(q3, min(j + 3, last));
            if (j < cnt) LIW_SLAB_BLOCK(q0);
            __builtin_amdgcn_sched_barrier(0);
            load_row(q0, min(j + 4, last));
            if (j + 1 < cnt) LIW_SLAB_BLOCK(q1);
            __builtin_amdgcn_sched_barrier(0);
            load_row(q1, min(j + 5, last));
            if (j + 2 < cnt) LIW_SLAB_BLOCK(q2);
            __builtin_amdgcn_sched_barrier(0);
            load_row(q2, min(j + 6, last));
            if (j + 3 < cnt) LIW_SLAB_BLOCK(q3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#endif
#if LIW_SLAB_ALG == 1
    if (__builtin_expect(__any(irregular), 0)) {
        // the blocks block_fast left out (their pair products join the lane's sums behind the regular ones): same test, general form.
        // Kept out of the main loop so that its live ranges do not count against the rows in flight there.
        for (int j = 0; j < maxc; ++j) {
            double q[NPL];
            load_row(q, j);
            const double d1x = q[0] - q[2], d1y = q[1] - q[3];
            const double ux = -(ra[0] * d1x + ra[1] * d1y), uy = -(ra[2] * d1x + ra[3] * d1y);
            if (irregular && j < cnt && !(ux * ux + uy * uy > 0.0)) block(q);
        }
    }
#endif
    SSTAMP(3);
#undef TA
#undef TB
#undef LIW_SLAB_BLOCK
#if defined(LIW_SLAB_DIRECT_STORE)
    if (in) {
        double* out = (psel ? A.PL[1] : A.PL[0]) + ((size_t)bb * n + f) * LP;
        slab_store<BOTH, 0>(out, acc);
    }
#else
    {
        static_assert(SLAB * STG + SLAB <= 2 * (TF2 - TFR) * SLAB && LP == 128, "staging area of the record epilogue");
        unsigned long long* pw = reinterpret_cast<unsigned long long*>(lds + SLAB * STG);
        lds_sync();                                  // every lane is past its last transform read
        pw[lane] = in ? (unsigned long long)((psel ? A.PL[1] : A.PL[0]) + ((size_t)bb * n + f) * LP) : 0ull;
        slab_flush<BOTH, 0>(lds, pw, acc, lane);
    }
#endif
    if (in && A.CS[0]) (psel ? A.CS[1] : A.CS[0])[cs_index(n, bb, CS_LASER, f)] = acc[BOTH ? slab_pairidx(8, 8) : slab_pairidx1(5, 5)];
    SSTAMP(4);
}
__global__ __launch_bounds__(64, 2) void k_lin_laser_slab(LinArgs A, DevParams P) { slab_item<true>(A, P, (int)blockIdx.x); }
// one pose free (MARG / TRACK): ~205 fp64 instructions per block instead of ~360, 21 accumulators; same rows, same epilogue
__global__ __launch_bounds__(64, 2) void k_lin_laser_slab1(LinArgs A, DevParams P) { slab_item<false>(A, P, (int)blockIdx.x); }
// (Measured late in round 5, tools/bracket_time.py: a grid capped at 768 ... 2 048 persistent work-groups, so that the IMU role's waves run
//  NEXT TO this kernel's instead of behind them, makes the linearise bracket slower — 4.45 - 4.65 against 4.38 ms per 49 152 windows: the
//  throughput of either kernel follows its resident waves, a memory-bound and a pipe-bound wave on one SIMD do not add up.)

// Per-frame order of the windows (round 6).  A (slab, frame) wave pads its 64 groups to the longest one; with the
// windows taken in batch order a ragged batch (per-window L from 500 to 4 000, frames with 0 .. 400 blocks) packed 4.1 rows per row of data
// and fell back to the lane-per-block kernel.  The 64 lanes of a wave need not be the same windows for every frame — a (slab, frame) item is
// independent of every other — so each frame gets its own order: windows sorted by the length of THEIR group of that frame, longest first
// (the heavy waves start first), perm[f][position] = window, -1 behind the last.  Counting sort by one work-group per frame (lengths
// clamped to 4 095; the cursor of a bin is an integer LDS atomic: windows of equal length may change places from run to run, which no result
// can see — a lane forms its (window, frame) record alone, in block order).
constexpr int SORT_BINS = 4096;
__global__ __launch_bounds__(1024) void k_laser_slab_order(int B, int n, const int* group_off, int* perm) {
    __shared__ int bin[SORT_BINS];
    __shared__ int part[1024];
    const int f = (int)blockIdx.x, t = (int)threadIdx.x;
    const size_t Bp = (size_t)((B + SLAB - 1) / SLAB) * SLAB;
    for (int e = t; e < SORT_BINS; e += 1024) bin[e] = 0;
    __syncthreads();
    auto key = [&](int b) { return SORT_BINS - 1 - min(group_off[b * (n + 1) + f + 1] - group_off[b * (n + 1) + f], SORT_BINS - 1); };   // bin 0 = longest
    for (int b = t; b < B; b += 1024) atomicAdd(&bin[key(b)], 1);
    __syncthreads();
    int sum = 0;
#pragma unroll
    for (int k = 0; k < SORT_BINS / 1024; ++k) sum += bin[t * (SORT_BINS / 1024) + k];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int base = part[t] - sum;
#pragma unroll
    for (int k = 0; k < SORT_BINS / 1024; ++k) { const int h = bin[t * (SORT_BINS / 1024) + k]; bin[t * (SORT_BINS / 1024) + k] = base; base += h; }
    __syncthreads();
    for (int b = t; b < B; b += 1024) perm[(size_t)f * Bp + atomicAdd(&bin[key(b)], 1)] = b;
    for (size_t e = (size_t)B + t; e < Bp; e += 1024) perm[(size_t)f * Bp + e] = -1;
}
// longest group of every (slab, frame): mx[s * n + f] = max over the slab's windows (in the frame's order) of the block count of (window, f)
__global__ void k_laser_slab_max(int B, int n, const int* group_off, const int* perm, int* mx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = (B + SLAB - 1) / SLAB;
    if (t >= S * n) return;
    const int s = t / n, f = t % n;
    int m = 0;
    for (int l = 0; l < SLAB; ++l) {
        const int b = perm[(size_t)f * ((size_t)S * SLAB) + (size_t)s * SLAB + l];
        if (b >= 0) m = max(m, group_off[b * (n + 1) + f + 1] - group_off[b * (n + 1) + f]);
    }
    mx[t] = m;
}
// exclusive prefix sum of mx (rows) by one work-group; off[N] = total rows, off[N + 1] = the z flag of k_laser_z_scan (one read-back for both)
__global__ __launch_bounds__(1024) void k_laser_slab_scan(int N, const int* mx, long long* off, const int* hz) {
    __shared__ long long part[1024];
    const int t = threadIdx.x, per = (N + 1023) / 1024;
    long long sum = 0;
    for (int k = 0; k < per; ++k) { const int e = t * per + k; if (e < N) sum += mx[e]; }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long base = part[t] - sum;
    for (int k = 0; k < per; ++k) { const int e = t * per + k; if (e < N) { off[e] = base; base += mx[e]; } }
    if (t == 1023) { off[N] = part[1023]; off[N + 1] = hz ? (long long)*hz : 0; }
}
// re-pack: work-group (s, f) turns, plane by plane, the 64 windows' runs of consecutive blocks (each contiguous in the caller's plane) into rows
// of 64 lanes through an LDS tile: 128-byte segments in (16 lanes along a window's run), 512-byte rows out.  (The first version read
// one window per lane: 64 scattered 8-byte loads per instruction, 4.9 ms per 24 576 C2 windows.)
__global__ __launch_bounds__(256) void k_laser_slab_pack(int B, int n, long Ltot, const int* group_off, const int* perm, const double* pts, const long long* off, const int* mx, double* pk) {
    __shared__ double tile[64 * 65];
    __shared__ int g0s[SLAB], cnts[SLAB];
    const int s = (int)blockIdx.x / n, f = (int)blockIdx.x % n, t = threadIdx.x;
    if (t < SLAB) {
        const int b = perm[(size_t)f * ((size_t)((B + SLAB - 1) / SLAB) * SLAB) + (size_t)s * SLAB + t];
        g0s[t] = b >= 0 ? group_off[b * (n + 1) + f] : 0;
        cnts[t] = b >= 0 ? group_off[b * (n + 1) + f + 1] - g0s[t] : 0;
    }
    __syncthreads();
    const int maxc = mx[(size_t)s * n + f];
    double* base = pk + (size_t)off[(size_t)s * n + f] * ROWD;
    for (int jb = 0; jb < maxc; jb += 64) {
        for (int c = 0; c < 8; ++c) {
            const double* plane = pts + (size_t)(c + c / 2) * (size_t)Ltot;     // planes 0 1 3 4 6 7 9 10 (x, y of the four end points)
            for (int pass = 0; pass < 4; ++pass) {
                const int l = pass * 16 + (t >> 4), jj = t & 15;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = jb + jj + 16 * i;
                    tile[(jj + 16 * i) * 65 + l] = j < cnts[l] ? plane[(size_t)g0s[l] + j] : 0.0;
                }
            }
            __syncthreads();
            const int rows = min(64, maxc - jb);
            for (int r = t >> 6; r < rows; r += 4) base[(size_t)(jb + r) * ROWD + c * SLAB + (t & 63)] = tile[r * 65 + (t & 63)];
            __syncthreads();
        }
#if LIW_SLAB_WPLANE
        {   // ninth plane: the weights of the rows just written (read back lane-linear: the lines are still in the cache)
            __threadfence_block();
            __syncthreads();
            const int rows = min(64, maxc - jb);
            for (int r = t >> 6; r < rows; r += 4) {
                const double* q = base + (size_t)(jb + r) * ROWD + (t & 63);
                base[(size_t)(jb + r) * ROWD + 8 * SLAB + (t & 63)] = slab_weight(q[0] - q[2 * SLAB], q[SLAB] - q[3 * SLAB], q[4 * SLAB] - q[6 * SLAB], q[5 * SLAB] - q[7 * SLAB]);
            }
        }
#endif
    }
}

#ifdef LIW_CLK
extern "C" void liw_debug_clk_slab(long long* out, int nn) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk_slab), sizeof(long long) * nn); }
#endif
// A/B aid (LIW_SLAB_BATCH_ORDER=1): the windows in batch order for every frame — the layout until round 5
__global__ void k_laser_slab_identity(int B, int n, int* perm) {
    const size_t Bp = (size_t)((B + SLAB - 1) / SLAB) * SLAB;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Bp * n) perm[t] = (int)(t % Bp) < B ? (int)(t % Bp) : -1;
}
void launch_laser_slab_prepare(int B, int n, const int* group_off, int* perm, int* mx, long long* off, const int* hz, hipStream_t s) {
    const int N = laser_slab_count(B) * n;
    if (std::getenv("LIW_SLAB_BATCH_ORDER")) hipLaunchKernelGGL(k_laser_slab_identity, dim3((unsigned)(((size_t)N * SLAB + 255) / 256)), dim3(256), 0, s, B, n, perm);
    else hipLaunchKernelGGL(k_laser_slab_order, dim3((unsigned)n), dim3(1024), 0, s, B, n, group_off, perm);
    hipLaunchKernelGGL(k_laser_slab_max, dim3((N + 255) / 256), dim3(256), 0, s, B, n, group_off, (const int*)perm, mx);
    hipLaunchKernelGGL(k_laser_slab_scan, dim3(1), dim3(1024), 0, s, N, (const int*)mx, off, hz);
}
void launch_laser_slab_pack(int B, int n, long Ltot, const int* group_off, const int* perm, const double* pts, const long long* off, const int* mx, double* pk, hipStream_t s) {
    hipLaunchKernelGGL(k_laser_slab_pack, dim3((unsigned)(laser_slab_count(B) * n)), dim3(256), 0, s, B, n, Ltot, group_off, perm, pts, off, mx, pk);
}
void launch_lin_laser_slab(const LinArgs& A, const DevParams& P, hipStream_t s) {
    if (A.mode == LIW_MODE_INIT) hipLaunchKernelGGL(k_lin_laser_slab, dim3((unsigned)(laser_slab_count(A.B) * A.n)), dim3(64), 0, s, A, P);
    else hipLaunchKernelGGL(k_lin_laser_slab1, dim3((unsigned)(laser_slab_count(A.B) * A.n)), dim3(64), 0, s, A, P);
}

}  // namespace liw
