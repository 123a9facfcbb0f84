// liw_capi.hip — the extern "C" boundary declared in include/liw_window.h (host side, HIP runtime).
// No CPU fallback: without a usable gfx950 device every compute entry point returns LIW_ENODEV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "liw_kernels.hpp"

using namespace liw;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return -1;
        cap = bytes ? bytes : 8;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct liw_ctx {
    liw_params prm;
    DevParams dp;
    std::string err;
    bool have_device = false;
    hipStream_t stream = nullptr;
    // single-window state
    liw_window hw{};
    bool have_window = false;
    int n = 0, L = 0;
    DevBuf arena;                 // every input array of the window, one allocation (one H2D copy per liw_set_window)
    // page-locked memory: [image 0 | image 1 | read-back record].  An image is the byte image of the arena; image `img_cur` mirrors what
    // the device holds (liw_solve folds the solved states back into it), the other one is where the next liw_set_window stages its
    // window — identical to the mirror means the device already has this window: no upload, and a marginalisation result computed
    // speculatively behind the solve stays valid (the lvio_2d::solver shim re-flattens the same frames for marginalization()).
    void* pinned = nullptr;
    size_t pinned_cap = 0, img_cap = 0, readback_cap = 0;
    int img_cur = 0;
    bool img_valid = false;
    size_t part_off[13] = {0}, part_bytes[13] = {0};
    hipEvent_t ev_upload = nullptr;   // completion of the last host-to-device copy (liw_set_window does not wait for it)
    DevBuf prior_X, prior_J, prior_R, has_prior, ws, scratch;
    // speculative marginalisation (TRACK solves): next prior + packed result record
    DevBuf priorn_X, priorn_J, priorn_R, has_priorn, result, marg_status;
    bool spec_marg = true, spec_valid = false;
    bool reattach = true;         // recognise a window whose bytes the device already holds (LIW_NO_REATTACH: always upload)
    double spec_out[36 + 225 + 15];
    liw_batch sb{};
    liw_ws_layout lay{};
    int hist_records = 0;
    int last_iters = 0;
    int solved_records = 0;       // history records of the last completed liw_solve on the current window (0: none)
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev_lin, ev_step, ev_x;
    size_t ev_lin_used = 0, ev_step_used = 0, xev_used = 0;
    bool time_exchange = false;         // liw_batch_exchange_timing
    int pack_seq = 0;                   // sequence number of the last k_pack_result (completion word of the read-back record)
    const double* last_x = nullptr;     // the exchanged buffer of the last liw_batch_solve_sharded exchange (active-window trailer)
    int last_x_copies = 1;
    // native peer-write exchange (liw_batch_p2p_setup)
    P2pPeers p2p{};
    int p2p_rank = 0, p2p_world = 0;
    unsigned long long p2p_epoch = 0;
    DevBuf p2p_err;
    LinFork fork{};
    bool have_fork = false;
    // lane-per-group laser role of large 2-D batches (k_laser_slab.hip): the batch's laser blocks re-packed once per solve into ctx-owned
    // memory (its size depends on the block counts, which liw_batch_ws_layout does not know), valid for the solve `lpk_key` names
    DevBuf lpk, lpk_off, lpk_mx, lpk_perm;
    struct { const void* ws; const void* pts; const void* frame; int B, n; long Ltot; } lpk_key{};
    bool lpk_on = false;
    long long lpk_rows = 0;
    int* rb_pin = nullptr;        // page-locked words for the small read-backs of the batched solve (a pageable destination sends hipMemcpyAsync down a slow, serialising path)
    // graph cache
    hipGraphExec_t gexec = nullptr;
    std::vector<unsigned char> gkey;
};

static int fail(liw_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        char buf[256];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        c->err = buf;
    }
    return code;
}
#define HIPCHK(c, call)                                                   \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fail((c), LIW_EHIP, #call, e_);      \
    } while (0)
#define NEEDDEV(c)                                                                                         \
    do {                                                                                                   \
        if (!(c)) return LIW_EINVAL;                                                                       \
        if (!(c)->have_device) return fail((c), LIW_ENODEV, "no usable gfx950 device (no CPU fallback)"); \
    } while (0)

// Quaternion(R).toRotationMatrix() round trip of the parameter loader (reference src/utilies/params.cpp:44-54,
// src/utilies/common.h:183-189); no normalisation of the quaternion in between, as there.
static void normalize_tf_host(double* R) {
    double c[4];
    double t = R[0] + R[4] + R[8];
    auto M = [&](int r, int cc) { return R[r * 3 + cc]; };
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        c[3] = 0.5 * t; t = 0.5 / t;
        c[0] = (M(2, 1) - M(1, 2)) * t; c[1] = (M(0, 2) - M(2, 0)) * t; c[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        c[i] = 0.5 * t; t = 0.5 / t;
        c[3] = (M(k, j) - M(j, k)) * t; c[j] = (M(j, i) + M(i, j)) * t; c[k] = (M(k, i) + M(i, k)) * t;
    }
    const double qw = c[3], qx = c[0], qy = c[1], qz = c[2];
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
void liw_normalize_rotation_host(double* R9) { normalize_tf_host(R9); }   // used by liw_laser.cpp
// context accessors for the other translation units of the library (k_posegraph.hip)
hipStream_t liw_ctx_stream(liw_ctx* c) { return c->stream; }
const DevParams* liw_ctx_devparams(liw_ctx* c) { return &c->dp; }
int liw_ctx_device(liw_ctx* c) { return c->prm.device; }
bool liw_ctx_has_device(liw_ctx* c) { return c && c->have_device; }
int liw_ctx_fail(liw_ctx* c, int code, const char* what) { return fail(c, code, what); }

void liw_fill_devparams(const liw_params* prm, DevParams* dp) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { dp->Riw[i * 3 + j] = prm->T_imu_to_wheel[i * 4 + j]; dp->Ril[i * 3 + j] = prm->T_imu_to_laser[i * 4 + j]; }
        dp->tiw[i] = prm->T_imu_to_wheel[i * 4 + 3];
        dp->til[i] = prm->T_imu_to_laser[i * 4 + 3];
    }
    if (prm->normalize_extrinsics) { normalize_tf_host(dp->Riw); normalize_tf_host(dp->Ril); }
    dp->g = prm->g;
    dp->laser_sqrt_info = 1.0 / prm->line_to_line_sigma;
    dp->ground_p_info = 1.0 / prm->manifold_p_sigma;
    dp->ground_q_info = 1.0 / prm->manifold_q_sigma;
    dp->fast_mode = prm->fast_mode;
}

extern "C" {

liw_ctx* liw_create(const liw_params* prm) {
    if (!prm) return nullptr;
    liw_ctx* c = new liw_ctx();
    c->prm = *prm;
    liw_fill_devparams(prm, &c->dp);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > prm->device && prm->device >= 0) {
        hipDeviceProp_t props;
        if (hipSetDevice(prm->device) == hipSuccess && hipGetDeviceProperties(&props, prm->device) == hipSuccess) {
            if (std::strstr(props.gcnArchName, "gfx950") != nullptr) {
                if (hipStreamCreate(&c->stream) == hipSuccess) c->have_device = true;
                // (role streams at the highest dispatch priority were tried: the dispatcher takes no notice while the laser kernel's waves
                // hold every register of the chip — same time line, tools/rocprof_timeline.py)
                if (c->have_device && hipStreamCreateWithFlags(&c->fork.side[0], hipStreamNonBlocking) == hipSuccess &&
                    hipStreamCreateWithFlags(&c->fork.side[1], hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&c->fork.ev_fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&c->fork.ev_join[0], hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&c->fork.ev_join[1], hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&c->fork.ev_compact, hipEventDisableTiming) == hipSuccess)
                    c->have_fork = true;
                if (std::getenv("LIW_SERIAL_ROLES")) c->have_fork = false;   // profiling aid: role kernels back to back
                if (std::getenv("LIW_NO_SPEC_MARG")) c->spec_marg = false;   // profiling / test aid: marginalise only when asked
                if (std::getenv("LIW_NO_REATTACH")) c->reattach = false;     // test aid: every liw_set_window uploads
                (void)hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming);
            } else {
                c->err = std::string("device is ") + props.gcnArchName + ", this library is built for gfx950 only";
            }
        }
    }
    if (!c->have_device && c->err.empty()) c->err = "no HIP device";
    return c;
}

void liw_destroy(liw_ctx* c) {
    if (!c) return;
    if (c->have_device) {
        (void)hipSetDevice(c->prm.device);
        DevBuf* bufs[] = {&c->lpk, &c->lpk_off, &c->lpk_mx, &c->lpk_perm, &c->p2p_err, &c->arena, &c->prior_X, &c->prior_J, &c->prior_R, &c->has_prior, &c->ws, &c->scratch,
                          &c->priorn_X, &c->priorn_J, &c->priorn_R, &c->has_priorn, &c->result, &c->marg_status};
        if (c->pinned) (void)hipHostFree(c->pinned);
        if (c->rb_pin) (void)hipHostFree(c->rb_pin);
        if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
        for (DevBuf* b : bufs) b->release();
        for (auto e : c->ev_lin) (void)hipEventDestroy(e);
        for (auto e : c->ev_step) (void)hipEventDestroy(e);
        for (auto e : c->ev_x) (void)hipEventDestroy(e);
        if (c->gexec) (void)hipGraphExecDestroy(c->gexec);
        if (c->have_fork) {
            (void)hipStreamDestroy(c->fork.side[0]); (void)hipStreamDestroy(c->fork.side[1]);
            (void)hipEventDestroy(c->fork.ev_fork); (void)hipEventDestroy(c->fork.ev_join[0]); (void)hipEventDestroy(c->fork.ev_join[1]); (void)hipEventDestroy(c->fork.ev_compact);
        }
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

const char* liw_last_error(const liw_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int liw_get_extrinsics(const liw_ctx* c, double* A, double* Bm) {
    if (!c) return LIW_EINVAL;
    auto put = [](const double* R, const double* t, double* m) {
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m[i * 4 + j] = R[i * 3 + j]; m[i * 4 + 3] = t[i]; }
        m[12] = m[13] = m[14] = 0.0; m[15] = 1.0;
    };
    if (A) put(c->dp.Riw, c->dp.tiw, A);
    if (Bm) put(c->dp.Ril, c->dp.til, Bm);
    return LIW_OK;
}

// ------------------------------------------------------------------------------------------ workspace
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
struct FullLayout {
    size_t PL[2], PI[2], PW[2], PG[2], CS[2], x_cand, group_off, lm, solve_ws, info, history, active, imu_pk, imu_pk_bad, bytes;
};
static FullLayout full_layout(int B, int n, int hist) {
    FullLayout f{};
    size_t o = 0;
    const int nm = n > 1 ? n - 1 : 1;
    for (int k = 0; k < 2; ++k) { f.PL[k] = o; o = al256(o + sizeof(double) * (size_t)B * n * LP); }
    for (int k = 0; k < 2; ++k) { f.PI[k] = o; o = al256(o + sizeof(double) * (size_t)B * pi_doubles_per_window(n)); }
    for (int k = 0; k < 2; ++k) { f.PW[k] = o; o = al256(o + sizeof(double) * (size_t)B * nm * PWS); }
    for (int k = 0; k < 2; ++k) { f.PG[k] = o; o = al256(o + sizeof(double) * (size_t)B * n * PGS); }
    f.x_cand = o; o = al256(o + sizeof(double) * (size_t)B * n * 15);
    f.group_off = o; o = al256(o + sizeof(int) * (size_t)B * (n + 1));
    f.active = o; o = al256(o + compact_list_bytes_host(B));
    f.lm = o; o = al256(o + sizeof(LmState) * (size_t)B);
    f.solve_ws = o; o = al256(o + sizeof(double) * (size_t)B * n * SOLVE_WS);
    f.info = o; o = al256(o + sizeof(liw_summary) * (size_t)B);
    f.history = hist > 0 ? o : 0;
    if (hist > 0) o = al256(o + sizeof(double) * (size_t)hist * B * n * 15);
    f.imu_pk_bad = o; o = al256(o + 2 * sizeof(int));
    f.imu_pk = o; o = al256(o + sizeof(double) * (size_t)B * nm * IMU_PK);
    for (int k = 0; k < 2; ++k) { f.CS[k] = o; o = al256(o + sizeof(double) * (size_t)B * 4 * n); }
    f.bytes = o;
    return f;
}
static WsView make_view(void* ws, int B, int n, int hist) {
    FullLayout f = full_layout(B, n, hist);
    char* base = (char*)ws;
    WsView v{};
    for (int k = 0; k < 2; ++k) {
        v.PL[k] = (double*)(base + f.PL[k]); v.PI[k] = (double*)(base + f.PI[k]);
        v.PW[k] = (double*)(base + f.PW[k]); v.PG[k] = (double*)(base + f.PG[k]);
    }
    v.x_cand = (double*)(base + f.x_cand);
    v.group_off = (int*)(base + f.group_off);
    v.lm = (LmState*)(base + f.lm);
    v.solve_ws = (double*)(base + f.solve_ws);
    v.info = (liw_summary*)(base + f.info);
    v.history = hist > 0 ? (double*)(base + f.history) : nullptr;
    v.history_records = hist;
    v.active = (int*)(base + f.active);
    v.pi_frame = pi_frame_format(B) ? 1 : 0;
    for (int k = 0; k < 2; ++k) v.CS[k] = (double*)(base + f.CS[k]);
    v.imu_pk = (double*)(base + f.imu_pk);
    v.imu_pk_bad = (int*)(base + f.imu_pk_bad);
    return v;
}

int liw_batch_ws_layout(int B, int n, int history_records, liw_ws_layout* out) {
    if (!out || B <= 0 || n <= 0 || n > 64) return LIW_EINVAL;
    FullLayout f = full_layout(B, n, history_records);
    out->bytes = f.bytes;
    out->laser_partial_off[0] = f.PL[0];
    out->laser_partial_off[1] = f.PL[1];
    out->laser_partial_bytes = sizeof(double) * (size_t)B * n * LP;
    out->info_off = f.info;
    out->history_off = f.history;
    return LIW_OK;
}

// ------------------------------------------------------------------------------------------ batch API
// min_n = 2 for the TRACK / MARG topologies: their prior block sits on frame n-2 (solver.cpp:234-254, :282-306), which does
// not exist in a 1-frame window (the reference never builds one: trajectory.cpp:525-560 always has the previous frame)
static int check_batch(liw_ctx* c, const liw_batch* b, int min_n = 1) {
    if (!b || b->B <= 0 || b->n <= 0 || b->n > 64) return fail(c, LIW_EINVAL, "bad batch (need 1 <= n <= 64, B >= 1)");
    if (b->n < min_n) return fail(c, LIW_EINVAL, "TRACK / MARG topology needs n >= 2 (the prior block is tied to frame n-2)");
    return LIW_OK;
}
static int min_frames(int mode) { return mode == LIW_MODE_INIT ? 1 : 2; }
// packed: the linearisation belongs to a solve opened by liw_batch_lm_begin (which packed the IMU block records into the workspace)
static bool lpk_matches(const liw_ctx* c, const liw_batch* b, const void* ws) {
    return c && c->lpk_on && c->lpk_key.ws == ws && c->lpk_key.pts == b->laser_pts && c->lpk_key.frame == b->laser_frame && c->lpk_key.B == b->B &&
           c->lpk_key.n == b->n && c->lpk_key.Ltot == (long)b->Ltot;
}
// Re-pack the laser blocks of a large 2-D batch for the lane-per-group kernel.  Runs behind group_offsets and the z scan of the solve being
// opened; ONE blocking 16-byte read-back (rows needed, z flag) sizes the ctx-owned buffer — only for batches that fill the chip that way
// (>= 2 048 (slab, frame) waves), never under stream capture.  LIW_NO_LASER_SLAB=1: always the lane-per-block kernel.
static int laser_slab_begin(liw_ctx* c, const liw_batch* b, int mode, const WsView& v, void* ws, hipStream_t s, bool may_sync) {
    c->lpk_on = false;
    if (!may_sync || std::getenv("LIW_NO_LASER_SLAB")) return LIW_OK;
    const int S = laser_slab_count(b->B), N = S * b->n;
    // INIT: every (slab, frame) is a wave -> 2 048 of them fill the chip.  TRACK (round 6, k_lin_laser_slab1): only the newest frame's
    // groups carry blocks, S working waves; from 256 of them on (16 384 windows) the lane-per-group walk beats the lane-per-block kernel's
    // per-group wave reductions (49 152 two-frame windows: 0.24 -> 0.0x ms per linearisation), and the marginalisation behind the solve
    // reuses the rows.
    if ((mode == LIW_MODE_INIT ? N < 2048 : S < 256) || b->Ltot <= 0) return LIW_OK;
    if (c->lpk_mx.ensure(sizeof(int) * (size_t)N) || c->lpk_off.ensure(sizeof(long long) * ((size_t)N + 2)) || c->lpk_perm.ensure(sizeof(int) * (size_t)N * 64))
        return fail(c, LIW_ENOMEM, "hipMalloc");
    launch_laser_slab_prepare(b->B, b->n, v.group_off, c->lpk_perm.as<int>(), c->lpk_mx.as<int>(), c->lpk_off.as<long long>(), v.imu_pk_bad + 1, s);
    if (!c->rb_pin) HIPCHK(c, hipHostMalloc((void**)&c->rb_pin, 64, hipHostMallocDefault));
    long long* const tail = reinterpret_cast<long long*>(c->rb_pin) + 2;      // (bytes 16 .. 31 of the page-locked block)
    tail[0] = 0; tail[1] = 1;
    HIPCHK(c, hipMemcpyAsync(tail, c->lpk_off.as<long long>() + N, 2 * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const long long rows = tail[0];
    if (tail[1] != 0 || rows <= 0) return LIW_OK;                               // 3-D end points: the lane-per-block kernel handles them
    if (rows * 64 > 4 * (long long)b->Ltot + 64ll * N) return LIW_OK;           // very ragged groups: padding would exceed 4x the data
    if (c->lpk.ensure(sizeof(double) * LASER_SLAB_ROWD * (size_t)rows)) { c->lpk.release(); return LIW_OK; }   // (no memory for the copy: not an error)
    launch_laser_slab_pack(b->B, b->n, (long)b->Ltot, v.group_off, c->lpk_perm.as<int>(), b->laser_pts, c->lpk_off.as<long long>(), c->lpk_mx.as<int>(), c->lpk.as<double>(), s);
    c->lpk_key = {ws, b->laser_pts, b->laser_frame, b->B, b->n, (long)b->Ltot};
    c->lpk_on = true;
    c->lpk_rows = rows;
    return LIW_OK;
}
static LinArgs lin_args(const liw_batch* b, int mode, const double* x, const WsView& v, int candidate, bool use_lm, bool packed = false, const liw_ctx* c = nullptr, const void* ws = nullptr) {
    LinArgs A{};
    A.B = b->B; A.n = b->n; A.mode = mode; A.eval_small = b->eval_small;
    A.x = x; A.group_off = v.group_off; A.laser_off = b->laser_off; A.laser_pts = b->laser_pts; A.Ltot = b->Ltot;
    A.match_pose = b->match_pose; A.has_match = b->has_match;
    A.imu_X = b->imu_X; A.imu_J = b->imu_J; A.imu_sqrtP = b->imu_sqrtP; A.imu_Dt = b->imu_Dt;
    A.wheel_T = b->wheel_T; A.wheel_sqrtP = b->wheel_sqrtP;
    for (int k = 0; k < 2; ++k) { A.PL[k] = v.PL[k]; A.PI[k] = v.PI[k]; A.PW[k] = v.PW[k]; A.PG[k] = v.PG[k]; }
    A.lm = use_lm ? v.lm : nullptr;
    A.active = use_lm ? v.active : nullptr;
    A.candidate = candidate;
    A.pi_frame = v.pi_frame;
    for (int k = 0; k < 2; ++k) A.CS[k] = v.pi_frame ? v.CS[k] : nullptr;
    if (packed && b->n > 1 && b->eval_small) { A.imu_pk = v.imu_pk; A.imu_pk_bad = v.imu_pk_bad; }
    if (packed) A.laser_hz = v.imu_pk_bad + 1;
    if (packed && lpk_matches(c, b, ws)) { A.laser_pk = c->lpk.as<double>(); A.laser_slab_off = c->lpk_off.as<long long>(); A.laser_perm = c->lpk_perm.as<int>(); }
    return A;
}
static StepArgs step_args(liw_ctx* c, const liw_batch* b, int mode, int max_iters, const WsView& v) {
    StepArgs a{};
    a.B = b->B; a.n = b->n; a.mode = mode; a.max_iters = max_iters; a.fast_mode = c->prm.fast_mode;
    a.x = b->x; a.match_pose = b->match_pose; a.has_match = b->has_match;
    a.prior_X = b->prior_X; a.prior_J = b->prior_J; a.has_prior = b->has_prior;
    a.w = v;
    a.use_active = lin_builds_active_list(b->B, b->eval_small) ? 1 : 0;   // every step launch follows a linearisation with LM state
    return a;
}
static int resolve_iters(liw_ctx* c, int mode, int max_iters) {
    if (max_iters > 0) return max_iters;
    return (mode == LIW_MODE_TRACK && c->prm.fast_mode) ? 10 : 50;   // solver.cpp:800-801
}

// ws carries the history record count in the ctx (single-window path) — batch callers pass 0 history

int liw_batch_lm_begin(liw_ctx* c, const liw_batch* b, int mode, int max_iters, void* ws, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    // the cap lives in the device-side LmState from here on: liw_batch_lm_step / _finish do not depend on host ctx state
    c->last_iters = resolve_iters(c, mode, max_iters);
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    hipStream_t s = (hipStream_t)stream;
    launch_group_offsets(b->B, b->n, b->laser_off, b->laser_frame, v.group_off, s);
    launch_lm_begin(b->B, b->n, v.lm, c->last_iters, s);
    (void)hipMemsetAsync(v.active + b->B + 1, 0, compact_list_bytes_host(b->B) - sizeof(int) * ((size_t)b->B + 1), s);   // ticket + publication words of k_compact_active
    // IMU block records of this solve, packed once (the role kernel of large batches is HBM-bound; k_lin_all on a few windows reads the
    // caller's arrays); *imu_pk_bad != 0 (set here on the device) sends the role back to the full arrays
    if (b->n > 1 && b->eval_small) {
        if ((long)b->B * (b->n - 1) >= 4096 && !std::getenv("LIW_NO_IMU_PACK"))
            launch_imu_pack(b->B, b->n, b->imu_X, b->imu_J, b->imu_sqrtP, b->imu_Dt, v.imu_pk, v.imu_pk_bad, s);
        else (void)hipMemsetAsync(v.imu_pk_bad, 0xff, sizeof(int), s);
    }
    // 2-D scans: the laser role skips the z planes.  Only where that pays: the scan is a full pass over the 12 planes, a small or
    // tracking-size batch saves less than that over its few linearisations (flag 1 = "has z": the role reads every plane)
    if ((long)b->B * (b->n - 1) >= 4096) {
        launch_laser_z_scan(b->Ltot, b->laser_pts, v.imu_pk_bad + 1, s);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (int r = laser_slab_begin(c, b, mode, v, ws, s, cap == hipStreamCaptureStatusNone)) return r;
    } else {
        (void)hipMemsetAsync(v.imu_pk_bad + 1, 0xff, sizeof(int), s);
        c->lpk_on = false;
    }
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_launch_paths(liw_ctx* c, const liw_batch* b, const void* ws, int* flags) {
    if (!c) return LIW_EINVAL;
    if (int r = check_batch(c, b)) return r;
    if (!flags) return fail(c, LIW_EINVAL, "liw_batch_launch_paths: null flags");
    *flags = (pi_frame_format(b->B) ? 1 : 0) | (lpk_matches(c, b, ws) ? 2 : 0);
    return LIW_OK;
}
int liw_batch_packed_rows(liw_ctx* c, const liw_batch* b, const void* ws, long long* rows, long long* blocks) {
    if (!c) return LIW_EINVAL;
    if (int r = check_batch(c, b)) return r;
    if (rows) *rows = lpk_matches(c, b, ws) ? c->lpk_rows : 0;
    if (blocks) *blocks = (long long)b->Ltot;
    return LIW_OK;
}
int liw_batch_lm_linearize(liw_ctx* c, const liw_batch* b, int mode, int candidate, void* ws, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    LinArgs A = lin_args(b, mode, candidate ? v.x_cand : b->x, v, candidate != 0, true, true, c, ws);
    launch_linearize(A, c->dp, (hipStream_t)stream, c->have_fork ? &c->fork : nullptr);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
/* linearisation without the final join of the role streams (see liw_window.h, factor-sharded exchange) */
int liw_batch_lm_linearize_async(liw_ctx* c, const liw_batch* b, int mode, int candidate, void* ws, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    LinArgs A = lin_args(b, mode, candidate ? v.x_cand : b->x, v, candidate != 0, true, true, c, ws);
    launch_linearize(A, c->dp, (hipStream_t)stream, c->have_fork ? &c->fork : nullptr, true);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_lm_join(liw_ctx* c, void* stream) {
    NEEDDEV(c);
    launch_linearize_join((hipStream_t)stream, c->have_fork ? &c->fork : nullptr);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_exchange_doubles(int B, int n, int mode) {
    if (B <= 0 || n <= 0 || n > 64) return LIW_EINVAL;
    return B * n * (mode == LIW_MODE_INIT ? 45 : 21) + 1;
}
int liw_batch_exchange_pack(liw_ctx* c, const liw_batch* b, int mode, int candidate, void* ws, double* buf, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    if (!buf) return fail(c, LIW_EINVAL, "liw_batch_exchange_pack: null buffer");
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    launch_exchange_pack(b->B, b->n, mode == LIW_MODE_INIT, v.PL[0], v.PL[1], candidate, mode == LIW_MODE_MARG ? nullptr : v.lm, buf, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_exchange_unpack(liw_ctx* c, const liw_batch* b, int mode, int candidate, void* ws, const double* buf, int copies, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    if (!buf || copies < 1) return fail(c, LIW_EINVAL, "liw_batch_exchange_unpack: null buffer / copies < 1");
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    const size_t stride = (size_t)liw_batch_exchange_doubles(b->B, b->n, mode);
    launch_exchange_unpack(b->B, b->n, mode == LIW_MODE_INIT, copies, stride, buf, v.PL[0], v.PL[1], v.pi_frame ? v.CS[0] : nullptr, v.pi_frame ? v.CS[1] : nullptr, candidate, mode == LIW_MODE_MARG ? nullptr : v.lm, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
static hipEvent_t next_event(std::vector<hipEvent_t>& pool, size_t& used);
/* The factor-sharded LM solve (reference caller src/trajectory/trajectory.cpp:446 -> solver::init_solve): the chunked
 * early-exit loop, owned by the library so that C++ and Python hosts share one implementation.  See include/liw_window.h. */
int liw_batch_solve_sharded(liw_ctx* c, const liw_batch* b, int mode, int max_iters, void* ws, void* stream,
                            double* xbuf, double* xall, int world, liw_exchange_fn exchange, void* user) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    if (mode != LIW_MODE_INIT && mode != LIW_MODE_TRACK) return fail(c, LIW_EINVAL, "liw_batch_solve_sharded: mode must be INIT or TRACK");
    const bool p2p = exchange == nullptr && c->p2p_world > 0;
    if (!xbuf || (!exchange && !p2p) || world < 1) return fail(c, LIW_EINVAL, "liw_batch_solve_sharded: exchange buffer / callback (or liw_batch_p2p_setup) / world");
    if (p2p && world != c->p2p_world) return fail(c, LIW_EINVAL, "liw_batch_solve_sharded: world differs from liw_batch_p2p_setup");
    const int K = resolve_iters(c, mode, max_iters);
    const size_t nd = (size_t)liw_batch_exchange_doubles(b->B, b->n, mode);
    hipStream_t s = (hipStream_t)stream;
    c->xev_used = 0;
    auto lin_exchange = [&](int cand) -> int {
        if (int r = liw_batch_lm_linearize_async(c, b, mode, cand, ws, stream)) return r;
        if (c->time_exchange) (void)hipEventRecord(next_event(c->ev_x, c->xev_used), s);
        if (int r = liw_batch_exchange_pack(c, b, mode, cand, ws, xbuf, stream)) return r;
        if (p2p) {   // native one-shot exchange: push into every peer's receive area, raise the flags, wait for the P flags of our own
            const unsigned long long e = ++c->p2p_epoch;
            launch_p2p_exchange(nd, xbuf, c->p2p, c->p2p_rank, world, e, c->p2p_err.as<int>(), s);
            c->last_x = c->p2p.area[c->p2p_rank] + (size_t)(e & 1ull) * world * nd;
            c->last_x_copies = world;
            WsView v = make_view(ws, b->B, b->n, b->history_records);
            launch_exchange_unpack(b->B, b->n, mode == LIW_MODE_INIT, world, nd, c->last_x, v.PL[0], v.PL[1], v.pi_frame ? v.CS[0] : nullptr, v.pi_frame ? v.CS[1] : nullptr, cand, v.lm, s, true);
            HIPCHK(c, hipGetLastError());
            if (c->time_exchange) (void)hipEventRecord(next_event(c->ev_x, c->xev_used), s);
            return liw_batch_lm_join(c, stream);
        }
        // the host's collective, ordered on `stream`: the elementwise sum over the ranks left in xbuf (returns 1), or the `world` images in
        // rank order in xall (returns world: the one-shot exchange; liw_batch_exchange_unpack adds them in that order on every rank)
        const int copies = exchange(user, xbuf, xall, nd, stream);
        if (copies != 1 && !(copies == world && xall)) return fail(c, LIW_EINVAL, "liw_batch_solve_sharded: the exchange callback must return 1 (sum in buf) or world (images in all)");
        c->last_x = copies == 1 ? xbuf : xall;
        c->last_x_copies = copies;
        if (int r = liw_batch_exchange_unpack(c, b, mode, cand, ws, c->last_x, copies, stream)) return r;
        if (c->time_exchange) (void)hipEventRecord(next_event(c->ev_x, c->xev_used), s);
        return liw_batch_lm_join(c, stream);
    };
    auto p2p_failed = [&]() -> int {   // blocking read of the peer-write error word (1 + rank whose flag never arrived)
        int e = 0;
        HIPCHK(c, hipMemcpyAsync(&e, c->p2p_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (e) return fail(c, LIW_EHIP, "peer-write exchange: a peer's flag never arrived (rank in p2p status)");
        return LIW_OK;
    };
    // windows still iterating at the last exchange: the trailer went through the same exchange, so every rank reads the same number and
    // takes the same decision without a second collective (blocking 8-byte read-backs, one per image)
    auto active = [&](long* out) -> int {
        double tot = 0.0;
        for (int k = 0; k < c->last_x_copies; ++k) {
            double v = 0.0;
            HIPCHK(c, hipMemcpyAsync(&v, c->last_x + (size_t)k * nd + (nd - 1), sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            tot += v;
        }
        *out = (long)(tot + 0.5) / world;
        if (p2p) return p2p_failed();
        return LIW_OK;
    };
    if (int r = liw_batch_lm_begin(c, b, mode, K, ws, stream)) return r;
    if (int r = lin_exchange(0)) return r;
    int k = 0, chunk = 4;
    while (k < K) {
        const int m = std::min(chunk, K - k);
        for (int i = 0; i < m; ++i) {
            if (int r = liw_batch_lm_step(c, b, mode, ws, stream)) return r;
            if (int r = lin_exchange(1)) return r;
        }
        k += m;
        if (k < K) {
            long act = 0;
            if (int r = active(&act)) return r;
            if (act == 0) break;
        }
        chunk *= 2;
    }
    if (int r = liw_batch_lm_step(c, b, mode, ws, stream)) return r;
    if (int r = liw_batch_lm_finish(c, b, mode, ws, stream)) return r;
    // a flag that timed out in the LAST chunk (or in a solve short enough to have no read-back between chunks) is seen here: after the
    // first timeout k_p2p_wait stops waiting, so the sums behind it may hold stale or partial peer images
    if (p2p) return p2p_failed();
    return LIW_OK;
}
/* native peer-write exchange: see include/liw_window.h */
size_t liw_batch_p2p_area_doubles(int B, int n, int mode, int world) {
    if (world < 1 || world > P2P_MAX) return 0;
    const int nd = liw_batch_exchange_doubles(B, n, mode);
    return nd > 0 ? (size_t)2 * world * nd : 0;
}
int liw_batch_p2p_setup(liw_ctx* c, int rank, int world, double* const* areas, unsigned long long* const* flags) {
    NEEDDEV(c);
    if (world == 0) { c->p2p_world = 0; return LIW_OK; }
    if (world < 1 || world > P2P_MAX || rank < 0 || rank >= world || !areas || !flags) return fail(c, LIW_EINVAL, "liw_batch_p2p_setup: rank / world / pointers");
    for (int r = 0; r < world; ++r) {
        if (!areas[r] || !flags[r]) return fail(c, LIW_EINVAL, "liw_batch_p2p_setup: null peer pointer");
        c->p2p.area[r] = areas[r]; c->p2p.flags[r] = flags[r];
    }
    if (c->p2p_err.ensure(sizeof(int))) return fail(c, LIW_ENOMEM, "hipMalloc");
    HIPCHK(c, hipMemset(c->p2p_err.p, 0, sizeof(int)));
    // The exchange counter continues from what this rank's flag words already hold (all zero in a fresh area): a second setup on the
    // same areas — or a new context on re-used ones — must not start again at 1 while the peers' words still carry the epochs of
    // earlier exchanges, or the first waits would pass at once on stale images.  Exchanges are collective, so every rank reads the same
    // maximum from its own words.
    unsigned long long seen[P2P_MAX] = {0}, e0 = 0;
    HIPCHK(c, hipMemcpy(seen, flags[rank], sizeof(unsigned long long) * (size_t)world, hipMemcpyDeviceToHost));
    for (int r = 0; r < world; ++r) e0 = std::max(e0, seen[r]);
    c->p2p_rank = rank; c->p2p_world = world; c->p2p_epoch = e0;
    return LIW_OK;
}
int liw_batch_p2p_status(liw_ctx* c, int* timed_out_rank_plus_1) {
    NEEDDEV(c);
    int e = 0;
    if (c->p2p_err.p) HIPCHK(c, hipMemcpy(&e, c->p2p_err.p, sizeof(int), hipMemcpyDeviceToHost));
    if (timed_out_rank_plus_1) *timed_out_rank_plus_1 = e;
    return LIW_OK;
}
/* average device time (ms) of one exchange (pack + collective + unpack) of the last liw_batch_solve_sharded, and their number */
int liw_batch_exchange_timing(liw_ctx* c, int enable, double* avg_ms, int* count) {
    NEEDDEV(c);
    if (avg_ms || count) {
        double tot = 0.0;
        int cnt = 0;
        for (size_t i = 0; i + 1 < c->xev_used; i += 2) {
            float ms = 0.f;
            (void)hipEventSynchronize(c->ev_x[i + 1]);
            if (hipEventElapsedTime(&ms, c->ev_x[i], c->ev_x[i + 1]) == hipSuccess) { tot += ms; ++cnt; }
        }
        if (avg_ms) *avg_ms = cnt ? tot / cnt : 0.0;
        if (count) *count = cnt;
        c->xev_used = 0;
    }
    c->time_exchange = enable != 0;
    return LIW_OK;
}
int liw_batch_lm_step(liw_ctx* c, const liw_batch* b, int mode, void* ws, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    StepArgs a = step_args(c, b, mode, c->last_iters, v);
    launch_lm_step(a, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_lm_finish(liw_ctx* c, const liw_batch* b, int mode, void* ws, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    StepArgs a = step_args(c, b, mode, c->last_iters, v);
    launch_lm_finish(a, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
/* the iteration cap the step kernel enforces (set before liw_batch_lm_step when driving the loop by hand) */
int liw_batch_set_max_iters(liw_ctx* c, int mode, int max_iters) {
    if (!c) return LIW_EINVAL;
    c->last_iters = resolve_iters(c, mode, max_iters);
    return c->last_iters;
}

static hipEvent_t next_event(std::vector<hipEvent_t>& pool, size_t& used) {
    if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
    return pool[used++];
}

static int enqueue_solve(liw_ctx* c, const liw_batch* b, int mode, int K, void* ws, hipStream_t s, bool timed) {
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    launch_group_offsets(b->B, b->n, b->laser_off, b->laser_frame, v.group_off, s);
    launch_lm_begin(b->B, b->n, v.lm, K, s);
    (void)hipMemsetAsync(v.active + b->B + 1, 0, compact_list_bytes_host(b->B) - sizeof(int) * ((size_t)b->B + 1), s);   // ticket + publication words of k_compact_active
    // (a few windows go through k_lin_all, whose IMU role reads the caller's arrays: nothing to pack)
    const bool pack = b->n > 1 && b->eval_small && !std::getenv("LIW_NO_IMU_PACK") && (long)b->B * (b->n - 1) >= 4096;
    if (pack) launch_imu_pack(b->B, b->n, b->imu_X, b->imu_J, b->imu_sqrtP, b->imu_Dt, v.imu_pk, v.imu_pk_bad, s);
    if (pack) launch_laser_z_scan(b->Ltot, b->laser_pts, v.imu_pk_bad + 1, s);
    c->lpk_on = false;
    if (pack) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (int r = laser_slab_begin(c, b, mode, v, ws, s, cap == hipStreamCaptureStatusNone)) return r;
    }
    StepArgs st = step_args(c, b, mode, K, v);
    auto lin = [&](int cand) {
        LinArgs A = lin_args(b, mode, cand ? v.x_cand : b->x, v, cand, true, pack, c, ws);
        if (timed) (void)hipEventRecord(next_event(c->ev_lin, c->ev_lin_used), s);
        launch_linearize(A, c->dp, s, c->have_fork ? &c->fork : nullptr);
        if (timed) (void)hipEventRecord(next_event(c->ev_lin, c->ev_lin_used), s);
    };
    auto step = [&]() {
        if (timed) (void)hipEventRecord(next_event(c->ev_step, c->ev_step_used), s);
        launch_lm_step(st, s);
        if (timed) (void)hipEventRecord(next_event(c->ev_step, c->ev_step_used), s);
    };
    lin(0);
    // Early exit (round 6): the launches of an iteration in which no window is still iterating cost ~0.2 ms per 49 152 windows (full grids
    // whose waves find nothing to do) — nothing next to a C2 init solve, whose slowest windows use the whole cap, but 2/3 of a batched
    // TRACKING frame (mean 5 LM iterations, the slowest of 49 152 robots 18, cap 50).  The linearisation leaves the number of windows
    // still iterating in active[0] (k_compact_active): read it back between chunks of iterations — 4 bytes, one stream drain — and stop
    // launching once it is 0.  Chunks grow while most windows are active (a check costs a pipeline drain, an empty iteration five empty
    // launches).  Not under stream capture (the captured launch sequence cannot branch), not without the compacted list (small batches).
    hipStreamCaptureStatus capst = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capst);
    static const bool no_exit = std::getenv("LIW_NO_EARLY_EXIT") != nullptr;
    const bool can_exit = capst == hipStreamCaptureStatusNone && !no_exit && lin_builds_active_list(b->B, b->eval_small);
    int next_check = can_exit ? 3 : K + 1;
    for (int k = 0; k < K; ++k) {
        step();
        lin(1);
        if (k + 1 == next_check && k + 1 < K) {
            if (!c->rb_pin) HIPCHK(c, hipHostMalloc((void**)&c->rb_pin, 64, hipHostMallocDefault));
            int* const act = c->rb_pin;                             // count, status word of the list (1 = complete: usable_active_list)
            act[0] = -1; act[1] = 0;
            HIPCHK(c, hipMemcpyAsync(act, v.active, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(act + 1, v.active + b->B + 2, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            if (act[1] != 1 || act[0] < 0 || act[0] > b->B) { next_check = K + 1; continue; }   // no complete list: run the full loop
            if (act[0] == 0) break;
            // (measured and dropped, round 6: handing the laser role back to the lane-per-block kernel once fewer than an eighth of the windows iterate —
            //  a tracking frame of 49 152 robots 6.78 ms with the switch, 6.65 ms without: the dead windows' waves of that kernel cost what the half-empty rows do)
            // (tracking solves end within a few iterations of each other — mean 5, 97 % by 8 —: a denser cadence there)
            next_check += act[0] > b->B / 4 ? (mode == LIW_MODE_TRACK ? 3 : 8) : (act[0] > b->B / 64 ? 3 : 2);
        }
    }
    step();
    launch_lm_finish(st, s);
    return LIW_OK;
}

int liw_batch_solve(liw_ctx* c, const liw_batch* b, int mode, int max_iters, void* ws, void* stream, int use_graph) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    if (mode != LIW_MODE_INIT && mode != LIW_MODE_TRACK) return fail(c, LIW_EINVAL, "liw_batch_solve: mode must be INIT or TRACK");
    const int K = resolve_iters(c, mode, max_iters);
    c->last_iters = K;
    hipStream_t s = (hipStream_t)stream;
    if (use_graph && !c->timing) {
        // cache key: everything the captured launches depend on
        std::vector<unsigned char> key(sizeof(liw_batch) + sizeof(int) * 2 + sizeof(void*));
        std::memcpy(key.data(), b, sizeof(liw_batch));
        std::memcpy(key.data() + sizeof(liw_batch), &mode, sizeof(int));
        std::memcpy(key.data() + sizeof(liw_batch) + sizeof(int), &K, sizeof(int));
        std::memcpy(key.data() + sizeof(liw_batch) + 2 * sizeof(int), &ws, sizeof(void*));
        if (!c->gexec || key != c->gkey) {
            if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            enqueue_solve(c, b, mode, K, ws, s, false);
            HIPCHK(c, hipStreamEndCapture(s, &g));
            HIPCHK(c, hipGraphInstantiate(&c->gexec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            c->gkey = key;
        }
        HIPCHK(c, hipGraphLaunch(c->gexec, s));
        return LIW_OK;
    }
    if (int r = enqueue_solve(c, b, mode, K, ws, s, c->timing)) return r;
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}

int liw_batch_marg_linearize(liw_ctx* c, const liw_batch* b, void* ws, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, 2)) return r;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    hipStream_t s = (hipStream_t)stream;
    launch_group_offsets(b->B, b->n, b->laser_off, b->laser_frame, v.group_off, s);
    LinArgs A = lin_args(b, LIW_MODE_MARG, b->x, v, 0, false);
    // the packed laser rows of the solve that ran on these very arrays (same allocations, same block counts: lpk_matches) serve the
    // marginalisation's one-pose linearisation too — solver::marginalization follows solver::solve / init_solve on the same frames
    // (trajectory.cpp:446-479, :534-544); include/liw_window.h states the contract (the arrays must not be rewritten in between)
    if (lpk_matches(c, b, ws)) { A.laser_pk = c->lpk.as<double>(); A.laser_slab_off = c->lpk_off.as<long long>(); A.laser_perm = c->lpk_perm.as<int>(); }
    if (c->timing) (void)hipEventRecord(next_event(c->ev_lin, c->ev_lin_used), s);
    launch_linearize(A, c->dp, s, c->have_fork ? &c->fork : nullptr);
    if (c->timing) (void)hipEventRecord(next_event(c->ev_lin, c->ev_lin_used), s);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_marg_schur(liw_ctx* c, const liw_batch* b, void* ws, double* sqrt_H, double* Delta_H, double* Delta_g, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, 2)) return r;
    if (c->prm.fast_mode) return LIW_OK;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    MargArgs a{};
    a.B = b->B; a.n = b->n; a.x = b->x;
    a.prior_X = b->prior_X; a.prior_J = b->prior_J; a.prior_R = b->prior_R; a.has_prior = b->has_prior;
    a.w = v; a.sqrt_H = sqrt_H; a.Delta_H = Delta_H; a.Delta_g = Delta_g; a.status = nullptr;
    launch_marg_schur(a, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_export_dense(liw_ctx* c, const liw_batch* b, int mode, int buf, void* ws, double* H, double* g, double* cost, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    ExportArgs a{};
    a.B = b->B; a.n = b->n; a.mode = mode; a.fast_mode = c->prm.fast_mode; a.buf = buf;
    a.x = b->x; a.prior_X = b->prior_X; a.prior_J = b->prior_J; a.has_prior = b->has_prior;
    a.w = v; a.H = H; a.g = g; a.cost = cost;
    launch_export_dense(a, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
/* standalone linearisation at b->x into buffer 0 (no LM state): liw_linearize / tests / bench kernel timing */
int liw_batch_linearize(liw_ctx* c, const liw_batch* b, int mode, void* ws, void* stream) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, min_frames(mode))) return r;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    hipStream_t s = (hipStream_t)stream;
    launch_group_offsets(b->B, b->n, b->laser_off, b->laser_frame, v.group_off, s);
    LinArgs A = lin_args(b, mode, b->x, v, 0, false);
    if (c->timing) (void)hipEventRecord(next_event(c->ev_lin, c->ev_lin_used), s);
    launch_linearize(A, c->dp, s, c->have_fork ? &c->fork : nullptr);
    if (c->timing) (void)hipEventRecord(next_event(c->ev_lin, c->ev_lin_used), s);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}

/* profiling aid, see include/liw_window.h */
int liw_batch_time_kernels(liw_ctx* c, const liw_batch* b, int mode, void* ws, void* stream, int reps, double* out_ms) {
    NEEDDEV(c);
    if (int r = check_batch(c, b, 2)) return r;
    if (mode != LIW_MODE_INIT && mode != LIW_MODE_TRACK) return fail(c, LIW_EINVAL, "liw_batch_time_kernels: mode must be INIT or TRACK");
    if (!out_ms || reps < 1) return fail(c, LIW_EINVAL, "liw_batch_time_kernels: out / reps");
    hipStream_t s = (hipStream_t)stream;
    WsView v = make_view(ws, b->B, b->n, b->history_records);
    const int K = resolve_iters(c, mode, 0);
    if (int r = liw_batch_lm_begin(c, b, mode, K, ws, stream)) return r;
    const bool packed = true;
    auto lin = [&](int cand, int mask) {
        LinArgs A = lin_args(b, mode, cand ? v.x_cand : b->x, v, cand, true, packed, c, ws);
        A.role_mask = mask;
        launch_linearize(A, c->dp, s, nullptr);          // (no fork: the kernel under the clock runs alone)
    };
    StepArgs st = step_args(c, b, mode, K, v);
    lin(0, 0);
    launch_lm_step(st, s);                                // the first step of a solve also builds the Jacobi scaling: not the one timed
    lin(1, 0);
    struct EventPool {   // destroyed on every return path (HIPCHK / ENOMEM below)
        std::vector<hipEvent_t> v;
        ~EventPool() { for (auto e : v) if (e) (void)hipEventDestroy(e); }
        hipEvent_t& operator[](size_t i) { return v[i]; }
    } ev;
    ev.v.assign((size_t)reps * 5 + 4, nullptr);
    for (auto& e : ev.v) HIPCHK(c, hipEventCreate(&e));
    for (int r = 0; r < reps; ++r) {
        hipEvent_t* e = &ev[(size_t)r * 5];
        lin(1, 8);                                        // (the list of windows still iterating, as every linearisation of a solve builds it)
        (void)hipEventRecord(e[0], s); lin(1, 1);
        if (getenv("LIW_KT_IMU_TWICE")) lin(1, 2);       // probe: the timed IMU role behind an IMU role instead of behind the laser role (the laser figure then includes it)
        (void)hipEventRecord(e[1], s); lin(1, 2);
        (void)hipEventRecord(e[2], s); lin(1, 4);
        (void)hipEventRecord(e[3], s); launch_lm_step(st, s);
        (void)hipEventRecord(e[4], s);
    }
    hipEvent_t* m = &ev[(size_t)reps * 5];
    {   // marginalisation: its laser role (one pose free), then the chain Schur complement + eigen square root
        launch_group_offsets(b->B, b->n, b->laser_off, b->laser_frame, v.group_off, s);
        LinArgs A = lin_args(b, LIW_MODE_MARG, b->x, v, 0, false);
        if (lpk_matches(c, b, ws)) { A.laser_pk = c->lpk.as<double>(); A.laser_slab_off = c->lpk_off.as<long long>(); A.laser_perm = c->lpk_perm.as<int>(); }   // as liw_batch_marg_linearize
        A.role_mask = 6;
        launch_linearize(A, c->dp, s, nullptr);
        (void)hipEventRecord(m[0], s);
        A.role_mask = 1;
        launch_linearize(A, c->dp, s, nullptr);
        (void)hipEventRecord(m[1], s);
    }
    if (!c->prm.fast_mode) {
        DevBuf tmp;
        if (tmp.ensure(sizeof(double) * (size_t)b->B * (36 + 225 + 15))) return fail(c, LIW_ENOMEM, "hipMalloc");
        // (a copy of the prior so that the measurement does not replace the caller's)
        DevBuf pX, pJ, pR, pH;
        if (pX.ensure(sizeof(double) * 15 * b->B) || pJ.ensure(sizeof(double) * 225 * b->B) || pR.ensure(sizeof(double) * 15 * b->B) || pH.ensure(sizeof(int) * b->B))
            return fail(c, LIW_ENOMEM, "hipMalloc");
        MargArgs a{};
        a.B = b->B; a.n = b->n; a.x = b->x;
        a.prior_X = b->prior_X; a.prior_J = b->prior_J; a.prior_R = b->prior_R; a.has_prior = b->has_prior;
        a.out_X = pX.as<double>(); a.out_J = pJ.as<double>(); a.out_R = pR.as<double>(); a.out_has = pH.as<int>();
        a.w = v; a.sqrt_H = tmp.as<double>(); a.Delta_H = a.sqrt_H + (size_t)36 * b->B; a.Delta_g = a.Delta_H + (size_t)225 * b->B; a.status = nullptr;
        (void)hipEventRecord(m[2], s);
        launch_marg_schur(a, s);
        (void)hipEventRecord(m[3], s);
        HIPCHK(c, hipStreamSynchronize(s));
        tmp.release(); pX.release(); pJ.release(); pR.release(); pH.release();
    } else {
        (void)hipEventRecord(m[2], s); (void)hipEventRecord(m[3], s);
    }
    HIPCHK(c, hipStreamSynchronize(s));
    for (int k = 0; k < 6; ++k) out_ms[k] = 0.0;
    for (int r = 0; r < reps; ++r)
        for (int k = 0; k < 4; ++k) {
            float f = 0.f;
            (void)hipEventElapsedTime(&f, ev[(size_t)r * 5 + k], ev[(size_t)r * 5 + k + 1]);
            out_ms[k] += f / reps;
        }
    float f = 0.f;
    (void)hipEventElapsedTime(&f, m[2], m[3]); out_ms[4] = f;
    (void)hipEventElapsedTime(&f, m[0], m[1]); out_ms[5] = f;
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}

static PreintNoise preint_noise(const liw_params& prm) {
    PreintNoise N{};
    for (int k = 0; k < 3; ++k) {
        N.q_na[k] = prm.imu_noise_acc_sigma[k] * prm.imu_noise_acc_sigma[k];
        N.q_nw[k] = prm.imu_noise_gyro_sigma[k] * prm.imu_noise_gyro_sigma[k];
        N.q_nba[k] = prm.imu_bias_acc_sigma[k] * prm.imu_bias_acc_sigma[k];
        N.q_nbw[k] = prm.imu_bias_gyro_sigma[k] * prm.imu_bias_gyro_sigma[k];
        N.wheel_cov[k] = prm.wheel_sigma[k] * prm.wheel_sigma[k];
    }
    return N;
}
int liw_batch_imu_preint(liw_ctx* c, int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end,
                         const double* bias6, double* X, double* J, double* P_scratch, double* sqrt_inverse_P, double* Dt, void* stream) {
    NEEDDEV(c);
    if (M < 0 || (M > 0 && (!sample_off || !samples || !t_start || !t_end || !bias6 || !X || !J || !P_scratch || !sqrt_inverse_P || !Dt)))
        return fail(c, LIW_EINVAL, "liw_batch_imu_preint: null argument");
    if (M == 0) return LIW_OK;
    launch_preint_imu(M, sample_off, samples, t_start, t_end, bias6, preint_noise(c->prm), X, J, P_scratch, sqrt_inverse_P, Dt, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}
int liw_batch_wheel_preint(liw_ctx* c, int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end,
                           double* delta_Tij, double* sqrt_inverse_P, double* Dt, void* stream) {
    NEEDDEV(c);
    if (M < 0 || (M > 0 && (!sample_off || !samples || !t_start || !t_end || !delta_Tij || !sqrt_inverse_P || !Dt)))
        return fail(c, LIW_EINVAL, "liw_batch_wheel_preint: null argument");
    if (M == 0) return LIW_OK;
    launch_preint_wheel(M, sample_off, samples, t_start, t_end, preint_noise(c->prm), delta_Tij, sqrt_inverse_P, Dt, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    return LIW_OK;
}

int liw_set_timing(liw_ctx* c, int enable) {
    if (!c) return LIW_EINVAL;
    c->timing = enable != 0;
    c->ev_lin_used = c->ev_step_used = 0;
    return LIW_OK;
}
int liw_get_timing(liw_ctx* c, double* lin_ms, int* lin_n, double* step_ms, int* step_n) {
    NEEDDEV(c);
    auto total = [&](std::vector<hipEvent_t>& pool, size_t used, double* ms, int* cnt) {
        double t = 0.0; int k = 0;
        for (size_t i = 0; i + 1 < used; i += 2) {
            float f = 0.f;
            if (hipEventSynchronize(pool[i + 1]) != hipSuccess) continue;
            if (hipEventElapsedTime(&f, pool[i], pool[i + 1]) == hipSuccess) { t += f; ++k; }
        }
        if (ms) *ms = k ? t / k : 0.0;
        if (cnt) *cnt = k;
    };
    total(c->ev_lin, c->ev_lin_used, lin_ms, lin_n);
    total(c->ev_step, c->ev_step_used, step_ms, step_n);
    c->ev_lin_used = c->ev_step_used = 0;
    return LIW_OK;
}

// ------------------------------------------------------------------------------------------ single window
int liw_set_window(liw_ctx* c, const liw_window* w) {
    NEEDDEV(c);
    if (!w || w->n < 1 || w->n > 64 || w->L < 0) return fail(c, LIW_EINVAL, "liw_set_window: need 1 <= n <= 64, L >= 0");
    if (!w->states || !w->match_pose || !w->has_match) return fail(c, LIW_EINVAL, "liw_set_window: states / match_pose / has_match are NULL");
    if (w->L > 0 && (!w->laser_frame || !w->laser_pts)) return fail(c, LIW_EINVAL, "liw_set_window: L > 0 but laser_frame / laser_pts are NULL");
    if (w->n > 1 && (!w->imu_X || !w->imu_J || !w->imu_sqrtP || !w->imu_Dt || !w->wheel_T || !w->wheel_sqrtP))
        return fail(c, LIW_EINVAL, "liw_set_window: n > 1 but an IMU / wheel array is NULL");
    HIPCHK(c, hipSetDevice(c->prm.device));
    const int n = w->n, L = w->L;
    // laser blocks must be sorted by owning frame
    for (int j = 1; j < L; ++j) if (w->laser_frame[j] < w->laser_frame[j - 1]) return fail(c, LIW_EINVAL, "laser_frame must be ascending");
    for (int j = 0; j < L; ++j) if (w->laser_frame[j] < 0 || w->laser_frame[j] >= n) return fail(c, LIW_EINVAL, "laser_frame out of range");
    // one page-locked staging image + ONE asynchronous host-to-device copy (a tracking window is a dozen arrays of a few hundred
    // bytes: a dozen pageable copies cost more than the solve); the copy is ordered before the solve on the ctx stream, nobody waits
    struct Part { const void* src; size_t bytes; };
    int off2[2] = {0, L};
    int goff[65];   // laser block range of every frame (what k_group_offsets computes for the batched path): first block owned by a frame >= i
    for (int i = 0, j = 0; i <= n; ++i) { while (j < L && w->laser_frame[j] < i) ++j; goff[i] = j; }
    const Part parts[13] = {
        {w->states, sizeof(double) * n * 15}, {off2, sizeof(off2)}, {w->laser_frame, sizeof(int) * (size_t)L},
        {nullptr, sizeof(double) * 12 * (size_t)L},   // laser_pts: transposed to component-major below
        {w->match_pose, sizeof(double) * n * 12}, {w->has_match, (size_t)n},
        {w->imu_X, sizeof(double) * (n - 1) * 15}, {w->imu_J, sizeof(double) * (n - 1) * 225},
        {w->imu_sqrtP, sizeof(double) * (n - 1) * 225}, {w->imu_Dt, sizeof(double) * (n - 1)},
        {w->wheel_T, sizeof(double) * (n - 1) * 12}, {w->wheel_sqrtP, sizeof(double) * (n - 1) * 9},
        {goff, sizeof(int) * (size_t)(n + 1)},
    };
    size_t off[13], tot = 0;
    for (int k = 0; k < 13; ++k) { off[k] = tot; tot = al256(tot + (parts[k].bytes ? parts[k].bytes : 8)); }
    const size_t readback = al256(sizeof(double) * (LIW_RESULT_HDR + (size_t)n * 27 + 276));
    if (c->img_cap < tot || c->readback_cap < readback) {
        (void)hipStreamSynchronize(c->stream);   // an upload out of the old block may still be in flight
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr; c->img_cap = c->readback_cap = c->pinned_cap = 0; c->img_valid = false;
        const size_t ic = std::max(tot * 2, (size_t)65536), rc = std::max(readback * 2, (size_t)8192);
        if (hipHostMalloc(&c->pinned, 2 * ic + rc, hipHostMallocDefault) != hipSuccess) return fail(c, LIW_ENOMEM, "hipHostMalloc");
        c->img_cap = ic; c->readback_cap = rc; c->pinned_cap = 2 * ic + rc;
        std::memset((char*)c->pinned + 2 * ic, 0, rc);   // the polled sequence word of a fresh read-back block must not hold a stale match
    }
    const int stage_id = c->img_valid ? 1 - c->img_cur : 0;
    char* stage = (char*)c->pinned + (size_t)stage_id * c->img_cap;
    // `stage` was the source of the upload two calls ago; uploads are ordered on the ctx stream and the latest one carries ev_upload
    if (c->ev_upload) (void)hipEventSynchronize(c->ev_upload);
    for (int k = 0; k < 13; ++k) {
        if (k == 3) {
            double* soa = (double*)(stage + off[k]);
            for (int j = 0; j < L; ++j) for (int q = 0; q < 12; ++q) soa[(size_t)q * L + j] = w->laser_pts[(size_t)j * 12 + q];
        } else if (parts[k].bytes) {
            std::memcpy(stage + off[k], parts[k].src, parts[k].bytes);
        }
    }
    // the same bytes as the device already holds (the lvio_2d::solver shim flattens the frames again for marginalization()):
    // nothing to upload, and what liw_solve left behind — history, the speculative marginalisation — stays valid
    bool same = c->reattach && c->img_valid && c->n == n && c->L == L;
    if (same) {
        const char* cur = (const char*)c->pinned + (size_t)c->img_cur * c->img_cap;
        for (int k = 0; k < 13 && same; ++k)
            if (parts[k].bytes && std::memcmp(stage + off[k], cur + off[k], parts[k].bytes) != 0) same = false;
    }
    if (same) { c->hw = *w; c->have_window = true; return LIW_OK; }
    c->solved_records = 0;
    c->spec_valid = false;
    c->have_window = false;
    if (c->arena.ensure(tot)) return fail(c, LIW_ENOMEM, "hipMalloc");
    HIPCHK(c, hipMemcpyAsync(c->arena.p, stage, tot, hipMemcpyHostToDevice, c->stream));
    if (c->ev_upload) HIPCHK(c, hipEventRecord(c->ev_upload, c->stream));
    // the mirror image only counts once EVERYTHING below has succeeded: a failed allocation further down must not leave an
    // image marked valid, or the next call with the same bytes would re-attach to device pointers that were never (re)established
    c->img_valid = false;
    for (int k = 0; k < 13; ++k) { c->part_off[k] = off[k]; c->part_bytes[k] = parts[k].bytes; }
    char* dev = (char*)c->arena.p;
    bool fresh_prior = c->prior_X.p == nullptr;
    if (c->prior_X.ensure(sizeof(double) * 15) || c->prior_J.ensure(sizeof(double) * 225) || c->prior_R.ensure(sizeof(double) * 15) ||
        c->has_prior.ensure(sizeof(int)) || c->priorn_X.ensure(sizeof(double) * 15) || c->priorn_J.ensure(sizeof(double) * 225) ||
        c->priorn_R.ensure(sizeof(double) * 15) || c->has_priorn.ensure(sizeof(int)) || c->marg_status.ensure(sizeof(int)) ||
        c->result.ensure(sizeof(double) * (LIW_RESULT_HDR + (size_t)n * 27 + 276)))
        return fail(c, LIW_ENOMEM, "hipMalloc");
    if (fresh_prior) {
        HIPCHK(c, hipMemsetAsync(c->prior_X.p, 0, sizeof(double) * 15, c->stream));
        HIPCHK(c, hipMemsetAsync(c->prior_J.p, 0, sizeof(double) * 225, c->stream));
        HIPCHK(c, hipMemsetAsync(c->prior_R.p, 0, sizeof(double) * 15, c->stream));
        HIPCHK(c, hipMemsetAsync(c->has_prior.p, 0, sizeof(int), c->stream));
    }
    c->hist_records = 64;
    FullLayout f = full_layout(1, n, c->hist_records);
    if (c->ws.ensure(f.bytes)) return fail(c, LIW_ENOMEM, "hipMalloc workspace");
    c->hw = *w; c->have_window = true; c->n = n; c->L = L;
    c->img_cur = stage_id; c->img_valid = true;
    liw_batch& b = c->sb;
    b.B = 1; b.n = n; b.Ltot = L;
    b.x = (double*)(dev + off[0]); b.laser_off = (int*)(dev + off[1]); b.laser_frame = (int*)(dev + off[2]);
    b.laser_pts = (double*)(dev + off[3]); b.match_pose = (double*)(dev + off[4]); b.has_match = (unsigned char*)(dev + off[5]);
    b.imu_X = (double*)(dev + off[6]); b.imu_J = (double*)(dev + off[7]); b.imu_sqrtP = (double*)(dev + off[8]);
    b.imu_Dt = (double*)(dev + off[9]); b.wheel_T = (double*)(dev + off[10]); b.wheel_sqrtP = (double*)(dev + off[11]);
    b.prior_X = c->prior_X.as<double>(); b.prior_J = c->prior_J.as<double>(); b.prior_R = c->prior_R.as<double>(); b.has_prior = c->has_prior.as<int>();
    b.eval_small = 1; b.history_records = c->hist_records;
    return LIW_OK;
}
/* Forget the uploaded window: afterwards every window-level call fails with LIW_ESTATE until the next liw_set_window.  For callers
 * (the lvio_2d::solver shim) whose flat arrays die with the calling scope, so that the ctx never holds dangling host pointers.
 * The device copy stays: a liw_set_window with the same bytes re-attaches to it (and to a speculative marginalisation result). */
int liw_clear_window(liw_ctx* c) {
    if (!c) return LIW_EINVAL;
    c->have_window = false;
    c->hw = liw_window{};
    c->solved_records = 0;
    return LIW_OK;
}
#define NEEDWIN(c)                                                                          \
    do {                                                                                    \
        NEEDDEV(c);                                                                         \
        if (!(c)->have_window) return fail((c), LIW_ESTATE, "liw_set_window was not called"); \
        HIPCHK(c, hipSetDevice((c)->prm.device));                                           \
    } while (0)

int liw_solve(liw_ctx* c, int mode, int max_iters, liw_summary* summary) {
    NEEDWIN(c);
    if (mode != LIW_MODE_INIT && mode != LIW_MODE_TRACK) return fail(c, LIW_EINVAL, "liw_solve: mode must be LIW_MODE_INIT or LIW_MODE_TRACK");
    if (c->n < min_frames(mode)) return fail(c, LIW_EINVAL, "liw_solve: TRACK topology needs n >= 2");
    // init topology ties every laser block to (frame 0, owning frame): a block owned by frame 0 would name the same
    // parameter block twice, which ceres::Problem::AddResidualBlock rejects (solver.cpp:93-106)
    if (mode == LIW_MODE_INIT && c->L > 0 && c->hw.laser_frame[0] == 0 && c->hw.has_match[0])
        return fail(c, LIW_EINVAL, "init topology: frame 0 must not own laser blocks (duplicate parameter blocks)");
    int K = resolve_iters(c, mode, max_iters);
    c->solved_records = 0;
    c->spec_valid = false;
    if (K + 1 > c->hist_records) {   // grow the history region
        c->hist_records = K + 1;
        c->sb.history_records = c->hist_records;
        FullLayout f = full_layout(1, c->n, c->hist_records);
        if (c->ws.ensure(f.bytes)) return fail(c, LIW_ENOMEM, "hipMalloc workspace");
    }
    // Same launch sequence as liw_batch_solve (lin, K x [step, lin], step, finish), enqueued in growing chunks.  Every chunk ends with
    // k_pack_result: the write-backs of k_lm_finish (they read the LM state as it is; the result only counts once the window is done)
    // and ONE packed read-back record, so a tracking solve that converges inside the first chunk costs one submission and one
    // synchronisation.
    // Launches enqueued behind the terminating step are exactly the ones whose kernels return immediately: same result.
    // TRACK solves also enqueue the marginalisation the reference runs next (trajectory.cpp:548-559: solver.solve();
    // solver.marginalization();) behind the solve, gated on the device by the window's `done` flag: its outputs wait in the read-back
    // record and its prior in a second set of buffers until liw_marginalize asks for them (or a new window / prior drops them).
    bool spec = c->spec_marg && mode == LIW_MODE_TRACK && !c->prm.fast_mode && c->n >= 2;
    if (spec && c->L > 0) {
        // The speculative marginalisation rides on LinArgs::marg_older: EVERY linearisation of the tracking solve also evaluates the laser
        // groups of the older frames (constants of the solve), so that no launch is needed behind it — 7 us less per 2-frame window.  In a
        // keep-N window (30 frames, ~2 000 blocks on the older frames against ~70 on the new one) that is n times the laser work per LM
        // iteration for records only the last buffer needs: there the marginalisation linearises for itself when it is asked for.
        const int* lf = c->hw.laser_frame;
        const long older = std::lower_bound(lf, lf + c->L, c->n - 1) - lf, newest = (long)c->L - older;
        static const char* lim_env = std::getenv("LIW_SPEC_OLDER_MAX");      // A/B aid: older-frame blocks up to which the records ride along
        const long lim = lim_env ? std::atol(lim_env) : 256;
        if (older > std::max(lim, 4 * newest)) spec = false;
    }
    const size_t rdoubles = LIW_RESULT_HDR + (size_t)c->n * 27 + 276;
    char* rb = (char*)c->pinned + 2 * c->img_cap;
    {
        c->last_iters = K;
        const liw_batch* b = &c->sb;
        hipStream_t s = c->stream;
        WsView v = make_view(c->ws.p, 1, c->n, b->history_records);
        v.group_off = (int*)((char*)c->arena.p + c->part_off[12]);   // computed on the host by liw_set_window, uploaded with the window
        StepArgs st = step_args(c, b, mode, K, v);
        bool first = true;
        auto lin = [&](int cand) {
            // the first linearisation of a solve needs no LM state (buffer 0, window live): its launch carries the state reset as one
            // more work-group instead of a k_lm_begin launch in front of it
            LinArgs A = lin_args(b, mode, cand ? v.x_cand : b->x, v, cand, !first);
            if (first) { A.reset_lm = v.lm; A.reset_iters = K; first = false; }
            A.active = nullptr;
            A.marg_older = spec ? 1 : 0;   // (the older frames' poses are constants of a tracking solve: their marginalisation records ride along)
            launch_linearize(A, c->dp, s, c->have_fork ? &c->fork : nullptr);
        };
        // the read-back record is written straight into the page-locked host block (device-visible under the same address): the packing
        // kernel's stores cross PCIe, no separate device-to-host copy command (LIW_NO_ZEROCOPY: device buffer + copy, the round-2 start)
        static const bool zero_copy = std::getenv("LIW_NO_ZEROCOPY") == nullptr;
        double* res = zero_copy ? (double*)rb : c->result.as<double>();
        double* marg_out = res + LIW_RESULT_HDR + (size_t)c->n * 27;
        PackArgs pk{};
        pk.n = c->n; pk.mode = mode; pk.lm = v.lm; pk.info = v.info; pk.x = b->x; pk.match_pose = b->match_pose; pk.has_match = b->has_match;
        pk.marg = nullptr; pk.marg_status = spec ? c->marg_status.as<int>() : nullptr; pk.out = res;
        MargArgs ma{};
        if (spec) {
            ma.B = 1; ma.n = c->n; ma.x = b->x;
            ma.prior_X = b->prior_X; ma.prior_J = b->prior_J; ma.prior_R = b->prior_R; ma.has_prior = b->has_prior;
            ma.out_X = c->priorn_X.as<double>(); ma.out_J = c->priorn_J.as<double>(); ma.out_R = c->priorn_R.as<double>(); ma.out_has = c->has_priorn.as<int>();
            ma.w = v; ma.sqrt_H = marg_out; ma.Delta_H = marg_out + 36; ma.Delta_g = marg_out + 36 + 225; ma.status = c->marg_status.as<int>();
            ma.gate = v.lm; ma.use_cur = 1;
        }
        // (linearise, step) pairs: pair 0 linearises at the initial point, pair j > 0 at the candidate of step j.  K + 1 pairs in all
        // (the last step takes the last candidate / meets the iteration cap); a solve of `it` iterations is done after it + 1 pairs.
        int k = 0, chunk = 4;
        bool done = false;
        while (!done) {
            const int m = std::min(chunk, K + 1 - k);
            for (int i = 0; i < m; ++i) { lin(k + i > 0 ? 1 : 0); launch_lm_step(st, s); }
            k += m;
            // the current LM buffer of a finished TRACK solve IS the marginalisation's linearisation (same states, same kernels): the laser
            // records of the older frames, which the tracking topology leaves out, were evaluated by the solve's own linearisations
            // (LinArgs::marg_older; until late round 5 a launch of its own here, 7 us of a 0.18 ms frame)
            if (spec) launch_marg_schur(ma, s);
            pk.seq = ++c->pack_seq;
            launch_pack_result(pk, s);
            HIPCHK(c, hipGetLastError());
            if (!zero_copy) HIPCHK(c, hipMemcpyAsync(rb, res, sizeof(double) * rdoubles, hipMemcpyDeviceToHost, s));
            // zero-copy: the record's sequence word is the completion signal (written last, system-scope release): polling it skips the
            // wake-up latency of the stream's completion interrupt — a few microseconds of a 0.2 ms tracking frame.  Bounded: a chunk that
            // takes longer than 2 ms (big windows) falls back to the blocking wait.  LIW_NO_POLL=1: always block.
            static const bool poll = zero_copy && std::getenv("LIW_NO_POLL") == nullptr;
            bool seen = false;
            if (poll) {
                const volatile int* h = (const volatile int*)rb;
                const auto t0 = std::chrono::steady_clock::now();
                for (int spin = 0; !seen; ++spin) {
                    if (h[3] == pk.seq) { seen = true; break; }
                    if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) break;
                }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            if (!seen) HIPCHK(c, hipStreamSynchronize(s));
            done = ((const int*)rb)[0] != 0;
            if (!done && k >= K + 1) return fail(c, LIW_EHIP, "liw_solve: the window did not terminate within its iteration cap");
            chunk *= 2;
        }
    }
    const int* hdr = (const int*)rb;
    const liw_summary* sp = (const liw_summary*)(rb + 16);
    const double* xs = (const double*)rb + LIW_RESULT_HDR;
    const double* mp = xs + (size_t)c->n * 15;
    std::memcpy(c->hw.states, xs, sizeof(double) * c->n * 15);
    std::memcpy(c->hw.match_pose, mp, sizeof(double) * c->n * 12);
    {   // the staging image keeps mirroring the device
        char* img = (char*)c->pinned + (size_t)c->img_cur * c->img_cap;
        std::memcpy(img + c->part_off[0], xs, sizeof(double) * c->n * 15);
        std::memcpy(img + c->part_off[4], mp, sizeof(double) * c->n * 12);
    }
    if (spec && hdr[1] == 0) {
        std::memcpy(c->spec_out, mp + (size_t)c->n * 12, sizeof(c->spec_out));
        c->spec_valid = true;
    }
    if (summary) *summary = *sp;
    c->solved_records = std::max(0, std::min(sp->iterations + 1, c->hist_records));
    return LIW_OK;
}
int liw_get_history(liw_ctx* c, double* x, int max_records) {
    NEEDWIN(c);
    // only a solve completed on the CURRENT window has a history: the record count is host state (cleared by
    // liw_set_window and at the start of every liw_solve), never read from a possibly uninitialised workspace
    if (c->solved_records <= 0) return fail(c, LIW_ESTATE, "liw_get_history: no completed liw_solve on the current window");
    if (!x || max_records <= 0) return fail(c, LIW_EINVAL, "liw_get_history: null buffer");
    FullLayout f = full_layout(1, c->n, c->hist_records);
    int rec = c->solved_records;
    if (rec > max_records) rec = max_records;
    HIPCHK(c, hipMemcpy(x, (char*)c->ws.p + f.history, sizeof(double) * (size_t)rec * c->n * 15, hipMemcpyDeviceToHost));
    return rec;
}
int liw_linearize(liw_ctx* c, int mode, double* H, double* g, double* cost) {
    NEEDWIN(c);
    const size_t N = (size_t)15 * c->n;
    if (c->scratch.ensure(sizeof(double) * (N * N + N + 1))) return fail(c, LIW_ENOMEM, "hipMalloc");
    double* dH = c->scratch.as<double>();
    double* dg = dH + N * N;
    double* dc = dg + N;
    if (int r = liw_batch_linearize(c, &c->sb, mode, c->ws.p, c->stream)) return r;
    if (int r = liw_batch_export_dense(c, &c->sb, mode, 0, c->ws.p, dH, dg, dc, c->stream)) return r;
    if (H) HIPCHK(c, hipMemcpyAsync(H, dH, sizeof(double) * N * N, hipMemcpyDeviceToHost, c->stream));
    if (g) HIPCHK(c, hipMemcpyAsync(g, dg, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    if (cost) HIPCHK(c, hipMemcpyAsync(cost, dc, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LIW_OK;
}
int liw_eval_factors(liw_ctx* c, int mode, double* laser_res, double* laser_jac, double* imu_res, double* imu_jac, double* wheel_res,
                     double* wheel_jac, double* ground_res, double* ground_jac) {
    NEEDWIN(c);
    const size_t n = c->n, L = c->L, nm = n - 1;
    const size_t sz[8] = {L * 2, L * 24, nm * 15, nm * 450, nm * 3, nm * 36, n * 2, n * 12};
    size_t tot = 0;
    for (size_t v : sz) tot += v;
    if (c->scratch.ensure(sizeof(double) * (tot + 8))) return fail(c, LIW_ENOMEM, "hipMalloc");
    HIPCHK(c, hipMemsetAsync(c->scratch.p, 0, sizeof(double) * (tot + 8), c->stream));
    double* base = c->scratch.as<double>();
    double* dptr[8];
    size_t o = 0;
    for (int k = 0; k < 8; ++k) { dptr[k] = base + o; o += sz[k]; }
    WsView v = make_view(c->ws.p, 1, c->n, c->sb.history_records);
    launch_group_offsets(1, c->n, c->sb.laser_off, c->sb.laser_frame, v.group_off, c->stream);
    LinArgs A = lin_args(&c->sb, mode, c->sb.x, v, 0, false);
    A.dbg_laser_res = dptr[0]; A.dbg_laser_jac = dptr[1]; A.dbg_imu_res = dptr[2]; A.dbg_imu_jac = dptr[3];
    A.dbg_wheel_res = dptr[4]; A.dbg_wheel_jac = dptr[5]; A.dbg_ground_res = dptr[6]; A.dbg_ground_jac = dptr[7];
    launch_linearize(A, c->dp, c->stream, c->have_fork ? &c->fork : nullptr);
    HIPCHK(c, hipGetLastError());
    double* hptr[8] = {laser_res, laser_jac, imu_res, imu_jac, wheel_res, wheel_jac, ground_res, ground_jac};
    for (int k = 0; k < 8; ++k)
        if (hptr[k] && sz[k]) HIPCHK(c, hipMemcpyAsync(hptr[k], dptr[k], sizeof(double) * sz[k], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LIW_OK;
}
int liw_marginalize(liw_ctx* c, double* sqrt_H36, double* Delta_H225, double* Delta_g15) {
    NEEDWIN(c);
    if (c->prm.fast_mode) return LIW_OK;   // solver.cpp:259-260
    if (c->n < 2) return fail(c, LIW_EINVAL, "liw_marginalize: needs n >= 2");
    if (c->spec_valid) {   // computed behind the solve of this very window (liw_solve): hand it over and make its prior the live one
        c->spec_valid = false;
        if (sqrt_H36) std::memcpy(sqrt_H36, c->spec_out, sizeof(double) * 36);
        if (Delta_H225) std::memcpy(Delta_H225, c->spec_out + 36, sizeof(double) * 225);
        if (Delta_g15) std::memcpy(Delta_g15, c->spec_out + 36 + 225, sizeof(double) * 15);
        std::swap(c->prior_X, c->priorn_X); std::swap(c->prior_J, c->priorn_J); std::swap(c->prior_R, c->priorn_R); std::swap(c->has_prior, c->has_priorn);
        c->sb.prior_X = c->prior_X.as<double>(); c->sb.prior_J = c->prior_J.as<double>();
        c->sb.prior_R = c->prior_R.as<double>(); c->sb.has_prior = c->has_prior.as<int>();
        return LIW_OK;
    }
    if (c->scratch.ensure(sizeof(double) * (36 + 225 + 15))) return fail(c, LIW_ENOMEM, "hipMalloc");
    double* d = c->scratch.as<double>();
    if (int r = liw_batch_marg_linearize(c, &c->sb, c->ws.p, c->stream)) return r;
    if (int r = liw_batch_marg_schur(c, &c->sb, c->ws.p, d, d + 36, d + 36 + 225, c->stream)) return r;
    double h[36 + 225 + 15];
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sqrt_H36) std::memcpy(sqrt_H36, h, sizeof(double) * 36);
    if (Delta_H225) std::memcpy(Delta_H225, h + 36, sizeof(double) * 225);
    if (Delta_g15) std::memcpy(Delta_g15, h + 36 + 225, sizeof(double) * 15);
    return LIW_OK;
}
int liw_get_prior(liw_ctx* c, double* X15, double* J225, double* R15) {
    NEEDDEV(c);
    if (!c->prior_X.p) return 0;
    int has = 0;
    HIPCHK(c, hipMemcpy(&has, c->has_prior.p, sizeof(int), hipMemcpyDeviceToHost));
    if (!has) return 0;
    if (X15) HIPCHK(c, hipMemcpy(X15, c->prior_X.p, sizeof(double) * 15, hipMemcpyDeviceToHost));
    if (J225) HIPCHK(c, hipMemcpy(J225, c->prior_J.p, sizeof(double) * 225, hipMemcpyDeviceToHost));
    if (R15) HIPCHK(c, hipMemcpy(R15, c->prior_R.p, sizeof(double) * 15, hipMemcpyDeviceToHost));
    return 1;
}
int liw_set_prior(liw_ctx* c, int has_prior, const double* X15, const double* J225, const double* R15) {
    if (c) c->spec_valid = false;
    NEEDDEV(c);
    HIPCHK(c, hipSetDevice(c->prm.device));
    if (c->prior_X.ensure(sizeof(double) * 15) || c->prior_J.ensure(sizeof(double) * 225) || c->prior_R.ensure(sizeof(double) * 15) ||
        c->has_prior.ensure(sizeof(int)))
        return fail(c, LIW_ENOMEM, "hipMalloc");
    int has = has_prior ? 1 : 0;
    HIPCHK(c, hipMemcpy(c->has_prior.p, &has, sizeof(int), hipMemcpyHostToDevice));
    if (has) {
        if (!X15 || !J225) return fail(c, LIW_EINVAL, "liw_set_prior: X and J required");
        double z[15] = {0};
        HIPCHK(c, hipMemcpy(c->prior_X.p, X15, sizeof(double) * 15, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->prior_J.p, J225, sizeof(double) * 225, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->prior_R.p, R15 ? R15 : z, sizeof(double) * 15, hipMemcpyHostToDevice));
    }
    if (c->have_window) {
        c->sb.prior_X = c->prior_X.as<double>(); c->sb.prior_J = c->prior_J.as<double>();
        c->sb.prior_R = c->prior_R.as<double>(); c->sb.has_prior = c->has_prior.as<int>();
    }
    return LIW_OK;
}

}  // extern "C"
