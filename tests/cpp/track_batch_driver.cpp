// C++ host of the reference's STEADY STATE as a batch (INTEGRATION.md "Batched tracking"): B robots tracking in lock-step, per laser frame
//     liw_batch_solve(LIW_MODE_TRACK) -> liw_batch_marg_linearize -> liw_batch_marg_schur
// on the 2-frame windows (previous frame, new frame), with the solved state / laser_match pose of the new frame carried into the next
// frame's window as its older frame and the prior arrays — the solver's persistent linearised block, src/factor/solver.h:31-37 — staying in
// place (src/trajectory/trajectory.cpp:525-560: solver.solve(); solver.marginalization(); once per laser frame).  No Python, no torch:
// hipMalloc'd arrays behind the C ABI of include/liw_window.h.
// Input: F frame dumps, each the batch dump of tests/test_gpu_cpp_sharded.py::dump_batch (int32 B, n = 2; per window int32 L + arrays).
// Output: int32 F, then per frame: states [B][2][15], match_pose [B][2][12], (iterations, termination, successful) int32 x3 per window,
//         Delta_H [B][225], Delta_g [B][15], prior_X [B][15], prior_J [B][225], prior_R [B][15], has_prior int32 [B].
// usage: track_batch_driver <out.bin> <frame0.bin> <frame1.bin> ...        (tests/test_gpu_cpp_track_batch.py compares with the Python mirror, bit for bit)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "liw_window.h"

static const double OFFICE_T_IMU_TO_WHEEL[16] = {0.0040697, -0.9998940, -0.0139789, -0.061, 0.0099712, 0.0140189, -0.9998520, 0.919,
                                                 0.9999420, 0.0039297, 0.0100272, -0.224, 0.0, 0.0, 0.0, 1.0};
static const double OFFICE_T_IMU_TO_LASER[16] = {0.0019070, -0.9999900, 0.0040438, 0.024, 0.0459794, -0.0039519, -0.9989346, -0.078,
                                                 0.9989406, 0.0020909, 0.0459714, -0.071, 0.0, 0.0, 0.0, 1.0};
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(4); } } while (0)
#define LIWOK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, r_, liw_last_error(ctx)); exit(r_ < 0 ? -r_ : r_); } } while (0)

template <class T> static std::vector<T> rd(FILE* f, size_t cnt) {
    std::vector<T> v(cnt);
    if (cnt && fread(v.data(), sizeof(T), cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}
template <class T> static T* up(const std::vector<T>& h) {
    T* d = nullptr;
    HIPOK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIPOK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
template <class T> static void down(FILE* out, const T* d, size_t cnt) {
    std::vector<T> h(cnt);
    HIPOK(hipMemcpy(h.data(), d, cnt * sizeof(T), hipMemcpyDeviceToHost));
    fwrite(h.data(), sizeof(T), cnt, out);
}

struct Frame {   // one frame's batch: the caller-side arrays of liw_batch, resident in HBM
    int B = 0, n = 0, Ltot = 0;
    double *x = nullptr, *match_pose = nullptr, *laser_pts = nullptr, *imu_X = nullptr, *imu_J = nullptr, *imu_P = nullptr, *imu_Dt = nullptr, *wheel_T = nullptr, *wheel_P = nullptr;
    int *laser_off = nullptr, *laser_frame = nullptr;
    unsigned char* has_match = nullptr;
};
static Frame load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    int hdr[2];
    if (fread(hdr, sizeof(int), 2, f) != 2) exit(2);
    Frame fr;
    fr.B = hdr[0]; fr.n = hdr[1];
    const int B = fr.B, n = fr.n;
    std::vector<double> x, mp, iX, iJ, iP, iD, wT, wP, rows;
    std::vector<unsigned char> hm;
    std::vector<int> off(1, 0), lf;
    for (int b = 0; b < B; ++b) {
        int L;
        if (fread(&L, sizeof(int), 1, f) != 1) exit(2);
        auto a = rd<double>(f, n * 15); x.insert(x.end(), a.begin(), a.end());
        auto l = rd<int>(f, L); lf.insert(lf.end(), l.begin(), l.end());
        auto p = rd<double>(f, (size_t)L * 12); rows.insert(rows.end(), p.begin(), p.end());
        a = rd<double>(f, n * 12); mp.insert(mp.end(), a.begin(), a.end());
        auto h = rd<unsigned char>(f, n); hm.insert(hm.end(), h.begin(), h.end());
        a = rd<double>(f, (n - 1) * 15); iX.insert(iX.end(), a.begin(), a.end());
        a = rd<double>(f, (n - 1) * 225); iJ.insert(iJ.end(), a.begin(), a.end());
        a = rd<double>(f, (n - 1) * 225); iP.insert(iP.end(), a.begin(), a.end());
        a = rd<double>(f, n - 1); iD.insert(iD.end(), a.begin(), a.end());
        a = rd<double>(f, (n - 1) * 12); wT.insert(wT.end(), a.begin(), a.end());
        a = rd<double>(f, (n - 1) * 9); wP.insert(wP.end(), a.begin(), a.end());
        (void)rd<double>(f, n - 1);                       // wheel_Dt: not part of the factor (wheel_factor.h:6-82)
        off.push_back((int)lf.size());
    }
    fclose(f);
    fr.Ltot = (int)lf.size();
    std::vector<double> soa((size_t)12 * std::max(fr.Ltot, 1));      // component-major [12][Ltot]
    for (int j = 0; j < fr.Ltot; ++j)
        for (int c = 0; c < 12; ++c) soa[(size_t)c * fr.Ltot + j] = rows[(size_t)j * 12 + c];
    if (lf.empty()) lf.push_back(0);
    fr.x = up(x); fr.match_pose = up(mp); fr.has_match = up(hm); fr.laser_off = up(off); fr.laser_frame = up(lf); fr.laser_pts = up(soa);
    fr.imu_X = up(iX); fr.imu_J = up(iJ); fr.imu_P = up(iP); fr.imu_Dt = up(iD); fr.wheel_T = up(wT); fr.wheel_P = up(wP);
    return fr;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    HIPOK(hipSetDevice(0));
    liw_params prm{};
    memcpy(prm.T_imu_to_wheel, OFFICE_T_IMU_TO_WHEEL, sizeof prm.T_imu_to_wheel);
    memcpy(prm.T_imu_to_laser, OFFICE_T_IMU_TO_LASER, sizeof prm.T_imu_to_laser);
    prm.g = 9.8; prm.line_to_line_sigma = 0.001; prm.manifold_p_sigma = 0.01; prm.manifold_q_sigma = 0.0005;
    for (int k = 0; k < 3; ++k) {
        prm.imu_noise_acc_sigma[k] = 0.0163; prm.imu_bias_acc_sigma[k] = 0.00499;
        prm.imu_noise_gyro_sigma[k] = 0.003208; prm.imu_bias_gyro_sigma[k] = 0.000499;
    }
    prm.wheel_sigma[0] = 0.5; prm.wheel_sigma[1] = 99999.0; prm.wheel_sigma[2] = 999.99;
    prm.fast_mode = 0; prm.normalize_extrinsics = 1; prm.device = 0;
    liw_ctx* ctx = liw_create(&prm);
    if (!ctx) { fprintf(stderr, "liw_create failed\n"); return 19; }

    const int F = argc - 2;
    std::vector<Frame> fr;
    for (int k = 0; k < F; ++k) fr.push_back(load(argv[2 + k]));
    const int B = fr[0].B, n = fr[0].n;
    if (n != 2) { fprintf(stderr, "tracking windows have two frames\n"); return 2; }
    // the solver's persistent linearised block: ONE set of arrays for all frames
    std::vector<double> z15((size_t)B * 15, 0.0), z225((size_t)B * 225, 0.0);
    std::vector<int> zi(B, 0);
    double *pX = up(z15), *pJ = up(z225), *pR = up(z15);
    int* pH = up(zi);
    double *sH = nullptr, *dH = nullptr, *dg = nullptr;
    HIPOK(hipMalloc(&sH, sizeof(double) * 36 * B)); HIPOK(hipMalloc(&dH, sizeof(double) * 225 * B)); HIPOK(hipMalloc(&dg, sizeof(double) * 15 * B));
    liw_ws_layout lay{};
    LIWOK(liw_batch_ws_layout(B, n, 0, &lay));
    void* ws = nullptr;
    HIPOK(hipMalloc(&ws, lay.bytes));
    HIPOK(hipMemset(ws, 0, lay.bytes));
    hipStream_t s;
    HIPOK(hipStreamCreate(&s));
    FILE* out = fopen(argv[1], "wb");
    if (!out) return 2;
    fwrite(&F, sizeof(int), 1, out);
    std::vector<liw_summary> sm(B);
    for (int k = 0; k < F; ++k) {
        Frame& f = fr[k];
        if (f.B != B || f.n != n) { fprintf(stderr, "frame %d: batch shape differs\n", k); return 2; }
        if (k > 0) {   // the older frame of this window is the previous window's solved new frame (states and the laser_match p2, q2)
            HIPOK(hipMemcpy2DAsync(f.x, sizeof(double) * 30, fr[k - 1].x + 15, sizeof(double) * 30, sizeof(double) * 15, B, hipMemcpyDeviceToDevice, s));
            HIPOK(hipMemcpy2DAsync(f.match_pose + 6, sizeof(double) * 24, fr[k - 1].match_pose + 12 + 6, sizeof(double) * 24, sizeof(double) * 6, B, hipMemcpyDeviceToDevice, s));
        }
        liw_batch b{};
        b.B = B; b.n = n; b.Ltot = f.Ltot;
        b.x = f.x; b.laser_off = f.laser_off; b.laser_frame = f.laser_frame; b.laser_pts = f.laser_pts; b.match_pose = f.match_pose; b.has_match = f.has_match;
        b.imu_X = f.imu_X; b.imu_J = f.imu_J; b.imu_sqrtP = f.imu_P; b.imu_Dt = f.imu_Dt; b.wheel_T = f.wheel_T; b.wheel_sqrtP = f.wheel_P;
        b.prior_X = pX; b.prior_J = pJ; b.prior_R = pR; b.has_prior = pH;
        b.eval_small = 1; b.history_records = 0;
        LIWOK(liw_batch_solve(ctx, &b, LIW_MODE_TRACK, 0, ws, s, 0));
        HIPOK(hipStreamSynchronize(s));
        HIPOK(hipMemcpy(sm.data(), (char*)ws + lay.info_off, sizeof(liw_summary) * B, hipMemcpyDeviceToHost));
        LIWOK(liw_batch_marg_linearize(ctx, &b, ws, s));
        LIWOK(liw_batch_marg_schur(ctx, &b, ws, sH, dH, dg, s));
        HIPOK(hipStreamSynchronize(s));
        down(out, f.x, (size_t)B * 30); down(out, f.match_pose, (size_t)B * 24);
        for (int w = 0; w < B; ++w) { int t[3] = {sm[w].iterations, sm[w].termination, sm[w].successful_steps}; fwrite(t, sizeof(int), 3, out); }
        down(out, dH, (size_t)B * 225); down(out, dg, (size_t)B * 15);
        down(out, pX, (size_t)B * 15); down(out, pJ, (size_t)B * 225); down(out, pR, (size_t)B * 15); down(out, pH, (size_t)B);
    }
    fclose(out);
    liw_destroy(ctx);
    return 0;
}
