// Test driver for the C++ host mirror (include/lvio_2d_solver.hpp): reads a flat window dumped by the Python tests,
// rebuilds std::deque<frame_info::ptr> like lvio_2d::trajectory would hold it, runs
//   solver.init_solve -> solver.marginalization        (mode "init")     or
//   solver.solve                                        (mode "track")
// through the reference-named C++ classes and writes the resulting states / laser_match poses / sqrt_H back.
// File format (little-endian): int32 n, L, then float64 arrays in liw_window order.
// usage: host_api_driver <in.bin> <out.bin> <init|track>      exit code = -last_status (19 = LIW_ENODEV)
#include <cstdio>
#include <cstring>
#include <vector>

#include "lvio_2d_solver.hpp"

static const double OFFICE_T_IMU_TO_WHEEL[16] = {0.0040697, -0.9998940, -0.0139789, -0.061, 0.0099712, 0.0140189, -0.9998520, 0.919,
                                                 0.9999420, 0.0039297, 0.0100272, -0.224, 0.0, 0.0, 0.0, 1.0};
static const double OFFICE_T_IMU_TO_LASER[16] = {0.0019070, -0.9999900, 0.0040438, 0.024, 0.0459794, -0.0039519, -0.9989346, -0.078,
                                                 0.9989406, 0.0020909, 0.0459714, -0.071, 0.0, 0.0, 0.0, 1.0};

template <class T> static std::vector<T> rd(FILE* f, size_t cnt) {
    std::vector<T> v(cnt);
    if (cnt && fread(v.data(), sizeof(T), cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[2];
    if (fread(hdr, sizeof(int), 2, f) != 2) return 2;
    const int n = hdr[0], L = hdr[1];
    auto states = rd<double>(f, n * 15);
    auto laser_frame = rd<int>(f, L);
    auto laser_pts = rd<double>(f, (size_t)L * 12);
    auto match_pose = rd<double>(f, n * 12);
    auto has_match = rd<unsigned char>(f, n);
    auto imu_X = rd<double>(f, (n - 1) * 15), imu_J = rd<double>(f, (n - 1) * 225), imu_P = rd<double>(f, (n - 1) * 225), imu_Dt = rd<double>(f, n - 1);
    auto wheel_T = rd<double>(f, (n - 1) * 12), wheel_P = rd<double>(f, (n - 1) * 9), wheel_Dt = rd<double>(f, n - 1);
    fclose(f);

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

    std::deque<lvio_2d::frame_info::ptr> frame_infos;
    int lpos = 0;
    for (int i = 0; i < n; ++i) {
        auto fr = std::make_shared<lvio_2d::frame_info>();
        for (int k = 0; k < 3; ++k) { fr->p[k] = states[i * 15 + k]; fr->q[k] = states[i * 15 + 3 + k]; fr->v[k] = states[i * 15 + 6 + k]; }
        for (int k = 0; k < 6; ++k) fr->bs[k] = states[i * 15 + 9 + k];
        if (i > 0) {
            auto r = std::make_shared<lvio_2d::imu_preint_result>();
            memcpy(r->X, &imu_X[(i - 1) * 15], sizeof r->X);
            memcpy(r->J, &imu_J[(i - 1) * 225], sizeof r->J);
            memcpy(r->sqrt_inverse_P, &imu_P[(i - 1) * 225], sizeof r->sqrt_inverse_P);
            r->Dt = imu_Dt[i - 1];
            fr->imu_observation_reslut = r;
            auto w = std::make_shared<lvio_2d::wheel_odom_preint_result>();
            memcpy(w->delta_Tij, &wheel_T[(i - 1) * 12], sizeof w->delta_Tij);
            memcpy(w->sqrt_inverse_P, &wheel_P[(i - 1) * 9], sizeof w->sqrt_inverse_P);
            w->Dt = wheel_Dt[i - 1];
            fr->wheel_observation_reslut = w;
        }
        if (has_match[i]) {
            auto lm = std::make_shared<lvio_2d::laser_match>();
            for (int k = 0; k < 3; ++k) {
                lm->p1[k] = match_pose[i * 12 + k]; lm->q1[k] = match_pose[i * 12 + 3 + k];
                lm->p2[k] = match_pose[i * 12 + 6 + k]; lm->q2[k] = match_pose[i * 12 + 9 + k];
            }
            while (lpos < L && laser_frame[lpos] == i) {
                lvio_2d::line a, b;
                memcpy(a.p1, &laser_pts[(size_t)lpos * 12], 24); memcpy(a.p2, &laser_pts[(size_t)lpos * 12 + 3], 24);
                memcpy(b.p1, &laser_pts[(size_t)lpos * 12 + 6], 24); memcpy(b.p2, &laser_pts[(size_t)lpos * 12 + 9], 24);
                lm->lines1.push_back(a); lm->lines2.push_back(b);
                ++lpos;
            }
            fr->add_laser_match(lm);
        }
        frame_infos.push_back(fr);
    }

    lvio_2d::solver opt_solver(prm);
    if (strcmp(argv[3], "init") == 0) {
        opt_solver.init_solve(frame_infos);
        if (opt_solver.last_status == 0) opt_solver.marginalization(frame_infos);
    } else {
        opt_solver.solve(frame_infos);
    }
    if (opt_solver.last_status != 0) {
        fprintf(stderr, "solver status %d: %s\n", opt_solver.last_status, opt_solver.last_error());
        return -opt_solver.last_status;
    }
    FILE* o = fopen(argv[2], "wb");
    for (int i = 0; i < n; ++i) {
        auto& fr = *frame_infos[i];
        fwrite(fr.p, 8, 3, o); fwrite(fr.q, 8, 3, o); fwrite(fr.v, 8, 3, o); fwrite(fr.bs, 8, 6, o);
    }
    for (int i = 0; i < n; ++i) {
        double mp[12] = {0};
        if (frame_infos[i]->laser_match_ptr) {
            auto& lm = *frame_infos[i]->laser_match_ptr;
            memcpy(mp, lm.p1, 24); memcpy(mp + 3, lm.q1, 24); memcpy(mp + 6, lm.p2, 24); memcpy(mp + 9, lm.q2, 24);
        }
        fwrite(mp, 8, 12, o);
    }
    fwrite(frame_infos.back()->sqrt_H, 8, 36, o);
    int it = opt_solver.last_summary.iterations;
    fwrite(&it, 4, 1, o);
    fclose(o);
    return 0;
}
