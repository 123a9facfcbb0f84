// replay_log — offline replay of a flat sensor log (format: 2dliw-slam_amd/replay.py) through lvio_2d::trajectory
// (include/lvio_2d_trajectory.hpp): the reference's front-end driver with the MI355X estimator underneath, without ROS.
//   usage: replay_log <log.bin> <output_dir/> [look_ahead] [--keep N] [--loops <file>] [--solve-period S] [--pg-iters K]
//     --keep N         frames kept in the window after a tracking solve (trajectory_params::keep_window_size; 1 = reference)
//     --loops <file>   enables the back-end (include/lvio_2d_keyframe_manager.hpp -> liw_posegraph_solve) with a loop-edge schedule
//                      standing in for loop detection: int32 count, then per edge int32 trigger key frame, int32 older key frame,
//                      float64 tf12[12]
// writes <output_dir>fornt_end.txt (TUM trajectory, the reference's file name), <output_dir>traj.md (record tables),
// <output_dir>result.bin: int32 status, frames, tracked, initializations, keyframes, solver_status; float64 time, state[15], and with
// --loops <output_dir>back_end.txt (TUM of the key frames, keyframe_manager.cpp:370-397) + <output_dir>backend.bin: int32 key frames,
// loop edges, solves, LM iterations of the last solve; float64 modify_delta_tf[12], current pose in the corrected frame [6], poses [N][6].
//   replay_log --backend-only <keyframes.bin> <output_dir/> --loops <file> [...]: no front-end; the key frames (int32 N, then per key
//   frame float64 time, p[3], q[3]) are handed to lvio_2d::keyframe_manager one by one (same outputs: back_end.txt, backend.bin).
// Parameters are the values of reference config/office.yaml.  Exit code 19 (LIW_ENODEV) when no MI355X is usable.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "lvio_2d_keyframe_manager.hpp"
#include "lvio_2d_trajectory.hpp"

static const double OFFICE_T_IMU_TO_WHEEL[16] = {0.0040697, -0.9998940, -0.0139789, -0.061, 0.0099712, 0.0140189, -0.9998520, 0.919,
                                                 0.9999420, 0.0039297, 0.0100272, -0.224, 0.0, 0.0, 0.0, 1.0};
static const double OFFICE_T_IMU_TO_LASER[16] = {0.0019070, -0.9999900, 0.0040438, 0.024, 0.0459794, -0.0039519, -0.9989346, -0.078,
                                                 0.9989406, 0.0020909, 0.0459714, -0.071, 0.0, 0.0, 0.0, 1.0};

int main(int argc, char** argv) {
    bool backend_only = false;
    if (argc > 1 && std::string(argv[1]) == "--backend-only") { backend_only = true; --argc; ++argv; }
    if (argc < 3) { fprintf(stderr, "usage: replay_log [--backend-only] <log.bin | keyframes.bin> <output_dir/> [look_ahead] [--keep N] [--loops <file>]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    liw_params prm{};
    for (int k = 0; k < 16; ++k) { prm.T_imu_to_wheel[k] = OFFICE_T_IMU_TO_WHEEL[k]; prm.T_imu_to_laser[k] = OFFICE_T_IMU_TO_LASER[k]; }
    prm.g = 9.8; prm.line_to_line_sigma = 0.001; prm.manifold_p_sigma = 0.01; prm.manifold_q_sigma = 0.0005;
    for (int k = 0; k < 3; ++k) { prm.imu_noise_acc_sigma[k] = 0.0163; prm.imu_bias_acc_sigma[k] = 0.00499; prm.imu_noise_gyro_sigma[k] = 0.003208; prm.imu_bias_gyro_sigma[k] = 0.000499; }
    prm.wheel_sigma[0] = 0.5; prm.wheel_sigma[1] = 99999.0; prm.wheel_sigma[2] = 999.99;
    prm.fast_mode = 0; prm.normalize_extrinsics = 1; prm.device = 0;
    liw_laser_params lp{};
    lp.w_laser_each_scan = 100.0; lp.h_laser_each_scan = 100.0; lp.laser_resolution = 0.05; lp.line_continuous_threshold = 0.1; lp.line_min_len = 0.05;
    lp.line_max_dis = 0.03; lp.line_max_tolerance_angle = 175.0; lp.ref_motion_filter_p = 0.01; lp.ref_motion_filter_q = 0.01; lp.ref_n_accumulation = 2;
    for (int k = 0; k < 16; ++k) lp.T_imu_to_laser[k] = OFFICE_T_IMU_TO_LASER[k];
    lp.normalize_extrinsics = 1;
    lvio_2d::trajectory_params tp;
    tp.output_tum = true;
    tp.output_dir = argv[2];
    int look_ahead = 40;
    const char* loops_path = nullptr;
    lvio_2d::keyframe_manager_params kp;
    for (int k = 0; k < 3; ++k) { kp.pg.loop_sigma_p[k] = 0.1; kp.pg.loop_sigma_q[k] = 0.01; }   // config/office.yaml:106-115
    kp.pg.loop_edge_k = 10.0; kp.pg.use_ground_p_factor = 1; kp.pg.use_ground_q_factor = 1;
    kp.output_tum = true; kp.output_dir = argv[2];
    for (int a = 3; a < argc; ++a) {
        const std::string s = argv[a];
        if (s == "--keep" && a + 1 < argc) tp.keep_window_size = atoi(argv[++a]);
        else if (s == "--loops" && a + 1 < argc) loops_path = argv[++a];
        else if (s == "--solve-period" && a + 1 < argc) kp.solve_period = atof(argv[++a]);
        else if (s == "--pg-iters" && a + 1 < argc) kp.max_iterations = atoi(argv[++a]);
        else if (a == 3 && s[0] != '-') look_ahead = atoi(argv[a]);
        else { fprintf(stderr, "unknown argument %s\n", argv[a]); return 2; }
    }
    struct loop_rec { int trigger, older; double tf12[12]; };
    std::vector<loop_rec> schedule;
    if (loops_path) {
        FILE* lf = fopen(loops_path, "rb");
        int cnt = 0;
        if (!lf || fread(&cnt, sizeof(int), 1, lf) != 1) { fprintf(stderr, "cannot read %s\n", loops_path); return 2; }
        schedule.resize((size_t)cnt);
        for (auto& l : schedule)
            if (fread(&l.trigger, sizeof(int), 1, lf) != 1 || fread(&l.older, sizeof(int), 1, lf) != 1 || fread(l.tf12, sizeof(double), 12, lf) != 12) { fprintf(stderr, "short loop file\n"); return 2; }
        fclose(lf);
    }
    int keyframes = 0;
    int rc = 0;
    {
        std::unique_ptr<lvio_2d::keyframe_manager> km;
        double backend_pose[6] = {0, 0, 0, 0, 0, 0};
        lvio_2d::trajectory traj(prm, lp, tp);
        if (loops_path) {
            km.reset(new lvio_2d::keyframe_manager(prm, kp));
            km->set_loop_detector([&](int index, const std::deque<lvio_2d::frame_info::ptr>&, lvio_2d::edge* e) {
                for (const auto& l : schedule)
                    if (l.trigger == index) { e->index1 = index; e->index2 = l.older; std::memcpy(e->tf12, l.tf12, sizeof l.tf12); return true; }
                return false;
            });
            traj.set_other_frame_sink([&](const std::deque<lvio_2d::frame_info::ptr>& fi) { km->update_other_frame(fi, backend_pose, backend_pose + 3); });
        }
        traj.set_keyframe_sink([&](const lvio_2d::frame_info::ptr& f) { ++keyframes; if (km) km->add_keyframe(f); });
        lvio_2d::dispatch_queue dq(&traj, look_ahead);
        int type;
        if (backend_only) {
            if (!km) { fprintf(stderr, "--backend-only needs --loops\n"); return 2; }
            int nk = 0;
            if (fread(&nk, sizeof(int), 1, f) != 1) return 2;
            for (int i = 0; i < nk && !km->last_status; ++i) {
                double v[7];
                if (fread(v, sizeof(double), 7, f) != 7) return 2;
                auto fr = std::make_shared<lvio_2d::frame_info>();
                fr->time = v[0]; fr->type = lvio_2d::frame_info::laser; fr->is_key_frame = true;
                for (int k = 0; k < 3; ++k) { fr->p[k] = v[1 + k]; fr->q[k] = v[4 + k]; }
                ++keyframes;
                km->add_keyframe(fr);
            }
            if (km->last_status) { rc = -km->last_status; fprintf(stderr, "back-end: %s\n", km->last_error()); }
        }
        while (!backend_only && fread(&type, sizeof(int), 1, f) == 1) {
            if (type == 0) {
                double v[7];
                if (fread(v, sizeof(double), 7, f) != 7) break;
                lvio_2d::sensor::imu m{v[0], {v[1], v[2], v[3]}, {v[4], v[5], v[6]}};
                dq.add(m);
            } else if (type == 1) {
                double v[13];
                if (fread(v, sizeof(double), 13, f) != 13) break;
                lvio_2d::sensor::wheel_odom m;
                m.time_stamp = v[0];
                std::memcpy(m.pose_R, v + 1, sizeof(double) * 9);
                std::memcpy(m.pose_t, v + 10, sizeof(double) * 3);
                dq.add(m);
            } else if (type == 3) {
                double t; float a[3]; int n;
                if (fread(&t, sizeof(double), 1, f) != 1 || fread(a, sizeof(float), 3, f) != 3 || fread(&n, sizeof(int), 1, f) != 1) break;
                std::vector<float> rg((size_t)n);
                if (n && fread(rg.data(), sizeof(float), (size_t)n, f) != (size_t)n) break;
                lvio_2d::sensor::laser m;
                m.time_stamp = t;
                m.points.resize((size_t)n * 3);
                m.times.resize((size_t)n);
                const int k = liw_laser_to_points(rg.data(), n, a[0], a[1], a[2], t, m.points.data(), m.times.data());   // sensor::laser ctor
                if (k < 0) { fprintf(stderr, "bad scan\n"); return 2; }
                m.points.resize((size_t)k * 3);
                m.times.resize((size_t)k);
                dq.add(m);
            } else { fprintf(stderr, "unknown record type %d\n", type); return 2; }
            if (traj.solver_status()) { rc = -traj.solver_status(); fprintf(stderr, "solver: %s\n", traj.solver_error()); break; }
            if (km && km->last_status) { rc = -km->last_status; fprintf(stderr, "back-end: %s\n", km->last_error()); break; }
        }
        if (!rc) dq.flush();
        if (!rc && traj.solver_status()) { rc = -traj.solver_status(); fprintf(stderr, "solver: %s\n", traj.solver_error()); }
        FILE* o = fopen((std::string(argv[2]) + "result.bin").c_str(), "wb");
        if (o) {
            const int cnt[6] = {(int)traj.get_status(), (int)traj.frames().size(), traj.tracked_frames, traj.initializations, keyframes, traj.solver_status()};
            fwrite(cnt, sizeof(int), 6, o);
            const double tm = traj.time();
            fwrite(&tm, sizeof(double), 1, o);
            fwrite(traj.p(), sizeof(double), 3, o); fwrite(traj.q(), sizeof(double), 3, o); fwrite(traj.v(), sizeof(double), 3, o); fwrite(traj.bs(), sizeof(double), 6, o);
            fclose(o);
        }
        if (km) {
            FILE* b = fopen((std::string(argv[2]) + "backend.bin").c_str(), "wb");
            if (b) {
                const int cnt[4] = {(int)km->keyframe_queue.size(), (int)km->loop_edges.size(), km->solves, km->last_summary.iterations};
                fwrite(cnt, sizeof(int), 4, b);
                fwrite(km->modify_delta_tf, sizeof(double), 12, b);
                fwrite(backend_pose, sizeof(double), 6, b);
                for (const auto& f : km->keyframe_queue) { fwrite(f->p, sizeof(double), 3, b); fwrite(f->q, sizeof(double), 3, b); }
                fclose(b);
            }
            fprintf(stderr, "back-end: %d key frames, %d loop edges, %d solve(s)\n", (int)km->keyframe_queue.size(), (int)km->loop_edges.size(), km->solves);
        }
        fprintf(stderr, "replay: dispatched %ld dropped %ld, %d initialisation(s), %d tracked frames, %d key frames\n", dq.dispatched, dq.dropped,
                traj.initializations, traj.tracked_frames, keyframes);
    }
    fclose(f);
    return rc;
}
