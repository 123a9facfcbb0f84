// lvio_2d_solver.hpp — C++ host mirror of the reference interfaces the hot path sits behind, header-only over the
// C ABI of liw_window.h.  Same class and method names as the reference so that `lvio_2d::trajectory` code reads the
// same; containers are plain arrays instead of Eigen members so that this header has no third-party dependency.
//   lvio_2d::solver                     reference src/factor/solver.h:28-79  (init_solve / solve / marginalization)
//   lvio_2d::frame_info / laser_match   reference src/trajectory/trajectory_type.h:9-75, src/trajectory/laser_type.h:13-21,76-85
//   lvio_2d::imu_preintegraption        reference src/factor/imu_preintegraption.h:105-208
//   lvio_2d::wheel_odom_preintegration  reference src/factor/wheel_odom_preintegration.h:44-152
// The Eigen-typed shim for the reference tree itself is in INTEGRATION.md; it differs from this file only in how the
// vectors are copied.  Error convention: like the reference, the three solver methods return void; `last_status`
// holds the C-ABI return code (LIW_ENODEV without an MI355X: there is no CPU fallback) and `last_error()` the text.
#pragma once
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "liw_window.h"

namespace lvio_2d {

struct line {
    double p1[3], p2[3];
};
struct scan;   // lvio_2d_laser.hpp
struct laser_match {
    using ptr = std::shared_ptr<laser_match>;
    std::vector<line> lines1, lines2;
    double p1[3], q1[3], p2[3], q2[3];
    std::shared_ptr<scan> scan2;   // the matched scan (set by lvio_2d::laser_manager)
};
struct imu_preint_result {
    using ptr = std::shared_ptr<imu_preint_result>;
    double X[15];
    double J[225];               // row-major
    double sqrt_inverse_P[225];  // row-major
    double Dt;
};
struct wheel_odom_preint_result {
    using ptr = std::shared_ptr<wheel_odom_preint_result>;
    double delta_Tij[12];        // R (row-major 3x3) then t
    double sqrt_inverse_P[9];
    double Dt;
};
struct frame_info {
    enum frame_type { laser = 0, camera = 1, unknow = 2 };
    using ptr = std::shared_ptr<frame_info>;
    double time = 0;
    double p[3] = {0, 0, 0}, q[3] = {0, 0, 0}, v[3] = {0, 0, 0}, bs[6] = {0, 0, 0, 0, 0, 0};
    imu_preint_result::ptr imu_observation_reslut;          // i-1 ~ i
    wheel_odom_preint_result::ptr wheel_observation_reslut; // i-1 ~ i
    laser_match::ptr laser_match_ptr;
    frame_type type = unknow;
    bool is_key_frame = false;
    std::vector<double> laser_concers;   // [k][3], only carried for key frames
    double sqrt_H[36];
    frame_info() { for (int k = 0; k < 36; ++k) sqrt_H[k] = (k % 7 == 0) ? 1.0 : 0.0; }
    void add_laser_match(const laser_match::ptr& m) { laser_match_ptr = m; type = laser; }
    void set_key_frame() { is_key_frame = true; }
    void set_acc_concers(const std::vector<double>& c) { laser_concers = c; }
    static ptr create(double time_, const double* p_, const double* q_, const double* v_, const double* bs_, const imu_preint_result::ptr& imu_,
                      const wheel_odom_preint_result::ptr& wheel_) {
        ptr r = std::make_shared<frame_info>();
        r->time = time_;
        for (int k = 0; k < 3; ++k) { r->p[k] = p_[k]; r->q[k] = q_[k]; r->v[k] = v_[k]; }
        for (int k = 0; k < 6; ++k) r->bs[k] = bs_[k];
        r->imu_observation_reslut = imu_;
        r->wheel_observation_reslut = wheel_;
        return r;
    }
};

struct feature_manger {};   // camera feature tracks (reference src/trajectory/camera_type.h): out of scope, kept for the signatures

class solver {
public:
    int last_status = 0;
    liw_summary last_summary{};

    explicit solver(const liw_params& prm) : ctx_(liw_create(&prm)), fast_mode_(prm.fast_mode != 0) {}
    ~solver() { liw_destroy(ctx_); }
    solver(const solver&) = delete;
    solver& operator=(const solver&) = delete;
    const char* last_error() const { return liw_last_error(ctx_); }
    // not in the reference (its driver constructs a new solver, solver.h:31: has_linearized_block = false): forget the stored prior
    void clear_prior() { last_status = liw_set_prior(ctx_, 0, nullptr, nullptr, nullptr); }

    // reference signatures (src/factor/solver.h:71-79): the feature_manger is the camera track store, unused with
    // enable_camera: false (every shipped config); the one-argument forms are what the camera-less driver calls
    void init_solve(std::deque<frame_info::ptr>& frame_infos, feature_manger&) { run(frame_infos, LIW_MODE_INIT); }
    void solve(std::deque<frame_info::ptr>& frame_infos, feature_manger&) { run(frame_infos, LIW_MODE_TRACK); }
    void marginalization(std::deque<frame_info::ptr>& frame_infos, feature_manger&) { marginalization(frame_infos); }
    void init_solve(std::deque<frame_info::ptr>& frame_infos) { run(frame_infos, LIW_MODE_INIT); }
    void solve(std::deque<frame_info::ptr>& frame_infos) { run(frame_infos, LIW_MODE_TRACK); }
    void marginalization(std::deque<frame_info::ptr>& frame_infos) {
        flat f(frame_infos);
        last_status = liw_set_window(ctx_, &f.w);
        if (last_status) return;
        double sqrt_H[36];
        last_status = liw_marginalize(ctx_, sqrt_H, nullptr, nullptr);
        if (last_status == 0 && !fast_mode_)
            for (int k = 0; k < 36; ++k) frame_infos.back()->sqrt_H[k] = sqrt_H[k];
        liw_clear_window(ctx_);   // `f` dies here: the ctx must not keep its host pointers
    }

private:
    struct flat {   // std::deque<frame_info::ptr> -> liw_window
        std::vector<double> states, laser_pts, match_pose, imu_X, imu_J, imu_P, imu_Dt, wheel_T, wheel_P, wheel_Dt;
        std::vector<int> laser_frame;
        std::vector<unsigned char> has_match;
        liw_window w{};
        explicit flat(std::deque<frame_info::ptr>& fi) {
            const int n = (int)fi.size();
            for (int i = 0; i < n; ++i) {
                const frame_info& f = *fi[i];
                states.insert(states.end(), f.p, f.p + 3);
                states.insert(states.end(), f.q, f.q + 3);
                states.insert(states.end(), f.v, f.v + 3);
                states.insert(states.end(), f.bs, f.bs + 6);
                const bool m = f.type == frame_info::laser && f.laser_match_ptr;
                has_match.push_back(m ? 1 : 0);
                match_pose.insert(match_pose.end(), 12, 0.0);
                if (m) {
                    const laser_match& lm = *f.laser_match_ptr;
                    for (int k = 0; k < 3; ++k) {
                        match_pose[i * 12 + k] = lm.p1[k]; match_pose[i * 12 + 3 + k] = lm.q1[k];
                        match_pose[i * 12 + 6 + k] = lm.p2[k]; match_pose[i * 12 + 9 + k] = lm.q2[k];
                    }
                    for (size_t j = 0; j < lm.lines1.size(); ++j) {
                        laser_frame.push_back(i);
                        laser_pts.insert(laser_pts.end(), lm.lines1[j].p1, lm.lines1[j].p1 + 3);
                        laser_pts.insert(laser_pts.end(), lm.lines1[j].p2, lm.lines1[j].p2 + 3);
                        laser_pts.insert(laser_pts.end(), lm.lines2[j].p1, lm.lines2[j].p1 + 3);
                        laser_pts.insert(laser_pts.end(), lm.lines2[j].p2, lm.lines2[j].p2 + 3);
                    }
                }
                if (i > 0) {
                    const imu_preint_result& r = *f.imu_observation_reslut;
                    imu_X.insert(imu_X.end(), r.X, r.X + 15);
                    imu_J.insert(imu_J.end(), r.J, r.J + 225);
                    imu_P.insert(imu_P.end(), r.sqrt_inverse_P, r.sqrt_inverse_P + 225);
                    imu_Dt.push_back(r.Dt);
                    const wheel_odom_preint_result& wr = *f.wheel_observation_reslut;
                    wheel_T.insert(wheel_T.end(), wr.delta_Tij, wr.delta_Tij + 12);
                    wheel_P.insert(wheel_P.end(), wr.sqrt_inverse_P, wr.sqrt_inverse_P + 9);
                    wheel_Dt.push_back(wr.Dt);
                }
            }
            auto ptr = [](std::vector<double>& v) { if (v.empty()) v.push_back(0.0); return v.data(); };
            if (laser_frame.empty()) laser_frame.push_back(0);
            const int L = (int)(laser_pts.size() / 12);
            w = liw_window{n, L, states.data(), laser_frame.data(), ptr(laser_pts), match_pose.data(), has_match.data(), ptr(imu_X),
                           ptr(imu_J), ptr(imu_P), ptr(imu_Dt), ptr(wheel_T), ptr(wheel_P), ptr(wheel_Dt)};
        }
        void scatter(std::deque<frame_info::ptr>& fi) {   // results back in place, as the reference mutates them
            for (size_t i = 0; i < fi.size(); ++i) {
                frame_info& f = *fi[i];
                for (int k = 0; k < 3; ++k) { f.p[k] = states[i * 15 + k]; f.q[k] = states[i * 15 + 3 + k]; f.v[k] = states[i * 15 + 6 + k]; }
                for (int k = 0; k < 6; ++k) f.bs[k] = states[i * 15 + 9 + k];
                if (has_match[i]) {
                    laser_match& lm = *f.laser_match_ptr;
                    for (int k = 0; k < 3; ++k) {
                        lm.p1[k] = match_pose[i * 12 + k]; lm.q1[k] = match_pose[i * 12 + 3 + k];
                        lm.p2[k] = match_pose[i * 12 + 6 + k]; lm.q2[k] = match_pose[i * 12 + 9 + k];
                    }
                }
            }
        }
    };
    void run(std::deque<frame_info::ptr>& fi, int mode) {
        flat f(fi);
        last_status = liw_set_window(ctx_, &f.w);
        if (last_status) return;
        last_status = liw_solve(ctx_, mode, 0, &last_summary);   // Summary is informational; the reference drops it
        if (last_status == 0) f.scatter(fi);
        liw_clear_window(ctx_);   // `f` dies here: the ctx must not keep its host pointers
    }
    liw_ctx* ctx_;
    bool fast_mode_;
};

class imu_preintegraption {
public:
    explicit imu_preintegraption(const liw_params& prm) : h_(liw_imu_preint_create(&prm)) {}
    ~imu_preintegraption() { liw_imu_preint_destroy(h_); }
    imu_preintegraption(const imu_preintegraption&) = delete;
    imu_preintegraption& operator=(const imu_preintegraption&) = delete;
    void reset_imu_measure(double time, const double* acc_bias, const double* gyr_bias) { liw_imu_preint_reset(h_, time, acc_bias, gyr_bias); }
    bool add_imu_measure(double time_stamp, const double* acc, const double* gyro) { return liw_imu_preint_add(h_, time_stamp, acc, gyro) != 0; }
    void update_only_t(double time) { liw_imu_preint_update_only_t(h_, time); }
    double Dt() const { return liw_imu_preint_Dt(h_); }
    imu_preint_result::ptr get_preintegraption_result() const {
        auto r = std::make_shared<imu_preint_result>();
        liw_imu_preint_result(h_, r->X, r->J, r->sqrt_inverse_P, &r->Dt);
        return r;
    }
private:
    liw_imu_preint* h_;
};

class wheel_odom_preintegration {
public:
    explicit wheel_odom_preintegration(const liw_params& prm) : h_(liw_wheel_preint_create(&prm)) {}
    ~wheel_odom_preintegration() { liw_wheel_preint_destroy(h_); }
    wheel_odom_preintegration(const wheel_odom_preintegration&) = delete;
    wheel_odom_preintegration& operator=(const wheel_odom_preintegration&) = delete;
    void reset_wheel_odom_measure(double time) { liw_wheel_preint_reset(h_, time); }
    bool add_wheel_odom_measure(double time_stamp, const double* pose_R9, const double* pose_t3) { return liw_wheel_preint_add(h_, time_stamp, pose_R9, pose_t3) != 0; }
    void update_only_t(double time) { liw_wheel_preint_update_only_t(h_, time); }
    wheel_odom_preint_result::ptr get_preintegraption_result() const {
        auto r = std::make_shared<wheel_odom_preint_result>();
        liw_wheel_preint_result(h_, r->delta_Tij, r->sqrt_inverse_P, &r->Dt);
        return r;
    }
private:
    liw_wheel_preint* h_;
};

}  // namespace lvio_2d
