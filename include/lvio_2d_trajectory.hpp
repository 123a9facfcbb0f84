// lvio_2d_trajectory.hpp — C++ host mirror of the reference's front-end driver, header-only over the
// other mirrors: lvio_2d::trajectory (reference src/trajectory/trajectory.h:16-84, trajectory.cpp) for the camera-less
// configuration every shipped config uses (config/office.yaml:6), and the offline form of the oldest-first message merge
// of lvio_2d::dispatch (src/trajectory/dispatch.h:192-257).  Same member / method names as the reference, so the code
// reads side by side with trajectory.cpp:
//   add_sensor_data(wheel_odom / imu / laser)      :69-80, :137-277
//   update_current_status                          :82-99
//   check_and_processing_initialize                :381-487   (init window -> solver.init_solve -> marginalization)
//   do_tracking                                    :525-560   (solver.solve + marginalization per laser frame, TUM line)
//   pop_frame / pop_frame_for_tracking             :488-524, :590-617
// Out of scope here: ROS transport, visualisation, camera branches, and the pose-graph back-end — popped key
// frames are handed to a sink callback instead of keyframe_manager.
#pragma once
#include <cmath>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "liw_io.h"
#include "liw_lie.h"
#include "lvio_2d_laser.hpp"
#include "lvio_2d_solver.hpp"

namespace lvio_2d {

namespace sensor {   // reference src/trajectory/sensor.h:14-125 without the ROS message constructors
struct imu { double time_stamp; double acc[3], gyro[3]; };
struct wheel_odom { double time_stamp; double pose_R[9], pose_t[3]; };   // nav_msgs/Odometry pose as rotation + translation
struct laser {
    double time_stamp = 0;
    std::vector<double> points;   // [n][3] laser frame
    std::vector<double> times;    // [n]
    void correct(const double* linear, const double* angular) {   // sensor.h:51-94
        liw_laser_correct(points.data(), times.data(), (int)times.size(), time_stamp, linear, angular);
    }
};
}  // namespace sensor

struct trajectory_params {   // the trajectory part of param::manager (config/office.yaml:73-75,94-95,127-128; FPS -> min_delta_t)
    int slide_window_size = 10;
    double p_motion_threshold = 0.1, q_motion_threshold = 0.05;
    double key_frame_p_motion_threshold = 0.05, key_frame_q_motion_threshold = 0.05;
    double min_delta_t = 0.001;
    // frames kept in the window after a tracking solve.  1 = the reference (pop_frame_for_tracking keeps the newest laser frame only,
    // trajectory.cpp:590-617: every tracking solve sees 2 frames); N keeps N, i.e. solver.solve / marginalization run on N + 1 frames —
    // the explicit keep-N policy of SURVEY 8 f3 (BASELINE configs C3 / C5: 30 / 50-key-frame windows).
    // N > 1 is NOT the reference's estimator and is an approximation: marginalization() still folds every frame but the newest into the
    // prior (solver.cpp:257-442 knows no other set), while the kept frames and their IMU / wheel blocks re-enter the next TRACK solve —
    // their information is counted in the prior AND as factors.  Parity claims are made for 1 only; the keep-N tests compare product and
    // oracle twin on the SAME policy (teacher-forced solves), not against the reference.
    int keep_window_size = 1;
    bool output_tum = false;
    std::string output_dir;
};

enum TRAJECTORY_STATUS { INITIALIZING = 0, TRACKING = 1 };

class trajectory {
public:
    using keyframe_sink = std::function<void(const frame_info::ptr&)>;

    trajectory(const liw_params& prm, const liw_laser_params& lprm, const trajectory_params& tprm)
        : prm_(prm), tprm_(tprm), imu_preintegraption_(prm), wheel_odom_preintegration_(prm), laser_manger_(lprm), opt_solver(prm) {
        liw_lie_from_matrix16(prm.T_imu_to_wheel, prm.normalize_extrinsics, T_imu_to_wheel);
        liw_lie_from_matrix16(prm.T_imu_to_laser, prm.normalize_extrinsics, T_imu_to_laser);
        recorder = liw_record_create();
        init_current_status();
        wheel_odom_inited = imu_inited = false;
        current_index = last_laser_index = -1;
        if (tprm_.output_tum) o_fstream = liw_tum_open((tprm_.output_dir + "fornt_end.txt").c_str(), &prm_);   // sic, trajectory.cpp:61
    }
    ~trajectory() {
        if (tprm_.output_tum) liw_record_write(recorder, (tprm_.output_dir + "traj.md").c_str());
        liw_tum_close(o_fstream);
        liw_record_destroy(recorder);
    }
    trajectory(const trajectory&) = delete;
    trajectory& operator=(const trajectory&) = delete;

    // ---- data entry points
    void add_sensor_data(const sensor::wheel_odom& d) {
        if (wheel_odom_preintegration_.add_wheel_odom_measure(d.time_stamp, d.pose_R, d.pose_t)) wheel_odom_inited = true;
    }
    void add_sensor_data(const sensor::imu& d) {
        if (imu_preintegraption_.add_imu_measure(d.time_stamp, d.acc, d.gyro)) imu_inited = true;
    }
    void add_sensor_data(sensor::laser& laser_data) {
        const double time = laser_data.time_stamp;
        if (status == TRACKING) {   // de-skew with the current body twist (:140-148)
            double T_w_i[12], T_w_laser[12], Rt_v[3], tmp_angular[3];
            liw_lie_make_tf(current_p, current_q, T_w_i);
            liw_lie_mul(T_w_i, T_imu_to_laser, T_w_laser);
            for (int i = 0; i < 3; ++i) Rt_v[i] = T_w_laser[0 * 3 + i] * current_v[0] + T_w_laser[1 * 3 + i] * current_v[1] + T_w_laser[2 * 3 + i] * current_v[2];
            // log_SO3(R_i_l^T exp_so3(current_angular_local) R_i_l)
            double E[9], Ril_t[12], A[12], B[12], C[12];
            liw_lie_exp_so3(current_angular_local, E);
            for (int k = 0; k < 9; ++k) { A[k] = E[k]; B[k] = T_imu_to_laser[k]; }
            for (int k = 9; k < 12; ++k) A[k] = B[k] = 0.0;
            liw_lie_inverse(B, Ril_t);
            liw_lie_mul(Ril_t, A, C);
            liw_lie_mul(C, B, A);
            liw_lie_log_SO3(A, tmp_angular);
            laser_data.correct(Rt_v, tmp_angular);
        }
        if (!imu_inited || !wheel_odom_inited) return;
        wheel_odom_preint_result::ptr wheel_result_filter = wheel_odom_preintegration_.get_preintegraption_result();
        double laser_delta_filter[12];
        wheel_delta_to(T_imu_to_laser, wheel_result_filter->delta_Tij, laser_delta_filter);
        if (status == INITIALIZING && is_static(laser_delta_filter, tprm_.p_motion_threshold, tprm_.q_motion_threshold)) return;
        if (status == TRACKING && imu_preintegraption_.Dt() < tprm_.min_delta_t) return;
        current_index++;
        // align the accumulators with the scan stamp (:176-184)
        wheel_odom_preintegration_.update_only_t(time);
        imu_preintegraption_.update_only_t(time);
        wheel_odom_preint_result::ptr wheel_result = wheel_odom_preintegration_.get_preintegraption_result();
        imu_preint_result::ptr imu_reuslt = imu_preintegraption_.get_preintegraption_result();
        wheel_odom_preintegration_.reset_wheel_odom_measure(time);
        imu_preintegraption_.reset_imu_measure(time, current_bs, current_bs + 3);
        for (int k = 0; k < 3; ++k) current_angular_local[k] = imu_reuslt->X[6 + k] / imu_reuslt->Dt;   // gamma / Dt
        double delta_tf[12];
        wheel_delta_to_imu_delta(wheel_result->delta_Tij, delta_tf);
        update_current_status(delta_tf, time);

        liw_record_begin(recorder);
        scan::ptr scan_ptr = laser_manger_.spawn_scan(laser_data.points.data(), (int)laser_data.times.size(), laser_data.times.empty() ? time : laser_data.times.front());
        liw_record_end(recorder, "spawn_scan");
        liw_record_add(recorder, "lines each frame", scan_ptr->lines.size());

        laser_match::ptr lm;
        if (status == INITIALIZING) {
            lm = laser_manger_.match_with_front(scan_ptr, current_p, current_q);
            laser_manger_.add_scan(scan_ptr, current_p, current_q);
        } else {
            liw_record_begin(recorder);
            lm = laser_manger_.match_with_ref(scan_ptr, current_p, current_q);
            liw_record_end(recorder, "match_line");
        }
        frame_info::ptr current_frame = frame_info::create(current_time, current_p, current_q, current_v, current_bs, imu_reuslt, wheel_result);
        current_frame->add_laser_match(lm);
        last_laser_index = current_index;
        frame_infos.push_back(current_frame);
        if (status == INITIALIZING) {
            if (check_and_processing_initialize()) status = TRACKING;
            return;
        }
        do_tracking();
        {
            double T_w_i[12], tf_w_l[12];
            liw_lie_make_tf(current_p, current_q, T_w_i);
            liw_lie_mul(T_w_i, T_imu_to_laser, tf_w_l);
            for (size_t i = 0; i + 2 < scan_ptr->concers.size(); i += 3) {
                double y[3];
                liw_lie_apply(tf_w_l, &scan_ptr->concers[i], y);
                acc_concers.insert(acc_concers.end(), y, y + 3);
            }
            double inv_last[12], delta_laser_tf[12];
            liw_lie_inverse(last_keyframe_tf, inv_last);
            liw_lie_mul(inv_last, tf_w_l, delta_laser_tf);
            const int n_match_size = lm ? (int)lm->lines1.size() : 0;
            liw_record_add(recorder, "match line size", (uint64_t)n_match_size);
            const int n_no_match_size = (int)scan_ptr->lines.size() - n_match_size;
            if (!is_static(delta_laser_tf, tprm_.key_frame_p_motion_threshold, tprm_.key_frame_q_motion_threshold) || n_match_size < n_no_match_size) {
                liw_record_add(recorder, "corner each keyframe", acc_concers.size() / 3);
                frame_infos.back()->set_acc_concers(acc_concers);
                frame_infos.back()->set_key_frame();
                acc_concers.clear();
                for (int k = 0; k < 12; ++k) last_keyframe_tf[k] = tf_w_l[k];
            }
        }
        liw_record_begin(recorder);
        laser_manger_.add_scan(scan_ptr, current_p, current_q);
        liw_record_end(recorder, "add scan to ref");
        last_time = current_time;
    }

    // ---- introspection (tests, replay tools)
    TRAJECTORY_STATUS get_status() const { return status; }
    const std::deque<frame_info::ptr>& frames() const { return frame_infos; }
    double time() const { return current_time; }
    const double* p() const { return current_p; }
    const double* q() const { return current_q; }
    const double* v() const { return current_v; }
    const double* bs() const { return current_bs; }
    int solver_status() const { return opt_solver.last_status; }
    const char* solver_error() const { return opt_solver.last_error(); }
    liw_record* get_recorder() { return recorder; }
    void set_keyframe_sink(keyframe_sink s) { sink_ = std::move(s); }
    // keyframe_manager::update_other_frame (trajectory.cpp:505-511): after every pop, the frames since the last key frame + the window
    using other_frame_sink = std::function<void(const std::deque<frame_info::ptr>&)>;
    void set_other_frame_sink(other_frame_sink s) { other_sink_ = std::move(s); }
    int tracked_frames = 0, initializations = 0;

private:
    static bool is_static(const double* delta_tf, double p_thr, double q_thr) {   // trajectory.cpp:5-12
        double dp[3], dq[3];
        liw_lie_log_SE3(delta_tf, dp, dq);
        return std::sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]) < p_thr && std::sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) < q_thr;
    }
    void wheel_delta_to_imu_delta(const double* wheel_delta, double* out) const {   // T_iw * d * T_iw^-1 (:13-16)
        double inv[12], tmp[12];
        liw_lie_inverse(T_imu_to_wheel, inv);
        liw_lie_mul(T_imu_to_wheel, wheel_delta, tmp);
        liw_lie_mul(tmp, inv, out);
    }
    void wheel_delta_to(const double* T_imu_to_x, const double* wheel_delta, double* out) const {   // (:18-27)
        double inv_x[12], T_x_to_wheel[12], inv[12], tmp[12];
        liw_lie_inverse(T_imu_to_x, inv_x);
        liw_lie_mul(inv_x, T_imu_to_wheel, T_x_to_wheel);
        liw_lie_inverse(T_x_to_wheel, inv);
        liw_lie_mul(T_x_to_wheel, wheel_delta, tmp);
        liw_lie_mul(tmp, inv, out);
    }
    void init_current_status() {   // (:38-57)
        status = INITIALIZING;
        for (int k = 0; k < 12; ++k) last_keyframe_tf[k] = (k < 9 && k % 4 == 0) ? 1.0 : 0.0;
        last_time = current_time = -std::numeric_limits<double>::max();
        double inv[12];
        liw_lie_inverse(T_imu_to_wheel, inv);
        liw_lie_log_SE3(inv, current_p, current_q);
        for (int k = 0; k < 3; ++k) current_v[k] = current_angular_local[k] = 0.0;
        for (int k = 0; k < 6; ++k) current_bs[k] = 0.0;
    }
    void update_current_status(const double* delta_tf, double time) {   // (:82-92; the code after the early return is dead there too)
        double T[12], N[12];
        liw_lie_make_tf(current_p, current_q, T);
        liw_lie_mul(T, delta_tf, N);
        liw_lie_log_SE3(N, current_p, current_q);
        current_time = time;
    }
    bool check_and_processing_initialize() {   // (:381-487)
        if ((int)frame_infos.size() < tprm_.slide_window_size) return false;
        int k = 0;
        bool is_first_laser = true;
        for (size_t i = 0; i < frame_infos.size(); i++) {
            if (frame_infos[i]->type != frame_info::laser) continue;
            if (is_first_laser) { is_first_laser = false; continue; }
            if (!frame_infos[i]->laser_match_ptr || frame_infos[i]->laser_match_ptr->lines2.size() < 2) { k = (int)i + 1; break; }
        }
        if (k > 0) {
            pop_frame((int)frame_infos.size());
            laser_manger_.clear_all_scan();
            init_current_status();
            return false;
        }
        ++initializations;
        liw_record_begin(recorder);
        opt_solver.init_solve(frame_infos);
        liw_record_end(recorder, "init_solve");
        int n_laser = 0;
        for (size_t i = 0; i < frame_infos.size(); i++)
            if (frame_infos[i]->type == frame_info::laser) laser_manger_.set_keyframe_pose(n_laser++, frame_infos[i]->p, frame_infos[i]->q);
        take_back_state();
        laser_manger_.clear_all_scan();
        for (size_t i = 0; i < frame_infos.size(); i++)
            if (frame_infos[i]->type == frame_info::laser && frame_infos[i]->laser_match_ptr && frame_infos[i]->laser_match_ptr->scan2)
                laser_manger_.add_scan(frame_infos[i]->laser_match_ptr->scan2, frame_infos[i]->p, frame_infos[i]->q);
        opt_solver.marginalization(frame_infos);
        acc_concers.clear();
        pop_frame_for_tracking();
        liw_lie_make_tf(current_p, current_q, last_keyframe_tf);
        return true;
    }
    void take_back_state() {
        const frame_info& b = *frame_infos.back();
        for (int k = 0; k < 3; ++k) { current_p[k] = b.p[k]; current_q[k] = b.q[k]; current_v[k] = b.v[k]; }
        for (int k = 0; k < 6; ++k) current_bs[k] = b.bs[k];
    }
    void pop_frame(int k) {   // (:488-524) key frames leave through the sink instead of keyframe_manager
        if (k <= 0) return;
        std::deque<frame_info::ptr> tmp_frame_infos;
        for (int i = 0; i < k; i++) {
            frame_info::ptr f = frame_infos.front();
            frame_infos.pop_front();
            if (f->is_key_frame) { if (sink_) sink_(f); tmp_frame_infos.clear(); }
            else tmp_frame_infos.push_back(f);
        }
        if (other_sink_) {
            for (const auto& f : frame_infos) tmp_frame_infos.push_back(f);
            other_sink_(tmp_frame_infos);
        }
        if (last_laser_index > -1) last_laser_index -= k;
        if (current_index > -1) current_index -= k;
    }
    void do_tracking() {   // (:525-560)
        if (status != TRACKING) return;
        liw_record_begin(recorder);
        opt_solver.solve(frame_infos);
        liw_record_end(recorder, "solve");
        take_back_state();
        liw_record_begin(recorder);
        opt_solver.marginalization(frame_infos);
        liw_record_end(recorder, "marginalization");
        pop_frame_for_tracking();
        ++tracked_frames;
        if (o_fstream) liw_tum_append(o_fstream, frame_infos.back()->time, current_p, current_q);
    }
    void pop_frame_for_tracking() {   // (:590-617) keep the last laser frame only (keep_window_size = 1) or the last N frames
        const int n = (int)frame_infos.size();
        int k = n - 1;
        for (int i = n - 1; i > -1; i--)
            if (frame_infos[i]->type == frame_info::laser) { k = i; break; }
        k -= tprm_.keep_window_size - 1;         // keep-N policy (1 = reference)
        pop_frame(k);
        while (laser_manger_.num_keyframes() > 1) laser_manger_.pop_scan();
    }

    liw_params prm_;
    trajectory_params tprm_;
    double T_imu_to_wheel[12], T_imu_to_laser[12];
    imu_preintegraption imu_preintegraption_;
    wheel_odom_preintegration wheel_odom_preintegration_;
    laser_manager laser_manger_;
    std::deque<frame_info::ptr> frame_infos;
    double last_keyframe_tf[12];
    double current_time, last_time;
    int current_index, last_laser_index;
    double current_p[3], current_q[3], current_v[3], current_bs[6], current_angular_local[3];
    TRAJECTORY_STATUS status;
    bool wheel_odom_inited, imu_inited;
    solver opt_solver;
    std::vector<double> acc_concers;
    liw_tum_writer* o_fstream = nullptr;
    liw_record* recorder = nullptr;
    keyframe_sink sink_;
    other_frame_sink other_sink_;
};

// Offline form of lvio_2d::dispatch (dispatch.h:192-257): messages are queued per sensor and handed over oldest-first once
// every queue holds `look_ahead` (40) of them; a message that is not newer than the last dispatched one is dropped; flush()
// drains what the look-ahead holds back when the log ends (the live node never does: it waits for more data).
class dispatch_queue {
public:
    explicit dispatch_queue(trajectory* t, int look_ahead = 40) : t_(t), look_ahead_(look_ahead) {}
    void add(const sensor::imu& m) { imu_.push_back(m); pump(false); }
    void add(const sensor::wheel_odom& m) { wheel_.push_back(m); pump(false); }
    void add(const sensor::laser& m) { laser_.push_back(m); pump(false); }
    void flush() { pump(true); }
    long dispatched = 0, dropped = 0;

private:
    void pump(bool drain) {
        for (;;) {
            if (!drain && ((int)laser_.size() < look_ahead_ || (int)wheel_.size() < look_ahead_ || (int)imu_.size() < look_ahead_)) return;
            // std::map iterates keys in order "imu" < "laser" < "wheel_odom"; an empty queue ends the scan (dispatch.h:217-218)
            int which = -1;
            double oldest = std::numeric_limits<double>::max();
            if (!imu_.empty()) {
                if (imu_.front().time_stamp < oldest) { oldest = imu_.front().time_stamp; which = 0; }
                if (!laser_.empty()) {
                    if (laser_.front().time_stamp < oldest) { oldest = laser_.front().time_stamp; which = 1; }
                    if (!wheel_.empty() && wheel_.front().time_stamp < oldest) { oldest = wheel_.front().time_stamp; which = 2; }
                } else if (drain && !wheel_.empty() && wheel_.front().time_stamp < oldest) { oldest = wheel_.front().time_stamp; which = 2; }
            } else if (drain) {
                if (!laser_.empty() && laser_.front().time_stamp < oldest) { oldest = laser_.front().time_stamp; which = 1; }
                if (!wheel_.empty() && wheel_.front().time_stamp < oldest) { oldest = wheel_.front().time_stamp; which = 2; }
            }
            if (which < 0) return;
            const bool stale = oldest <= last_dispatch_time_;
            if (!stale) last_dispatch_time_ = oldest;
            if (which == 0) { sensor::imu m = imu_.front(); imu_.pop_front(); if (!stale) t_->add_sensor_data(m); }
            else if (which == 1) { sensor::laser m = std::move(laser_.front()); laser_.pop_front(); if (!stale) t_->add_sensor_data(m); }
            else { sensor::wheel_odom m = wheel_.front(); wheel_.pop_front(); if (!stale) t_->add_sensor_data(m); }
            if (stale) ++dropped; else ++dispatched;
        }
    }
    trajectory* t_;
    int look_ahead_;
    std::deque<sensor::imu> imu_;
    std::deque<sensor::wheel_odom> wheel_;
    std::deque<sensor::laser> laser_;
    double last_dispatch_time_ = -std::numeric_limits<double>::max();
};

}  // namespace lvio_2d
