// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of lvio_2d::trajectory (src/trajectory/trajectory.h:16-84, trajectory.cpp) for the camera-less
// configuration (config/office.yaml:6) on top of the other restatements (solver.h, preint.h, laser_frontend.h,
// io_formats.h), and of the oldest-first merge of lvio_2d::dispatch (src/trajectory/dispatch.h:192-257) in offline form.
// ROS transport, visualisation, camera branches and keyframe_manager are out of scope (popped key frames are counted).
#pragma once
#include <deque>
#include <limits>
#include <string>

#include "io_formats.h"
#include "keyframe_manager.h"
#include "laser_frontend.h"
#include "solver.h"

namespace oracle {

struct trajectory_params {
    int slide_window_size = 10;
    double p_motion_threshold = 0.1, q_motion_threshold = 0.05;
    double key_frame_p_motion_threshold = 0.05, key_frame_q_motion_threshold = 0.05;
    double min_delta_t = 0.001;
    // frames kept in the window after a tracking solve.  1 = the reference (pop_frame_for_tracking keeps the newest laser frame only,
    // trajectory.cpp:590-617, so every tracking solve sees 2 frames); N keeps N, i.e. TRACK-topology solves on N + 1 frames — the
    // explicit keep-N policy SURVEY 8 f3 asks for (BASELINE configs C3 / C5: 30 / 50-key-frame windows)
    int keep_window_size = 1;
};
struct laser_msg { double time_stamp; std::vector<Vec3<double>> points; std::vector<double> times; };

class trajectory {
public:
    enum TRAJECTORY_STATUS { INITIALIZING = 0, TRACKING = 1 };
    struct frame {   // frame_info (trajectory_type.h:9-75): the solver's part + what only the trajectory reads
        frame_info::ptr f;
        double time;
        bool is_key_frame = false;
        lfe::laser_match_lines::ptr match;
    };

    trajectory(const params* prm_, const laser_params* lprm_, const trajectory_params& tp)
        : prm(prm_), lprm(lprm_), tprm(tp), imu_preintegraption_(prm_), wheel_odom_preintegration_(prm_), laser_manger_(lprm_), opt_solver(prm_) {
        init_current_status();
        wheel_odom_inited = imu_inited = false;
        current_index = last_laser_index = -1;
        tum = tum_header();
    }

    void add_sensor_data(const wheel_sample& d) { if (wheel_odom_preintegration_.add_wheel_odom_measure(d)) wheel_odom_inited = true; }
    void add_sensor_data(const imu_sample& d) { if (imu_preintegraption_.add_imu_measure(d)) imu_inited = true; }
    void add_sensor_data(laser_msg& laser_data) {
        double time = laser_data.time_stamp;
        if (status == TRACKING) {
            Iso3<double> T_w_laser = lie::make_tf(current_p, current_q) * prm->T_imu_to_laser;
            Mat3<double> R_w_laser = T_w_laser.R, R_i_l = prm->T_imu_to_laser.R;
            Vec3<double> tmp_angular = lie::log_SO3<double>(R_i_l.transpose() * lie::exp_so3(current_angular_local) * R_i_l);
            laser_correct(laser_data.points, laser_data.times, laser_data.time_stamp, R_w_laser.transpose() * current_v, tmp_angular);
        }
        if (!imu_inited) return;
        if (!wheel_odom_inited) return;
        auto wheel_result_filter = wheel_odom_preintegration_.get_preintegraption_result();
        auto laser_delta_filter = wheel_delta_to_laser_delta(wheel_result_filter.delta_Tij);
        if (status == INITIALIZING && is_static(laser_delta_filter, tprm.p_motion_threshold, tprm.q_motion_threshold)) return;
        if (status == TRACKING && imu_preintegraption_.Dt < tprm.min_delta_t) return;
        current_index++;
        wheel_odom_preintegration_.update_only_t(time);
        imu_preintegraption_.update_only_t(time);
        auto wheel_result = std::make_shared<wheel_odom_preint_result>(wheel_odom_preintegration_.get_preintegraption_result());
        auto imu_reuslt = std::make_shared<imu_preint_result>(imu_preintegraption_.get_preintegraption_result());
        wheel_odom_preintegration_.reset_wheel_odom_measure(time);
        imu_preintegraption_.reset_imu_measure(time, current_bs, current_bs + 3);
        for (int k = 0; k < 3; ++k) current_angular_local(k) = imu_reuslt->X[gamma_index + k] / imu_reuslt->Dt;
        Iso3<double> delta_tf = wheel_delta_to_imu_delta(wheel_result->delta_Tij);
        update_current_status(delta_tf, time);

        auto scan_ptr = laser_manger_.spawn_scan(laser_data.points, laser_data.times.empty() ? time : laser_data.times.front());
        rec.add_record("lines each frame", scan_ptr->lines.size());
        lfe::laser_match_lines::ptr lm = nullptr;
        if (status == INITIALIZING) {
            lm = laser_manger_.match_with_front(scan_ptr, current_p, current_q);
            laser_manger_.add_scan(scan_ptr, current_p, current_q);
        } else {
            lm = laser_manger_.match_with_ref(scan_ptr, current_p, current_q);
        }
        frame fr;
        fr.f = std::make_shared<frame_info>();
        for (int k = 0; k < 3; ++k) { fr.f->p[k] = current_p(k); fr.f->q[k] = current_q(k); fr.f->v[k] = current_v(k); }
        for (int k = 0; k < 6; ++k) fr.f->bs[k] = current_bs[k];
        fr.f->imu_observation_reslut = imu_reuslt;
        fr.f->wheel_observation_reslut = wheel_result;
        for (int k = 0; k < 36; ++k) fr.f->sqrt_H[k] = (k % 7 == 0) ? 1.0 : 0.0;
        fr.time = current_time;
        fr.match = lm;
        // add_laser_match: the solver's view of the match
        fr.f->laser_match_ptr = std::make_shared<laser_match>();
        for (size_t i = 0; i < lm->lines1.size(); ++i) {
            fr.f->laser_match_ptr->lines1.push_back(line{lm->lines1[i]->p1, lm->lines1[i]->p2});
            fr.f->laser_match_ptr->lines2.push_back(line{lm->lines2[i]->p1, lm->lines2[i]->p2});
        }
        for (int k = 0; k < 3; ++k) {
            fr.f->laser_match_ptr->p1[k] = lm->p1(k); fr.f->laser_match_ptr->q1[k] = lm->q1(k);
            fr.f->laser_match_ptr->p2[k] = lm->p2(k); fr.f->laser_match_ptr->q2[k] = lm->q2(k);
        }
        fr.f->type = frame_info::laser;
        last_laser_index = current_index;
        frame_infos.push_back(fr);
        if (status == INITIALIZING) {
            if (check_and_processing_initialize()) status = TRACKING;
            return;
        }
        do_tracking();
        {
            Iso3<double> tf_w_l = lie::make_tf(current_p, current_q) * prm->T_imu_to_laser;
            for (size_t i = 0; i < lm->scan2->concers.size(); i++) acc_concers.push_back(tf_w_l * lm->scan2->concers[i]);
        }
        Iso3<double> current_laser_tf = lie::make_tf(current_p, current_q) * prm->T_imu_to_laser;
        Iso3<double> delta_laser_tf = last_keyframe_tf.inverse() * current_laser_tf;
        int n_match_size = 0;
        if (lm) n_match_size = (int)lm->lines1.size();
        rec.add_record("match line size", n_match_size);
        int n_no_match_size = (int)scan_ptr->lines.size() - n_match_size;
        if (!is_static(delta_laser_tf, tprm.key_frame_p_motion_threshold, tprm.key_frame_q_motion_threshold) || n_match_size < n_no_match_size) {
            rec.add_record("corner each keyframe", acc_concers.size());
            frame_infos.back().is_key_frame = true;
            acc_concers.clear();
            last_keyframe_tf = current_laser_tf;
        }
        laser_manger_.add_scan(scan_ptr, current_p, current_q);
        last_time = current_time;
    }

    // state (public: test infrastructure)
    const params* prm;
    const laser_params* lprm;
    trajectory_params tprm;
    imu_preintegraption imu_preintegraption_;
    wheel_odom_preintegration wheel_odom_preintegration_;
    lfe::laser_manager laser_manger_;
    std::deque<frame> frame_infos;
    Iso3<double> last_keyframe_tf;
    double current_time, last_time;
    int current_index, last_laser_index;
    Vec3<double> current_p, current_q, current_v, current_angular_local;
    double current_bs[6];
    TRAJECTORY_STATUS status;
    bool wheel_odom_inited, imu_inited;
    solver opt_solver;
    std::vector<Vec3<double>> acc_concers;
    std::string tum;
    record rec;
    int tracked_frames = 0, initializations = 0, keyframes_out = 0;
    // teacher-forcing capture (tests): the exact input of every tracking solve (flat window + prior) and its outputs, so that the
    // product can be run on the oracle's own windows one solve at a time (a free-running replay is chaotic beyond ~10 frames)
    struct capture_rec {
        int n = 0, L = 0, has_prior = 0, iterations = 0, termination = 0;
        std::vector<double> states, laser_pts, match_pose, imu_X, imu_J, imu_P, imu_Dt, wheel_T, wheel_P, wheel_Dt;
        std::vector<int> laser_frame;
        std::vector<unsigned char> has_match;
        std::vector<double> prior_X, prior_J, prior_R, states_after, match_after, Delta_H, Delta_g, post_X, post_J, post_R;
    };
    bool capture = false;
    std::vector<capture_rec> captures;
    keyframe_manager* backend = nullptr;      // optional back-end (BASELINE C5); not owned
    Vec3<double> backend_p, backend_q;        // newest front-end pose in the corrected map frame (update_other_frame)

private:
    static bool is_static(const Iso3<double>& delta_tf, double p_thr, double q_thr) {
        Vec3<double> dp, dq;
        lie::log_SE3(delta_tf, dp, dq);
        return norm(dp) < p_thr && norm(dq) < q_thr;
    }
    Iso3<double> wheel_delta_to_imu_delta(const Iso3<double>& wheel_delta) const { return prm->T_imu_to_wheel * wheel_delta * prm->T_imu_to_wheel.inverse(); }
    Iso3<double> wheel_delta_to_laser_delta(const Iso3<double>& wheel_delta) const {
        auto T_laser_to_wheel = prm->T_imu_to_laser.inverse() * prm->T_imu_to_wheel;
        return T_laser_to_wheel * wheel_delta * T_laser_to_wheel.inverse();
    }
    void init_current_status() {
        status = INITIALIZING;
        last_keyframe_tf = Iso3<double>();
        last_time = current_time = -std::numeric_limits<double>::max();
        lie::log_SE3(prm->T_imu_to_wheel.inverse(), current_p, current_q);
        current_v = Vec3<double>(); current_angular_local = Vec3<double>();
        for (int k = 0; k < 6; ++k) current_bs[k] = 0.0;
    }
    void update_current_status(const Iso3<double>& delta_tf, double time) {
        Iso3<double> new_tf = lie::make_tf(current_p, current_q) * delta_tf;
        lie::log_SE3(new_tf, current_p, current_q);
        current_time = time;
    }
    solver::frames solver_frames() const {
        solver::frames fi;
        for (const auto& fr : frame_infos) fi.push_back(fr.f);
        return fi;
    }
    static void capture_states(const solver::frames& fi, std::vector<double>& st, std::vector<double>& mp) {
        st.clear(); mp.clear();
        for (const auto& f : fi) {
            st.insert(st.end(), f->p, f->p + 3); st.insert(st.end(), f->q, f->q + 3); st.insert(st.end(), f->v, f->v + 3); st.insert(st.end(), f->bs, f->bs + 6);
            double m[12] = {0};
            if (f->laser_match_ptr) for (int k = 0; k < 3; ++k) { m[k] = f->laser_match_ptr->p1[k]; m[3 + k] = f->laser_match_ptr->q1[k]; m[6 + k] = f->laser_match_ptr->p2[k]; m[9 + k] = f->laser_match_ptr->q2[k]; }
            mp.insert(mp.end(), m, m + 12);
        }
    }
    void capture_window(const solver::frames& fi, capture_rec& c) const {
        c.n = (int)fi.size();
        capture_states(fi, c.states, c.match_pose);
        for (int i = 0; i < c.n; ++i) {
            const frame_info& f = *fi[i];
            const bool m = f.type == frame_info::laser && f.laser_match_ptr;
            c.has_match.push_back(m ? 1 : 0);
            if (m)
                for (size_t j = 0; j < f.laser_match_ptr->lines1.size(); ++j) {
                    c.laser_frame.push_back(i);
                    const line& a = f.laser_match_ptr->lines1[j]; const line& b = f.laser_match_ptr->lines2[j];
                    for (int k = 0; k < 3; ++k) c.laser_pts.push_back(a.p1(k));
                    for (int k = 0; k < 3; ++k) c.laser_pts.push_back(a.p2(k));
                    for (int k = 0; k < 3; ++k) c.laser_pts.push_back(b.p1(k));
                    for (int k = 0; k < 3; ++k) c.laser_pts.push_back(b.p2(k));
                }
            if (i > 0) {
                const imu_preint_result& r = *f.imu_observation_reslut;
                c.imu_X.insert(c.imu_X.end(), r.X, r.X + 15); c.imu_J.insert(c.imu_J.end(), &r.J[0][0], &r.J[0][0] + 225);
                c.imu_P.insert(c.imu_P.end(), &r.sqrt_inverse_P[0][0], &r.sqrt_inverse_P[0][0] + 225); c.imu_Dt.push_back(r.Dt);
                const wheel_odom_preint_result& w = *f.wheel_observation_reslut;
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) c.wheel_T.push_back(w.delta_Tij.R(a, b));
                for (int a = 0; a < 3; ++a) c.wheel_T.push_back(w.delta_Tij.t(a));
                c.wheel_P.insert(c.wheel_P.end(), &w.sqrt_inverse_P[0][0], &w.sqrt_inverse_P[0][0] + 9); c.wheel_Dt.push_back(w.Dt);
            }
        }
        c.L = (int)c.laser_frame.size();
        c.has_prior = opt_solver.has_linearized_block ? 1 : 0;
        if (c.has_prior) { c.prior_X = opt_solver.linearized_X; c.prior_J = opt_solver.linearized_jacobians.d; c.prior_R = opt_solver.linearized_residuals; }
    }
    void take_back_state() {
        const frame_info& b = *frame_infos.back().f;
        for (int k = 0; k < 3; ++k) { current_p(k) = b.p[k]; current_q(k) = b.q[k]; current_v(k) = b.v[k]; }
        for (int k = 0; k < 6; ++k) current_bs[k] = b.bs[k];
    }
    bool check_and_processing_initialize() {
        if ((int)frame_infos.size() < tprm.slide_window_size) return false;
        int k = 0;
        bool is_first_laser = true;
        for (size_t i = 0; i < frame_infos.size(); i++) {
            if (frame_infos[i].f->type == frame_info::laser) {
                if (is_first_laser) { is_first_laser = false; continue; }
                if (frame_infos[i].f->laser_match_ptr == nullptr) { k = (int)i + 1; break; }
                if (frame_infos[i].f->laser_match_ptr->lines2.size() < 2) { k = (int)i + 1; break; }
            }
        }
        if (k > 0) {
            pop_frame((int)frame_infos.size());
            laser_manger_.clear_all_scan();
            init_current_status();
            return false;
        }
        ++initializations;
        solver::frames fi = solver_frames();
        opt_solver.init_solve(fi);
        int n_laser = 0;
        auto& key_frame = laser_manger_.key_frame;
        for (size_t i = 0; i < frame_infos.size(); i++)
            if (frame_infos[i].f->type == frame_info::laser) {
                key_frame[n_laser]->current_p = Vec3<double>(frame_infos[i].f->p[0], frame_infos[i].f->p[1], frame_infos[i].f->p[2]);
                key_frame[n_laser]->current_q = Vec3<double>(frame_infos[i].f->q[0], frame_infos[i].f->q[1], frame_infos[i].f->q[2]);
                n_laser++;
            }
        take_back_state();
        laser_manger_.clear_all_scan();
        for (size_t i = 0; i < frame_infos.size(); i++)
            if (frame_infos[i].f->type == frame_info::laser && frame_infos[i].match && frame_infos[i].match->scan2)
                laser_manger_.add_scan(frame_infos[i].match->scan2, Vec3<double>(frame_infos[i].f->p[0], frame_infos[i].f->p[1], frame_infos[i].f->p[2]),
                                       Vec3<double>(frame_infos[i].f->q[0], frame_infos[i].f->q[1], frame_infos[i].f->q[2]));
        opt_solver.marginalization(fi);
        acc_concers.clear();
        pop_frame_for_tracking();
        last_keyframe_tf = lie::make_tf(current_p, current_q);
        return true;
    }
    void pop_frame(int k) {   // trajectory.cpp:488-524: popped key frames go to the back-end, which then sees the newest pose
        if (k <= 0) return;
        for (int i = 0; i < k; i++) {
            if (frame_infos.front().is_key_frame) {
                ++keyframes_out;
                const frame_info& f = *frame_infos.front().f;
                if (backend) backend->add_keyframe(frame_infos.front().time, Vec3<double>(f.p[0], f.p[1], f.p[2]), Vec3<double>(f.q[0], f.q[1], f.q[2]), f.type == frame_info::laser);
            }
            frame_infos.pop_front();
        }
        if (backend && !frame_infos.empty()) {
            const frame_info& b = *frame_infos.back().f;
            backend->update_other_frame(Vec3<double>(b.p[0], b.p[1], b.p[2]), Vec3<double>(b.q[0], b.q[1], b.q[2]), backend_p, backend_q);
        }
        if (last_laser_index > -1) last_laser_index -= k;
        if (current_index > -1) current_index -= k;
    }
    void do_tracking() {
        if (status != TRACKING) return;
        solver::frames fi = solver_frames();
        if (capture) { captures.emplace_back(); capture_window(fi, captures.back()); }
        opt_solver.solve(fi);
        take_back_state();
        if (capture) {
            capture_rec& c = captures.back();
            capture_states(fi, c.states_after, c.match_after);
            c.iterations = opt_solver.last_summary.num_iterations; c.termination = opt_solver.last_summary.termination;
        }
        opt_solver.marginalization(fi);
        if (capture && !prm->fast_mode) {
            capture_rec& c = captures.back();
            c.Delta_H = opt_solver.Delta_H.d; c.Delta_g = opt_solver.Delta_g;
            c.post_X = opt_solver.linearized_X; c.post_J = opt_solver.linearized_jacobians.d; c.post_R = opt_solver.linearized_residuals;
        }
        pop_frame_for_tracking();
        ++tracked_frames;
        tum += tum_line(prm->T_imu_to_wheel, frame_infos.back().time, current_p, current_q);
    }
    void pop_frame_for_tracking() {
        int n = (int)frame_infos.size();
        int k = n - 1;
        for (int i = n - 1; i > -1; i--)
            if (frame_infos[i].f->type == frame_info::laser) { k = i; break; }
        k -= tprm.keep_window_size - 1;          // keep-N policy (1 = reference)
        pop_frame(k);
        while (laser_manger_.key_frame.size() > 1) laser_manger_.pop_scan();
    }
};

}  // namespace oracle
