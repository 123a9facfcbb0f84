// lvio_2d_keyframe_manager.hpp — C++ host mirror of the reference's back-end bookkeeping around the pose-graph solve
// (BASELINE config C5), header-only over liw_posegraph.h / liw_lie.h / liw_io.h.  Same member names as
// reference src/trajectory/keyframe_manager.{h,cpp}:
//   add_keyframe / do_add_keyframe     :400-407, :419-482   key-frame queue, tracking poses, sequential edges, corrected pose
//   update_other_frame                 :408-418             current front-end pose carried into the corrected map frame
//   solve                              :722-838             -> liw_posegraph_solve (the MI355X relinearisation)
//   is_time_to_solve                   :839-848             "a loop is pending and 10 s have passed since the last solve"
//   ~keyframe_manager                  :370-397             back_end.txt (TUM) of every key frame
// Out of scope: loop DETECTION (laser_loop_detect, :642-712 and the descriptor code behind it).  A detector is a
// callback here: when key frame `index` arrives it may return an edge (index1 = index, index2 = an older key frame, tf12) — the
// shape laser_loop_detect returns (:664-665, :702).  Two deliberate differences of this offline form: the back-end runs on the
// caller's thread (the reference has its own thread, keyframe_manager.cpp:859-881) and is_time_to_solve compares key-frame
// STAMPS instead of ros::WallTime, so that a replay is deterministic.
#pragma once
#include <cstring>
#include <deque>
#include <functional>
#include <string>
#include <vector>

#include "liw_io.h"
#include "liw_lie.h"
#include "liw_posegraph.h"
#include "lvio_2d_solver.hpp"

namespace lvio_2d {

struct edge {   // reference src/trajectory/keyframe_type.h:12-32
    int index1, index2;
    double tf12[12];   // R (row-major 9) then t
};

struct keyframe_manager_params {
    liw_pg_params pg{};                // loop_sigma_p / _q, loop_edge_k, use_ground_{p,q}_factor (config/office.yaml:106-115)
    double solve_period = 10.0;        // seconds between back-end solves (is_time_to_solve, :843); key-frame time here
    int max_iterations = 0;            // <= 0: Ceres default (50), as :813-818
    bool output_tum = false;
    std::string output_dir;
};

class keyframe_manager {
public:
    // detector(index of the new key frame, the queue so far) -> true and an edge (index1 = index, index2 = older) if a loop closes
    using loop_detector = std::function<bool(int, const std::deque<frame_info::ptr>&, edge*)>;

    keyframe_manager(const liw_params& prm, const keyframe_manager_params& kp) : prm_(prm), kp_(kp), ctx_(liw_create(&prm)) {
        liw_lie_make_tf(zero3_, zero3_, modify_delta_tf);
    }
    ~keyframe_manager() {
        if (kp_.output_tum) write_tum((kp_.output_dir + "back_end.txt").c_str());
        liw_destroy(ctx_);
    }
    keyframe_manager(const keyframe_manager&) = delete;
    keyframe_manager& operator=(const keyframe_manager&) = delete;

    void set_loop_detector(loop_detector d) { detector_ = std::move(d); }

    // add_keyframe (:400-407) + do_add_keyframe (:419-482) in one call: no worker thread in the offline form
    void add_keyframe(const frame_info::ptr& frame_ptr) {
        keyframe_queue.push_back(frame_ptr);
        tfs_tracking.emplace_back();
        double* tr = tfs_tracking.back().v;
        liw_lie_make_tf(frame_ptr->p, frame_ptr->q, tr);
        double corrected[12];
        liw_lie_mul(modify_delta_tf, tr, corrected);
        liw_lie_log_SE3(corrected, frame_ptr->p, frame_ptr->q);
        if (keyframe_queue.size() > 1) {
            const int index1 = (int)keyframe_queue.size() - 2, index2 = index1 + 1;
            edge e{index1, index2, {}};
            double inv1[12];
            liw_lie_inverse(tfs_tracking[index1].v, inv1);
            liw_lie_mul(inv1, tfs_tracking[index2].v, e.tf12);
            seq_edges.push_back(e);
        }
        if (frame_ptr->type == frame_info::laser && detector_) {
            edge lp{};
            if (detector_((int)keyframe_queue.size() - 1, keyframe_queue, &lp)) {
                loop_edges.push_back(lp);
                has_loop_wait_for_solve = true;
                last_loop_index = (int)keyframe_queue.size() - 1;
            }
        }
        double last_frame_tf[12];
        std::memcpy(last_frame_tf, tr, sizeof last_frame_tf);
        if (is_time_to_solve(frame_ptr->time)) {
            last_solve_time = frame_ptr->time;
            solve();
            if (last_status == 0) {
                double current_frame_tf[12], inv_last[12];
                liw_lie_make_tf(frame_ptr->p, frame_ptr->q, current_frame_tf);
                liw_lie_inverse(last_frame_tf, inv_last);
                liw_lie_mul(current_frame_tf, inv_last, modify_delta_tf);   // :468-473
            }
            has_loop_wait_for_solve = false;
            ++solves;
        }
    }

    // update_other_frame (:408-418): the front-end's newest pose expressed in the corrected map frame (what the reference shows)
    void update_other_frame(const std::deque<frame_info::ptr>& frame_infos, double* p3, double* q3) const {
        if (frame_infos.empty()) return;
        double tf[12], cur[12];
        liw_lie_make_tf(frame_infos.back()->p, frame_infos.back()->q, tf);
        liw_lie_mul(modify_delta_tf, tf, cur);
        liw_lie_log_SE3(cur, p3, q3);
    }

    // keyframe_manager::solve (:722-838) on the MI355X
    void solve() {
        const int N = (int)keyframe_queue.size();
        if (N < 2) return;
        std::vector<double> poses((size_t)N * 6);
        for (int i = 0; i < N; ++i) {
            std::memcpy(&poses[(size_t)i * 6], keyframe_queue[i]->p, 24);
            std::memcpy(&poses[(size_t)i * 6 + 3], keyframe_queue[i]->q, 24);
        }
        std::vector<int> si, li;
        std::vector<double> st, lt;
        for (const edge& e : seq_edges) { si.push_back(e.index1); si.push_back(e.index2); st.insert(st.end(), e.tf12, e.tf12 + 12); }
        for (const edge& e : loop_edges) { li.push_back(e.index1); li.push_back(e.index2); lt.insert(lt.end(), e.tf12, e.tf12 + 12); }
        last_status = liw_posegraph_solve(ctx_, &kp_.pg, N, poses.data(), (int)seq_edges.size(), si.data(), st.data(), (int)loop_edges.size(),
                                          li.empty() ? nullptr : li.data(), lt.empty() ? nullptr : lt.data(), kp_.max_iterations, &last_summary);
        if (last_status) return;
        for (int i = 0; i < N; ++i) {
            std::memcpy(keyframe_queue[i]->p, &poses[(size_t)i * 6], 24);
            std::memcpy(keyframe_queue[i]->q, &poses[(size_t)i * 6 + 3], 24);
        }
    }

    bool write_tum(const char* path) const {   // (:370-397) every key frame's base pose, 10 decimals
        liw_tum_writer* w = liw_tum_open(path, &prm_);
        if (!w) return false;
        for (const auto& f : keyframe_queue) liw_tum_append(w, f->time, f->p, f->q);
        liw_tum_close(w);
        return true;
    }

    const char* last_error() const { return liw_last_error(ctx_); }

    std::deque<frame_info::ptr> keyframe_queue;
    struct tf12 { double v[12]; };
    std::vector<tf12> tfs_tracking;
    std::vector<edge> seq_edges, loop_edges;
    double modify_delta_tf[12];
    bool has_loop_wait_for_solve = false;
    int last_loop_index = -1, solves = 0;
    double last_solve_time = -1e300;
    int last_status = 0;
    liw_summary last_summary{};

private:
    bool is_time_to_solve(double time_now) const { return has_loop_wait_for_solve && time_now - last_solve_time > kp_.solve_period; }
    liw_params prm_;
    keyframe_manager_params kp_;
    liw_ctx* ctx_;
    loop_detector detector_;
    double zero3_[3] = {0, 0, 0};
};

}  // namespace lvio_2d
