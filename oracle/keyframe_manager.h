// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of the back-end bookkeeping around keyframe_manager::solve (src/trajectory/keyframe_manager.cpp):
//   do_add_keyframe        :419-482  queue, tfs_tracking, pose <- modify_delta_tf * tracking pose, sequential edge tf1^-1 tf2,
//                                    solve when is_time_to_solve, modify_delta_tf <- current * last^-1
//   update_other_frame     :408-418  modify_delta_tf * front-end pose
//   is_time_to_solve       :839-848  loop pending and > 10 s since the last solve — compared on key-frame STAMPS here (offline replay
//                                    must be deterministic; the reference uses ros::WallTime)
// Loop detection (:642-712) is out of scope: loop edges come from a schedule (trigger key-frame index -> older index, tf12).
#pragma once
#include <deque>
#include <vector>

#include "posegraph.h"

namespace oracle {

struct backend_loop { int trigger, older; Iso3<double> tf12; };

class keyframe_manager {
public:
    struct keyframe { double time; Vec3<double> p, q; bool is_laser; };
    keyframe_manager(const params* prm_, const pg_params& P_, double solve_period_, int max_iterations_)
        : prm(prm_), P(P_), solve_period(solve_period_), max_iterations(max_iterations_) {}

    void add_keyframe(double time, const Vec3<double>& p, const Vec3<double>& q, bool is_laser) {
        keyframe kf{time, p, q, is_laser};
        tfs_tracking.push_back(lie::make_tf(p, q));
        lie::log_SE3<double>(modify_delta_tf * tfs_tracking.back(), kf.p, kf.q);
        keyframe_queue.push_back(kf);
        if (keyframe_queue.size() > 1) {
            int index1 = (int)keyframe_queue.size() - 2, index2 = (int)keyframe_queue.size() - 1;
            seq_idx.emplace_back(index1, index2);
            seq_tf.push_back(tfs_tracking[index1].inverse() * tfs_tracking[index2]);
        }
        if (is_laser) {
            const int idx = (int)keyframe_queue.size() - 1;
            for (const backend_loop& l : schedule)
                if (l.trigger == idx) {
                    loop_idx.emplace_back(idx, l.older);
                    loop_tf.push_back(l.tf12);
                    has_loop_wait_for_solve = true;
                    last_loop_index = idx;
                    break;
                }
        }
        Iso3<double> last_frame_tf = tfs_tracking.back();
        if (has_loop_wait_for_solve && time - last_solve_time > solve_period) {
            last_solve_time = time;
            solve();
            Iso3<double> current_frame_tf = lie::make_tf(keyframe_queue.back().p, keyframe_queue.back().q);
            modify_delta_tf = current_frame_tf * last_frame_tf.inverse();
            has_loop_wait_for_solve = false;
            ++solves;
        }
    }
    void update_other_frame(const Vec3<double>& p, const Vec3<double>& q, Vec3<double>& po, Vec3<double>& qo) const {
        lie::log_SE3<double>(modify_delta_tf * lie::make_tf(p, q), po, qo);
    }
    void solve() {
        const int N = (int)keyframe_queue.size();
        if (N < 2) return;
        std::vector<double> poses((size_t)N * 6);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 3; ++k) { poses[i * 6 + k] = keyframe_queue[i].p(k); poses[i * 6 + 3 + k] = keyframe_queue[i].q(k); }
        keyframe_manager_solve(prm, P, N, poses.data(), seq_idx, seq_tf, loop_idx, loop_tf, max_iterations, &last_summary);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 3; ++k) { keyframe_queue[i].p(k) = poses[i * 6 + k]; keyframe_queue[i].q(k) = poses[i * 6 + 3 + k]; }
    }

    const params* prm;
    pg_params P;
    double solve_period;
    int max_iterations;
    std::vector<backend_loop> schedule;
    std::deque<keyframe> keyframe_queue;
    std::vector<Iso3<double>> tfs_tracking;
    std::vector<std::pair<int, int>> seq_idx, loop_idx;
    std::vector<Iso3<double>> seq_tf, loop_tf;
    Iso3<double> modify_delta_tf;
    bool has_loop_wait_for_solve = false;
    int last_loop_index = -1, solves = 0;
    double last_solve_time = -1e300;
    miniceres::Summary last_summary;
};

}  // namespace oracle
