// lvio_2d_laser.hpp — C++ host mirror of the reference's laser front-end classes, header-only over the C ABI of
// liw_laser.h.  Same class / method names as the reference so that lvio_2d::trajectory code reads the same:
//   lvio_2d::scan            reference src/trajectory/laser_type.h:23-61   (lines, concers; the grid stays in the library)
//   lvio_2d::laser_manager   reference src/trajectory/laser_manager.h:9-49 (spawn_scan, add_scan, match_with_front /
//                            _back / _ref, do_match, pop_scan, clear_all_scan)
// The matches come back as lvio_2d::laser_match of lvio_2d_solver.hpp, i.e. directly what frame_info::add_laser_match
// and lvio_2d::solver consume.
#pragma once
#include <memory>
#include <vector>

#include "liw_laser.h"
#include "lvio_2d_solver.hpp"

namespace lvio_2d {

struct scan {
    using ptr = std::shared_ptr<scan>;
    double time = 0;
    std::vector<line> lines;                 // p1, p2 (abc / len: lines_abc_len)
    std::vector<double> lines_abc_len;       // [lines.size()][4]
    std::vector<double> concers;             // [k][3]
    liw_scan* handle = nullptr;
    ~scan() { liw_scan_destroy(handle); }
    scan() = default;
    scan(const scan&) = delete;
    scan& operator=(const scan&) = delete;
    void refresh() {
        const int n = liw_scan_num_lines(handle);
        std::vector<double> raw((size_t)n * 10);
        if (n) liw_scan_get_lines(handle, raw.data());
        lines.resize(n);
        lines_abc_len.resize((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 3; ++k) { lines[i].p1[k] = raw[i * 10 + k]; lines[i].p2[k] = raw[i * 10 + 3 + k]; }
            for (int k = 0; k < 4; ++k) lines_abc_len[i * 4 + k] = raw[i * 10 + 6 + k];
        }
        concers.resize((size_t)liw_scan_num_concers(handle) * 3);
        if (!concers.empty()) liw_scan_get_concers(handle, concers.data());
    }
};

class laser_manager {
public:
    explicit laser_manager(const liw_laser_params& prm) : prm_(prm), h_(liw_laser_manager_create(&prm)) {}
    ~laser_manager() { liw_laser_manager_destroy(h_); }
    laser_manager(const laser_manager&) = delete;
    laser_manager& operator=(const laser_manager&) = delete;

    // points: [n][3] in the laser frame (already de-skewed, as the reference assumes)
    scan::ptr spawn_scan(const double* points, int n, double time) {
        auto s = std::make_shared<scan>();
        s->time = time;
        s->handle = liw_scan_spawn(&prm_, points, n, time);
        s->refresh();
        return s;
    }
    void add_scan(const scan::ptr& s, const double* current_p, const double* current_q) { liw_laser_manager_add_scan(h_, s->handle, current_p, current_q); }
    laser_match::ptr match_with_front(const scan::ptr& s, const double* p, const double* q) { return take(liw_laser_manager_match_with_front(h_, s->handle, p, q), s); }
    laser_match::ptr match_with_back(const scan::ptr& s, const double* p, const double* q) { return take(liw_laser_manager_match_with_back(h_, s->handle, p, q), s); }
    laser_match::ptr match_with_ref(const scan::ptr& s, const double* p, const double* q) { return take(liw_laser_manager_match_with_ref(h_, s->handle, p, q), s); }
    static laser_match::ptr do_match(const liw_laser_params& prm, const scan::ptr& scan1, const scan::ptr& scan2, const double* p1, const double* q1,
                                     const double* p2, const double* q2, int kk = 0) {
        return take(liw_laser_do_match(&prm, scan1->handle, scan2->handle, p1, q1, p2, q2, kk), scan2);
    }
    bool pop_scan() { return liw_laser_manager_pop_scan(h_) != 0; }
    void clear_all_scan() { liw_laser_manager_clear_all_scan(h_); }
    int num_keyframes() const { return liw_laser_manager_num_keyframes(h_); }
    // get_keyframs()[i]->current_p / current_q = ... (trajectory.cpp:452-461)
    void set_keyframe_pose(int i, const double* p, const double* q) { liw_laser_manager_set_keyframe_pose(h_, i, p, q); }

private:
    static laser_match::ptr take(liw_laser_match* m, const scan::ptr& scan2) {
        auto r = std::make_shared<laser_match>();
        r->scan2 = scan2;
        const int n = liw_laser_match_size(m);
        std::vector<double> pts((size_t)n * 12);
        double pose[12];
        liw_laser_match_get(m, pts.data(), pose, nullptr, nullptr);
        r->lines1.resize(n);
        r->lines2.resize(n);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                r->lines1[i].p1[k] = pts[i * 12 + k]; r->lines1[i].p2[k] = pts[i * 12 + 3 + k];
                r->lines2[i].p1[k] = pts[i * 12 + 6 + k]; r->lines2[i].p2[k] = pts[i * 12 + 9 + k];
            }
        for (int k = 0; k < 3; ++k) { r->p1[k] = pose[k]; r->q1[k] = pose[3 + k]; r->p2[k] = pose[6 + k]; r->q2[k] = pose[9 + k]; }
        liw_laser_match_destroy(m);
        return r;
    }
    liw_laser_params prm_;
    liw_laser_manager* h_;
};

}  // namespace lvio_2d
