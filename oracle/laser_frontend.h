// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of the reference's 2D laser front-end, same names and control flow:
//   helper functions / line / scan::add_line   src/trajectory/laser_manager.cpp:6-223, laser_type.h:13-61
//   laser_manager::{do_match, spawn_scan, add_scan, match_with_*, pop_scan, clear_all_scan}   :226-565
//   convert::laser_to_point_times               src/utilies/common.cpp:5-40
//   sensor::laser::correct                      src/trajectory/sensor.h:51-94
// Eigen::JacobiSVD(A).matrixV().col(2) of the n x 3 design matrix is restated as a one-sided (Hestenes) Jacobi SVD;
// the dense grid my_2d_vec<std::vector<line::ptr>> (src/utilies/my_struct.h) as an ordered map of occupied cells.
#pragma once
#include <cmath>
#include <deque>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "lie.h"

namespace oracle {

struct laser_params {   // the laser part of param::manager (src/utilies/params.h; config/office.yaml:78-122)
    double w_laser_each_scan = 100.0, h_laser_each_scan = 100.0, laser_resolution = 0.05;
    double line_continuous_threshold = 0.1, line_min_len = 0.05, line_max_dis = 0.03, line_max_tolerance_angle = 175.0;
    double ref_motion_filter_p = 0.01, ref_motion_filter_q = 0.01;
    int ref_n_accumulation = 2;
    Iso3<double> T_imu_to_laser;
};

namespace lf {
typedef Vec3<double> V;
constexpr double epsilo = 0.0008;
inline double angle_to_rad(double a) { return a / 180.0 * M_PI; }
inline double rad_to_angle(double a) { return a / M_PI * 180.0; }

inline V project_to_line(const V& p, const V& start_point, const V& end_point) {
    if (norm(end_point - start_point) < epsilo) return p;
    V se_unit_vec = normalized(end_point - start_point);
    V sp_vec = p - start_point;
    double line_project_norm = dot(sp_vec, se_unit_vec);
    return start_point + line_project_norm * se_unit_vec;
}

// right singular vector of the smallest singular value of A (num x 3), one-sided Jacobi on the columns
inline V smallest_right_singular_vector(std::vector<double>& A, int num) {
    double Vm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 100; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double app = 0, aqq = 0, apq = 0;
                for (int i = 0; i < num; ++i) { app += A[i * 3 + p] * A[i * 3 + p]; aqq += A[i * 3 + q] * A[i * 3 + q]; apq += A[i * 3 + p] * A[i * 3 + q]; }
                if (std::fabs(apq) <= 1e-18 * std::sqrt(app * aqq) || apq == 0.0) continue;
                rotated = true;
                const double zeta = (aqq - app) / (2.0 * apq);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < num; ++i) {
                    const double ap = A[i * 3 + p], aq = A[i * 3 + q];
                    A[i * 3 + p] = c * ap - s * aq; A[i * 3 + q] = s * ap + c * aq;
                }
                for (int i = 0; i < 3; ++i) {
                    const double vp = Vm[i][p], vq = Vm[i][q];
                    Vm[i][p] = c * vp - s * vq; Vm[i][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sv[3];
    for (int j = 0; j < 3; ++j) { double s = 0; for (int i = 0; i < num; ++i) s += A[i * 3 + j] * A[i * 3 + j]; sv[j] = s; }
    int m = 0;
    if (sv[1] < sv[m]) m = 1;
    if (sv[2] < sv[m]) m = 2;
    return V(Vm[0][m], Vm[1][m], Vm[2][m]);
}
inline V fit_line_by_least_square(const std::vector<V>& points, int index1, int index2) {
    int num = index2 - index1 + 1;
    std::vector<double> A((size_t)num * 3);
    for (int i = index1; i <= index2; i++) { A[(i - index1) * 3] = points[i](0); A[(i - index1) * 3 + 1] = points[i](1); A[(i - index1) * 3 + 2] = 1; }
    return smallest_right_singular_vector(A, num);
}
inline std::tuple<V, V, V, double> create_line(const std::vector<V>& points, const V& line_abc, int index1, int index2) {
    const V& ret = line_abc;
    V point1(0, 0, 0), point2(0, 0, 0);
    if (std::fabs(ret(1)) < 0.5) {
        point1(1) = 0; point1(0) = -ret(2) / ret(0);
        point2(1) = 1; point2(0) = (-ret(2) - ret(1)) / ret(0);
    } else {
        point1(0) = 0; point2(0) = 1;
        point1(1) = -ret(2) / ret(1);
        point2(1) = (-ret(2) - ret(0)) / ret(1);
    }
    double max_dis = 0;
    for (int i = index1; i <= index2; i++) {
        double tmp_error = e_laser::dis_from_line(points[i], point1, point2);
        if (tmp_error > max_dis) max_dis = tmp_error;
    }
    return {project_to_line(points[index1], point1, point2), project_to_line(points[index2], point1, point2), ret, max_dis};
}
inline bool is_continuous(const laser_params& P, const V& a, const V& b) { return norm(a - b) <= P.line_continuous_threshold; }
inline double clac_cos(const V& point_j, const V& point_i, const V& point_k) {
    if (norm(point_i - point_j) < epsilo) return -1;
    if (norm(point_j - point_k) < epsilo) return -1;
    V ji = normalized(point_i - point_j), jk = normalized(point_k - point_j);
    return dot(ji, jk);
}
inline double clac_angle(const V& j, const V& i, const V& k) { return std::acos(clac_cos(j, i, k)); }
}  // namespace lf

// the classes below live in oracle::lfe: solver.h has its own plain `line` / `laser_match` (the solver's view of them)
namespace lfe {

struct line {
    using ptr = std::shared_ptr<line>;
    Vec3<double> p1, p2, abc;
    double len;
    line(const Vec3<double>& p1_, const Vec3<double>& p2_, const Vec3<double>& abc_) : p1(p1_), p2(p2_), abc(abc_) { len = norm(p1 - p2); }
};

inline std::tuple<double, Vec3<double>> calc_angle_and_intersection(const line::ptr& l1, const line::ptr& l2) {
    Vec3<double> v1 = l1->p1 - l1->p2, v2 = l2->p1 - l2->p2;
    double angle = std::acos(dot(normalized(v1), normalized(v2)));
    // 2x2 solve A x = b (JacobiSVD::solve in the reference; the system is well conditioned whenever the result is used):
    // Gaussian elimination with partial pivoting
    double a00 = l1->abc(0), a01 = l1->abc(1), a10 = l2->abc(0), a11 = l2->abc(1), b0 = -l1->abc(2), b1 = -l2->abc(2);
    if (std::fabs(a10) > std::fabs(a00)) { std::swap(a00, a10); std::swap(a01, a11); std::swap(b0, b1); }
    const double f = a10 / a00;
    a11 -= f * a01; b1 -= f * b0;
    const double y = b1 / a11, x = (b0 - a01 * y) / a00;
    return {angle, Vec3<double>(x, y, 0)};
}

struct scan {
    using ptr = std::shared_ptr<scan>;
    const laser_params* P;
    double time;
    int w, h;
    double resolution;
    std::map<std::pair<int, int>, std::vector<line::ptr>> line_map;   // (r, c) -> lines, only touched cells exist
    std::vector<line::ptr> lines;
    std::vector<Vec3<double>> concers;
    bool is_index_valid(int r, int c) const { return r >= 0 && r < h && c >= 0 && c < w; }
    std::tuple<int, int> xy_to_index(double x, double y) const { return {(int)(x / resolution + w / 2), (int)(y / resolution + h / 2)}; }
    scan(const laser_params* P_, int w_, int h_, double resolution_, double time_) : P(P_), time(time_), w(w_), h(h_), resolution(resolution_) {}

    void add_line(const std::vector<Vec3<double>>& points, int index1, int index2, bool add_concers = true) {
        if (index2 - index1 < 2) return;
        auto line_abc = lf::fit_line_by_least_square(points, index1, index2);
        auto [p1, p2, abc, error] = lf::create_line(points, line_abc, index1, index2);
        line::ptr l = std::make_shared<line>(p1, p2, abc);
        double len = norm(p1 - p2);
        if (error > P->line_max_dis) return;
        if (len < P->line_min_len) return;
        if (add_concers) {
            for (int i = index1; i <= index2; i++) {
                auto [c, r] = xy_to_index(points[i](0), points[i](1));
                if (is_index_valid(r, c)) {
                    auto& cell = line_map[{r, c}];
                    if (cell.empty()) {
                        cell.push_back(l);
                        if (lines.empty() || lines.back() != l) lines.push_back(l);
                    } else if (l != cell.back()) {
                        cell.push_back(l);
                        if (lines.empty() || lines.back() != l) lines.push_back(l);
                    } else
                        continue;
                    if (cell.size() == 2) {
                        if (cell[0]->len > 0.1 && cell[1]->len > 0.1) {
                            auto [angle, inter] = calc_angle_and_intersection(cell[0], cell[1]);
                            if (angle < lf::angle_to_rad(150) && angle > lf::angle_to_rad(30)) {
                                auto [concer_c, concer_r] = xy_to_index(inter(0), inter(1));
                                if (std::abs(concer_r - r) <= 1 && std::abs(concer_c - c) <= 1) concers.push_back(inter);
                            }
                        }
                    }
                }
            }
        } else {
            Vec3<double> unit = normalized(p2 - p1);
            for (double tr = 0; tr <= len; tr += 0.05) {
                Vec3<double> tmp = p1 + unit * tr;
                auto [c, r] = xy_to_index(tmp(0), tmp(1));
                if (is_index_valid(r, c)) {
                    auto& cell = line_map[{r, c}];
                    if (cell.empty() || l != cell.back()) {
                        cell.push_back(l);
                        if (lines.empty() || lines.back() != l) lines.push_back(l);
                    }
                }
            }
        }
    }
    void add_line(const Vec3<double>& p1, const Vec3<double>& p2, bool add_concers) {
        std::vector<Vec3<double>> fake_points;
        Vec3<double> mid_point = (p2 + p1) / 2.0;
        fake_points.push_back(p1);
        fake_points.push_back(mid_point);
        fake_points.push_back(p2);
        add_line(fake_points, 0, 2, add_concers);
    }
    const std::vector<line::ptr>* cell(int r, int c) const {
        auto it = line_map.find({r, c});
        return it == line_map.end() ? nullptr : &it->second;
    }
};

struct laser_submap {
    using ptr = std::shared_ptr<laser_submap>;
    Vec3<double> current_p, current_q;
    scan::ptr scan_ptr;
    laser_submap(const scan::ptr& s, const Vec3<double>& p, const Vec3<double>& q) : current_p(p), current_q(q), scan_ptr(s) {}
};
struct laser_match_lines {
    using ptr = std::shared_ptr<laser_match_lines>;
    std::vector<line::ptr> lines1, lines2;
    Vec3<double> p1, q1, p2, q2;
    scan::ptr scan2;
};

class laser_manager {
public:
    explicit laser_manager(const laser_params* P_) : P(P_) {
        w = P->w_laser_each_scan / P->laser_resolution + 1;
        h = P->h_laser_each_scan / P->laser_resolution + 1;
        resolution = P->laser_resolution;
        line_max_tolerance_angle = lf::angle_to_rad(P->line_max_tolerance_angle);
        current_count = 0;
    }

    static laser_match_lines::ptr do_match(const laser_params* P, const scan::ptr& scan1, const scan::ptr& scan2, const Vec3<double>& p1_,
                                           const Vec3<double>& q1_, const Vec3<double>& p2_, const Vec3<double>& q2_, int kk = 0) {
        auto inv = [](const Iso3<double>& T) { Iso3<double> r; r.R = T.R.transpose(); r.t = -(r.R * T.t); return r; };
        Iso3<double> T_1_2 = inv(lie::make_tf(p1_, q1_) * P->T_imu_to_laser) * (lie::make_tf(p2_, q2_) * P->T_imu_to_laser);
        laser_match_lines::ptr ret(new laser_match_lines);
        ret->p1 = p1_; ret->q1 = q1_; ret->p2 = p2_; ret->q2 = q2_; ret->scan2 = scan2;
        for (size_t i = 0; i < scan2->lines.size(); i++) {
            Vec3<double> p1 = scan2->lines[i]->p1, p2 = scan2->lines[i]->p2;
            std::vector<line::ptr> tmp_lines;
            Vec3<double> mid_p = (p1 + p2) / 2.0;
            {
                Vec3<double> transform_mid_p = T_1_2 * mid_p;
                auto [c, r] = scan1->xy_to_index(transform_mid_p(0), transform_mid_p(1));
                int a = 1 + kk;
                for (int dr = -a; dr <= a; dr++)
                    for (int dc = -a; dc <= a; dc++) {
                        int tmp_r = r + dr, tmp_c = c + dc;
                        if (scan1->is_index_valid(tmp_r, tmp_c))
                            if (auto* cell = scan1->cell(tmp_r, tmp_c)) tmp_lines.insert(tmp_lines.end(), cell->begin(), cell->end());
                    }
            }
            if (tmp_lines.empty()) continue;
            line::ptr best_match_line = nullptr;
            double best_angle = M_PI * 2;
            Vec3<double> v2 = T_1_2 * scan2->lines[i]->p2 - T_1_2 * scan2->lines[i]->p1;
            for (size_t j = 0; j < tmp_lines.size(); j++) {
                Vec3<double> v1 = tmp_lines[j]->p2 - tmp_lines[j]->p1;
                double angle = std::acos(std::fabs(dot(normalized(v1), normalized(v2))));
                if (angle < best_angle) { best_match_line = tmp_lines[j]; best_angle = angle; }
            }
            if (lf::rad_to_angle(best_angle) > 10) continue;
            ret->lines1.push_back(best_match_line);
            ret->lines2.push_back(scan2->lines[i]);
        }
        double aver_dis = 0;
        std::vector<double> diss(ret->lines1.size(), 0);
        for (size_t i = 0; i < ret->lines1.size(); i++) {
            Vec3<double> p1 = T_1_2 * ret->lines2[i]->p1, p2 = T_1_2 * ret->lines2[i]->p2;
            double dis = 0.5 * (e_laser::dis_from_line(p1, ret->lines1[i]->p1, ret->lines1[i]->p2) +
                                e_laser::dis_from_line(p2, ret->lines1[i]->p1, ret->lines1[i]->p2));
            aver_dis += dis;
            diss[i] = dis;
        }
        aver_dis /= ret->lines1.size();
        laser_match_lines::ptr ret2(new laser_match_lines);
        ret2->p1 = p1_; ret2->q1 = q1_; ret2->p2 = p2_; ret2->q2 = q2_; ret2->scan2 = scan2;
        double k = 1.2;
        for (size_t i = 0; i < ret->lines1.size(); i++)
            if (diss[i] < aver_dis * k) { ret2->lines1.push_back(ret->lines1[i]); ret2->lines2.push_back(ret->lines2[i]); }
        return ret2;
    }

    scan::ptr spawn_scan(const std::vector<Vec3<double>>& points, double time) {
        scan::ptr current_scan = std::make_shared<scan>(P, w, h, resolution, time);
        std::vector<std::tuple<int, int>> line_start_end_indexs;
        {
            int start_index = 0;
            int end_index = (int)points.size() - 1;
            for (size_t i = 1; i < points.size(); i++)
                if (!lf::is_continuous(*P, points[i - 1], points[i])) {
                    end_index = (int)i - 1;
                    line_start_end_indexs.emplace_back(start_index, end_index);
                    start_index = (int)i;
                }
            end_index = (int)points.size() - 1;
            line_start_end_indexs.emplace_back(start_index, end_index);
        }
        int step = 3;
        std::vector<double> responses(points.size(), -1);
        for (const auto& [start_index, end_index] : line_start_end_indexs) {
            std::vector<int> maybe_end_points;
            for (int i = start_index + 1; i <= end_index - 1; i++)
                responses[i] = lf::clac_cos(points[i], points[std::max(i - step, start_index)], points[std::min(i + step, end_index)]);
            maybe_end_points.push_back(start_index);
            for (int i = start_index + 1; i <= end_index - 1; i++) {
                bool is_max = true;
                int begin_j = std::max(i - step, start_index + 1), end_j = std::min(i + step, end_index - 1);
                for (int j = begin_j; j <= end_j; j++)
                    if (responses[j] >= responses[i] && j != i) { is_max = false; break; }
                if (is_max) { maybe_end_points.push_back(i); i += step; }
            }
            maybe_end_points.push_back(end_index);
            int last_end_index = 0;
            for (int i = 1; i < (int)maybe_end_points.size() - 1; i++) {
                double angle = lf::clac_angle(points[maybe_end_points[i]], points[maybe_end_points[last_end_index]], points[maybe_end_points[i + 1]]);
                if (std::abs(angle) < line_max_tolerance_angle) {
                    current_scan->add_line(points, maybe_end_points[last_end_index], maybe_end_points[i]);
                    last_end_index = i;
                }
            }
            current_scan->add_line(points, maybe_end_points[last_end_index], maybe_end_points.back());
        }
        return current_scan;
    }

    void add_scan(const scan::ptr& scan_ptr, const Vec3<double>& current_p, const Vec3<double>& current_q) {
        auto inv = [](const Iso3<double>& T) { Iso3<double> r; r.R = T.R.transpose(); r.t = -(r.R * T.t); return r; };
        laser_submap::ptr submap_ptr(new laser_submap(scan_ptr, current_p, current_q));
        key_frame.push_back(submap_ptr);
        auto current_tf = lie::make_tf(current_p, current_q);
        if (ref_submap_ptr != nullptr) {
            Vec3<double> dp, dq;
            lie::log_SE3<double>(inv(last_add_tf) * current_tf, dp, dq);
            if (norm(dp) < P->ref_motion_filter_p && norm(dq) < P->ref_motion_filter_q) return;
        } else {
            scan::ptr tmp_scan = std::make_shared<scan>(P, w, h, resolution, 0);
            ref_submap_ptr = laser_submap::ptr(new laser_submap(tmp_scan, current_p, current_q));
            last_add_tf = current_tf;
            current_count = 1;
            for (size_t i = 0; i < scan_ptr->lines.size(); i++) ref_submap_ptr->scan_ptr->add_line(scan_ptr->lines[i]->p1, scan_ptr->lines[i]->p2, false);
            return;
        }
        for (size_t i = 0; i < scan_ptr->lines.size(); i++) {
            {
                Iso3<double> tf_ref = lie::make_tf(ref_submap_ptr->current_p, ref_submap_ptr->current_q);
                Iso3<double> tf_ref_current = inv(tf_ref) * current_tf;
                Iso3<double> l_tf_ref_current = inv(P->T_imu_to_laser) * tf_ref_current * P->T_imu_to_laser;
                ref_submap_ptr->scan_ptr->add_line(l_tf_ref_current * scan_ptr->lines[i]->p1, l_tf_ref_current * scan_ptr->lines[i]->p2, false);
            }
            if (spawnning_ref_submap_ptr) {
                Iso3<double> tf_s = lie::make_tf(spawnning_ref_submap_ptr->current_p, spawnning_ref_submap_ptr->current_q);
                Iso3<double> l_tf = inv(P->T_imu_to_laser) * (inv(tf_s) * current_tf) * P->T_imu_to_laser;
                spawnning_ref_submap_ptr->scan_ptr->add_line(l_tf * scan_ptr->lines[i]->p1, l_tf * scan_ptr->lines[i]->p2, false);
            }
        }
        current_count++;
        if (spawnning_ref_submap_ptr == nullptr) {
            if (current_count == P->ref_n_accumulation / 2) {
                scan::ptr tmp_scan = std::make_shared<scan>(P, w, h, resolution, 0);
                spawnning_ref_submap_ptr = laser_submap::ptr(new laser_submap(tmp_scan, current_p, current_q));
                last_add_tf = lie::make_tf(current_p, current_q);
                for (size_t i = 0; i < scan_ptr->lines.size(); i++)
                    spawnning_ref_submap_ptr->scan_ptr->add_line(scan_ptr->lines[i]->p1, scan_ptr->lines[i]->p2, false);
            }
        }
        if (current_count == P->ref_n_accumulation) {
            ref_submap_ptr = spawnning_ref_submap_ptr;
            scan::ptr tmp_scan = std::make_shared<scan>(P, w, h, resolution, 0);
            spawnning_ref_submap_ptr = laser_submap::ptr(new laser_submap(tmp_scan, current_p, current_q));
            last_add_tf = lie::make_tf(current_p, current_q);
            for (size_t i = 0; i < scan_ptr->lines.size(); i++)
                spawnning_ref_submap_ptr->scan_ptr->add_line(scan_ptr->lines[i]->p1, scan_ptr->lines[i]->p2, false);
            current_count = P->ref_n_accumulation / 2;
        }
        last_add_tf = current_tf;
    }

    laser_match_lines::ptr empty_match(const scan::ptr& s, const Vec3<double>& p, const Vec3<double>& q) {
        laser_match_lines::ptr ret(new laser_match_lines);
        ret->p1 = p; ret->p2 = p; ret->q1 = q; ret->q2 = q; ret->scan2 = s;
        return ret;
    }
    laser_match_lines::ptr match_with_front(const scan::ptr s, const Vec3<double>& p, const Vec3<double>& q) {
        if (key_frame.empty()) return empty_match(s, p, q);
        return do_match(P, key_frame.front()->scan_ptr, s, key_frame.front()->current_p, key_frame.front()->current_q, p, q);
    }
    laser_match_lines::ptr match_with_back(const scan::ptr s, const Vec3<double>& p, const Vec3<double>& q) {
        if (key_frame.empty()) return empty_match(s, p, q);
        return do_match(P, key_frame.back()->scan_ptr, s, key_frame.back()->current_p, key_frame.back()->current_q, p, q);
    }
    laser_match_lines::ptr match_with_ref(const scan::ptr s, const Vec3<double>& p, const Vec3<double>& q) {
        if (ref_submap_ptr == nullptr) return empty_match(s, p, q);
        return do_match(P, ref_submap_ptr->scan_ptr, s, ref_submap_ptr->current_p, ref_submap_ptr->current_q, p, q);
    }
    laser_submap::ptr pop_scan() {
        if (key_frame.empty()) return nullptr;
        laser_submap::ptr ret = key_frame.front();
        key_frame.pop_front();
        return ret;
    }
    void clear_all_scan() { key_frame.clear(); ref_submap_ptr = nullptr; spawnning_ref_submap_ptr = nullptr; }

    const laser_params* P;
    int w, h;
    double resolution, line_max_tolerance_angle;
    std::deque<laser_submap::ptr> key_frame;
    laser_submap::ptr ref_submap_ptr, spawnning_ref_submap_ptr;
    Iso3<double> last_add_tf;
    int current_count;
};

}  // namespace lfe

// convert::laser_to_point_times (src/utilies/common.cpp:5-40): float angles, cosf/sinf
inline void laser_to_point_times(const float* ranges, int n, float angle_start, float angle_increment, float time_increment, double time,
                                 std::vector<Vec3<double>>& points, std::vector<double>& times) {
    for (size_t i = 0; i < (size_t)n; i++) {
        if (!std::isnan(ranges[i]) && !std::isinf(ranges[i]) && ranges[i] > 0.1) {
            // the reference is built for baseline x86-64 (no FMA): product and sum round separately
            const volatile float prod = i * angle_increment;
            const float ang = angle_start + prod;
            Vec3<double> point(std::cos(ang) * ranges[i], std::sin(ang) * ranges[i], 0);
            if (!points.empty())
                if (norm(point - points.back()) < 0.01) continue;
            points.emplace_back(point);
            const volatile float tinc = i * time_increment;
            times.emplace_back(time + tinc);
        }
    }
}
// sensor::laser::correct (src/trajectory/sensor.h:51-94)
inline void laser_correct(std::vector<Vec3<double>>& points, const std::vector<double>& times, double time_stamp, const Vec3<double>& linear,
                          const Vec3<double>& angular) {
    for (size_t i = 0; i < points.size(); i++) {
        double dt = times[i] - time_stamp;
        Iso3<double> T_i_j = lie::make_tf<double>(dt * linear, dt * angular);
        points[i] = T_i_j * points[i];
    }
}

}  // namespace oracle
