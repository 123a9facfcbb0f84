// liw_laser.cpp — 2D laser front-end of libliw_window.so (host C++): scan -> line segments ->
// laser_match, i.e. the producer of the laser_factor blocks the GPU path consumes.
//
// Native replacement of reference src/trajectory/laser_manager.cpp (+ laser_type.h, my_struct.h), same decisions,
// different machinery:
//   * line_map is a SPARSE grid: an open-addressing hash from cell index to a small run of line indices, instead of
//     a lazily allocated 2001 x 2001 array of std::vector<shared_ptr<line>>;
//   * lines are plain structs addressed by index; a laser_match is two index lists plus the copied end points, laid
//     out as the liw_window.laser_pts record ([lines1.p1 lines1.p2 lines2.p1 lines2.p2]);
//   * the homogeneous least-squares line  min |[x y 1] abc|, |abc| = 1  (fit_line_by_least_square, :19-37, JacobiSVD
//     there) is the smallest eigenvector of the 3x3 moment matrix, accumulated in one pass and diagonalised by
//     cyclic Jacobi rotations.
// Thresholds, tie-breaking and iteration orders follow the reference line by line (cited at each step).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include "../../include/liw_laser.h"
#include "liw_dual.hpp"

using liw::Iso;
using liw::M3;
using liw::V3;
typedef V3<double> Vec;

void liw_normalize_rotation_host(double* R9);   // liw_capi.hip (params.cpp:44-54 round trip)

namespace {

constexpr double kEps = 0.0008;             // epsilo, laser_manager.cpp:3
constexpr double kPi = 3.14159265358979323846;
inline double deg2rad(double a) { return a / 180.0 * kPi; }    // convert::angle_to_rad
inline double rad2deg(double a) { return a / kPi * 180.0; }

inline Vec vsub(const Vec& a, const Vec& b) { return Vec(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec vadd(const Vec& a, const Vec& b) { return Vec(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec vscale(const Vec& a, double s) { return Vec(a.x * s, a.y * s, a.z * s); }
inline double vdot(const Vec& a, const Vec& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double vnorm(const Vec& a) { return std::sqrt(vdot(a, a)); }
inline Vec vunit(const Vec& a) { const double z = vdot(a, a); return z > 0.0 ? vscale(a, 1.0 / std::sqrt(z)) : a; }
// Eigen's normalized() divides (x / n), it does not multiply by a reciprocal: keep the same rounding
inline Vec vunit_div(const Vec& a) { const double z = vdot(a, a); if (!(z > 0.0)) return a; const double n = std::sqrt(z); return Vec(a.x / n, a.y / n, a.z / n); }
inline Vec apply(const Iso<double>& T, const Vec& p) { return vadd(liw::mul(T.R, p), T.t); }

struct Line { Vec p1, p2, abc; double len; };

// e_laser::dis_from_line (src/utilies/common.h:86-95)
double dis_from_line(const Vec& p, const Vec& p1, const Vec& p2) {
    const Vec line = vunit_div(vsub(p2, p1));
    const Vec p2p = vsub(p, p2);
    const double t = vdot(vunit_div(line), p2p);
    return vnorm(vsub(p2p, vscale(line, t)));
}
// project_to_line (:8-17)
Vec project_to_line(const Vec& p, const Vec& a, const Vec& b) {
    if (vnorm(vsub(b, a)) < kEps) return p;
    const Vec u = vunit_div(vsub(b, a));
    return vadd(a, vscale(u, vdot(vsub(p, a), u)));
}

// smallest eigenvector of a symmetric 3x3 (cyclic Jacobi)
Vec smallest_eigvec3(double M[3][3]) {
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = std::fabs(M[0][1]) + std::fabs(M[0][2]) + std::fabs(M[1][2]);
        const double diag = std::fabs(M[0][0]) + std::fabs(M[1][1]) + std::fabs(M[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (M[p][q] == 0.0) continue;
                const double theta = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double mkp = M[k][p], mkq = M[k][q];
                    M[k][p] = c * mkp - s * mkq; M[k][q] = s * mkp + c * mkq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double mpk = M[p][k], mqk = M[q][k];
                    M[p][k] = c * mpk - s * mqk; M[q][k] = s * mpk + c * mqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    if (M[1][1] < M[m][m]) m = 1;
    if (M[2][2] < M[m][m]) m = 2;
    return Vec(V[0][m], V[1][m], V[2][m]);
}

// fit_line_by_least_square (:19-37): right singular vector of [x y 1] for the smallest singular value
Vec fit_line(const double* pts, int i1, int i2) {
    double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = i1; i <= i2; ++i) {
        const double x = pts[i * 3], y = pts[i * 3 + 1];
        M[0][0] += x * x; M[0][1] += x * y; M[0][2] += x; M[1][1] += y * y; M[1][2] += y; M[2][2] += 1.0;
    }
    M[1][0] = M[0][1]; M[2][0] = M[0][2]; M[2][1] = M[1][2];
    return smallest_eigvec3(M);
}

}  // namespace

struct liw_scan_impl {
    int w, h;
    double res, time;
    std::vector<Line> pool;          // every accepted line (index = line id)
    std::vector<int> lines;          // scan::lines: ids in order of first registration in the grid
    std::vector<int> pos;            // pool id -> index in `lines` (-1: never registered)
    std::vector<Vec> concers;
    // sparse line_map: open addressing, key = r * w + c; value = ids in push order (chunked list in `cells`)
    struct Cell { int64_t key; std::vector<int> ids; };
    std::vector<int> slot;           // hash table of indices into cells (-1 empty)
    std::vector<Cell> cells;

    liw_scan_impl(int w_, int h_, double r_, double t_) : w(w_), h(h_), res(r_), time(t_), slot(1024, -1) {}
    bool valid(int r, int c) const { return r >= 0 && r < h && c >= 0 && c < w; }
    // scan::xy_to_index (laser_type.h:38-41): {x / resolution + w / 2, y / resolution + h / 2} truncated to int -> (c, r)
    void xy_to_index(double x, double y, int& c, int& r) const { c = (int)(x / res + (double)(w / 2)); r = (int)(y / res + (double)(h / 2)); }
    static uint64_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; return k; }
    const Cell* find(int r, int c) const {
        const int64_t key = (int64_t)r * w + c;
        const size_t mask = slot.size() - 1;
        for (size_t p = mix((uint64_t)key) & mask;; p = (p + 1) & mask) {
            const int s = slot[p];
            if (s < 0) return nullptr;
            if (cells[s].key == key) return &cells[s];
        }
    }
    Cell& at(int r, int c) {
        const int64_t key = (int64_t)r * w + c;
        if ((cells.size() + 1) * 2 > slot.size()) {
            std::vector<int> ns(slot.size() * 2, -1);
            const size_t mask = ns.size() - 1;
            for (size_t i = 0; i < cells.size(); ++i) {
                size_t p = mix((uint64_t)cells[i].key) & mask;
                while (ns[p] >= 0) p = (p + 1) & mask;
                ns[p] = (int)i;
            }
            slot.swap(ns);
        }
        const size_t mask = slot.size() - 1;
        size_t p = mix((uint64_t)key) & mask;
        for (;; p = (p + 1) & mask) {
            const int s = slot[p];
            if (s < 0) break;
            if (cells[s].key == key) return cells[s];
        }
        slot[p] = (int)cells.size();
        cells.push_back(Cell{key, {}});
        return cells.back();
    }
    void register_line(int id) {
        if (!lines.empty() && lines.back() == id) return;
        if ((int)pos.size() <= id) pos.resize(id + 1, -1);
        pos[id] = (int)lines.size();
        lines.push_back(id);
    }
    int line_index(int id) const { return id < (int)pos.size() ? pos[id] : -1; }

    // scan::add_line(points, index1, index2, add_concers) (:137-212)
    bool add_line(const double* pts, int i1, int i2, bool add_concers, double line_max_dis, double line_min_len) {
        if (i2 - i1 < 2) return false;
        const Vec abc = fit_line(pts, i1, i2);
        // create_line (:61-96)
        Vec a(0, 0, 0), b(0, 0, 0);
        if (std::fabs(abc.y) < 0.5) {
            a.y = 0; a.x = -abc.z / abc.x; b.y = 1; b.x = (-abc.z - abc.y) / abc.x;
        } else {
            a.x = 0; b.x = 1; a.y = -abc.z / abc.y; b.y = (-abc.z - abc.x) / abc.y;
        }
        double max_dis = 0;
        for (int i = i1; i <= i2; ++i) max_dis = std::max(max_dis, dis_from_line(liw::cast_v3<double>(pts + 3 * i), a, b));
        const Vec p1 = project_to_line(liw::cast_v3<double>(pts + 3 * i1), a, b), p2 = project_to_line(liw::cast_v3<double>(pts + 3 * i2), a, b);
        const double len = vnorm(vsub(p1, p2));
        if (max_dis > line_max_dis) return false;
        if (len < line_min_len) return false;
        const int id = (int)pool.size();
        pool.push_back(Line{p1, p2, abc, len});
        if (add_concers) {
            for (int i = i1; i <= i2; ++i) {
                int c, r;
                xy_to_index(pts[3 * i], pts[3 * i + 1], c, r);
                if (!valid(r, c)) continue;
                Cell& cell = at(r, c);
                if (!cell.ids.empty() && cell.ids.back() == id) continue;
                cell.ids.push_back(id);
                register_line(id);
                if (cell.ids.size() == 2) {
                    const Line& l0 = pool[cell.ids[0]];
                    const Line& l1 = pool[cell.ids[1]];
                    if (l0.len > 0.1 && l1.len > 0.1) {
                        // calc_angle_and_intersection (:38-60)
                        const double angle = std::acos(vdot(vunit_div(vsub(l0.p1, l0.p2)), vunit_div(vsub(l1.p1, l1.p2))));
                        if (angle < deg2rad(150) && angle > deg2rad(30)) {
                            const double det = l0.abc.x * l1.abc.y - l0.abc.y * l1.abc.x;
                            const double ix = (-l0.abc.z * l1.abc.y + l1.abc.z * l0.abc.y) / det;
                            const double iy = (-l0.abc.x * l1.abc.z + l1.abc.x * l0.abc.z) / det;
                            int cc, cr;
                            xy_to_index(ix, iy, cc, cr);
                            if (std::abs(cr - r) <= 1 && std::abs(cc - c) <= 1) concers.push_back(Vec(ix, iy, 0.0));
                        }
                    }
                }
            }
        } else {
            const Vec unit = vunit_div(vsub(p2, p1));
            for (double tr = 0; tr <= len; tr += 0.05) {
                const Vec t = vadd(p1, vscale(unit, tr));
                int c, r;
                xy_to_index(t.x, t.y, c, r);
                if (!valid(r, c)) continue;
                Cell& cell = at(r, c);
                if (cell.ids.empty() || cell.ids.back() != id) { cell.ids.push_back(id); register_line(id); }
            }
        }
        return true;
    }
    // scan::add_line(p1, p2, add_concers) (:213-222)
    bool add_segment(const Vec& p1, const Vec& p2, bool add_concers, double line_max_dis, double line_min_len) {
        const Vec mid = Vec((p2.x + p1.x) / 2, (p2.y + p1.y) / 2, (p2.z + p1.z) / 2);
        const double fake[9] = {p1.x, p1.y, p1.z, mid.x, mid.y, mid.z, p2.x, p2.y, p2.z};
        return add_line(fake, 0, 2, add_concers, line_max_dis, line_min_len);
    }
};

struct liw_scan { std::shared_ptr<liw_scan_impl> s; liw_laser_params prm; };

struct liw_laser_match {
    std::vector<double> pts;       // [size][12]
    std::vector<int> idx1, idx2;
    double pose[12];
};

namespace {

struct Extr { Iso<double> T_il; };
Extr extrinsics(const liw_laser_params& prm) {
    double R[9], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = prm.T_imu_to_laser[i * 4 + j]; t[i] = prm.T_imu_to_laser[i * 4 + 3]; }
    if (prm.normalize_extrinsics) liw_normalize_rotation_host(R);
    Extr e;
    e.T_il = liw::cast_iso<double>(R, t);
    return e;
}
inline void grid_dims(const liw_laser_params& prm, int& w, int& h) {
    // laser_manager ctor (:229-241): int <- double expression
    w = (int)(prm.w_laser_each_scan / prm.laser_resolution + 1);
    h = (int)(prm.h_laser_each_scan / prm.laser_resolution + 1);
}
inline double clac_cos(const Vec& pj, const Vec& pi, const Vec& pk) {   // (:102-113)
    if (vnorm(vsub(pi, pj)) < kEps) return -1;
    if (vnorm(vsub(pj, pk)) < kEps) return -1;
    return vdot(vunit_div(vsub(pi, pj)), vunit_div(vsub(pk, pj)));
}

std::shared_ptr<liw_scan_impl> spawn(const liw_laser_params& prm, const double* pts, int N, double time) {
    int w, h;
    grid_dims(prm, w, h);
    auto sc = std::make_shared<liw_scan_impl>(w, h, prm.laser_resolution, time);
    auto P = [&](int i) { return liw::cast_v3<double>(pts + 3 * i); };
    // continuous runs (:361-374)
    std::vector<std::pair<int, int>> runs;
    {
        int start = 0;
        for (int i = 1; i < N; ++i)
            if (!(vnorm(vsub(P(i - 1), P(i))) <= prm.line_continuous_threshold)) { runs.emplace_back(start, i - 1); start = i; }
        runs.emplace_back(start, N - 1);
    }
    const int step = 3;
    const double tol = deg2rad(prm.line_max_tolerance_angle);
    std::vector<double> resp((size_t)std::max(N, 0), -1.0);
    std::vector<int> ends;
    for (const auto& run : runs) {
        const int s = run.first, e = run.second;
        for (int i = s + 1; i <= e - 1; ++i) resp[i] = clac_cos(P(i), P(std::max(i - step, s)), P(std::min(i + step, e)));
        ends.clear();
        ends.push_back(s);
        for (int i = s + 1; i <= e - 1; ++i) {       // strict local maxima of the corner response (:388-405)
            bool is_max = true;
            const int bj = std::max(i - step, s + 1), ej = std::min(i + step, e - 1);
            for (int j = bj; j <= ej; ++j)
                if (resp[j] >= resp[i] && j != i) { is_max = false; break; }
            if (is_max) { ends.push_back(i); i += step; }
        }
        ends.push_back(e);
        int last = 0;
        for (int i = 1; i + 1 < (int)ends.size(); ++i) {
            const double angle = std::acos(clac_cos(P(ends[i]), P(ends[last]), P(ends[i + 1])));
            if (std::fabs(angle) < tol) {
                sc->add_line(pts, ends[last], ends[i], true, prm.line_max_dis, prm.line_min_len);
                last = i;
            }
        }
        sc->add_line(pts, ends[last], ends.back(), true, prm.line_max_dis, prm.line_min_len);
    }
    return sc;
}

liw_laser_match* empty_match(const double* p, const double* q) {
    liw_laser_match* m = new liw_laser_match();
    for (int k = 0; k < 3; ++k) { m->pose[k] = p[k]; m->pose[3 + k] = q[k]; m->pose[6 + k] = p[k]; m->pose[9 + k] = q[k]; }
    return m;
}

// laser_manager::do_match (:244-348)
liw_laser_match* do_match(const liw_laser_params& prm, const liw_scan_impl& s1, const liw_scan_impl& s2, const double* p1_, const double* q1_,
                          const double* p2_, const double* q2_, int kk) {
    const Extr ex = extrinsics(prm);
    const Iso<double> T1 = liw::mul(liw::make_tf(liw::cast_v3<double>(p1_), liw::cast_v3<double>(q1_)), ex.T_il);
    const Iso<double> T2 = liw::mul(liw::make_tf(liw::cast_v3<double>(p2_), liw::cast_v3<double>(q2_)), ex.T_il);
    const Iso<double> T12 = liw::mul(liw::inverse(T1), T2);
    std::vector<int> m1, m2;
    std::vector<int> cand;
    for (size_t i = 0; i < s2.lines.size(); ++i) {
        const Line& l2 = s2.pool[s2.lines[i]];
        const Vec mid((l2.p1.x + l2.p2.x) / 2, (l2.p1.y + l2.p2.y) / 2, (l2.p1.z + l2.p2.z) / 2);
        const Vec tm = apply(T12, mid);
        int c, r;
        s1.xy_to_index(tm.x, tm.y, c, r);
        cand.clear();
        const int a = 1 + kk;
        for (int dr = -a; dr <= a; ++dr)
            for (int dc = -a; dc <= a; ++dc) {
                const int rr = r + dr, cc = c + dc;
                if (!s1.valid(rr, cc)) continue;
                if (const liw_scan_impl::Cell* cell = s1.find(rr, cc)) cand.insert(cand.end(), cell->ids.begin(), cell->ids.end());
            }
        if (cand.empty()) continue;
        int best = -1;
        double best_angle = kPi * 2;
        const Vec v2 = vsub(apply(T12, l2.p2), apply(T12, l2.p1));
        for (int id : cand) {
            const Line& l1 = s1.pool[id];
            const double angle = std::acos(std::fabs(vdot(vunit_div(vsub(l1.p2, l1.p1)), vunit_div(v2))));
            if (angle < best_angle) { best = id; best_angle = angle; }
        }
        if (rad2deg(best_angle) > 10) continue;
        m1.push_back(best);
        m2.push_back(s2.lines[i]);
    }
    // keep the pairs closer than 1.2 x the mean line-to-line distance (:316-345)
    std::vector<double> diss(m1.size(), 0.0);
    double aver = 0;
    for (size_t i = 0; i < m1.size(); ++i) {
        const Line& l1 = s1.pool[m1[i]];
        const Line& l2 = s2.pool[m2[i]];
        const double d = 0.5 * (dis_from_line(apply(T12, l2.p1), l1.p1, l1.p2) + dis_from_line(apply(T12, l2.p2), l1.p1, l1.p2));
        aver += d;
        diss[i] = d;
    }
    aver /= (double)m1.size();
    liw_laser_match* out = new liw_laser_match();
    for (int k = 0; k < 3; ++k) { out->pose[k] = p1_[k]; out->pose[3 + k] = q1_[k]; out->pose[6 + k] = p2_[k]; out->pose[9 + k] = q2_[k]; }
    for (size_t i = 0; i < m1.size(); ++i) {
        if (!(diss[i] < aver * 1.2)) continue;
        const Line& l1 = s1.pool[m1[i]];
        const Line& l2 = s2.pool[m2[i]];
        const double rec[12] = {l1.p1.x, l1.p1.y, l1.p1.z, l1.p2.x, l1.p2.y, l1.p2.z, l2.p1.x, l2.p1.y, l2.p1.z, l2.p2.x, l2.p2.y, l2.p2.z};
        out->pts.insert(out->pts.end(), rec, rec + 12);
        // index in scan::lines (registration order), which is what a caller of the reference sees
        out->idx1.push_back(s1.line_index(m1[i]));
        out->idx2.push_back(s2.line_index(m2[i]));
    }
    return out;
}

struct Submap { std::shared_ptr<liw_scan_impl> scan; Vec p, q; };

}  // namespace

struct liw_laser_manager {
    liw_laser_params prm;
    std::deque<Submap> key_frame;
    std::shared_ptr<Submap> ref, spawning;
    Iso<double> last_add_tf;
    int current_count = 0;
    liw_scan ref_view;   // borrowed handle returned by liw_laser_manager_ref_scan

    std::shared_ptr<Submap> fresh_submap(const liw_scan_impl& from, const Vec& p, const Vec& q) {
        int w, h;
        grid_dims(prm, w, h);
        auto sm = std::make_shared<Submap>(Submap{std::make_shared<liw_scan_impl>(w, h, prm.laser_resolution, 0.0), p, q});
        for (int id : from.lines) sm->scan->add_segment(from.pool[id].p1, from.pool[id].p2, false, prm.line_max_dis, prm.line_min_len);
        return sm;
    }
    // laser_manager::add_scan (:424-496)
    void add_scan(const std::shared_ptr<liw_scan_impl>& sc, const Vec& p, const Vec& q) {
        key_frame.push_back(Submap{sc, p, q});
        const Iso<double> cur = liw::make_tf(p, q);
        if (ref) {
            const Iso<double> d = liw::mul(liw::inverse(last_add_tf), cur);
            const Vec dq = liw::log_SO3(d.R);
            if (vnorm(d.t) < prm.ref_motion_filter_p && vnorm(dq) < prm.ref_motion_filter_q) return;
        } else {
            ref = fresh_submap(*sc, p, q);
            last_add_tf = cur;
            current_count = 1;
            return;
        }
        const Extr ex = extrinsics(prm);
        auto accumulate = [&](Submap& sm) {
            const Iso<double> rel = liw::mul(liw::inverse(liw::make_tf(sm.p, sm.q)), cur);
            const Iso<double> lrel = liw::mul(liw::mul(liw::inverse(ex.T_il), rel), ex.T_il);
            return lrel;
        };
        const Iso<double> l_ref = accumulate(*ref);
        Iso<double> l_sp = l_ref;
        if (spawning) l_sp = accumulate(*spawning);
        for (int id : sc->lines) {
            const Line& l = sc->pool[id];
            ref->scan->add_segment(apply(l_ref, l.p1), apply(l_ref, l.p2), false, prm.line_max_dis, prm.line_min_len);
            if (spawning) spawning->scan->add_segment(apply(l_sp, l.p1), apply(l_sp, l.p2), false, prm.line_max_dis, prm.line_min_len);
        }
        ++current_count;
        if (!spawning) {
            if (current_count == prm.ref_n_accumulation / 2) {
                spawning = fresh_submap(*sc, p, q);
                last_add_tf = cur;
            }
        }
        if (current_count == prm.ref_n_accumulation) {
            ref = spawning;
            spawning = fresh_submap(*sc, p, q);
            last_add_tf = cur;
            current_count = prm.ref_n_accumulation / 2;
        }
        last_add_tf = cur;
    }
};

extern "C" {

int liw_laser_to_points(const float* ranges, int n, float angle_min, float angle_increment, float time_increment, double stamp,
                        double* points, double* times) {
    if (!ranges || !points || !(angle_increment > 0)) return LIW_EINVAL;   // the reference exits on angle_increment <= 0
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float rg = ranges[i];
        if (std::isnan(rg) || std::isinf(rg) || !(rg > 0.1)) continue;
        // float arithmetic as in common.cpp:22-24 (float angle, cosf/sinf, float product)
        // product and sum round separately (the reference is built without FMA contraction)
        const volatile float prod = (float)(size_t)i * angle_increment;
        const float ang = angle_min + prod;
        const double x = (double)(std::cos(ang) * rg), y = (double)(std::sin(ang) * rg);
        if (m > 0) {
            const double dx = x - points[3 * (m - 1)], dy = y - points[3 * (m - 1) + 1], dz = 0.0 - points[3 * (m - 1) + 2];
            if (std::sqrt(dx * dx + dy * dy + dz * dz) < 0.01) continue;
        }
        points[3 * m] = x; points[3 * m + 1] = y; points[3 * m + 2] = 0.0;
        if (times) times[m] = stamp + (double)((float)(size_t)i * time_increment);
        ++m;
    }
    return m;
}

void liw_laser_correct(double* points, const double* times, int n, double stamp, const double* lin, const double* ang) {
    for (int i = 0; i < n; ++i) {
        const double dt = times[i] - stamp;
        const Iso<double> T = liw::make_tf(Vec(dt * lin[0], dt * lin[1], dt * lin[2]), Vec(dt * ang[0], dt * ang[1], dt * ang[2]));
        const Vec r = apply(T, liw::cast_v3<double>(points + 3 * i));
        points[3 * i] = r.x; points[3 * i + 1] = r.y; points[3 * i + 2] = r.z;
    }
}

liw_scan* liw_scan_spawn(const liw_laser_params* prm, const double* points, int n, double time) {
    if (!prm || (n > 0 && !points) || n < 0) return nullptr;
    liw_scan* s = new liw_scan();
    s->prm = *prm;
    s->s = spawn(*prm, points, n, time);
    return s;
}
liw_scan* liw_scan_create_empty(const liw_laser_params* prm, double time) {
    if (!prm) return nullptr;
    int w, h;
    grid_dims(*prm, w, h);
    liw_scan* s = new liw_scan();
    s->prm = *prm;
    s->s = std::make_shared<liw_scan_impl>(w, h, prm->laser_resolution, time);
    return s;
}
int liw_scan_add_segment(liw_scan* s, const double* p1, const double* p2, int add_concers) {
    if (!s || !p1 || !p2) return LIW_EINVAL;
    return s->s->add_segment(liw::cast_v3<double>(p1), liw::cast_v3<double>(p2), add_concers != 0, s->prm.line_max_dis, s->prm.line_min_len) ? 1 : 0;
}
void liw_scan_destroy(liw_scan* s) { delete s; }
int liw_scan_num_lines(const liw_scan* s) { return s ? (int)s->s->lines.size() : 0; }
void liw_scan_get_lines(const liw_scan* s, double* out) {
    if (!s || !out) return;
    for (size_t i = 0; i < s->s->lines.size(); ++i) {
        const Line& l = s->s->pool[s->s->lines[i]];
        const double rec[10] = {l.p1.x, l.p1.y, l.p1.z, l.p2.x, l.p2.y, l.p2.z, l.abc.x, l.abc.y, l.abc.z, l.len};
        std::memcpy(out + 10 * i, rec, sizeof rec);
    }
}
int liw_scan_num_concers(const liw_scan* s) { return s ? (int)s->s->concers.size() : 0; }
void liw_scan_get_concers(const liw_scan* s, double* out) {
    if (!s || !out) return;
    for (size_t i = 0; i < s->s->concers.size(); ++i) { out[3 * i] = s->s->concers[i].x; out[3 * i + 1] = s->s->concers[i].y; out[3 * i + 2] = s->s->concers[i].z; }
}
int liw_scan_cell_lines(const liw_scan* s, double x, double y, int* ids, int cap) {
    if (!s) return -1;
    int c, r;
    s->s->xy_to_index(x, y, c, r);
    if (!s->s->valid(r, c)) return -1;
    const liw_scan_impl::Cell* cell = s->s->find(r, c);
    if (!cell) return 0;
    for (int k = 0; k < (int)cell->ids.size() && k < cap && ids; ++k)
        ids[k] = s->s->line_index(cell->ids[k]);
    return (int)cell->ids.size();
}

liw_laser_match* liw_laser_do_match(const liw_laser_params* prm, const liw_scan* s1, const liw_scan* s2, const double* p1, const double* q1,
                                    const double* p2, const double* q2, int kk) {
    if (!prm || !s1 || !s2 || !p1 || !q1 || !p2 || !q2) return nullptr;
    return do_match(*prm, *s1->s, *s2->s, p1, q1, p2, q2, kk);
}
void liw_laser_match_destroy(liw_laser_match* m) { delete m; }
int liw_laser_match_size(const liw_laser_match* m) { return m ? (int)m->idx1.size() : 0; }
void liw_laser_match_get(const liw_laser_match* m, double* pts, double* pose12, int* idx1, int* idx2) {
    if (!m) return;
    if (pts && !m->pts.empty()) std::memcpy(pts, m->pts.data(), sizeof(double) * m->pts.size());
    if (pose12) std::memcpy(pose12, m->pose, sizeof m->pose);
    if (idx1 && !m->idx1.empty()) std::memcpy(idx1, m->idx1.data(), sizeof(int) * m->idx1.size());
    if (idx2 && !m->idx2.empty()) std::memcpy(idx2, m->idx2.data(), sizeof(int) * m->idx2.size());
}

liw_laser_manager* liw_laser_manager_create(const liw_laser_params* prm) {
    if (!prm) return nullptr;
    liw_laser_manager* m = new liw_laser_manager();
    m->prm = *prm;
    return m;
}
void liw_laser_manager_destroy(liw_laser_manager* m) { delete m; }
void liw_laser_manager_add_scan(liw_laser_manager* m, liw_scan* scan, const double* p, const double* q) {
    if (!m || !scan || !p || !q) return;
    m->add_scan(scan->s, liw::cast_v3<double>(p), liw::cast_v3<double>(q));
}
static liw_laser_match* match_against(liw_laser_manager* m, const Submap* sm, const liw_scan* scan, const double* p, const double* q) {
    if (!m || !scan || !p || !q) return nullptr;
    if (!sm) return empty_match(p, q);
    const double p1[3] = {sm->p.x, sm->p.y, sm->p.z}, q1[3] = {sm->q.x, sm->q.y, sm->q.z};
    return do_match(m->prm, *sm->scan, *scan->s, p1, q1, p, q, 0);
}
liw_laser_match* liw_laser_manager_match_with_front(liw_laser_manager* m, const liw_scan* scan, const double* p, const double* q) {
    return match_against(m, m && !m->key_frame.empty() ? &m->key_frame.front() : nullptr, scan, p, q);
}
liw_laser_match* liw_laser_manager_match_with_back(liw_laser_manager* m, const liw_scan* scan, const double* p, const double* q) {
    return match_against(m, m && !m->key_frame.empty() ? &m->key_frame.back() : nullptr, scan, p, q);
}
liw_laser_match* liw_laser_manager_match_with_ref(liw_laser_manager* m, const liw_scan* scan, const double* p, const double* q) {
    return match_against(m, m && m->ref ? m->ref.get() : nullptr, scan, p, q);
}
int liw_laser_manager_pop_scan(liw_laser_manager* m) {
    if (!m || m->key_frame.empty()) return 0;
    m->key_frame.pop_front();
    return 1;
}
void liw_laser_manager_clear_all_scan(liw_laser_manager* m) {
    if (!m) return;
    m->key_frame.clear();
    m->ref.reset();
    m->spawning.reset();
}
int liw_laser_manager_num_keyframes(const liw_laser_manager* m) { return m ? (int)m->key_frame.size() : 0; }
int liw_laser_manager_set_keyframe_pose(liw_laser_manager* m, int i, const double* p, const double* q) {
    if (!m || !p || !q || i < 0 || i >= (int)m->key_frame.size()) return LIW_EINVAL;
    m->key_frame[i].p = liw::cast_v3<double>(p);
    m->key_frame[i].q = liw::cast_v3<double>(q);
    return LIW_OK;
}
const liw_scan* liw_laser_manager_ref_scan(const liw_laser_manager* m, double* p3, double* q3) {
    if (!m || !m->ref) return nullptr;
    liw_laser_manager* mm = const_cast<liw_laser_manager*>(m);
    mm->ref_view.s = m->ref->scan;
    mm->ref_view.prm = m->prm;
    if (p3) { p3[0] = m->ref->p.x; p3[1] = m->ref->p.y; p3[2] = m->ref->p.z; }
    if (q3) { q3[0] = m->ref->q.x; q3[1] = m->ref->q.y; q3[2] = m->ref->q.z; }
    return &mm->ref_view;
}

}  // extern "C"
