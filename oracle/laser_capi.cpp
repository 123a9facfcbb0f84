// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  C entry points of the laser front-end restatement
// (laser_frontend.h) for oracle/pyoracle.py.  The params struct has the layout of liw_laser_params (include/liw_laser.h).
#include <algorithm>
#include <cstring>

#include "laser_frontend.h"

using namespace oracle;
using namespace oracle::lfe;

extern "C" {

struct oracle_laser_params_c {
    double w_laser_each_scan, h_laser_each_scan, laser_resolution;
    double line_continuous_threshold, line_min_len, line_max_dis, line_max_tolerance_angle;
    double ref_motion_filter_p, ref_motion_filter_q;
    int ref_n_accumulation;
    double T_imu_to_laser[16];
    int normalize_extrinsics;
};
struct oracle_laser_ctx { laser_params prm; laser_manager* mgr; };
struct oracle_scan_h { scan::ptr s; };
struct oracle_match_h { laser_match_lines::ptr m; scan::ptr s1; };

void* oracle_laser_create(const oracle_laser_params_c* c) {
    oracle_laser_ctx* h = new oracle_laser_ctx();
    laser_params& p = h->prm;
    p.w_laser_each_scan = c->w_laser_each_scan; p.h_laser_each_scan = c->h_laser_each_scan; p.laser_resolution = c->laser_resolution;
    p.line_continuous_threshold = c->line_continuous_threshold; p.line_min_len = c->line_min_len; p.line_max_dis = c->line_max_dis;
    p.line_max_tolerance_angle = c->line_max_tolerance_angle;
    p.ref_motion_filter_p = c->ref_motion_filter_p; p.ref_motion_filter_q = c->ref_motion_filter_q; p.ref_n_accumulation = c->ref_n_accumulation;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) p.T_imu_to_laser.R(i, j) = c->T_imu_to_laser[i * 4 + j]; p.T_imu_to_laser.t(i) = c->T_imu_to_laser[i * 4 + 3]; }
    if (c->normalize_extrinsics) lie::normalize_tf<double>(p.T_imu_to_laser);
    h->mgr = new laser_manager(&h->prm);
    return h;
}
void oracle_laser_destroy(void* hv) { oracle_laser_ctx* h = (oracle_laser_ctx*)hv; delete h->mgr; delete h; }

static Vec3<double> v3(const double* p) { return Vec3<double>(p[0], p[1], p[2]); }

void* oracle_scan_spawn(void* hv, const double* pts, int n, double time) {
    oracle_laser_ctx* h = (oracle_laser_ctx*)hv;
    std::vector<Vec3<double>> points;
    for (int i = 0; i < n; ++i) points.push_back(v3(pts + 3 * i));
    return new oracle_scan_h{h->mgr->spawn_scan(points, time)};
}
void* oracle_scan_create_empty(void* hv, double time) {
    oracle_laser_ctx* h = (oracle_laser_ctx*)hv;
    return new oracle_scan_h{std::make_shared<scan>(&h->prm, h->mgr->w, h->mgr->h, h->mgr->resolution, time)};
}
int oracle_scan_add_segment(void* sv, const double* p1, const double* p2, int add_concers) {
    scan::ptr s = ((oracle_scan_h*)sv)->s;
    const size_t before = s->lines.size();
    s->add_line(v3(p1), v3(p2), add_concers != 0);
    return s->lines.size() > before ? 1 : 0;
}
void oracle_scan_destroy(void* sv) { delete (oracle_scan_h*)sv; }
int oracle_scan_num_lines(void* sv) { return (int)((oracle_scan_h*)sv)->s->lines.size(); }
void oracle_scan_get_lines(void* sv, double* out) {
    scan::ptr s = ((oracle_scan_h*)sv)->s;
    for (size_t i = 0; i < s->lines.size(); ++i) {
        const line& l = *s->lines[i];
        const double rec[10] = {l.p1(0), l.p1(1), l.p1(2), l.p2(0), l.p2(1), l.p2(2), l.abc(0), l.abc(1), l.abc(2), l.len};
        std::memcpy(out + 10 * i, rec, sizeof rec);
    }
}
int oracle_scan_num_concers(void* sv) { return (int)((oracle_scan_h*)sv)->s->concers.size(); }
void oracle_scan_get_concers(void* sv, double* out) {
    scan::ptr s = ((oracle_scan_h*)sv)->s;
    for (size_t i = 0; i < s->concers.size(); ++i) { out[3 * i] = s->concers[i](0); out[3 * i + 1] = s->concers[i](1); out[3 * i + 2] = s->concers[i](2); }
}
static int index_of(const scan::ptr& s, const line::ptr& l) {
    auto it = std::find(s->lines.begin(), s->lines.end(), l);
    return it == s->lines.end() ? -1 : (int)(it - s->lines.begin());
}
int oracle_scan_cell_lines(void* sv, double x, double y, int* ids, int cap) {
    scan::ptr s = ((oracle_scan_h*)sv)->s;
    auto [c, r] = s->xy_to_index(x, y);
    if (!s->is_index_valid(r, c)) return -1;
    const auto* cell = s->cell(r, c);
    if (!cell) return 0;
    for (int k = 0; k < (int)cell->size() && k < cap && ids; ++k) ids[k] = index_of(s, (*cell)[k]);
    return (int)cell->size();
}

static void* wrap_match(const laser_match_lines::ptr& m, const scan::ptr& s1) { return new oracle_match_h{m, s1}; }
void* oracle_laser_do_match(void* hv, void* s1v, void* s2v, const double* p1, const double* q1, const double* p2, const double* q2, int kk) {
    oracle_laser_ctx* h = (oracle_laser_ctx*)hv;
    scan::ptr s1 = ((oracle_scan_h*)s1v)->s, s2 = ((oracle_scan_h*)s2v)->s;
    return wrap_match(laser_manager::do_match(&h->prm, s1, s2, v3(p1), v3(q1), v3(p2), v3(q2), kk), s1);
}
void oracle_laser_match_destroy(void* mv) { delete (oracle_match_h*)mv; }
int oracle_laser_match_size(void* mv) { return (int)((oracle_match_h*)mv)->m->lines1.size(); }
void oracle_laser_match_get(void* mv, double* pts, double* pose12, int* idx1, int* idx2) {
    oracle_match_h* h = (oracle_match_h*)mv;
    const laser_match_lines& m = *h->m;
    for (size_t i = 0; i < m.lines1.size(); ++i) {
        if (pts) {
            const double rec[12] = {m.lines1[i]->p1(0), m.lines1[i]->p1(1), m.lines1[i]->p1(2), m.lines1[i]->p2(0), m.lines1[i]->p2(1), m.lines1[i]->p2(2),
                                    m.lines2[i]->p1(0), m.lines2[i]->p1(1), m.lines2[i]->p1(2), m.lines2[i]->p2(0), m.lines2[i]->p2(1), m.lines2[i]->p2(2)};
            std::memcpy(pts + 12 * i, rec, sizeof rec);
        }
        if (idx1) idx1[i] = h->s1 ? index_of(h->s1, m.lines1[i]) : -1;
        if (idx2) idx2[i] = index_of(m.scan2, m.lines2[i]);
    }
    if (pose12)
        for (int k = 0; k < 3; ++k) { pose12[k] = m.p1(k); pose12[3 + k] = m.q1(k); pose12[6 + k] = m.p2(k); pose12[9 + k] = m.q2(k); }
}

void oracle_laser_add_scan(void* hv, void* sv, const double* p, const double* q) {
    ((oracle_laser_ctx*)hv)->mgr->add_scan(((oracle_scan_h*)sv)->s, v3(p), v3(q));
}
void* oracle_laser_match_with(void* hv, int which, void* sv, const double* p, const double* q) {
    laser_manager* m = ((oracle_laser_ctx*)hv)->mgr;
    scan::ptr s = ((oracle_scan_h*)sv)->s;
    if (which == 0) return wrap_match(m->match_with_front(s, v3(p), v3(q)), m->key_frame.empty() ? nullptr : m->key_frame.front()->scan_ptr);
    if (which == 1) return wrap_match(m->match_with_back(s, v3(p), v3(q)), m->key_frame.empty() ? nullptr : m->key_frame.back()->scan_ptr);
    return wrap_match(m->match_with_ref(s, v3(p), v3(q)), m->ref_submap_ptr ? m->ref_submap_ptr->scan_ptr : nullptr);
}
int oracle_laser_pop_scan(void* hv) { return ((oracle_laser_ctx*)hv)->mgr->pop_scan() ? 1 : 0; }
void oracle_laser_clear_all_scan(void* hv) { ((oracle_laser_ctx*)hv)->mgr->clear_all_scan(); }
int oracle_laser_num_keyframes(void* hv) { return (int)((oracle_laser_ctx*)hv)->mgr->key_frame.size(); }
void* oracle_laser_ref_scan(void* hv, double* p3, double* q3) {   // new handle (caller destroys) or NULL
    laser_manager* m = ((oracle_laser_ctx*)hv)->mgr;
    if (!m->ref_submap_ptr) return nullptr;
    for (int k = 0; k < 3; ++k) { if (p3) p3[k] = m->ref_submap_ptr->current_p(k); if (q3) q3[k] = m->ref_submap_ptr->current_q(k); }
    return new oracle_scan_h{m->ref_submap_ptr->scan_ptr};
}

int oracle_laser_to_points(const float* ranges, int n, float angle_min, float angle_increment, float time_increment, double stamp, double* points,
                           double* times) {
    std::vector<Vec3<double>> pts;
    std::vector<double> ts;
    laser_to_point_times(ranges, n, angle_min, angle_increment, time_increment, stamp, pts, ts);
    for (size_t i = 0; i < pts.size(); ++i) { points[3 * i] = pts[i](0); points[3 * i + 1] = pts[i](1); points[3 * i + 2] = pts[i](2); times[i] = ts[i]; }
    return (int)pts.size();
}
void oracle_laser_correct(double* points, const double* times, int n, double stamp, const double* lin, const double* ang) {
    std::vector<Vec3<double>> pts;
    std::vector<double> ts(times, times + n);
    for (int i = 0; i < n; ++i) pts.push_back(v3(points + 3 * i));
    laser_correct(pts, ts, stamp, v3(lin), v3(ang));
    for (int i = 0; i < n; ++i) { points[3 * i] = pts[i](0); points[3 * i + 1] = pts[i](1); points[3 * i + 2] = pts[i](2); }
}

}  // extern "C"

// ---- on-disk formats (io_formats.h)
#include "io_formats.h"
extern "C" {
int oracle_tum_line(const double* T_iw16, int normalize, double time, const double* p, const double* q, char* buf, int cap) {
    Iso3<double> T;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T.R(i, j) = T_iw16[i * 4 + j]; T.t(i) = T_iw16[i * 4 + 3]; }
    if (normalize) lie::normalize_tf<double>(T);
    const std::string s = tum_line(T, time, Vec3<double>(p[0], p[1], p[2]), Vec3<double>(q[0], q[1], q[2]));
    if (buf && cap > 0) { const size_t k = std::min((size_t)cap - 1, s.size()); std::memcpy(buf, s.data(), k); buf[k] = 0; }
    return (int)s.size();
}
void* oracle_record_create() { return new record(); }
void oracle_record_destroy(void* r) { delete (record*)r; }
void oracle_record_add_time(void* r, const char* name, unsigned long long us) { ((record*)r)->add_time(name, us); }
void oracle_record_add(void* r, const char* name, unsigned long long v) { ((record*)r)->add_record(name, v); }
int oracle_record_dump(void* r, char* buf, int cap) {
    const std::string s = ((record*)r)->dump();
    if (buf && cap > 0) { const size_t k = std::min((size_t)cap - 1, s.size()); std::memcpy(buf, s.data(), k); buf[k] = 0; }
    return (int)s.size();
}
}
