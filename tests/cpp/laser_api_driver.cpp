// Test driver for include/lvio_2d_laser.hpp: two scans (points dumped by the Python test) -> spawn_scan -> add_scan of
// the first -> match_with_front of the second, through the reference-named C++ classes; writes line and match records.
// File format (little-endian): int32 n1, n2; float64 pose1[6], pose2[6]; points1 [n1][3]; points2 [n2][3].
// Output: int32 nl1, nl2, nm; lines1 [nl1][6]; lines2 [nl2][6]; match [nm][12]; match pose [12].
#include <cstdio>
#include <vector>

#include "lvio_2d_laser.hpp"

static const double OFFICE_T_IMU_TO_LASER[16] = {0.0019070, -0.9999900, 0.0040438, 0.024, 0.0459794, -0.0039519, -0.9989346, -0.078,
                                                 0.9989406, 0.0020909, 0.0459714, -0.071, 0.0, 0.0, 0.0, 1.0};

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[2];
    double pose[12];
    if (fread(hdr, sizeof(int), 2, f) != 2 || fread(pose, sizeof(double), 12, f) != 12) return 2;
    std::vector<double> a((size_t)hdr[0] * 3), b((size_t)hdr[1] * 3);
    if (fread(a.data(), sizeof(double), a.size(), f) != a.size() || fread(b.data(), sizeof(double), b.size(), f) != b.size()) return 2;
    fclose(f);
    liw_laser_params prm{};   // config/office.yaml:78-122
    prm.w_laser_each_scan = 100.0; prm.h_laser_each_scan = 100.0; prm.laser_resolution = 0.05;
    prm.line_continuous_threshold = 0.1; prm.line_min_len = 0.05; prm.line_max_dis = 0.03; prm.line_max_tolerance_angle = 175.0;
    prm.ref_motion_filter_p = 0.01; prm.ref_motion_filter_q = 0.01; prm.ref_n_accumulation = 2;
    for (int k = 0; k < 16; ++k) prm.T_imu_to_laser[k] = OFFICE_T_IMU_TO_LASER[k];
    prm.normalize_extrinsics = 1;
    lvio_2d::laser_manager mgr(prm);
    lvio_2d::scan::ptr s1 = mgr.spawn_scan(a.data(), hdr[0], 0.0), s2 = mgr.spawn_scan(b.data(), hdr[1], 0.1);
    mgr.add_scan(s1, pose, pose + 3);
    lvio_2d::laser_match::ptr m = mgr.match_with_front(s2, pose + 6, pose + 9);
    lvio_2d::frame_info fi;
    fi.add_laser_match(m);   // what lvio_2d::trajectory does with it
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    const int cnt[3] = {(int)s1->lines.size(), (int)s2->lines.size(), (int)m->lines1.size()};
    fwrite(cnt, sizeof(int), 3, o);
    for (const auto& l : s1->lines) { fwrite(l.p1, sizeof(double), 3, o); fwrite(l.p2, sizeof(double), 3, o); }
    for (const auto& l : s2->lines) { fwrite(l.p1, sizeof(double), 3, o); fwrite(l.p2, sizeof(double), 3, o); }
    for (size_t i = 0; i < m->lines1.size(); ++i) {
        fwrite(m->lines1[i].p1, sizeof(double), 3, o); fwrite(m->lines1[i].p2, sizeof(double), 3, o);
        fwrite(m->lines2[i].p1, sizeof(double), 3, o); fwrite(m->lines2[i].p2, sizeof(double), 3, o);
    }
    fwrite(m->p1, sizeof(double), 3, o); fwrite(m->q1, sizeof(double), 3, o); fwrite(m->p2, sizeof(double), 3, o); fwrite(m->q2, sizeof(double), 3, o);
    fclose(o);
    return fi.type == lvio_2d::frame_info::laser ? 0 : 3;
}
