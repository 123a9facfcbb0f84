/* liw_laser.h — C ABI of the 2D laser front-end that feeds the sliding-window estimator.
 *
 * Host-side (CPU) replacement of the step immediately before the hot path: it turns a 2D scan into line
 * segments, keeps the reference sub-maps, and produces the `laser_match` whose (lines1, lines2) pairs are the
 * laser_factor blocks of liw_window.laser_pts.  Stands in for
 *   convert::laser_to_point_times        reference src/utilies/common.cpp:5-40
 *   sensor::laser::correct (de-skew)     reference src/trajectory/sensor.h:51-94
 *   lvio_2d::line / scan / scan::add_line reference src/trajectory/laser_type.h:13-61, laser_manager.cpp:126-223
 *   laser_manager::spawn_scan            reference src/trajectory/laser_manager.cpp:350-422
 *   laser_manager::do_match              reference src/trajectory/laser_manager.cpp:244-348
 *   laser_manager::add_scan / match_with_{front,back,ref} / pop_scan / clear_all_scan   :424-565
 * Conventions as liw_window.h: fp64, row-major, plain pointers; points are [N][3] in the LASER frame;
 * poses are (p, q = rotation vector) of the IMU in the world, as in frame_info.  The grid `line_map` is sparse
 * here (hash of occupied cells), lines are addressed by their index in scan::lines.
 */
#ifndef LIW_LASER_H
#define LIW_LASER_H
#include "liw_window.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the laser part of param::manager (reference src/utilies/params.h, config/office.yaml:78-122) */
typedef struct liw_laser_params {
    double w_laser_each_scan, h_laser_each_scan, laser_resolution;
    double line_continuous_threshold, line_min_len, line_max_dis;
    double line_max_tolerance_angle;               /* degrees, as in the YAML */
    double ref_motion_filter_p, ref_motion_filter_q;
    int ref_n_accumulation;
    double T_imu_to_laser[16];                     /* 4x4 row-major, as in the YAML */
    int normalize_extrinsics;                      /* 1: quaternion round trip like src/utilies/params.cpp:44-54 */
} liw_laser_params;

/* LaserScan -> points (+ per-point stamps).  float arithmetic on the angles like the reference; keeps a range if it
 * is finite and > 0.1 and the point is >= 0.01 m away from the previously kept one.  Returns the number of points
 * written (<= n_ranges); points [n_ranges][3], times [n_ranges]. */
int liw_laser_to_points(const float* ranges, int n_ranges, float angle_min, float angle_increment, float time_increment,
                        double stamp, double* points, double* times);
/* de-skew in place with the body twist (linear, angular) at `stamp`: p_i <- make_tf(dt_i*linear, dt_i*angular) * p_i */
void liw_laser_correct(double* points, const double* times, int n, double stamp, const double* linear3, const double* angular3);

typedef struct liw_scan liw_scan;           /* lvio_2d::scan (shared ownership: the manager keeps its own reference) */
liw_scan* liw_scan_spawn(const liw_laser_params* prm, const double* points, int n_points, double time);
liw_scan* liw_scan_create_empty(const liw_laser_params* prm, double time);
/* scan::add_line(p1, p2, add_concers) — the segment overload the sub-maps use; returns 1 if a line was registered */
int liw_scan_add_segment(liw_scan* s, const double* p1, const double* p2, int add_concers);
void liw_scan_destroy(liw_scan* s);
int liw_scan_num_lines(const liw_scan* s);
void liw_scan_get_lines(const liw_scan* s, double* out /* [num_lines][10] = p1 p2 abc len */);
int liw_scan_num_concers(const liw_scan* s);
void liw_scan_get_concers(const liw_scan* s, double* out /* [num_concers][3] */);
/* line_map(r, c) of the cell containing (x, y): writes up to cap line indices (push order), returns the cell's size
 * (-1 if the cell is outside the grid) */
int liw_scan_cell_lines(const liw_scan* s, double x, double y, int* ids, int cap);

typedef struct liw_laser_match liw_laser_match;   /* lvio_2d::laser_match */
/* laser_manager::do_match(scan1, scan2, p1, q1, p2, q2, kk) (static in the reference) */
liw_laser_match* liw_laser_do_match(const liw_laser_params* prm, const liw_scan* scan1, const liw_scan* scan2, const double* p1,
                                    const double* q1, const double* p2, const double* q2, int kk);
void liw_laser_match_destroy(liw_laser_match* m);
int liw_laser_match_size(const liw_laser_match* m);
/* pts [size][12] = lines1[j].p1, lines1[j].p2, lines2[j].p1, lines2[j].p2 (the liw_window.laser_pts record);
 * pose12 = p1 q1 p2 q2; idx1 / idx2 [size] = indices into scan1->lines / scan2->lines (any may be NULL) */
void liw_laser_match_get(const liw_laser_match* m, double* pts, double* pose12, int* idx1, int* idx2);

typedef struct liw_laser_manager liw_laser_manager;   /* lvio_2d::laser_manager */
liw_laser_manager* liw_laser_manager_create(const liw_laser_params* prm);
void liw_laser_manager_destroy(liw_laser_manager* m);
void liw_laser_manager_add_scan(liw_laser_manager* m, liw_scan* scan, const double* p, const double* q);
liw_laser_match* liw_laser_manager_match_with_front(liw_laser_manager* m, const liw_scan* scan, const double* p, const double* q);
liw_laser_match* liw_laser_manager_match_with_back(liw_laser_manager* m, const liw_scan* scan, const double* p, const double* q);
liw_laser_match* liw_laser_manager_match_with_ref(liw_laser_manager* m, const liw_scan* scan, const double* p, const double* q);
int liw_laser_manager_pop_scan(liw_laser_manager* m);            /* 1 if a key frame was popped */
void liw_laser_manager_clear_all_scan(liw_laser_manager* m);
int liw_laser_manager_num_keyframes(const liw_laser_manager* m);
/* get_keyframs()[i]->current_p / current_q = p, q (what trajectory.cpp:452-461 does after init_solve) */
int liw_laser_manager_set_keyframe_pose(liw_laser_manager* m, int i, const double* p, const double* q);
/* the reference sub-map's scan (NULL before the first add_scan); borrowed, valid until the next add_scan / clear */
const liw_scan* liw_laser_manager_ref_scan(const liw_laser_manager* m, double* p3, double* q3);

#ifdef __cplusplus
}
#endif
#endif
