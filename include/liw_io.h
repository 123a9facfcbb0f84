/* liw_io.h — on-disk outputs of the front-end, C ABI, host code.
 *   TUM trajectory   reference src/trajectory/trajectory.cpp:59-67 (header, std::fixed, 10 decimals) and :549-559
 *                    (pose written = T_w_imu * T_imu_to_wheel, quaternion of its rotation, "time x y z qx qy qz qw")
 *   record           reference src/utilies/record.h:13-126: named duration / value series dumped as two markdown tables
 *                    ("time_recorder", "others_recorder": size, max, min, aver, variance) — used for the "solve" and
 *                    "marginalization" timing tables (trajectory.cpp:533-545).
 */
#ifndef LIW_IO_H
#define LIW_IO_H
#include <stdint.h>

#include "liw_window.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pose of the base: make_tf(p, q) * T_imu_to_wheel -> out7 = x y z qx qy qz qw (Eigen::Quaterniond(Matrix3d) branches) */
void liw_tum_pose(const liw_params* prm, const double* p3, const double* q3, double* out7);
/* one trajectory line "time x y z qx qy qz qw\n" with std::fixed / setprecision(10); returns the length (excluding NUL) */
int liw_tum_format_line(double time, const double* pose7, char* buf, int cap);

typedef struct liw_tum_writer liw_tum_writer;
/* opens <path> for writing and emits the "#Time px py pz qx qy qz qw" header; NULL if the file cannot be opened */
liw_tum_writer* liw_tum_open(const char* path, const liw_params* prm);
/* appends the base pose of an IMU state (p, q) at `time`; returns 0, or LIW_ESTATE if time does not increase
 * (the reference logs "error output time" and still writes the line — so does this) */
int liw_tum_append(liw_tum_writer* w, double time, const double* p3, const double* q3);
void liw_tum_close(liw_tum_writer* w);

typedef struct liw_record liw_record;
liw_record* liw_record_create(void);
void liw_record_destroy(liw_record* r);
void liw_record_begin(liw_record* r);                              /* record::begin_record (nestable) */
uint64_t liw_record_end(liw_record* r, const char* type_name);     /* record::end_record -> microseconds recorded */
void liw_record_add_time(liw_record* r, const char* type_name, uint64_t us);   /* a duration measured elsewhere (HIP events) */
void liw_record_add(liw_record* r, const char* type_name, uint64_t v);        /* record::add_record */
/* the markdown the destructor of `record` writes; returns the length needed (excluding NUL), writes at most cap bytes */
int liw_record_format(const liw_record* r, char* buf, int cap);
int liw_record_write(const liw_record* r, const char* path);       /* 0 or LIW_EINVAL */

#ifdef __cplusplus
}
#endif
#endif
