// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement with the reference's own iostream calls of
//   the TUM trajectory line   src/trajectory/trajectory.cpp:59-67 (std::fixed, setprecision(10)), :549-559
//   record::~record           src/utilies/record.h:19-93 (two markdown tables)
#pragma once
#include <iomanip>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "lie.h"

namespace oracle {

inline std::string tum_header() { return "#Time px py pz qx qy qz qw\n"; }
inline std::string tum_line(const Iso3<double>& T_imu_to_wheel, double time, const Vec3<double>& current_p, const Vec3<double>& current_q) {
    std::ostringstream o_fstream;
    o_fstream << std::fixed;
    o_fstream << std::setprecision(10);
    Iso3<double> current_tf = lie::make_tf(current_p, current_q);
    Iso3<double> current_tf_base = current_tf * T_imu_to_wheel;
    Quat<double> q = rotmat_to_quat(current_tf_base.R);
    double x = current_tf_base.t(0), y = current_tf_base.t(1), z = current_tf_base.t(2);
    o_fstream << time << " " << x << " " << y << " " << z << " " << q.x << " " << q.y << " " << q.z << " " << q.w << std::endl;
    return o_fstream.str();
}

struct record {
    std::map<std::string, std::vector<uint64_t>> time_recorder, others_recorder;
    void add_time(const std::string& type_name, uint64_t dt) { time_recorder[type_name].push_back(dt); }
    void add_record(const std::string& type_name, uint64_t v) { others_recorder[type_name].push_back(v); }
    static void table(std::ostream& of, std::map<std::string, std::vector<uint64_t>>& rec) {
        for (auto& [type_name, records] : rec) {
            of << "| " << type_name << " | " << records.size() << " | ";
            uint64_t total = 0, max = 0, min = 999999999;
            for (auto item : records) {
                total += item;
                if (item > max) max = item;
                if (item < min) min = item;
            }
            double aver = 0, variance = 0;
            if (records.size() > 0) aver = double(total) / records.size();
            for (auto item : records) variance += (item - aver) * (item - aver);
            variance /= records.size();
            of << max << " |" << min << " |" << aver << " |" << variance << " |" << std::endl;
        }
    }
    std::string dump() {
        std::ostringstream of;
        of << "time_recorder" << std::endl;
        of << "size of total record type:" << time_recorder.size() << std::endl << std::endl;
        of << "| type name | record size | max(us) | min(us) | aver(us) | variance(${us}^2$) |" << std::endl << "| --- | --- | --- | --- | --- | --- |" << std::endl;
        table(of, time_recorder);
        of << std::endl;
        of << "others_recorder" << std::endl;
        of << "size of total record type:" << others_recorder.size() << std::endl << std::endl;
        of << "| type name | record size | max | min | aver | variance |" << std::endl << "| --- | --- | --- | --- | --- | --- |" << std::endl;
        table(of, others_recorder);
        return of.str();
    }
};

}  // namespace oracle
