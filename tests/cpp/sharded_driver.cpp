// C++ host of the factor-sharded LM solve: libliw_window's liw_batch_solve_sharded driven with RCCL collectives — the
// sequence of INTEGRATION.md 5, compiled.  One process per rank (rank / world / a shared ncclUniqueId file on the command line); every
// rank loads the same batch dump, keeps its contiguous share of every window's laser blocks (the small factors are evaluated by every
// rank), and runs
//     plain        liw_batch_solve                                  (world = 1 reference: the un-sharded path)
//     allreduce    liw_batch_solve_sharded + ncclAllReduce(ncclDouble, ncclSum) on the compact record
//     allgather    liw_batch_solve_sharded + ncclAllGather into `world` images, summed in rank order by liw_batch_exchange_unpack
// and writes states + summaries of each variant.  At world = 1 the three must agree bit for bit (tests/test_gpu_cpp_sharded.py).
// Batch dump (little-endian): int32 B, n; then per window: int32 L, and the arrays of tests/test_cpp_host.py::dump_window without its header.
// usage: sharded_driver <batch.bin> <out.bin> <rank> <world> <id_file> [iters]     exit: 0 ok, 3 = RCCL refused the communicator
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "liw_window.h"

static const double OFFICE_T_IMU_TO_WHEEL[16] = {0.0040697, -0.9998940, -0.0139789, -0.061, 0.0099712, 0.0140189, -0.9998520, 0.919,
                                                 0.9999420, 0.0039297, 0.0100272, -0.224, 0.0, 0.0, 0.0, 1.0};
static const double OFFICE_T_IMU_TO_LASER[16] = {0.0019070, -0.9999900, 0.0040438, 0.024, 0.0459794, -0.0039519, -0.9989346, -0.078,
                                                 0.9989406, 0.0020909, 0.0459714, -0.071, 0.0, 0.0, 0.0, 1.0};
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(4); } } while (0)
#define LIWOK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, r_, liw_last_error(ctx)); exit(r_ < 0 ? -r_ : r_); } } while (0)

template <class T> static std::vector<T> rd(FILE* f, size_t cnt) {
    std::vector<T> v(cnt);
    if (cnt && fread(v.data(), sizeof(T), cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}
template <class T> static T* up(const std::vector<T>& h) {
    T* d = nullptr;
    HIPOK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIPOK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

struct Xch { ncclComm_t comm; int world; bool gather; };
static int exchange_cb(void* user, double* buf, double* all, size_t doubles, void* stream) {
    Xch* x = (Xch*)user;
    if (x->gather) {
        if (ncclAllGather(buf, all, doubles, ncclDouble, x->comm, (hipStream_t)stream) != ncclSuccess) return -1;
        return x->world;
    }
    if (ncclAllReduce(buf, buf, doubles, ncclDouble, ncclSum, x->comm, (hipStream_t)stream) != ncclSuccess) return -1;
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const int rank = atoi(argv[3]), world = atoi(argv[4]);
    const int iters = argc > 6 ? atoi(argv[6]) : 50;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[2];
    if (fread(hdr, sizeof(int), 2, f) != 2) return 2;
    const int B = hdr[0], n = hdr[1];
    std::vector<double> x, match_pose, imu_X, imu_J, imu_P, imu_Dt, wheel_T, wheel_P;
    std::vector<unsigned char> has_match;
    std::vector<int> laser_off(1, 0), laser_frame;
    std::vector<std::vector<double>> pts_rows;   // per kept block: 12 doubles
    for (int b = 0; b < B; ++b) {
        int L;
        if (fread(&L, sizeof(int), 1, f) != 1) return 2;
        auto st = rd<double>(f, n * 15);
        auto lf = rd<int>(f, L);
        auto lp = rd<double>(f, (size_t)L * 12);
        auto mp = rd<double>(f, n * 12);
        auto hm = rd<unsigned char>(f, n);
        auto iX = rd<double>(f, (n - 1) * 15), iJ = rd<double>(f, (n - 1) * 225), iP = rd<double>(f, (n - 1) * 225), iD = rd<double>(f, n - 1);
        auto wT = rd<double>(f, (n - 1) * 12), wP = rd<double>(f, (n - 1) * 9), wD = rd<double>(f, n - 1);
        x.insert(x.end(), st.begin(), st.end());
        match_pose.insert(match_pose.end(), mp.begin(), mp.end());
        has_match.insert(has_match.end(), hm.begin(), hm.end());
        imu_X.insert(imu_X.end(), iX.begin(), iX.end()); imu_J.insert(imu_J.end(), iJ.begin(), iJ.end());
        imu_P.insert(imu_P.end(), iP.begin(), iP.end()); imu_Dt.insert(imu_Dt.end(), iD.begin(), iD.end());
        wheel_T.insert(wheel_T.end(), wT.begin(), wT.end()); wheel_P.insert(wheel_P.end(), wP.begin(), wP.end());
        // this rank's contiguous share of the window's blocks (2dliw-slam_amd/batch.py shard_laser: blocks stay sorted by owning frame)
        const int lo = (int)((long)L * rank / world), hi = (int)((long)L * (rank + 1) / world);
        for (int j = lo; j < hi; ++j) {
            laser_frame.push_back(lf[j]);
            pts_rows.emplace_back(lp.begin() + (size_t)j * 12, lp.begin() + (size_t)j * 12 + 12);
        }
        laser_off.push_back((int)laser_frame.size());
    }
    fclose(f);
    const int Ltot = (int)laser_frame.size();
    std::vector<double> pts_soa((size_t)12 * std::max(Ltot, 1));   // component-major, as liw_batch wants it
    for (int j = 0; j < Ltot; ++j)
        for (int c = 0; c < 12; ++c) pts_soa[(size_t)c * Ltot + j] = pts_rows[j][c];

    HIPOK(hipSetDevice(0));
    liw_params prm{};
    memcpy(prm.T_imu_to_wheel, OFFICE_T_IMU_TO_WHEEL, sizeof prm.T_imu_to_wheel);
    memcpy(prm.T_imu_to_laser, OFFICE_T_IMU_TO_LASER, sizeof prm.T_imu_to_laser);
    prm.g = 9.8; prm.line_to_line_sigma = 0.001; prm.manifold_p_sigma = 0.01; prm.manifold_q_sigma = 0.0005;
    for (int k = 0; k < 3; ++k) {
        prm.imu_noise_acc_sigma[k] = 0.0163; prm.imu_bias_acc_sigma[k] = 0.00499;
        prm.imu_noise_gyro_sigma[k] = 0.003208; prm.imu_bias_gyro_sigma[k] = 0.000499;
    }
    prm.wheel_sigma[0] = 0.5; prm.wheel_sigma[1] = 99999.0; prm.wheel_sigma[2] = 999.99;
    prm.fast_mode = 0; prm.normalize_extrinsics = 1; prm.device = 0;
    liw_ctx* ctx = liw_create(&prm);
    if (!ctx) { fprintf(stderr, "liw_create failed\n"); return 19; }

    // ---- the communicator: rank 0 writes the unique id, the others wait for the file
    ncclUniqueId id;
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) { fprintf(stderr, "ncclGetUniqueId failed\n"); return 3; }
        std::string tmp = std::string(argv[5]) + ".tmp";
        FILE* g = fopen(tmp.c_str(), "wb");
        fwrite(&id, sizeof id, 1, g);
        fclose(g);
        rename(tmp.c_str(), argv[5]);
    } else {
        FILE* g = nullptr;
        for (int t = 0; t < 600 && !(g = fopen(argv[5], "rb")); ++t) std::this_thread::sleep_for(std::chrono::milliseconds(50));
        if (!g || fread(&id, sizeof id, 1, g) != 1) { fprintf(stderr, "no unique id\n"); return 3; }
        fclose(g);
    }
    ncclComm_t comm = nullptr;
    ncclResult_t cr = ncclCommInitRank(&comm, world, id, rank);
    if (cr != ncclSuccess) {   // e.g. two ranks on one device: RCCL refuses ("duplicate GPU"), the caller asserts this exit code
        fprintf(stderr, "ncclCommInitRank refused: %s\n", ncclGetErrorString(cr));
        return 3;
    }

    liw_batch bt{};
    bt.B = B; bt.n = n; bt.Ltot = Ltot;
    double* dx = up(x);
    bt.x = dx;
    bt.laser_off = up(laser_off); bt.laser_frame = up(laser_frame); bt.laser_pts = up(pts_soa);
    double* dmp = up(match_pose);
    bt.match_pose = dmp; bt.has_match = up(has_match);
    bt.imu_X = up(imu_X); bt.imu_J = up(imu_J); bt.imu_sqrtP = up(imu_P); bt.imu_Dt = up(imu_Dt);
    bt.wheel_T = up(wheel_T); bt.wheel_sqrtP = up(wheel_P);
    std::vector<double> z15((size_t)B * 15, 0.0), z225((size_t)B * 225, 0.0);
    std::vector<int> zi(B, 0);
    bt.prior_X = up(z15); bt.prior_J = up(z225); bt.prior_R = up(z15); bt.has_prior = up(zi);
    bt.eval_small = 1; bt.history_records = 0;
    liw_ws_layout lay{};
    LIWOK(liw_batch_ws_layout(B, n, 0, &lay));
    void* ws = nullptr;
    HIPOK(hipMalloc(&ws, lay.bytes));
    HIPOK(hipMemset(ws, 0, lay.bytes));
    const size_t nd = (size_t)liw_batch_exchange_doubles(B, n, LIW_MODE_INIT);
    double *xbuf = nullptr, *xall = nullptr;
    HIPOK(hipMalloc(&xbuf, nd * sizeof(double)));
    HIPOK(hipMalloc(&xall, nd * sizeof(double) * world));
    hipStream_t s;
    HIPOK(hipStreamCreate(&s));

    FILE* out = fopen(argv[2], "wb");
    if (!out) return 2;
    const int variants = world == 1 ? 3 : 2;
    fwrite(&variants, sizeof(int), 1, out);
    std::vector<double> xs(x.size());
    std::vector<liw_summary> sm(B);
    for (int v = 0; v < 3; ++v) {
        if (v == 0 && world > 1) continue;       // the un-sharded path only makes sense with every block on this rank
        HIPOK(hipMemcpy(dx, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(dmp, match_pose.data(), match_pose.size() * sizeof(double), hipMemcpyHostToDevice));
        if (v == 0) {
            LIWOK(liw_batch_solve(ctx, &bt, LIW_MODE_INIT, iters, ws, s, 0));
        } else {
            Xch xc{comm, world, v == 2};
            LIWOK(liw_batch_solve_sharded(ctx, &bt, LIW_MODE_INIT, iters, ws, s, xbuf, xall, world, exchange_cb, &xc));
        }
        HIPOK(hipStreamSynchronize(s));
        HIPOK(hipMemcpy(xs.data(), dx, xs.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(sm.data(), (char*)ws + lay.info_off, sizeof(liw_summary) * B, hipMemcpyDeviceToHost));
        fwrite(&v, sizeof(int), 1, out);
        fwrite(xs.data(), sizeof(double), xs.size(), out);
        for (int b = 0; b < B; ++b) { int t[3] = {sm[b].iterations, sm[b].termination, sm[b].successful_steps}; fwrite(t, sizeof(int), 3, out); }
    }
    fclose(out);
    ncclCommDestroy(comm);
    liw_destroy(ctx);
    return 0;
}
