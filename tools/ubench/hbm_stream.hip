// Chip-wide read rate of the laser slab kernel's access pattern (gfx950): W waves per CU, every wave streams its own run of ROWS rows of
// 4 KiB (64 lanes), R rows in flight in registers, as  (a) 8 loads of 8 bytes per lane (planes of 512 B; k_lin_laser_slab until round 5),
// (b) 4 loads of 16 bytes per lane (pair planes of 1 KiB).  A wave does a handful of FMAs per row, so this is the memory side alone.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/hbm_stream tools/ubench/hbm_stream.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double __attribute__((ext_vector_type(2))) dbl2;
template <int WIDE, int R, int WPS>
__global__ __launch_bounds__(64, WPS) void k_stream(const double* src, int rows, double* out) {
    __shared__ double pad[WPS == 2 ? 2560 : 64];      // 20 kB of LDS per wave at two waves per SIMD: eight waves per CU like the laser kernel
    const int lane = threadIdx.x;
    const double* base = src + (size_t)blockIdx.x * rows * 512;
    double q[R][8];
    auto load = [&](double* d, int j) {
        const double* r = base + (size_t)j * 512;
        if constexpr (WIDE) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const dbl2 v = *reinterpret_cast<const dbl2*>(r + c * 128 + lane * 2); d[2 * c] = v.x; d[2 * c + 1] = v.y; }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) d[c] = r[c * 64 + lane];
        }
    };
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < R - 1; ++k) load(q[k], k < rows ? k : rows - 1);
    for (int j = 0; j < rows; j += R) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int nx = j + k + R - 1;
            load(q[(k + R - 1) % R], nx < rows ? nx : rows - 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc = __builtin_fma(q[k][c], 1.0000001, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    pad[lane] = acc;
    out[(size_t)blockIdx.x * 64 + lane] = acc + pad[63 - lane];
}
template <int WIDE, int R, int WPS> void run(const double* src, double* out, int waves, int rows) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_stream<WIDE, R, WPS>), dim3(waves), dim3(64), 0, 0, src, rows, out);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double bytes = (double)waves * rows * 4096;
    printf("%s rows in flight %d, %d waves/SIMD: %.3f ms  %.0f GB/s\n", WIDE ? "16 B/lane" : " 8 B/lane", R, WPS, best, bytes / best * 1e-6);
}
// Mixed stream: a wave reads RD rows and writes WR rows of 4 KiB per round (16 bytes per lane, consecutive addresses; NT: nontemporal
// hint on loads and stores) -- the practical chip-wide rate of a read / write mix, which is what bounds the record-passing kernels.
template <int RD, int WR, int NT>
__global__ __launch_bounds__(64, 2) void k_mix(const double* src, double* dst, int rounds, double* out) {
    __shared__ double pad[2560];
    const int lane = threadIdx.x;
    const dbl2* rp = reinterpret_cast<const dbl2*>(src + (size_t)blockIdx.x * rounds * RD * 512) + lane;
    dbl2* wp = reinterpret_cast<dbl2*>(dst + (size_t)blockIdx.x * rounds * WR * 512) + lane;
    dbl2 acc = {0.0, 0.0};
    dbl2 q[2][RD * 4];
    auto load = [&](dbl2* d, int r) {
#pragma unroll
        for (int c = 0; c < RD * 4; ++c) d[c] = NT ? __builtin_nontemporal_load(rp + (size_t)r * RD * 256 + c * 64) : rp[(size_t)r * RD * 256 + c * 64];
    };
    load(q[0], 0);
    for (int r = 0; r < rounds; r += 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            load(q[1 - k], r + k + 1 < rounds ? r + k + 1 : rounds - 1);
#pragma unroll
            for (int c = 0; c < RD * 4; ++c) acc += q[k][c];
#pragma unroll
            for (int c = 0; c < WR * 4; ++c) {
                dbl2* a = wp + (size_t)(r + k) * WR * 256 + c * 64;
                if (NT) __builtin_nontemporal_store(acc, a); else *a = acc;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    pad[lane] = acc.x + acc.y;
    out[(size_t)blockIdx.x * 64 + lane] = pad[63 - lane];
}
template <int RD, int WR, int NT> void run_mix(const double* src, double* dst, double* out, int waves, int rounds) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mix<RD, WR, NT>), dim3(waves), dim3(64), 0, 0, src, dst, rounds, out);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double bytes = (double)waves * rounds * (RD + WR) * 4096;
    printf("mix %d read : %d written%s: %.3f ms  %.0f GB/s\n", RD, WR, NT ? " (nontemporal)" : "", best, bytes / best * 1e-6);
}
int main() {
    const int waves = 22272, rows = 67;
    double* src; double* out;
    (void)hipMalloc(&src, (size_t)waves * rows * 4096); (void)hipMemset(src, 0, (size_t)waves * rows * 4096);
    (void)hipMalloc(&out, (size_t)waves * 512);
    run<0, 2, 2>(src, out, waves, rows); run<1, 2, 2>(src, out, waves, rows);
    run<0, 3, 2>(src, out, waves, rows); run<1, 3, 2>(src, out, waves, rows);
    run<0, 4, 2>(src, out, waves, rows); run<1, 4, 2>(src, out, waves, rows);
    run<0, 6, 2>(src, out, waves, rows); run<1, 6, 2>(src, out, waves, rows);
    run<0, 2, 4>(src, out, waves, rows); run<1, 2, 4>(src, out, waves, rows);
    run<0, 4, 4>(src, out, waves, rows); run<1, 4, 4>(src, out, waves, rows);
    run<1, 4, 8>(src, out, waves, rows);
    double* dst; (void)hipMalloc(&dst, (size_t)waves * rows * 4096);
    run_mix<4, 1, 0>(src, dst, out, waves, 16); run_mix<4, 1, 1>(src, dst, out, waves, 16);
    run_mix<2, 1, 0>(src, dst, out, waves, 22); run_mix<2, 1, 1>(src, dst, out, waves, 22);
    run_mix<1, 1, 0>(src, dst, out, waves, 33); run_mix<1, 1, 1>(src, dst, out, waves, 33);
    run_mix<1, 2, 0>(src, dst, out, waves, 22); run_mix<1, 2, 1>(src, dst, out, waves, 22);
    return 0;
}
