// DP-ALU DPP micro-benchmark (gfx950): v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast — the only DPP control the 64-bit ALU takes.
// One instruction computes acc += (-)bcast_row(src, lane N) * mul for every lane of a 16-lane row, i.e. four independent 16-lane
// problems per wave with no v_readlane.  Checks (1) the semantics incl. the neg modifier, (2) whether a VALU write -> DPP read of the
// same VGPR needs explicit wait states when the DPP instruction sits in inline asm (the hazard recogniser does not look inside),
// (3) issue rate of independent DPP-FMAs and latency of a dependent chain.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dpp64 tools/ubench/dpp64.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define TICK(t, var) asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(var) :: "memory")

template <int L> __device__ __forceinline__ void fnma_bc(double& acc, double src, double mul) {   // acc -= src@lane L of the row * mul
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}
template <int L> __device__ __forceinline__ double bc(double src) {
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(L));
    return r;
}

__global__ void k_sem(const double* in, double* out) {
    const int lane = threadIdx.x;
    const double w = in[lane], m = in[64 + lane];
    double a0 = in[128 + lane], a1 = a0, a2 = a0, a3 = a0;
    fnma_bc<3>(a0, w, m);
    fnma_bc<15>(a1, w, m);
    fnma_bc<0>(a2, w, m);
    a3 = bc<7>(w);
    out[lane] = a0; out[64 + lane] = a1; out[128 + lane] = a2; out[192 + lane] = a3;
    // hazard probe: the DPP source is produced by the VALU instruction immediately before
    double h0 = in[128 + lane], h1 = h0;
    {
        double t;
        asm volatile("v_mul_f64 %0, %2, %3\n v_fmac_f64_dpp %1, -%0, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                     : "=&v"(t), "+v"(h0) : "v"(w), "v"(m));
    }
    {
        double t;
        asm volatile("v_mul_f64 %0, %2, %3\n s_nop 1\n v_fmac_f64_dpp %1, -%0, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                     : "=&v"(t), "+v"(h1) : "v"(w), "v"(m));
    }
    out[256 + lane] = h0; out[320 + lane] = h1;
}

__global__ void k_rate(double* out, long long* clk, double x0) {
    const int lane = threadIdx.x;
    double w = x0 + 1e-3 * lane, m = 1.0 + 1e-9 * lane;
    double a[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) a[r] = x0 + r;
    long long t0, t1;
    TICK(t0, w);
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        fnma_bc<0>(a[0], w, m); fnma_bc<1>(a[1], w, m); fnma_bc<2>(a[2], w, m); fnma_bc<3>(a[3], w, m); fnma_bc<4>(a[4], w, m);
        fnma_bc<5>(a[5], w, m); fnma_bc<6>(a[6], w, m); fnma_bc<7>(a[7], w, m); fnma_bc<8>(a[8], w, m); fnma_bc<9>(a[9], w, m);
        fnma_bc<10>(a[10], w, m); fnma_bc<11>(a[11], w, m); fnma_bc<12>(a[12], w, m); fnma_bc<13>(a[13], w, m); fnma_bc<14>(a[14], w, m);
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 15; ++r) s += a[r];
    TICK(t1, s);
    if (lane == 0) clk[0] = t1 - t0;   // 480 independent-ish DPP FMAs (+ 15 adds)
    // dependent chain: acc feeds the next instruction's DPP source
    double c = x0;
    TICK(t0, c);
#pragma unroll
    for (int it = 0; it < 256; ++it) { double nw = c; asm volatile("s_nop 1\n v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(nw), "v"(m)); }
    TICK(t1, c);
    if (lane == 0) clk[1] = t1 - t0;
    // plain fma reference, 15 independent accumulators
#pragma unroll
    for (int r = 0; r < 15; ++r) a[r] = x0 + r;
    TICK(t0, w);
#pragma unroll
    for (int it = 0; it < 32; ++it) {
#pragma unroll
        for (int r = 0; r < 15; ++r) a[r] = __builtin_fma(-w, m, a[r]);
    }
    s = 0.0;
#pragma unroll
    for (int r = 0; r < 15; ++r) s += a[r];
    TICK(t1, s);
    if (lane == 0) clk[2] = t1 - t0;
    out[lane] = s + c;
}

int main() {
    double hin[192], hout[384];
    for (int i = 0; i < 192; ++i) hin[i] = 0.37 + 0.011 * i + 1e-3 * (i % 7);
    double *din, *dout; long long* dclk;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout)); hipMalloc(&dclk, 64);
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(din, dout);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    int bad = 0, haz0 = 0, haz1 = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15;
        const double w3 = hin[row + 3], w15 = hin[row + 15], w0 = hin[row], m = hin[64 + l], a = hin[128 + l];
        if (hout[l] != fma(-w3, m, a)) ++bad;
        if (hout[64 + l] != fma(-w15, m, a)) ++bad;
        if (hout[128 + l] != fma(-w0, m, a)) ++bad;
        if (hout[192 + l] != hin[row + 7]) ++bad;
        const double t5 = hin[row + 5] * hin[64 + row + 5];
        if (hout[256 + l] != fma(-t5, m, a)) ++haz0;
        if (hout[320 + l] != fma(-t5, m, a)) ++haz1;
    }
    printf("semantics mismatches: %d (of 256)\n", bad);
    printf("hazard probe: back-to-back VALU->DPP mismatches %d, with s_nop 1: %d\n", haz0, haz1);
    k_rate<<<1, 64>>>(dout, dclk, 1.25);
    long long hclk[4];
    hipMemcpy(hclk, dclk, 32, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx950 (see tools/ubench/lat.hip); report raw ticks and per-instruction ratios
    printf("480 independent DPP-FMA: %lld ticks; 256 dependent (with s_nop 1): %lld ticks; 480 plain FMA: %lld ticks\n", hclk[0], hclk[1], hclk[2]);
    printf("ratio DPP-FMA / FMA issue: %.3f\n", (double)hclk[0] / (double)hclk[2]);
    return bad ? 1 : 0;
}
