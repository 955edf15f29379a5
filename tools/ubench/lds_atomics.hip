// Microbenchmark: LDS atomics as cross-lane reductions (gfx950): cost of ds_add_f64 / ds_min_f64 / ds_min_u32 from k lanes of one wave to ONE address
// (issue -> the wave reads the result back), and whether ds_add_f64's result is the sum in LANE ORDER (deterministic) over many repetitions.
//   make -C tools/ubench && gpurun -- tools/ubench/_build/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

__global__ __launch_bounds__(256) void k(const double* in, double* out, long long* cyc, int nact)
{
    __shared__ __align__(16) double lds[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 512; i += 256) lds[i] = 0.0;
    __syncthreads();
    double v = in[tid];
    long long t0, t1;
    const int a0 = (int)(unsigned long long)(&lds[16 * wave]); // one address per wave (the low half of a generic LDS pointer is the LDS offset)
    double r = 0.0;
    // 0: ds_add_f64 from nact lanes, then read back, x16
    t0 = clock64();
    if (lane < nact) {
        asm volatile(REP16("ds_add_f64 %1, %0\n") : : "v"(v), "v"(a0));
    }
    asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(a0));
    t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
    if (lane == 0) out[wave] = r;
    // 1: ds_min_f64 x16 + read
    const int a1 = a0 + 8;
    t0 = clock64();
    if (lane < nact) {
        asm volatile(REP16("ds_min_f64 %1, %0\n") : : "v"(v), "v"(a1));
    }
    asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(a1));
    t1 = clock64();
    if (tid == 0) cyc[1] = t1 - t0;
    if (lane == 0) out[4 + wave] = r;
    // 2: ds_min_u32 x16 + read
    const int a2 = a0 + 16;
    int key = 1000 - lane, ri = 0;
    t0 = clock64();
    if (lane < nact) {
        asm volatile(REP16("ds_min_u32 %1, %0\n") : : "v"(key), "v"(a2));
    }
    asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "=v"(ri) : "v"(a2));
    t1 = clock64();
    if (tid == 0) cyc[2] = t1 - t0;
    // 3: ONE ds_min_f64 + read back + ONE ds_min_u32 by the lanes that hold the minimum + read back (the argmin sequence)
    const int a3 = a0 + 24, a4 = a0 + 32;
    if (lane == 0) { lds[16 * wave + 3] = 1e300; reinterpret_cast<unsigned*>(lds)[2 * (16 * wave + 4)] = 0xffffffffu; }
    __syncthreads();
    t0 = clock64();
    double m;
    if (lane < nact) asm volatile("ds_min_f64 %1, %0\n" : : "v"(v), "v"(a3));
    asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "=v"(m) : "v"(a3));
    if (lane < nact && v == m) asm volatile("ds_min_u32 %1, %0\n" : : "v"(lane), "v"(a4));
    asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "=v"(ri) : "v"(a4));
    t1 = clock64();
    if (tid == 0) cyc[3] = t1 - t0;
    if (lane == 0) { out[8 + wave] = m; out[12 + wave] = ri; }
}

int main()
{
    double *in, *out; long long* cyc;
    hipMalloc(&in, 256 * 8); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
    std::vector<double> h(256);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((double)(s >> 8) / (1 << 24) - 0.5) * 1e3; }
    hipMemcpy(in, h.data(), 256 * 8, hipMemcpyHostToDevice);
    for (int nact : {64, 32, 16, 4}) {
        double first[16]; bool same = true, inorder = true;
        long long c[4];
        for (int rep = 0; rep < 2000; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, in, out, cyc, nact);
            double o[16];
            hipMemcpy(o, out, 16 * 8, hipMemcpyDeviceToHost);
            if (rep == 0) memcpy(first, o, sizeof o);
            else if (memcmp(first, o, sizeof o) != 0) same = false;
        }
        hipMemcpy(c, cyc, 32, hipMemcpyDeviceToHost);
        // lane-order reference of the 16-fold add of wave 0
        double ref = 0.0;
        for (int r = 0; r < 16; ++r) for (int l = 0; l < nact; ++l) ref += h[l];
        double ref2 = 0.0; // instruction-major alternative is the same order here (each instruction finishes before the next)
        inorder = (ref == first[0]);
        printf("%2d lanes: 16 ds_add_f64 + read %5lld cycles, 16 ds_min_f64 + read %5lld, 16 ds_min_u32 + read %5lld, argmin (min_f64, read, min_u32, read) %5lld; "
               "ds_add_f64 bit-identical over 2000 launches: %s, equals the lane-order sum: %s (%.17g vs %.17g)\n",
               nact, c[0], c[1], c[2], c[3], same ? "yes" : "NO", inorder ? "yes" : "no", first[0], ref);
        (void)ref2;
        printf("      out:"); for (int i = 0; i < 16; ++i) printf(" %g", first[i]); printf("\n");
    }
    return 0;
}
