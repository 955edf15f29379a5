// Microbenchmark: DEPENDENT-chain latencies of the instructions the active-set loop's critical path is made of (gfx950), one wave
// alone on its SIMD.  Cycles per link of a chain of N dependent instructions, s_memtime around the chain.
//   make -C tools/ubench && gpurun -- tools/ubench/_build/chain_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int nwaves_active)
{
    __shared__ __align__(16) double lds[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    // pointer chase table: lds[i] as int holds the BYTE address of the next element (stride 24 doubles, wraps)
    int* li = reinterpret_cast<int*>(lds);
    if (tid == 0) for (int i = 0; i < 64; ++i) li[2 * ((i * 24) % 1536)] = 8 * (((i + 1) * 24) % 1536);
    __syncthreads();
    if ((tid >> 6) >= nwaves_active) return;
    double a = 1.0 + lane * 1e-9, x = 1.0000001, b = 0.5;
    long long t0, t1;
    int m = 0;
    auto rec = [&](long long d) { if (tid == 0) cyc[m] = d; ++m; };
    // 0: v_fma_f64 dependent
    t0 = clock64();
    asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(x), "v"(b));
    t1 = clock64(); rec(t1 - t0);
    // 1: v_add_f64 dependent
    t0 = clock64();
    asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(a) : "v"(b));
    t1 = clock64(); rec(t1 - t0);
    // 2: v_mul_f64 dependent
    t0 = clock64();
    asm volatile(REP64("v_mul_f64 %0, %0, %1\n") : "+v"(a) : "v"(x));
    t1 = clock64(); rec(t1 - t0);
    // 3: four independent fma chains interleaved (64 instructions)
    {
        double c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3;
        t0 = clock64();
        asm volatile(REP16("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n")
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(b));
        t1 = clock64(); rec(t1 - t0);
        a = (c0 + c1) + (c2 + c3);
    }
    // 4: DPP reduction step: two v_mov_dpp + v_add_f64, dependent, x16 (compiler-scheduled: the chain itself orders it)
    {
        t0 = clock64();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int lo = __double2loint(a), hi = __double2hiint(a);
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false);
            a = a + __hiloint2double(hi, lo);
        }
        asm volatile("" : "+v"(a));
        t1 = clock64(); rec(t1 - t0);
    }
    // 5: v_readlane_b32 -> v_mov_b32 -> v_readlane ... (SGPR round trip), x16 pairs
    {
        int v = lane, s = 0;
        t0 = clock64();
        asm volatile(REP16("v_readlane_b32 %1, %0, 3\ns_nop 3\nv_mov_b32 %0, %1\n") : "+v"(v), "+s"(s));
        t1 = clock64(); rec(t1 - t0);
        a += v;
    }
    // 6: v_rcp_f64 dependent x16
    t0 = clock64();
    asm volatile(REP16("v_rcp_f64 %0, %0\n") : "+v"(a));
    t1 = clock64(); rec(t1 - t0);
    // 7: v_rsq_f64 dependent x16
    a = fabs(a) + 1.0;
    t0 = clock64();
    asm volatile(REP16("v_rsq_f64 %0, %0\n") : "+v"(a));
    t1 = clock64(); rec(t1 - t0);
    // 8: ds_read_b64 pointer chase x16 (uniform address: broadcast)
    {
        int addr = 0;
        double v = 0.0;
        t0 = clock64();
        asm volatile(REP16("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(addr));
        t1 = clock64(); rec(t1 - t0);
        a += addr + v;
    }
    // 9: ds_write_b64 then ds_read_b64 of it, x16
    {
        int addr = 8 * (1600 + lane);
        t0 = clock64();
        asm volatile(REP16("ds_write_b64 %1, %0\nds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(addr));
        t1 = clock64(); rec(t1 - t0);
    }
    // 10: v_cmp_lt_f64 + s_cbranch_vccnz never taken, x16
    {
        double big = 1e300;
        t0 = clock64();
        asm volatile(REP16("v_cmp_lt_f64 vcc, %1, %0\ns_cbranch_vccnz 1f\n1:\n") : : "v"(a), "v"(big) : "vcc");
        t1 = clock64(); rec(t1 - t0);
    }
    // 11: compare + select of a double, dependent, x16
    t0 = clock64();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        a = (b < a) ? a * 1.0000001 : b;
        asm volatile("" : "+v"(a));
    }
    t1 = clock64(); rec(t1 - t0);
    // 12: s_barrier x16 (all active waves of the workgroup take part)
    t0 = clock64();
    asm volatile(REP16("s_waitcnt lgkmcnt(0)\ns_barrier\n"));
    t1 = clock64(); rec(t1 - t0);
    // 13: LDS exchange through a barrier: write, barrier, read another wave's value, x16
    {
        const int wr = 8 * (1700 + tid), rd = 8 * (1700 + ((tid + 64) & (nwaves_active * 64 - 1)));
        t0 = clock64();
        asm volatile(REP16("ds_write_b64 %1, %0\ns_waitcnt lgkmcnt(0)\ns_barrier\nds_read_b64 %0, %2\ns_waitcnt lgkmcnt(0)\ns_barrier\n") : "+v"(a) : "v"(wr), "v"(rd));
        t1 = clock64(); rec(t1 - t0);
    }
    // 14: 32 independent ds_read_b64 issued back to back, one wait
    {
        double r[8];
        const int base = 8 * lane;
        t0 = clock64();
        asm volatile(REP4("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:512\nds_read_b64 %2, %8 offset:1024\nds_read_b64 %3, %8 offset:1536\n"
                          "ds_read_b64 %4, %8 offset:2048\nds_read_b64 %5, %8 offset:2560\nds_read_b64 %6, %8 offset:3072\nds_read_b64 %7, %8 offset:3584\n")
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(base));
        t1 = clock64(); rec(t1 - t0);
        a += r[0] + r[7];
    }
    // 15: 16 independent ds_read_b128
    {
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v r[4];
        const int base = 16 * lane;
        t0 = clock64();
        asm volatile(REP4("ds_read_b128 %0, %4\nds_read_b128 %1, %4 offset:1024\nds_read_b128 %2, %4 offset:2048\nds_read_b128 %3, %4 offset:3072\n")
                     "s_waitcnt lgkmcnt(0)\n" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "v"(base));
        t1 = clock64(); rec(t1 - t0);
        a += r[0].x + r[3].y;
    }
    // 16: 16 ds_write_b64 back to back + wait;  17: 8 ds_write_b128
    {
        const int base = 8 * lane;
        t0 = clock64();
        asm volatile(REP4("ds_write_b64 %1, %0\nds_write_b64 %1, %0 offset:512\nds_write_b64 %1, %0 offset:1024\nds_write_b64 %1, %0 offset:1536\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(a), "v"(base));
        t1 = clock64(); rec(t1 - t0);
        typedef double d2w __attribute__((ext_vector_type(2)));
        d2w w2; w2.x = a; w2.y = a;
        const int b2 = 16 * lane;
        t0 = clock64();
        asm volatile(REP4("ds_write_b128 %1, %0\nds_write_b128 %1, %0 offset:1024\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(w2), "v"(b2));
        t1 = clock64(); rec(t1 - t0);
    }
    // 18: s_memtime back to back (its own cost)
    t0 = clock64();
    t1 = clock64(); rec(t1 - t0);
    if (tid == 0) out[0] = a;
}

int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 64 * 8);
    const char* names[] = {"v_fma_f64 dependent, per link (64)", "v_add_f64 dependent (64)", "v_mul_f64 dependent (64)", "4 independent fma chains, per instruction (64)",
                           "DPP step: 2 v_mov_dpp + v_add_f64, per step (16)", "v_readlane -> v_mov round trip (16)", "v_rcp_f64 dependent (16)", "v_rsq_f64 dependent (16)",
                           "ds_read_b32 pointer chase, per hop (16)", "ds_write_b64 + ds_read_b64 of it (16)", "v_cmp_f64 + s_cbranch not taken (16)",
                           "v_cmp_f64 + 2 v_cndmask dependent (16)", "s_waitcnt + s_barrier (16)", "write | barrier | read | barrier (16)",
                           "32 independent ds_read_b64 + wait, per read (32)", "16 independent ds_read_b128 + wait (16)", "16 ds_write_b64 + wait (16)", "8 ds_write_b128 + wait (8)",
                           "s_memtime pair (1)"};
    const int cnt[] = {64, 64, 64, 64, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 32, 16, 16, 8, 1};
    for (int nw : {1, 4}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, cyc, nw);
        hipDeviceSynchronize();
        std::vector<long long> h(64);
        hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
        printf("%d wave(s) of the workgroup active (wave 0's clock):\n", nw);
        for (int i = 0; i < 19; ++i) printf("  %-52s %8.1f cycles\n", names[i], (double)h[i] / cnt[i]);
    }
    return 0;
}
