// How does the hardware hand out workgroups that fill a CU each (160 KB of LDS)?  Every workgroup spins for its own
// given time; the launch's duration is measured for several launch orders of the same set of times, and the XCC of
// every workgroup is recorded.  Input: a text file, one line per order: a name, then one spin time per workgroup in
// ticks of the 100 MHz wall clock.  Output: name, measured microseconds, share of workgroups with XCC == index % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

__global__ __launch_bounds__(256) void spin_k(const int* ticks, unsigned* xcc_out, long long* trace)
{
    extern __shared__ double lds[];
    const long long t0 = wall_clock64();
    const int want = ticks[blockIdx.x];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double x = threadIdx.x;
    while (wall_clock64() - t0 < want) x = fma(x, 1.0000001, 1e-9);
    lds[threadIdx.x] = x;
    if (threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        xcc_out[blockIdx.x] = xcc & 0xf;
        trace[blockIdx.x * 3] = t0;
        trace[blockIdx.x * 3 + 1] = wall_clock64();
        trace[blockIdx.x * 3 + 2] = hw;
    }
}

// the same work handed out by a queue: 256 resident workgroups take the next index from a counter
__global__ __launch_bounds__(256) void spin_queue_k(const int* ticks, int* queue, int total)
{
    extern __shared__ double lds[];
    __shared__ int next;
    double x = threadIdx.x;
    for (;;) {
        if (threadIdx.x == 0) next = atomicAdd(queue, 1);
        __syncthreads();
        const int i = next;
        if (i >= total) break;
        const long long t0 = wall_clock64();
        const int want = ticks[i];
        while (wall_clock64() - t0 < want) x = fma(x, 1.0000001, 1e-9);
        __syncthreads();
    }
    lds[threadIdx.x] = x;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    std::ifstream in(argv[1]);
    std::string line;
    (void)hipFuncSetAttribute((const void*)spin_k, hipFuncAttributeMaxDynamicSharedMemorySize, 163000);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string name;
        ss >> name;
        std::vector<int> t;
        for (int v; ss >> v;) t.push_back(v);
        if (t.empty()) continue;
        int* d;
        unsigned* dx;
        long long* dt;
        (void)hipMalloc(&dt, sizeof(long long) * 3 * t.size());
        (void)hipMalloc(&d, sizeof(int) * t.size());
        (void)hipMalloc(&dx, sizeof(unsigned) * t.size());
        (void)hipMemcpy(d, t.data(), sizeof(int) * t.size(), hipMemcpyHostToDevice);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin_k, dim3((unsigned)t.size()), dim3(256), 163000, 0, d, dx, dt);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        int* dq;
        (void)hipMalloc(&dq, sizeof(int));
        float bestq = 1e30f;
        (void)hipFuncSetAttribute((const void*)spin_queue_k, hipFuncAttributeMaxDynamicSharedMemorySize, 163000);
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(dq, 0, sizeof(int));
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin_queue_k, dim3(256), dim3(256), 163000, 0, d, dq, (int)t.size());
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < bestq) bestq = ms;
        }
        (void)hipFree(dq);
        printf("%-24s %9.1f us   through a queue of 256 resident workgroups\n", name.c_str(), bestq * 1e3);
        std::vector<unsigned> x(t.size());
        (void)hipMemcpy(x.data(), dx, sizeof(unsigned) * t.size(), hipMemcpyDeviceToHost);
        size_t rr = 0;
        for (size_t i = 0; i < x.size(); ++i) rr += x[i] == i % 8;
        printf("%-24s %9.1f us   xcc==i%%8: %.3f   first xccs:", name.c_str(), best * 1e3, (double)rr / x.size());
        for (int i = 0; i < 16 && i < (int)x.size(); ++i) printf(" %u", x[i]);
        printf("\n");
        if (argc > 2) {  // trace of the last repetition: index, xcc, se, cu, start and end ticks relative to the first start
            std::vector<long long> tr(3 * t.size());
            (void)hipMemcpy(tr.data(), dt, sizeof(long long) * tr.size(), hipMemcpyDeviceToHost);
            long long t00 = tr[0];
            for (size_t i = 0; i < t.size(); ++i) t00 = tr[3 * i] < t00 ? tr[3 * i] : t00;
            FILE* f = fopen((std::string(argv[2]) + "." + name).c_str(), "w");
            for (size_t i = 0; f && i < t.size(); ++i) {
                const unsigned hw = (unsigned)tr[3 * i + 2];
                fprintf(f, "%zu %u %u %u %lld %lld\n", i, x[i], (hw >> 13) & 7, (hw >> 8) & 15, tr[3 * i] - t00, tr[3 * i + 1] - t00);
            }
            if (f) fclose(f);
        }
        (void)hipFree(dt);
        (void)hipFree(d);
        (void)hipFree(dx);
    }
    return 0;
}
