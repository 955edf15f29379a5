// Device-copy bandwidth on this part: what a plain streaming kernel reaches, to put the 8 TB/s roofline figure in context.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void copy_k(const double4* __restrict__ a, double4* __restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_k(const double4* __restrict__ a, double* out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const double4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678) out[0] = s;
}

int main()
{
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(double4);
    double4 *a, *b;
    double* o;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 8);
    (void)hipMemset(a, 1, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n);
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("copy  grid %6d: %.2f TB/s (read + write)\n", grid, 2.0 * bytes * 5 / (ms * 1e-3) / 1e12);
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, o, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("read  grid %6d: %.2f TB/s\n", grid, 1.0 * bytes * 5 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
