// In what granule does a CU hand out LDS, i.e. how large may a workgroup's dynamic block be for THREE (or four, five) of them to share the 160 KB?
// The runtime's occupancy answer for a 256-thread kernel that needs few registers, over a sweep of dynamic LDS sizes; prints the sizes at which the
// answer changes.  (wbcqp_api.hip asks the runtime the same question at launch; this probe is where the 512-byte rule of set_lds() comes from.)
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/lds_granule.hip -o /tmp/lds_granule && /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(double* out)
{
    extern __shared__ double lds[];
    __shared__ int word;
    if (threadIdx.x == 0) word = 1;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[255 - threadIdx.x] + word;
}

int main()
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    int last = -1;
    for (int bytes = 16 * 1024; bytes <= 160 * 1024 - 64; bytes += 16) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, (size_t)bytes) != hipSuccess) break;
        if (occ != last) {
            std::printf("dynamic LDS %6d B (+ 4 B static): %d workgroups per CU\n", bytes, occ);
            last = occ;
        }
    }
    return 0;
}
