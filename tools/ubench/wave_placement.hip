// Where do the waves of co-resident 256-thread workgroups land?  Records (XCC, SE, CU, SIMD) of every wave of a
// 1024-workgroup launch that keeps each workgroup alive long enough for all of them to be resident together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void place_k(unsigned* out, int spin)
{
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned la;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 1e-9);
    lds[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + wave) * 2] = hw; out[(blockIdx.x * 4 + wave) * 2 + 1] = (xcc & 0xf) | (la << 8); }
    if (x == 12345.0) out[0] = 0;
}

int main(int argc, char** argv)
{
    // default: the rows kernel's shape (four workgroups per CU); `wave_placement 512 82000`: the compact solve kernel's (two per CU)
    const int nblk = argc > 1 ? atoi(argv[1]) : 1024;
    const int ldsb = argc > 2 ? atoi(argv[2]) : 18000;
    (void)hipFuncSetAttribute((const void*)place_k, hipFuncAttributeMaxDynamicSharedMemorySize, 163000);
    unsigned* d;
    (void)hipMalloc(&d, sizeof(unsigned) * nblk * 8);
    hipLaunchKernelGGL(place_k, dim3(nblk), dim3(256), ldsb, 0, d, 20000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(nblk * 8);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned) * nblk * 8, hipMemcpyDeviceToHost);
    // per CU: which SIMD did wave 0 of each resident workgroup get
    std::map<unsigned, std::vector<std::pair<int, int>>> cu;
    int same = 0;
    for (int b = 0; b < nblk; ++b) {
        int simd[4];
        unsigned key = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            simd[w] = (hw >> 4) & 3;
            key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 15);
        }
        cu[key].push_back({b, simd[0]});
        if (b < 8) printf("block %d: xcc %u, key %x, SIMD of waves 0..3 = %d %d %d %d\n", b, h[b * 8 + 1] & 0xf, key, simd[0], simd[1], simd[2], simd[3]);
        if (simd[0] == simd[1] || simd[1] == simd[2]) ++same;
    }
    printf("%zu distinct CUs; workgroups whose waves share a SIMD: %d\n", cu.size(), same);
    int shown = 0, clash = 0;
    for (auto& kv : cu) {
        int cnt[4] = {0, 0, 0, 0};
        for (auto& pr : kv.second) cnt[pr.second]++;
        if (cnt[0] > 1 || cnt[1] > 1 || cnt[2] > 1 || cnt[3] > 1) ++clash;
        if (shown++ < 6) {
            printf("CU %x:", kv.first);
            for (auto& pr : kv.second) {
                const int b = pr.first;
                printf(" (blk %d, waves 0..3 on SIMD %u %u %u %u, LDS_ALLOC %x)", b, (h[(b * 4 + 0) * 2] >> 4) & 3, (h[(b * 4 + 1) * 2] >> 4) & 3,
                       (h[(b * 4 + 2) * 2] >> 4) & 3, (h[(b * 4 + 3) * 2] >> 4) & 3, h[(b * 4 + 0) * 2 + 1] >> 8);
            }
            printf("\n");
        }
    }
    printf("CUs where two workgroups put wave 0 on the same SIMD: %d of %zu\n", clash, cu.size());
    return 0;
}
