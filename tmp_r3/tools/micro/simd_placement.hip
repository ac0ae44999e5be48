// Which SIMD of its CU does wavefront w of a workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] cu_id[11:8]
// sh_id[12] se_id[15:13] on gfx9.)  Build: hipcc --offload-arch=gfx950 -O2 simd_placement.hip -o simd_placement
// Run: ./simd_placement <threads per workgroup> <dynamic LDS bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    const int w = threadIdx.x / 64, nw = blockDim.x / 64;
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;   // keep the workgroups resident together
    lds[threadIdx.x] = x;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * nw + w] = id | (lds[threadIdx.x] == 12345.0 ? 1u << 31 : 0);
}

int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 192, lds = argc > 2 ? atoi(argv[2]) : 38752, blocks = 8192, nw = nt / 64;
    unsigned* d; hipMalloc(&d, sizeof(unsigned) * blocks * nw);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(nt), lds, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * nw);
    hipMemcpy(h.data(), d, sizeof(unsigned) * blocks * nw, hipMemcpyDeviceToHost);
    long hist[8][4] = {};
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) hist[w][(h[b * nw + w] >> 4) & 3]++;
    printf("threads %d, LDS %d B: wavefront index x SIMD id\n", nt, lds);
    for (int w = 0; w < nw; ++w) printf("  wave %d: %6ld %6ld %6ld %6ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    long tot[4] = {}; for (int w = 0; w < nw; ++w) for (int s = 0; s < 4; ++s) tot[s] += hist[w][s];
    printf("  all   : %6ld %6ld %6ld %6ld\n", tot[0], tot[1], tot[2], tot[3]);
    return 0;
}
