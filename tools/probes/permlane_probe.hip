// v_permlane32_swap_b32 semantics on gfx950 (used by store_block_bf16 / load_block_bf16): prints, for lanes 0, 1, 32, 33, the two
// results of swap(a = lane, b = 100 + lane).  Expected: a' = (a.low | b.low) -> lanes 0,1,32,33 = 0 1 100 101;
// b' = (a.high | b.high) -> 32 33 132 133.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a' lanes 0 1 32 33: %u %u %u %u\n", h[0], h[1], h[32], h[33]);
    printf("b' lanes 0 1 32 33: %u %u %u %u\n", h[64], h[65], h[96], h[97]);
    return 0;
}
