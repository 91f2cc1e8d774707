// tr16_probe.hip -- what ds_read_b64_tr_b16 delivers (round 6: LDS transpose reads for the K = rows operands of the weight gradient).
// Every lane passes the address of 4 consecutive halfwords; lds[i] = i, so the value a lane receives names the halfword it came from.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o tools/probes/tr16_probe && tools/probes/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int hw;   // halfword index this lane points at
    if (mode == 0) hw = 4 * l;                                   // lane-linear
    else if (mode == 1) hw = (l & 15) * 64 + (l >> 4) * 4;       // 16-lane group g reads column group g of 16 rows with a pitch of 64 halfwords
    else hw = (l >> 2) * 64 + (l & 3) * 4;                       // lane -> (row = l / 4, 4-column piece l % 4), pitch 64 halfwords
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + hw));
    for (int q = 0; q < 4; ++q) out[l * 4 + q] = (unsigned short)r[q];
}

int main() {
    unsigned short* d;
    unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d: lane -> the four source halfword indices (lane address = %s)\n", mode,
               mode == 0 ? "4*lane" : mode == 1 ? "(l&15)*64 + (l>>4)*4" : "(l>>2)*64 + (l&3)*4");
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
