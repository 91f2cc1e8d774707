// Round trip of store_block_bf16 / load_block_bf16_raw / block_from_bf16_raw (nlam_wbf.inc) on a 32 x 32 block in MFMA accumulator
// layout: lane (j, hi) holds element r = 4 tt + c of row j at column 8 tt + 4 hi + c.  Prints mismatches of the stored rows and of the
// block read back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define NLAM_TU 99
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};
template <int NS> struct BfFrag { u32x4 t[NS]; };
template <int NS>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[NS]) {
    f32x2 v = {a, b};
    for (int p = 0; p < NS; ++p) {
        const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        out[p] = bits;
    }
}
template <int NS>
__device__ __forceinline__ BfFrag<NS> split8(const float (&x)[8]) {
    BfFrag<NS> f;
    for (int pr = 0; pr < 4; ++pr) {
        unsigned o[NS];
        split_pair<NS>(x[2 * pr], x[2 * pr + 1], o);
        for (int p = 0; p < NS; ++p) f.t[p][pr] = o[p];
    }
    return f;
}
#include "bf16_block_helpers.inc"
__global__ void k(unsigned short* buf, float* back) {
    const int lane = threadIdx.x, j = lane & 31, hi = lane >> 5;
    f32x16 v;
    for (int r = 0; r < 16; ++r) v[r] = (float)(j * 64 + 8 * (r >> 2) + 4 * hi + (r & 3));   // row * 64 + col: exact in bf16? (< 2048: 8 bits no) use small
    for (int r = 0; r < 16; ++r) v[r] = (float)((j % 8) * 32 + 8 * (r >> 2) + 4 * hi + (r & 3));   // <= 255: exact in bf16
    store_block_bf16(buf + j * 32, 0, hi, true, v);
    __syncthreads();
    u32x4 raw[2];
    load_block_bf16_raw(buf + j * 32, 0, hi, true, raw);
    const f32x16 w = block_from_bf16_raw(raw);
    for (int r = 0; r < 16; ++r) back[lane * 16 + r] = w[r] - v[r];
}
int main() {
    unsigned short* d; float* b;
    hipMalloc(&d, 32 * 32 * 2); hipMalloc(&b, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, b);
    unsigned short h[32 * 32]; float hb[64 * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int j = 0; j < 32; ++j) for (int c = 0; c < 32; ++c) {
        unsigned u = (unsigned)h[j * 32 + c] << 16; float f; memcpy(&f, &u, 4);
        if (f != (float)((j % 8) * 32 + c)) { if (bad < 8) printf("store mismatch row %d col %d: %g\n", j, c, f); ++bad; }
    }
    int bad2 = 0;
    for (int i = 0; i < 64 * 16; ++i) if (hb[i] != 0.f) { if (bad2 < 8) printf("round trip mismatch lane %d r %d: %g\n", i / 16, i % 16, hb[i]); ++bad2; }
    printf("store mismatches %d, round-trip mismatches %d\n", bad, bad2);
    return 0;
}
