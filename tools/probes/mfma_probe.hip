// Micro-probe: how fast does the mma_chunk pattern (A operand = one ds_read_b128 per 4
// dependent v_mfma_f32_32x32x2_f32 per block, B operand from registers) issue on gfx950,
// alone and with a VALU-only epilogue phase, at 1 / 2 / 4 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
// same shape of work on the bf16 matrix pipe: 32x32x16 bf16 MFMAs (8 passes) + a VALU phase
template <int NM, int VALU_REPS>
__global__ __launch_bounds__(512) void probe_bf16(float* out, int tiles_per_wave) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane); b[i] = (short)0x3f80; }
    float sink = 0.f;
    for (int it = 0; it < tiles_per_wave; ++it) {
        f32x16 acc[2];
        for (int hb = 0; hb < 2; ++hb)
            for (int r = 0; r < 16; ++r) acc[hb][r] = 0.f;
#pragma unroll 8
        for (int t = 0; t < NM / 2; ++t) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
        }
        float v = 0.f;
        for (int hb = 0; hb < 2; ++hb)
            for (int r = 0; r < 16; ++r) v += acc[hb][r];
#pragma unroll 1
        for (int k = 0; k < VALU_REPS; ++k) {
#pragma unroll
            for (int q = 0; q < 32; ++q) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
        }
        sink += v;
        a[0] += (short)(v > 1e30f);
    }
    if (sink == 12345.f) out[threadIdx.x] = sink;
}

template <int NM, int VR>
void run_bf16(const char* name, int waves_per_block, float* out) {
    const int tiles = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_bf16<NM, VR>), dim3(256), dim3(64 * waves_per_block), 0, 0, out, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_bf16<NM, VR>), dim3(256), dim3(64 * waves_per_block), 0, 0, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 256.0 * waves_per_block * tiles * NM;
    const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/block %2d : %8.1f us  %7.1f TFLOP/s bf16 (%.0f %% of 2500)\n", name, waves_per_block, ms * 1e3, tf, 100.0 * tf / 2500.0);
}

template <int HB, int VALU_REPS>
__global__ __launch_bounds__(1024) void probe(const float* W, float* out, int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int s = threadIdx.x; s < 64 * 192; s += blockDim.x) smem[s] = W[s];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 x = {1.f + lane, 2.f, 3.f, 4.f};
    float sink = 0.f;
    for (int it = 0; it < tiles_per_wave; ++it) {
        f32x16 acc[HB];
        for (int hb = 0; hb < HB; ++hb)
            for (int r = 0; r < 16; ++r) acc[hb][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 32; ++t) {   // 32 chunks x HB blocks x 4 = 256 MFMAs at HB = 2
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&smem[(((hb * 32 + t) % 48) * 64 + lane) * 4]);
                acc[hb] = MFMA32(a[0], x[0], acc[hb]);
                acc[hb] = MFMA32(a[1], x[1], acc[hb]);
                acc[hb] = MFMA32(a[2], x[2], acc[hb]);
                acc[hb] = MFMA32(a[3], x[3], acc[hb]);
            }
        }
        float v = 0.f;
        for (int hb = 0; hb < HB; ++hb)
            for (int r = 0; r < 16; ++r) v += acc[hb][r];
        // VALU-only phase: VALU_REPS x 32 dependent-ish FMAs per lane
#pragma unroll 1
        for (int k = 0; k < VALU_REPS; ++k) {
#pragma unroll
            for (int q = 0; q < 32; ++q) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
        }
        sink += v;
        x[0] += 1e-9f * v;
    }
    if (sink == 12345.f) out[threadIdx.x] = sink;
}

template <int HB, int VR>
void run(const char* name, int waves_per_block, int blocks_per_cu, const float* W, float* out) {
    const int tiles = 16;
    const size_t lds = 64 * 192 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<HB, VR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((probe<HB, VR>), dim3(grid), dim3(64 * waves_per_block), lds, 0, W, out, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<HB, VR>), dim3(grid), dim3(64 * waves_per_block), lds, 0, W, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * waves_per_block * tiles * 32 * HB * 4;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/block %2d blocks/CU %d : %8.1f us  %6.1f TFLOP/s (%.0f %% of 157.3)\n", name, waves_per_block, blocks_per_cu,
           ms * 1e3, tf, 100.0 * tf / 157.3);
}

int main() {
    float *W, *out;
    hipMalloc(&W, 64 * 192 * 4);
    hipMalloc(&out, 4096);
    hipMemset(W, 0, 64 * 192 * 4);
    for (int wpb : {4, 8}) {
        run<2, 0>("mfma only, HB=2", wpb, 1, W, out);
        run<1, 0>("mfma only, HB=1 (one chain)", wpb, 1, W, out);
        run<2, 20>("mfma + 640 VALU/tile", wpb, 1, W, out);
        run<2, 45>("mfma + 1440 VALU/tile", wpb, 1, W, out);
    }
    run<2, 45>("mfma + 1440 VALU/tile", 8, 2, W, out);
    run<2, 45>("mfma + 1440 VALU/tile", 16, 1, W, out);
    run<2, 0>("mfma only", 16, 1, W, out);
    // bf16: 96 MFMAs (= the bf16x3 equivalent of 256 fp32 MFMAs) with and without the VALU phase
    for (int wpb : {4, 8}) {
        run_bf16<96, 0>("bf16 96 mfma/tile only", wpb, out);
        run_bf16<96, 45>("bf16 96 mfma + 1440 VALU/tile", wpb, out);
        run_bf16<96, 65>("bf16 96 mfma + 2080 VALU/tile", wpb, out);
    }
    return 0;
}
