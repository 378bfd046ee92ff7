// How much of the fp32 MFMA rate does a GEMM-like loop (one accumulator chain per wave: 2 ds_read_b128 + 4 v_mfma_f32_32x32x2_f32 per 8 k, as pw_gemm's
// 32 x 32 wave tile) reach with W waves per SIMD? W = 1 ... 6 (workgroups of 4 waves, W per CU), and the same loop without the LDS reads.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_occupancy_f32.hip -o tools/micro/bin/mfma_occupancy_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int W, bool READS>
__global__ __launch_bounds__(256, W) void loop(const f4 *src, float *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) f4 lds[1536];              // 24 KB: six workgroups fit a CU
    for (int i = threadIdx.x; i < 1536; i += 256) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16v acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    f4 a = lds[lane], b = lds[64 + lane];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (READS) {
                const int o = ((it * 4 + g) * 7 + wave * 64) % 1280;
                a = lds[o + lane]; b = lds[o + 128 + lane];
            }
#pragma unroll
            for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
    }
    float total = 0.f;
    for (int r = 0; r < 16; r++) total += acc[r];
    if (total == 123.456f) sink[threadIdx.x] = total;
}

// wave tiles 32 x 64 (NI = 2: 1 A + 2 B reads per 8 MFMAs) and 64 x 64 (MI = NI = 2: 2 + 2 reads per 16 MFMAs)
template <int W, int MI, int NI>
__global__ __launch_bounds__(256, W) void loop_big(const f4 *src, float *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) f4 lds[1536];
    for (int i = threadIdx.x; i < 1536; i += 256) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16v acc[MI][NI];
    for (int i = 0; i < MI; i++) for (int j = 0; j < NI; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int o = ((it * 4 + g) * 7 + wave * 64) % 1024;
            f4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = lds[o + i * 64 + lane];
#pragma unroll
            for (int j = 0; j < NI; j++) b[j] = lds[o + 128 + j * 64 + lane];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NI; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }
    float total = 0.f;
    for (int i = 0; i < MI; i++) for (int j = 0; j < NI; j++) for (int r = 0; r < 16; r++) total += acc[i][j][r];
    if (total == 123.456f) sink[threadIdx.x] = total;
}

template <int W, int MI, int NI>
static void run_big(hipStream_t st, const f4 *src, float *sink)
{
    const int iters = 2000 / (MI * NI);
    const int grid = 256 * W;
    const double flops = (double)grid * 4 * iters * (2.0 * 32 * MI * 32 * NI * 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((loop_big<W, MI, NI>), dim3(grid), dim3(256), 0, st, src, sink, iters);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("  %d waves per SIMD, wave tile %d x %d (%d reads per %d MFMAs): %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", W, 32 * MI, 32 * NI, MI + NI, 4 * MI * NI, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
}

template <int W, bool READS>
static void run(hipStream_t st, const f4 *src, float *sink)
{
    const int iters = 2000;
    const int grid = 256 * W;
    const double flops = (double)grid * 4 * iters * (2.0 * 32 * 32 * 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((loop<W, READS>), dim3(grid), dim3(256), 0, st, src, sink, iters);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("  %d waves per SIMD, %s: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", W, READS ? "2 ds_read_b128 per 4 MFMAs" : "no LDS reads in the loop     ", best, flops / best / 1e9, flops / best / 1e9 / 157.3);
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    f4 *src; float *sink;
    CK(hipMalloc(&src, 32768)); CK(hipMalloc(&sink, 4096));
    std::vector<float> h(8192);
    srand(1);
    for (auto &v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMemcpy(src, h.data(), 32768, hipMemcpyHostToDevice));
    for (int pass = 0; pass < 2; pass++) {
        run<1, true>(st, src, sink); run<2, true>(st, src, sink); run<3, true>(st, src, sink); run<4, true>(st, src, sink); run<5, true>(st, src, sink); run<6, true>(st, src, sink);
        run<1, false>(st, src, sink); run<4, false>(st, src, sink);
        run_big<2, 1, 2>(st, src, sink); run_big<3, 1, 2>(st, src, sink); run_big<4, 1, 2>(st, src, sink);
        run_big<1, 2, 2>(st, src, sink); run_big<2, 2, 2>(st, src, sink); run_big<3, 2, 2>(st, src, sink); run_big<4, 2, 2>(st, src, sink);
    }
    return 0;
}
