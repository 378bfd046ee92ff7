// Microbenchmark: inside ONE wave, do VALU ops placed between independent fp32 MFMAs execute in the MFMA's shadow?
// One wave per SIMD (workgroup of 4 waves, 256 workgroups); loop body = 4 x { 1 MFMA 32x32x2 f32 + K v_pk_fma_f32 }.
// Prints cycles per MFMA for K = 0, 2, 4, 6, 8, 12 packed fp32 FMAs, and for integer / scalar-fp32 VALU ops: a flat 64
// would mean the K VALU ops are free.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_inwave.hip -o gpurun_out/mfma_valu_inwave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int K, int WAVES, int KIND>   // KIND 0: v_pk_fma_f32, 1: v_add_u32 (integer), 2: v_fma_f32
__global__ __launch_bounds__(64 * WAVES) void k(long long *out, float *sink, int nm)
{
    f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 0.001f, y = 1.0f + x;
    f2 v[12];
    for (int i = 0; i < 12; i++) v[i] = f2{1.f + i, 2.f + i};
    unsigned iv[12];
    for (int i = 0; i < 12; i++) iv[i] = threadIdx.x + i;
    const f2 m = {1.0001f, 0.9999f}, c = {0.001f, 0.002f};
    const long long t0 = clock64();
    for (int i = 0; i < nm; i += 4) {
#define STEP(acc)                                                                            \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);                       \
        _Pragma("unroll") for (int j = 0; j < K; j++) {                                           \
            if (KIND == 0) v[j] = __builtin_elementwise_fma(v[j], m, c);                           \
            else if (KIND == 1) iv[j] = iv[j] * 3u + 7u;                                           \
            else v[j].x = fmaf(v[j].x, m.x, c.x);                                                  \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3)
    }
    const long long t1 = clock64();
    float s = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 12; i++) s += v[i].x + v[i].y + (float)iv[i];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K, int WAVES, int KIND = 0>
void run(long long *d_out, float *sink)
{
    const int nm = 4096;
    long long h;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<K, WAVES, KIND>), dim3(256), dim3(64 * WAVES), 0, 0, d_out, sink, nm);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("waves/SIMD %d, %2d %s per MFMA: %.1f cycles per MFMA per wave\n", WAVES / 4, K, KIND == 0 ? "v_pk_fma_f32" : KIND == 1 ? "v_mad_u32 (int)" : "v_fma_f32", (double)h / nm);
}

int main()
{
    long long *d_out;
    float *sink;
    (void)hipMalloc(&d_out, 64);
    (void)hipMalloc(&sink, 8192);
    run<0, 4>(d_out, sink); run<2, 4>(d_out, sink); run<4, 4>(d_out, sink); run<6, 4>(d_out, sink); run<8, 4>(d_out, sink); run<12, 4>(d_out, sink);
    run<4, 4, 1>(d_out, sink); run<8, 4, 1>(d_out, sink); run<12, 4, 1>(d_out, sink);
    run<4, 4, 2>(d_out, sink); run<8, 4, 2>(d_out, sink); run<12, 4, 2>(d_out, sink);
    return 0;
}
