// Microbenchmark: do v_mfma_f32_32x32x2_f32 (wave A) and v_pk_fma_f32 / v_fma_f32 (wave B on the same SIMD) overlap?
// One workgroup of 8 waves per CU; waves 0-3 (one per SIMD) issue NM MFMAs, waves 4-7 (one per SIMD) issue NV VALU ops.
// mode 1 = MFMA only, 2 = VALU only, 3 = both; prio 1 = the VALU waves run at s_setprio 3. Prints cycles (s_memtime).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o gpurun_out/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>   // 0: fp32 MFMA 32x32x2, 1: bf16 MFMA 32x32x16
__global__ __launch_bounds__(512) void k(long long *out, float *sink, int mode, int nm, int nv, int prio)
{
    const int wave = threadIdx.x >> 6;
    __shared__ int dummy;
    if (threadIdx.x == 0) dummy = 0;
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) {
        if (mode & 1) {
            f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            float x = threadIdx.x * 0.001f, y = 1.0f + x;
            typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
            bf8 bx, by;
            for (int i = 0; i < 8; i++) { bx[i] = (__bf16)x; by[i] = (__bf16)y; }
            for (int i = 0; i < nm; i += 4) {
                if (KIND == 0) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a3, 0, 0, 0);
                }
            }
            sink[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else {
        if (mode & 2) {
            if (prio) __builtin_amdgcn_s_setprio(3);      // VALU wave above the (older) MFMA wave in the issue arbitration
            f2 v0 = {1.f, 2.f}, v1 = {3.f, 4.f}, v2 = {5.f, 6.f}, v3 = {7.f, 8.f};
            const f2 m = {1.0001f, 0.9999f}, c = {0.001f, 0.002f};
            for (int i = 0; i < nv; i += 4) {
                v0 = __builtin_elementwise_fma(v0, m, c);
                v1 = __builtin_elementwise_fma(v1, m, c);
                v2 = __builtin_elementwise_fma(v2, m, c);
                v3 = __builtin_elementwise_fma(v3, m, c);
            }
            sink[threadIdx.x] = v0.x + v1.y + v2.x + v3.y;
        }
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main()
{
    long long *d_out, h[8];
    float *sink;
    hipMalloc(&d_out, 64);
    hipMalloc(&sink, 4096);
    const int nm = 4096, nv = 16384;
    for (int prio = 0; prio < 2; prio++)
    for (int kind = 0; kind < 2; kind++)
        for (int mode = 1; mode <= 3; mode++) {
            if (prio && mode != 3) continue;
            for (int rep = 0; rep < 2; rep++) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d_out, sink, mode, nm, nv, prio);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d_out, sink, mode, nm, nv, prio);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost);
            printf("%s prio %d mode %d (%s): mfma wave %lld cycles (%.1f/MFMA), valu wave %lld cycles (%.2f/op)\n", kind ? "bf16 32x32x16" : "fp32 32x32x2 ", prio,
                   mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", h[0], (double)h[0] / nm, h[4], (double)h[4] / nv);
        }
    return 0;
}
