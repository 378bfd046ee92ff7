// How fast can the output of a bf16 pointwise layer be WRITTEN, by store shape? One launch writes the [M][N] bf16 output of layer 15 at batch 512
// (100352 x 512, 103 MB) tile by tile like the GEMM's epilogue (128 x 128 tiles, 8 waves of 32 x 64, persistent 2 workgroups per CU, same tile order):
//   epi4   the shipped epilogue's shape: 16 stores per wave tile, 4 bytes per lane = two 128-byte row segments per instruction
//   row16  the same tile as 4 stores of 16 bytes per lane: 8 lanes = one 128-byte row segment, 8 rows per instruction (what a transpose through LDS would give)
//   lin16  the whole buffer linearly, 16 bytes per lane (the ceiling)
// and the same three with a READ of as many bytes beside them (lin16 loads of another buffer) to see what the pair reaches together.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o tools/micro/bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int N = 512, BM = 128, BN = 128;

template <int MODE, bool RD>
__global__ __launch_bounds__(512) void writer(unsigned *out, const f4 *src, float *sink, int mt, int nt)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
    const int ntiles = mt * nt;
    f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = (t / nt) * BM, n0 = (t % nt) * BN;
        if (RD) {      // as many bytes read as written: 128 x 128 x 2 = 32 KB per tile = 4 x 16 B per thread
#pragma unroll
            for (int i = 0; i < 4; i++) acc += src[((size_t)t * 4 + i) * 512 + threadIdx.x];
        }
        if (MODE == 0) {
            const int li = lane & 31, lh = lane >> 5;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[((size_t)row * N + n0 + wn) / 2 + li] = 0x3f803f80u + r;
            }
        } else if (MODE == 1) {
            const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = m0 + wm + r * 8 + rr;
                *reinterpret_cast<f4 *>(out + ((size_t)row * N + n0 + wn) / 2 + ch * 4) = f4{ 1.f, 2.f, 3.f, (float)r };
            }
        } else {
            // the tile's 32 KB as one linear piece of the buffer
#pragma unroll
            for (int r = 0; r < 4; r++)
                *reinterpret_cast<f4 *>(out + ((size_t)t * 8192) + (r * 512 + threadIdx.x) * 4) = f4{ 1.f, 2.f, 3.f, (float)r };
        }
    }
    if (RD && acc.x == 123.25f) sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

static double time_ms(hipStream_t s, const std::function<void()> &f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    std::vector<float> ts;
    for (int i = 0; i < 9; i++) {
        CK(hipEventRecord(a, s)); f(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const long M = 100352;
    const size_t bytes = (size_t)M * N * 2;
    unsigned *out; f4 *src; float *sink;
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(src, 0x3c, bytes));
    const int mt = (int)(M / BM), nt = N / BN;
    const char *names[] = { "epi4  (4 B/lane, 2 row segments per instr)", "row16 (16 B/lane, 8 row segments per instr)", "lin16 (16 B/lane, linear)" };
    for (int wg = 1; wg <= 4; wg *= 2) {
        const dim3 grid(256 * wg), block(512);
        printf("persistent grid: %d workgroups of 8 waves per CU\n", wg);
        double ms;
        ms = time_ms(st, [&] { hipLaunchKernelGGL((writer<0, false>), grid, block, 0, st, out, src, sink, mt, nt); });
        printf("  write only  %-46s %.4f ms  %6.0f GB/s\n", names[0], ms, bytes / ms / 1e6);
        ms = time_ms(st, [&] { hipLaunchKernelGGL((writer<1, false>), grid, block, 0, st, out, src, sink, mt, nt); });
        printf("  write only  %-46s %.4f ms  %6.0f GB/s\n", names[1], ms, bytes / ms / 1e6);
        ms = time_ms(st, [&] { hipLaunchKernelGGL((writer<2, false>), grid, block, 0, st, out, src, sink, mt, nt); });
        printf("  write only  %-46s %.4f ms  %6.0f GB/s\n", names[2], ms, bytes / ms / 1e6);
        ms = time_ms(st, [&] { hipLaunchKernelGGL((writer<0, true>), grid, block, 0, st, out, src, sink, mt, nt); });
        printf("  read+write  %-46s %.4f ms  %6.0f GB/s (both directions)\n", names[0], ms, 2 * bytes / ms / 1e6);
        ms = time_ms(st, [&] { hipLaunchKernelGGL((writer<1, true>), grid, block, 0, st, out, src, sink, mt, nt); });
        printf("  read+write  %-46s %.4f ms  %6.0f GB/s (both directions)\n", names[1], ms, 2 * bytes / ms / 1e6);
        ms = time_ms(st, [&] { hipLaunchKernelGGL((writer<2, true>), grid, block, 0, st, out, src, sink, mt, nt); });
        printf("  read+write  %-46s %.4f ms  %6.0f GB/s (both directions)\n", names[2], ms, 2 * bytes / ms / 1e6);
    }
    return 0;
}
