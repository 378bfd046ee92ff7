// What is the ceiling of the L2 -> CU operand stream of a tiled GEMM? Every workgroup re-reads an L2-resident region (the
// filter of a pointwise layer, or an activation panel its XCD neighbours fetched) and does nothing with it:
//   dma    buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction into a per-wave LDS ring, INFL pieces in flight per wave
//   vgpr   buffer_load_dwordx4 into registers, INFL loads in flight per lane (no LDS)
//   dma64  as dma with 64-byte rows per lane group (16 rows x 64 B per instruction: the k32 stage shape)
// Region sizes: 512 KB (one filter, every workgroup the same bytes), 64 MB (Infinity Cache), 1 GB (HBM stream).
// Prints GB/s per CU and chip-wide. build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_fill_rate.hip -o tools/micro/bin/lds_fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// each wave walks `pieces` 1-KiB pieces of the region starting at its own offset (wrapping), INFL in flight
template <int INFL, int WAVES, bool ROW64>
__global__ __launch_bounds__(64 * WAVES) void fill_dma(const float *src, unsigned region_bytes, int pieces, float *sink)
{
    __shared__ __attribute__((aligned(16))) float lds[WAVES * INFL * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *ring = lds + wave * INFL * 256;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, region_bytes);
    const unsigned npieces = region_bytes / 1024;
    unsigned p = ((unsigned)blockIdx.x * WAVES + wave) * 37u % npieces;
    // ROW64: lane l reads 16 B at row (l / 4), chunk (l % 4) of rows that are 128 B apart in memory (half lines)
    const unsigned lane_off = ROW64 ? (unsigned)((lane >> 2) * 128 + (lane & 3) * 16) : (unsigned)lane * 16;
    const unsigned piece_bytes = ROW64 ? 2048u : 1024u;
    const unsigned np2 = region_bytes / piece_bytes;
    if (ROW64) p %= np2;
    int slot = 0;
    for (int i = 0; i < pieces; i++) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ring + slot * 256), 16, p * piece_bytes + lane_off, 0, 0, 0);
        slot = slot + 1 == INFL ? 0 : slot + 1;
        p = p + 1 == (ROW64 ? np2 : npieces) ? 0 : p + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pieces < 0) sink[threadIdx.x] = ring[lane];
}

template <int INFL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fill_vgpr(const float *src, unsigned region_bytes, int pieces, float *sink)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, region_bytes);
    const unsigned npieces = region_bytes / 1024;
    unsigned p = ((unsigned)blockIdx.x * WAVES + wave) * 37u % npieces;
    f4 acc = f4{ 0.f, 0.f, 0.f, 0.f };
    for (int i = 0; i < pieces; i += INFL) {
        f4 v[INFL];
#pragma unroll
        for (int j = 0; j < INFL; j++) {
            v[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, p * 1024u + lane * 16, 0, 0));
            p = p + 1 == npieces ? 0 : p + 1;
        }
#pragma unroll
        for (int j = 0; j < INFL; j++) acc += v[j];
    }
    if (acc.x == 123.456f) sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

static double time_ms(hipStream_t s, const std::function<void()> &f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) f();
    std::vector<float> ts;
    for (int i = 0; i < 7; i++) {
        CK(hipEventRecord(a, s)); f(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <typename K>
static void run(const char *what, K kern, int wg_per_cu, int waves, const float *src, unsigned region, double piece_bytes, float *sink, hipStream_t st)
{
    const int pieces = 2048;
    const unsigned grid = 256u * wg_per_cu;
    const double ms = time_ms(st, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), 0, st, src, region, pieces, sink); });
    const double bytes = (double)grid * waves * pieces * piece_bytes;
    printf("  %-44s wg/CU=%d waves=%d  %8.4f ms  %7.1f GB/s per CU  %6.2f TB/s chip\n", what, wg_per_cu, waves, ms, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float *src, *sink;
    const size_t big = 1ull << 30;
    CK(hipMalloc(&src, big)); CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemset(src, 0x3c, big));
    const unsigned regions[] = { 512u << 10, 64u << 20, 1u << 30 };
    const char *names[] = { "512 KB (L2)", "64 MB (Infinity Cache)", "1 GB (HBM)" };
    for (int r = 0; r < 3; r++) {
        printf("region %s\n", names[r]);
        const unsigned reg = regions[r];
        run("dma  1 KiB pieces, 2 in flight per wave", fill_dma<2, 8, false>, 1, 8, src, reg, 1024, sink, st);
        run("dma  1 KiB pieces, 4 in flight per wave", fill_dma<4, 8, false>, 1, 8, src, reg, 1024, sink, st);
        run("dma  1 KiB pieces, 8 in flight per wave", fill_dma<8, 8, false>, 1, 8, src, reg, 1024, sink, st);
        run("dma  1 KiB pieces, 4 in flight per wave", fill_dma<4, 8, false>, 2, 8, src, reg, 1024, sink, st);
        run("dma  1 KiB pieces, 8 in flight per wave", fill_dma<8, 4, false>, 1, 4, src, reg, 1024, sink, st);
        run("dma  1 KiB pieces, 8 in flight per wave", fill_dma<8, 4, false>, 4, 4, src, reg, 1024, sink, st);
        run("dma64 16 rows x 64 B, 4 in flight per wave", fill_dma<4, 8, true>, 1, 8, src, reg, 1024, sink, st);
        run("dma64 16 rows x 64 B, 8 in flight per wave", fill_dma<8, 8, true>, 2, 8, src, reg, 1024, sink, st);
        run("vgpr 16 B per lane, 4 in flight per lane", fill_vgpr<4, 8>, 1, 8, src, reg, 1024, sink, st);
        run("vgpr 16 B per lane, 8 in flight per lane", fill_vgpr<8, 8>, 1, 8, src, reg, 1024, sink, st);
        run("vgpr 16 B per lane, 8 in flight per lane", fill_vgpr<8, 8>, 2, 8, src, reg, 1024, sink, st);
        run("vgpr 16 B per lane, 8 in flight per lane", fill_vgpr<8, 4>, 8, 4, src, reg, 1024, sink, st);
    }
    return 0;
}
