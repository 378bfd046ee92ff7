// Control for the depthwise stages: what does a pure 16-B-per-lane streaming kernel reach on THIS box
// over exactly the input + output footprint of a depthwise layer? Three forms of the stream:
//   plain   each lane: one 16-B global load -> one 16-B global store, U independent loads in flight per lane, one-shot grid
//   lds     each wave: LDS-DMA (buffer_load_dwordx4 ... lds) of 1 KiB pieces DEPTH pieces ahead, ds_read_b128, 16-B store
//           (the data path of the LDS-staged depthwise kernel, without its arithmetic)
//   s2      stride-2 footprint: read 4 units, write 1 (layer 4 / 8 / 12 / 24)
// Also checks what an out-of-range LDS-DMA lane writes into LDS (the zero padding of the staged kernels relies on 0).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/dw_copy_control.hip -o tools/micro/bin/dw_copy_control
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

// ---- plain: U float4 per lane, lanes of a wave contiguous, the U pieces of a workgroup 4 KiB apart (one-shot grid).
// RATIO > 1 = the stride-2 footprint: RATIO float4 read per float4 written (the input is RATIO regions of n_out float4).
template <int U, int RATIO>
__global__ __launch_bounds__(256) void copy_plain(f4 *__restrict__ out, const f4 *__restrict__ in, long n_out)
{
    const long base = ((long)blockIdx.x * U) * 256 + threadIdx.x;
    f4 v[U][RATIO];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const long i = base + (long)u * 256;
#pragma unroll
        for (int r = 0; r < RATIO; r++) v[u][r] = (i < n_out) ? in[(long)r * n_out + i] : f4{ 0.f, 0.f, 0.f, 0.f };
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const long i = base + (long)u * 256;
        f4 s = v[u][0];
#pragma unroll
        for (int r = 1; r < RATIO; r++) s += v[u][r];
        if (i < n_out) out[i] = s;
    }
}

// ---- lds: persistent workgroups; each wave owns a ring of DEPTH 1-KiB LDS slots; piece p of the wave's stream is loaded by
// LDS-DMA DEPTH-1 pieces ahead of its use, read back with ds_read_b128 and stored. No barrier: a wave reads only its own slots.
template <int DEPTH>
__global__ __launch_bounds__(256) void copy_lds(f4 *__restrict__ out, const f4 *__restrict__ in, long n, unsigned in_bytes)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * DEPTH * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const ring = lds + wave * DEPTH * 256;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, in_bytes);
    const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    const long npieces = n / 64;                       // pieces of 64 float4 = 1 KiB
    // piece index of this wave's j-th piece: wave_id + j * nwaves
    long p_issue = wave_id;
    int slot_i = 0;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) {
        const unsigned off = p_issue < npieces ? (unsigned)(p_issue * 1024 + lane * 16) : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ring + slot_i * 256), 16, off, 0, 0, 0);
        p_issue += nwaves;
        slot_i = slot_i + 1 == DEPTH ? 0 : slot_i + 1;
    }
    int slot_r = 0;
    for (long p = wave_id; p < npieces; p += nwaves) {
        const unsigned off = p_issue < npieces ? (unsigned)(p_issue * 1024 + lane * 16) : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ring + slot_i * 256), 16, off, 0, 0, 0);
        p_issue += nwaves;
        slot_i = slot_i + 1 == DEPTH ? 0 : slot_i + 1;
        // the piece read now was issued DEPTH-1 issues ago: all but the DEPTH-1 youngest DMA (and the stores in between) must be done.
        // stores count in vmcnt too (in order): DEPTH-1 DMA + DEPTH-1 stores are younger than the awaited DMA.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1) > 63 ? 63 : 2 * (DEPTH - 1)) : "memory");
        const f4 v = *reinterpret_cast<const f4 *>(ring + slot_r * 256 + lane * 4);
        slot_r = slot_r + 1 == DEPTH ? 0 : slot_r + 1;
        out[p * 64 + lane] = v;
    }
}

// ---- OOB check: lanes with an out-of-range offset — what lands in LDS?
__global__ void oob_check(float *out, const float *in, unsigned in_bytes)
{
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 777.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, in_bytes);
    const unsigned off = (threadIdx.x & 1) ? 0xffffffffu : threadIdx.x * 16;     // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

static double time_ms(hipStream_t s, int iters, const std::function<void()> &f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    std::vector<float> ts;
    for (int i = 0; i < iters; i++) {
        CK(hipEventRecord(a, s));
        f();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <int U, int RATIO>
static void run_plain(const char *what, f4 *out, const f4 *in, long n_out, hipStream_t st)
{
    const long per_wg = 256L * U;
    const unsigned grid = (unsigned)((n_out + per_wg - 1) / per_wg);
    const double ms = time_ms(st, 15, [&] { hipLaunchKernelGGL((copy_plain<U, RATIO>), dim3(grid), dim3(256), 0, st, out, in, n_out); });
    const double bytes = 16.0 * n_out * (RATIO + 1);
    printf("  %-34s plain U=%d            %8.4f ms  %7.1f GB/s  (%.3f of 8.0 TB/s)\n", what, U, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}

template <int DEPTH>
static void run_lds(const char *what, f4 *out, const f4 *in, long n, int wg_per_cu, hipStream_t st)
{
    const unsigned grid = 256u * wg_per_cu;
    const double ms = time_ms(st, 15, [&] { hipLaunchKernelGGL((copy_lds<DEPTH>), dim3(grid), dim3(256), 0, st, out, in, n, (unsigned)(n * 16)); });
    const double bytes = 32.0 * n;
    printf("  %-34s lds  DEPTH=%d wg/CU=%d  %8.4f ms  %7.1f GB/s  (%.3f of 8.0 TB/s)\n", what, DEPTH, wg_per_cu, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // ---- what does an out-of-range LDS-DMA lane write?
    {
        float *din, *dout;
        CK(hipMalloc(&din, 4096)); CK(hipMalloc(&dout, 1024));
        std::vector<float> h(1024);
        for (int i = 0; i < 1024; i++) h[i] = 1000.f + i;
        CK(hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(oob_check, dim3(1), dim3(64), 0, st, dout, din, 4096u);
        CK(hipStreamSynchronize(st));
        std::vector<float> o(256);
        CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
        int zeros = 0, kept = 0, data_ok = 0;
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 4; j++) {
                const float v = o[l * 4 + j];
                if (l & 1) { zeros += v == 0.f; kept += v == 777.f; }
                else data_ok += v == 1000.f + l * 4 + j;
            }
        printf("LDS-DMA out-of-range lanes: %d of 128 dwords read back 0, %d kept the old LDS value; in-range dwords correct: %d of 128\n", zeros, kept, data_ok);
    }
    // ---- footprints of the depthwise layers at batch 256 fp32 (float4 counts): L2 112x112x32 s1, L6 56x56x128 s1, L14 14x14x512 s1;
    //      stride 2: L4 (112x112x64 -> 56x56x64), L8 (56x56x128 -> 28x28x128)
    struct F { const char *name; long n_out; int ratio; };
    const F fs[] = { { "L2  s1 411+411 MB", 256L * 112 * 112 * 32 / 4, 1 }, { "L6  s1 205+205 MB", 256L * 56 * 56 * 128 / 4, 1 },
                     { "L10 s1 103+103 MB", 256L * 28 * 28 * 256 / 4, 1 }, { "L14 s1  51+51 MB", 256L * 14 * 14 * 512 / 4, 1 },
                     { "L4  s2 822+206 MB", 256L * 56 * 56 * 64 / 4, 4 }, { "L8  s2 411+103 MB", 256L * 28 * 28 * 128 / 4, 4 } };
    long maxin = 0, maxout = 0;
    for (const F &f : fs) { maxin = std::max(maxin, f.n_out * f.ratio); maxout = std::max(maxout, f.n_out); }
    f4 *in, *out;
    CK(hipMalloc(&in, maxin * 16)); CK(hipMalloc(&out, maxout * 16));
    CK(hipMemset(in, 0x3c, maxin * 16));
    for (const F &f : fs) {
        printf("%s\n", f.name);
        if (f.ratio == 1) {
            run_plain<1, 1>(f.name, out, in, f.n_out, st);
            run_plain<2, 1>(f.name, out, in, f.n_out, st);
            run_plain<4, 1>(f.name, out, in, f.n_out, st);
            run_plain<8, 1>(f.name, out, in, f.n_out, st);
            run_lds<2>(f.name, out, in, f.n_out, 4, st);
            run_lds<4>(f.name, out, in, f.n_out, 4, st);
            run_lds<8>(f.name, out, in, f.n_out, 4, st);
            run_lds<8>(f.name, out, in, f.n_out, 8, st);
            run_lds<16>(f.name, out, in, f.n_out, 2, st);
        } else {
            run_plain<1, 4>(f.name, out, in, f.n_out, st);
            run_plain<2, 4>(f.name, out, in, f.n_out, st);
            run_plain<4, 4>(f.name, out, in, f.n_out, st);
        }
    }
    return 0;
}
