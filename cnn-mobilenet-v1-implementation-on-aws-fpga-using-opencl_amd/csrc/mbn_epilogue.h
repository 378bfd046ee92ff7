// mbn_epilogue.h — accumulator -> global store of the 32x32 MFMA tiles, shared by the GEMM-shaped kernels
// (mbn_f32_pw.hip, mbn_f32_dwpw.hip): out[row][col] = relu6(acc * scale[col] + shift[col]).
// C/D layout of v_mfma_f32_32x32x2_f32: lane l holds column l&31 and rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// Stores are buffer stores: the output's descriptor sits in 4 SGPRs, the per-lane byte offset is ONE VGPR that is the
// same for all 16*MI*NI stores, and each store's wave-uniform part (tile origin + r's row) is its scalar offset. A store
// then costs fma + clamp + one scalar add — no per-element 64-bit address registers (which cost a whole occupancy step
// when tried with flat stores) and no bounds branches. The output must be smaller than 4 GiB.
#pragma once

typedef float mbn_f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mbn_make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

// MODE 0: the MI*32 x NI*32 block lies inside the matrix, no checks. MODE 1: columns inside, rows may run past m — the
// descriptor (num_records = m*ldc*4 bytes) drops them: the whole offset goes through the VGPR so the hardware range
// check sees it. MODE 2: explicit row and column checks (ragged column tiles). row0/col0 must be wave-uniform.
// TO = float or __bf16 (the element type of `out`; bf16 is rounded to nearest-even like the plain cast).
template <int MI, int NI, int MODE, typename TO = float>
__device__ __forceinline__ void mbn_store_relu6_f32(__amdgpu_buffer_rsrc_t out, unsigned ldc, unsigned row0, int col0,
                                                    int lane, const mbn_f16v (&acc)[MI][NI],
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    unsigned m, int n)
{
    const int li = lane & 31, lh = lane >> 5;
    constexpr unsigned ES = sizeof(TO);
    const unsigned lane_off = ((unsigned)(4 * lh) * ldc + (unsigned)li) * ES;          // bytes
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        const int col = col0 + ni * 32 + li;
        const bool cok = MODE != 2 || col < n;
        const float sc = scale[cok ? col : n - 1], sh = shift[cok ? col : n - 1];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned ro = row0 + mi * 32 + (r & 3) + 8 * (r >> 2);           // + 4*lh per lane
                const float f = fminf(fmaxf(fmaf(acc[mi][ni][r], sc, sh), 0.f), 6.f);
                const unsigned soff = (ro * ldc + (unsigned)(col0 + ni * 32)) * ES;    // wave-uniform bytes
                if constexpr (ES == 4) {
                    const unsigned v = __builtin_bit_cast(unsigned, f);
                    if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b32(v, out, lane_off, soff, 0);
                    else if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b32(v, out, lane_off + soff, 0, 0);
                    else if (cok && ro + 4 * lh < m) __builtin_amdgcn_raw_buffer_store_b32(v, out, lane_off, soff, 0);
                } else {
                    const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)f);
                    if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b16(v, out, lane_off, soff, 0);
                    else if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b16(v, out, lane_off + soff, 0, 0);
                    else if (cok && ro + 4 * lh < m) __builtin_amdgcn_raw_buffer_store_b16(v, out, lane_off, soff, 0);
                }
            }
    }
}

// bf16 output with CHANNEL-PAIRED column blocks. A bf16 kernel stages its filter rows into LDS in the order
//   LDS row rho of a 64-column group  <-  output channel 2*(rho & 31) + (rho >> 5)   (mbn_pair_channel below),
// so lane l's accumulators of the two 32x32 blocks of a group (ni = 2t, 2t+1) are the ADJACENT channels 2*(l&31) and
// 2*(l&31)+1 of the same pixel rows: they round and pack into one dword and a store instruction writes 128 contiguous
// bytes per pixel row (32 lanes x 4 B), two rows per instruction — 8*MI*NI buffer_store_dword per wave instead of
// 16*MI*NI buffer_store_short with 64-byte rows. Measured alternatives (profiles/r02): the transposed product with one
// 16-byte store per lane (every lane a different pixel row) is 20 % SLOWER than the 2-byte stores on the 512->512 GEMM and
// 2.5x slower on the 16->32 layer — the memory pipeline wants contiguity ACROSS lanes, not width per lane.
// NI must be even. MODE 0: block inside the matrix; MODE 1: rows may run past m (descriptor range check).
__device__ __forceinline__ int mbn_pair_channel(int rho) { return (rho & ~63) | (2 * (rho & 31) + ((rho >> 5) & 1)); }

// The fp32 form of the channel-paired store: lane l's accumulators of blocks 2t and 2t+1 are channels 2l and 2l+1, so a
// store instruction writes 8 bytes per lane = 256 contiguous bytes per pixel row, two rows per instruction — 8*MI*NI
// buffer_store_dwordx2 per wave instead of 16*MI*NI buffer_store_dword. The values are those of mbn_store_relu6_f32 bit
// for bit (same fma, same clamp). Used by the fused block kernels, where the epilogue stores measured 13 % of a step.
template <int MI, int NI, int MODE>
__device__ __forceinline__ void mbn_store_relu6_f32_pair(__amdgpu_buffer_rsrc_t out, unsigned ldc, unsigned row0, int col0, int lane,
                                                         const mbn_f16v (&acc)[MI][NI], const float *__restrict__ scale,
                                                         const float *__restrict__ shift)
{
    static_assert((NI & 1) == 0, "channel-paired epilogue needs an even number of 32-column blocks");
    typedef float f2e __attribute__((ext_vector_type(2)));
    typedef unsigned u2e __attribute__((ext_vector_type(2)));
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lane_off = ((unsigned)(4 * lh) * ldc + (unsigned)(2 * li)) * 4u;        // bytes
#pragma unroll
    for (int t = 0; t < NI / 2; t++) {
        const f2e sc = *reinterpret_cast<const f2e *>(scale + col0 + 64 * t + 2 * li);
        const f2e sh = *reinterpret_cast<const f2e *>(shift + col0 + 64 * t + 2 * li);
        const f2e scx = f2e{ sc.x, sc.x }, scy = f2e{ sc.y, sc.y }, shx = f2e{ sh.x, sh.x }, shy = f2e{ sh.y, sh.y };
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // round 5: the BN of rows r, r + 1 of one channel is ONE v_pk_fma_f32 (the two accumulators are adjacent registers, scale and shift
                // broadcast); the clamps write each value where its store wants it, so the channel pairing costs nothing. Same fma, same clamp: same bits.
                const f2e p0 = __builtin_elementwise_fma(f2e{ acc[mi][2 * t][r], acc[mi][2 * t][r + 1] }, scx, shx);
                const f2e p1 = __builtin_elementwise_fma(f2e{ acc[mi][2 * t + 1][r], acc[mi][2 * t + 1][r + 1] }, scy, shy);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int rr = r + h;
                    const unsigned ro = row0 + mi * 32 + (rr & 3) + 8 * (rr >> 2);         // + 4*lh per lane
                    const f2e v = f2e{ fminf(fmaxf(h ? p0.y : p0.x, 0.f), 6.f), fminf(fmaxf(h ? p1.y : p1.x, 0.f), 6.f) };
                    const unsigned soff = (ro * ldc + (unsigned)(col0 + 64 * t)) * 4u;     // wave-uniform bytes
                    if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2e, v), out, lane_off, soff, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2e, v), out, lane_off + soff, 0, 0);
                }
            }
    }
}

template <int MI, int NI, int MODE>
__device__ __forceinline__ void mbn_store_relu6_bf16_pair(__amdgpu_buffer_rsrc_t out, unsigned ldc, unsigned row0, int col0, int lane,
                                                          const mbn_f16v (&acc)[MI][NI], const float *__restrict__ scale,
                                                          const float *__restrict__ shift)
{
    static_assert((NI & 1) == 0, "channel-paired epilogue needs an even number of 32-column blocks");
    typedef float f2e __attribute__((ext_vector_type(2)));
    typedef __bf16 bf2e __attribute__((ext_vector_type(2)));
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lane_off = ((unsigned)(4 * lh) * ldc + (unsigned)(2 * li)) * 2u;        // bytes
#pragma unroll
    for (int t = 0; t < NI / 2; t++) {
        const f2e sc = *reinterpret_cast<const f2e *>(scale + col0 + 64 * t + 2 * li);
        const f2e sh = *reinterpret_cast<const f2e *>(shift + col0 + 64 * t + 2 * li);
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned ro = row0 + mi * 32 + (r & 3) + 8 * (r >> 2);               // + 4*lh per lane
                const float v0 = fminf(fmaxf(fmaf(acc[mi][2 * t][r], sc.x, sh.x), 0.f), 6.f);
                const float v1 = fminf(fmaxf(fmaf(acc[mi][2 * t + 1][r], sc.y, sh.y), 0.f), 6.f);
                const unsigned v = __builtin_bit_cast(unsigned, bf2e{ (__bf16)v0, (__bf16)v1 });   // RNE (v_cvt_pk_bf16_f32)
                const unsigned soff = (ro * ldc + (unsigned)(col0 + 64 * t)) * 2u;         // wave-uniform bytes
                if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b32(v, out, lane_off, soff, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(v, out, lane_off + soff, 0, 0);
            }
    }
}

// The same channel-paired bf16 store for accumulators of v_mfma_f32_16x16x32_bf16 blocks (the shape the chip holds a higher clock under,
// profiles/r03/x_bf16_mfma_shape.txt). C/D of a 16 x 16 block: lane (c = l & 15, q = l >> 4) holds column c, rows 4 q .. 4 q + 3. With the filter
// rows staged channel-paired (mbn_pair_channel), the 16-row LDS blocks j and j + 2 of a 64-column group hold channels 2 (16 j + c) and
// 2 (16 j + c) + 1 (j = 0, 1): packed 4-byte stores, 64 contiguous bytes per pixel row and instruction, four rows per instruction.
// acc[i][j]: i = 16-row block of the wave tile (MI16 of them), j = 16-column LDS block (NI16 = 4 per 64-column group, NI16 % 4 == 0).
typedef float mbn_f4v __attribute__((ext_vector_type(4)));
template <int MI16, int NI16, int MODE>
__device__ __forceinline__ void mbn_store_relu6_bf16_pair16(__amdgpu_buffer_rsrc_t out, unsigned ldc, unsigned row0, int col0, int lane,
                                                            const mbn_f4v (&acc)[MI16][NI16], const float *__restrict__ scale,
                                                            const float *__restrict__ shift)
{
    static_assert((NI16 & 3) == 0, "channel-paired 16x16 epilogue needs whole 64-column groups");
    typedef float f2e __attribute__((ext_vector_type(2)));
    typedef __bf16 bf2e __attribute__((ext_vector_type(2)));
    const int c16 = lane & 15, q16 = lane >> 4;
    const unsigned lane_off = ((unsigned)(4 * q16) * ldc + (unsigned)(2 * c16)) * 2u;      // bytes
#pragma unroll
    for (int t = 0; t < NI16 / 4; t++)
#pragma unroll
        for (int jp = 0; jp < 2; jp++) {
            const f2e sc = *reinterpret_cast<const f2e *>(scale + col0 + 64 * t + 32 * jp + 2 * c16);
            const f2e sh = *reinterpret_cast<const f2e *>(shift + col0 + 64 * t + 32 * jp + 2 * c16);
#pragma unroll
            for (int i = 0; i < MI16; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned ro = row0 + 16 * i + r;                                 // + 4 * q16 per lane
                    const float v0 = fminf(fmaxf(fmaf(acc[i][4 * t + jp][r], sc.x, sh.x), 0.f), 6.f);
                    const float v1 = fminf(fmaxf(fmaf(acc[i][4 * t + jp + 2][r], sc.y, sh.y), 0.f), 6.f);
                    const unsigned v = __builtin_bit_cast(unsigned, bf2e{ (__bf16)v0, (__bf16)v1 });
                    const unsigned soff = (ro * ldc + (unsigned)(col0 + 64 * t + 32 * jp)) * 2u;   // wave-uniform bytes
                    if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b32(v, out, lane_off, soff, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(v, out, lane_off + soff, 0, 0);
                }
        }
}
