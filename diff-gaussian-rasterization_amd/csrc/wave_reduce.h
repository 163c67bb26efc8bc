// wave_reduce.h -- multi-value wave64 reductions for gfx950.
//
// Reducing V values across 64 lanes one at a time costs 6 cross-lane steps each.  The butterfly below
// reduces up to 16 values TOGETHER in 30 instructions: every stage halves the lane span of each value and
// packs two registers into one (v_permlane32_swap / v_permlane16_swap move half a register in one
// instruction; the narrower stages use DPP row_mirror / row_half_mirror / quad_perm).  Afterwards each
// 4-lane quad holds the grand total of ONE input value:
//     quad q = lane >> 2 holds value  kWaveReduce16Comp[q]
// so a single store / atomic with 16 (here 14) active lanes scatters all totals at once.
//
// The permlane swaps are emitted as inline asm: __builtin_amdgcn_permlane{16,32}_swap returns a pair whose
// second element this ROCm's compiler aliases to the first.  A VALU write of an operand must be >= 2 wait
// states ahead of the swap that reads it, hence the leading s_nop 1 of each asm block.
#pragma once
#include <hip/hip_runtime.h>

namespace dgr {

#define DGR_SWAP32(a, b) "v_permlane32_swap_b32 %" #a ", %" #b "\n\t"
#define DGR_SWAP16(a, b) "v_permlane16_swap_b32 %" #a ", %" #b "\n\t"

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141, DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E;

// value index held by lane quad q after wave_reduce16
__device__ __forceinline__ int wave_reduce16_comp(int lane) {
    const int r = lane >> 4;
    const int perm = (r == 1) ? 2 : (r == 2) ? 1 : r;  // rows come out as 0,2,1,3
    return 4 * (2 * ((lane >> 2) & 1) + ((lane >> 3) & 1)) + perm;
}

// x[0..15] per lane -> total of value wave_reduce16_comp(lane) in every lane.  x[14], x[15] may be anything
// the caller does not read back (pass zeros).
__device__ __forceinline__ float wave_reduce16(float (&x)[16]) {
    // stage A: span 64 -> 32, 16 registers -> 8
    asm volatile("s_nop 1\n\t" DGR_SWAP32(0, 1) DGR_SWAP32(2, 3) DGR_SWAP32(4, 5) DGR_SWAP32(6, 7) DGR_SWAP32(8, 9)
                 DGR_SWAP32(10, 11) DGR_SWAP32(12, 13) DGR_SWAP32(14, 15)
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
    float y0 = x[0] + x[1], y1 = x[2] + x[3], y2 = x[4] + x[5], y3 = x[6] + x[7];
    float y4 = x[8] + x[9], y5 = x[10] + x[11], y6 = x[12] + x[13], y7 = x[14] + x[15];
    // stage B: span 32 -> 16, 8 -> 4
    asm volatile("s_nop 1\n\t" DGR_SWAP16(0, 1) DGR_SWAP16(2, 3) DGR_SWAP16(4, 5) DGR_SWAP16(6, 7)
                 : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
    const float z0 = y0 + y1, z1 = y2 + y3, z2 = y4 + y5, z3 = y6 + y7;
    // stage C: span 16 -> 8, 4 -> 2.  The DPP adds write through bank masks (a bank = 4 lanes of a row): lanes 0-7 of
    // every row take the first source, lanes 8-15 the second -- a merge costs two instructions, no v_cndmask.
    float t0, t1;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %1, %4, %4 row_mirror row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 : "=&v"(t0), "=&v"(t1)
                 : "v"(z0), "v"(z1), "v"(z2), "v"(z3));
    // stage D: span 8 -> 4, 2 -> 1: lanes 0-3 of every 8 take t0, lanes 4-7 take t1
    float u;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                 : "=&v"(u)
                 : "v"(t0), "v"(t1));
    // stage E: span 4 -> 1
    u += dpp_mov<DPP_QUAD_XOR1>(u);
    u += dpp_mov<DPP_QUAD_XOR2>(u);
    return u;
}

// Same network for 12 values (x[0..11]): 25 instructions.  Lane quads whose wave_reduce16_comp() is >= 12 hold garbage.
// TEN = true: x[10] and x[11] are known to be zero (the caller need not set them): their swap and add are left out.
template <bool TEN = false>
__device__ __forceinline__ float wave_reduce12(float (&x)[12]) {
    float y0, y1, y2, y3, y4, y5;
    if (TEN) {
        asm volatile("s_nop 1\n\t" DGR_SWAP32(0, 1) DGR_SWAP32(2, 3) DGR_SWAP32(4, 5) DGR_SWAP32(6, 7) DGR_SWAP32(8, 9)
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                       "+v"(x[8]), "+v"(x[9]));
        y5 = 0.f;
    } else {
        asm volatile("s_nop 1\n\t" DGR_SWAP32(0, 1) DGR_SWAP32(2, 3) DGR_SWAP32(4, 5) DGR_SWAP32(6, 7) DGR_SWAP32(8, 9)
                     DGR_SWAP32(10, 11)
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                       "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
        y5 = x[10] + x[11];
    }
    y0 = x[0] + x[1]; y1 = x[2] + x[3]; y2 = x[4] + x[5]; y3 = x[6] + x[7]; y4 = x[8] + x[9];
    asm volatile("s_nop 1\n\t" DGR_SWAP16(0, 1) DGR_SWAP16(2, 3) DGR_SWAP16(4, 5)
                 : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5));
    const float z0 = y0 + y1, z1 = y2 + y3, z2 = y4 + y5;
    float t0, t1;  // (upper half rows of t1 would carry values 12..15: unused)
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %1, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0xc\n\t"
                 : "=&v"(t0), "=&v"(t1)
                 : "v"(z0), "v"(z1), "v"(z2));
    float u;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                 : "=&v"(u)
                 : "v"(t0), "v"(t1));
    u += dpp_mov<DPP_QUAD_XOR1>(u);
    u += dpp_mov<DPP_QUAD_XOR2>(u);
    return u;
}

// x[0..3] per lane -> total of value {0,2,1,3}[lane >> 4] in every lane of that 16-lane row (10 instructions).
__device__ __forceinline__ int wave_reduce4_comp(int lane) {
    const int r = lane >> 4;
    return (r == 1) ? 2 : (r == 2) ? 1 : r;
}
__device__ __forceinline__ float wave_reduce4(float (&x)[4]) {
    asm volatile("s_nop 1\n\t" DGR_SWAP32(0, 1) DGR_SWAP32(2, 3) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    float y0 = x[0] + x[1], y1 = x[2] + x[3];
    asm volatile("s_nop 1\n\t" DGR_SWAP16(0, 1) : "+v"(y0), "+v"(y1));
    float z = y0 + y1;
    z += dpp_mov<DPP_ROW_MIRROR>(z);
    z += dpp_mov<DPP_ROW_HALF_MIRROR>(z);
    z += dpp_mov<DPP_QUAD_XOR1>(z);
    z += dpp_mov<DPP_QUAD_XOR2>(z);
    return z;
}

}  // namespace dgr
