// wave_reduce.h -- multi-value wave64 reductions for gfx950.
//
// Reducing V values across 64 lanes one at a time costs 6 cross-lane steps each.  The butterflies below reduce 12 or 16
// values TOGETHER: every stage halves the lane span of each value and packs two registers into one (DPP row_mirror /
// row_half_mirror adds written through bank masks inside a 16-lane row, v_permlane16_swap / v_permlane32_swap -- which
// move half a register in one instruction -- across rows, quad_perm inside a quad).  Afterwards each 4-lane quad holds
// the grand total of ONE input value (wave_reduce12d_comp / wave_reduce16d_comp), so a single LDS atomic with 12 (15)
// active lanes scatters all totals at once.  The within-row stages come first: with the measured issue costs (a DPP add
// 4.2 cycles, a permlane swap 8.1, a plain add 2.5) a merge stage costs one DPP add per input register whatever the
// register count, and only the two cross-row stages need swaps (3 swaps for 12 values; a swap-first network needs 9).
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
// f + f[lane ^ 1] + f[lane ^ 2] + f[lane ^ 3] in every lane of a quad: two DPP adds.  Written as asm because the compiler fuses
// only the first `f += dpp_mov(f)` into a v_add_f32_dpp and sinks the second add into the caller's branch behind a v_mov_b32_dpp
// (4.2 + 2.4 cycles of issue instead of 4.2).  The s_nop: a DPP read needs its source written >= 2 wait states earlier.
__device__ __forceinline__ float quad_sum(float f) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "+v"(f));
    return f;
}

// Twelve sums, within-row stages first.  A DPP add issues at 1.4x a plain VALU instruction, a permlane
// swap at 2.7x (profiles/microbench/valu_rates.hip), and a merge stage needs one DPP add per input register whatever the
// register count -- so the two stages that halve the register count inside a 16-lane row (row_mirror, row_half_mirror,
// written through bank masks: 12 -> 6 -> 3 registers, 18 DPP adds) run on all twelve registers, and the two cross-row
// stages, which only the swaps can do, run on the three that are left: 3 swaps instead of 9.
//   value held by lane l afterwards:  4 min(l >> 4, 2) + {0, 2, 1, 3}[(l >> 2) & 3]   (rows 2 and 3 both hold values 8..11)
__device__ __forceinline__ int wave_reduce12d_comp(int lane) {
    const int r = lane >> 4, b = (lane >> 2) & 3;
    return 4 * (r < 2 ? r : 2) + ((b == 1) ? 2 : (b == 2) ? 1 : b);
}
#define DGR_MERGE(d, s, ctrl, m0, m1) \
    "v_add_f32_dpp %" #d ", %" #d ", %" #d " " ctrl " row_mask:0xf bank_mask:" m0 "\n\t" \
    "v_add_f32_dpp %" #d ", %" #s ", %" #s " " ctrl " row_mask:0xf bank_mask:" m1 "\n\t"
__device__ __forceinline__ void wave_reduce12d_head(float (&x)[12], float& u0, float& u1) {
    // stage 1 (in place, span 16 -> 8 inside every row): lanes 0-7 of x[2k] <- x[2k], lanes 8-15 <- x[2k+1];
    // stage 2 (span 8 -> 4): banks {0, 2} of x[4m] <- x[4m] (values 4m, 4m+1), banks {1, 3} <- x[4m+2] (values 4m+2, 4m+3).
    // A DPP read needs its source written >= 2 instructions earlier: the order below guarantees it after the leading nop.
    asm volatile("s_nop 1\n\t"
                 DGR_MERGE(0, 1, "row_mirror", "0x3", "0xc") DGR_MERGE(2, 3, "row_mirror", "0x3", "0xc")
                 DGR_MERGE(4, 5, "row_mirror", "0x3", "0xc") DGR_MERGE(6, 7, "row_mirror", "0x3", "0xc")
                 DGR_MERGE(8, 9, "row_mirror", "0x3", "0xc") DGR_MERGE(10, 11, "row_mirror", "0x3", "0xc")
                 DGR_MERGE(0, 2, "row_half_mirror", "0x5", "0xa") DGR_MERGE(4, 6, "row_half_mirror", "0x5", "0xa")
                 DGR_MERGE(8, 10, "row_half_mirror", "0x5", "0xa")
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
    // stage 3 (rows 0+1, 2+3): the swap exchanges the odd rows of its first operand with the even rows of its second
    float z0 = x[0], z1 = x[4], z2 = x[8], z3 = x[8];
    asm volatile("s_nop 1\n\t" DGR_SWAP16(0, 1) DGR_SWAP16(2, 3) : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3));
    u0 = z0 + z1;  // rows 0, 2: values 0..3 ; rows 1, 3: values 4..7
    u1 = z2 + z3;  // every row: values 8..11 (rows 0, 1: rows 0+1 ; rows 2, 3: rows 2+3)
}
// stage 4 (halves): afterwards u0 + u1 holds values 0..3 in row 0, 4..7 in row 1, 8..11 in rows 2 and 3
__device__ __forceinline__ float wave_reduce12d_tail(float u0, float u1) {
    asm volatile("s_nop 1\n\t" DGR_SWAP32(0, 1) : "+v"(u0), "+v"(u1));
    return quad_sum(u0 + u1);
}
__device__ __forceinline__ float wave_reduce12d(float (&x)[12]) {
    float u0, u1;
    wave_reduce12d_head(x, u0, u1);
    return wave_reduce12d_tail(u0, u1);
}
// ... or, without stage 4, the twelve sums of each HALF of the wave on its own (the paired lists of the mapping backward: a
// step that serves one entry on lanes 0-31 and another on lanes 32-63): r0 = quad_sum(u0) holds the half's values 0..3 in its
// even row and 4..7 in its odd row, r1 = quad_sum(u1) its values 8..11 in both rows; quad b holds value {0, 2, 1, 3}[b].
// The same additions as the full network up to its last stage -- which, for an entry that lives in one half, adds exact zeros.
__device__ __forceinline__ int wave_reduce12d_half_slot0(int lane) {  // r0's value in this lane
    const int b = (lane >> 2) & 3;
    return 4 * ((lane >> 4) & 1) + ((b == 1) ? 2 : (b == 2) ? 1 : b);
}
__device__ __forceinline__ int wave_reduce12d_half_slot1(int lane) {  // r1's value in this lane (taken from the half's even row)
    const int b = (lane >> 2) & 3;
    return ((lane >> 4) & 1) ? -1 : 8 + ((b == 1) ? 2 : (b == 2) ? 1 : b);
}

// Sixteen values, within-row stages first: 16 -> 8 -> 4 registers with 24 DPP adds, two swaps + one swap across rows.
//   value held by lane l afterwards:  4 (l >> 4) + {0, 2, 1, 3}[(l >> 2) & 3]
__device__ __forceinline__ int wave_reduce16d_comp(int lane) {
    const int b = (lane >> 2) & 3;
    return 4 * (lane >> 4) + ((b == 1) ? 2 : (b == 2) ? 1 : b);
}
__device__ __forceinline__ void wave_reduce16d_head(float (&x)[16], float& u0, float& u1) {
    asm volatile("s_nop 1\n\t"
                 DGR_MERGE(0, 1, "row_mirror", "0x3", "0xc") DGR_MERGE(2, 3, "row_mirror", "0x3", "0xc")
                 DGR_MERGE(4, 5, "row_mirror", "0x3", "0xc") DGR_MERGE(6, 7, "row_mirror", "0x3", "0xc")
                 DGR_MERGE(8, 9, "row_mirror", "0x3", "0xc") DGR_MERGE(10, 11, "row_mirror", "0x3", "0xc")
                 DGR_MERGE(12, 13, "row_mirror", "0x3", "0xc") DGR_MERGE(14, 15, "row_mirror", "0x3", "0xc")
                 DGR_MERGE(0, 2, "row_half_mirror", "0x5", "0xa") DGR_MERGE(4, 6, "row_half_mirror", "0x5", "0xa")
                 DGR_MERGE(8, 10, "row_half_mirror", "0x5", "0xa") DGR_MERGE(12, 14, "row_half_mirror", "0x5", "0xa")
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
    float z0 = x[0], z1 = x[4], z2 = x[8], z3 = x[12];
    asm volatile("s_nop 1\n\t" DGR_SWAP16(0, 1) DGR_SWAP16(2, 3) : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3));
    u0 = z0 + z1;  // rows 0, 2: values 0..3 ; rows 1, 3: values 4..7
    u1 = z2 + z3;  // rows 0, 2: values 8..11 ; rows 1, 3: values 12..15
}
__device__ __forceinline__ float wave_reduce16d_tail(float u0, float u1) {
    asm volatile("s_nop 1\n\t" DGR_SWAP32(0, 1) : "+v"(u0), "+v"(u1));
    return quad_sum(u0 + u1);   // row r: values 4 r .. 4 r + 3
}
__device__ __forceinline__ float wave_reduce16d(float (&x)[16]) {
    float u0, u1;
    wave_reduce16d_head(x, u0, u1);
    return wave_reduce16d_tail(u0, u1);
}
// ... or, without the last stage, the sixteen sums of each HALF of the wave on its own (paired lists of the full backward, as
// wave_reduce12d_half_slot*): r0 = quad_sum(u0) holds the half's values 0..3 in its even row and 4..7 in its odd row, r1 =
// quad_sum(u1) its values 8..11 / 12..15; quad b holds value {0, 2, 1, 3}[b].
__device__ __forceinline__ int wave_reduce16d_half_slot0(int lane) {
    const int b = (lane >> 2) & 3;
    return 4 * ((lane >> 4) & 1) + ((b == 1) ? 2 : (b == 2) ? 1 : b);
}
__device__ __forceinline__ int wave_reduce16d_half_slot1(int lane) { return 8 + wave_reduce16d_half_slot0(lane); }

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

// Three values, each HALF of the wave (lanes 0-31, lanes 32-63) on its own: afterwards lane 32 h holds the half's total of a,
// lane 32 h + 16 that of b, lane 32 h + 8 that of c (half_reduce3_comp).  Two swaps across the two rows of a half, one
// row_mirror merge that packs (a | b) and c into one register, three DPP adds inside the 8-lane groups.
__device__ __forceinline__ int half_reduce3_comp(int lane) {
    const int l = lane & 31;
    return l == 0 ? 0 : l == 16 ? 1 : l == 8 ? 2 : -1;
}
__device__ __forceinline__ float half_reduce3(float a, float b, float c) {
    float c2 = c;
    // the swap exchanges the odd rows of its first operand with the even rows of its second
    asm volatile("s_nop 1\n\t" DGR_SWAP16(0, 1) DGR_SWAP16(2, 3) : "+v"(a), "+v"(b), "+v"(c), "+v"(c2));
    float u = a + b;   // even rows: a over the half's two rows; odd rows: b over them
    float v = c + c2;  // every row: c over the half's two rows
    // lanes 0-7 of a row <- u + mirrored u, lanes 8-15 <- v + mirrored v; then 8 -> 4 -> 2 -> 1 inside the groups
    asm volatile("s_nop 1\n\t"
                 DGR_MERGE(0, 1, "row_mirror", "0x3", "0xc")
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(u), "+v"(v));
    return quad_sum(u);
}

}  // namespace dgr
