#!/usr/bin/env python3
"""Generates csrc/tile_sort_net.h: the compare-exchange networks with which ONE wave sorts up to 1024 32-bit words in its
registers (csrc/tile_sort.h: sort_wave_trunc).  Run:  python3 gen_tile_sort_net.py > tile_sort_net.h

Layout.  A wave holds N = 64 NCH words, NCH = 1, 2, 4, 8, 16 per lane; word index i = lane * NCH + c (register c of the lane): the
LOW index bits select the register, the high six the lane.  The network is the all-ascending bitonic one (per level of size s a
"flip" step i <-> i ^ (s - 1), then "disperse" steps i <-> i ^ d for d = s/4 .. 1; the smaller word always moves to the smaller
index, so trailing 0xffffffff padding never moves).  A distance d is used in every level above it -- distance 1 in all of them --
so the most frequent distances (1 .. NCH/2) cost nothing but a v_min / v_max between two registers of the same lane, and only
the rarer large ones cross lanes.  (Round 8 had i = 64 c + lane: every level ended in six cross-lane steps per register.)

Cross-lane steps by what decides "lower / upper" (lane bit lb) and how the partner lane (lane ^ ml) is reached:
  lb = 1, 2  inside a quad: v_mov_b32_dpp quad_perm, then v_med3_u32 (word, partner, bound) with bound = 0 in lower lanes and
             0xffffffff in upper ones -- med3(a, b, 0) = min(a, b), med3(a, b, ~0) = max(a, b): no v_cndmask, no lane-role logic;
  lb = 4, 8  inside a 16-lane row: ONE v_min_u32_dpp written through the bank mask of the lower lanes and ONE v_max_u32_dpp
             through the bank mask of the upper ones (row_shl / row_shr by 4 or 8 for the disperse steps, row_half_mirror /
             row_mirror for the flips) -- two instructions, emitted as inline asm (the compiler does not fuse masked DPP moves);
  lb = 16, 32  across rows: ds_bpermute + v_med3_u32.  Three such steps per register in a 1024-word sort.
Every step is emitted for all registers at once (step-major), so that a lane's NCH independent chains interleave: the compiler
kept round 8's chain-major source order and left a dependent DPP -> min -> max -> select sequence with s_nop between the steps.
"""
import sys


def quad_perm(ml):
    return {1: "quad_perm:[1,0,3,2]", 2: "quad_perm:[2,3,0,1]", 3: "quad_perm:[3,2,1,0]"}[ml]


def quad_ctrl(ml):
    return {1: "0xB1", 2: "0x4E", 3: "0x1B"}[ml]


def emit_net(nch, out):
    r = nch.bit_length() - 1
    N = 64 * nch
    w = out.append
    w(f"// ---- {N} words, {nch} per lane")
    w(f"__device__ __forceinline__ void sort_net_{nch}(uint32_t (&v)[{nch}], const NetLane& k) {{")
    w(f"    uint32_t t[{nch}];")
    w("    (void)t;")

    def reg_step(m, lowbit):
        # partner register c ^ m, lower = bit `lowbit` of c clear
        for c in range(nch):
            p = c ^ m
            if c < p:
                lo, hi = (c, p) if (c & lowbit) == 0 else (p, c)
                w(f"    {{ const uint32_t a = min(v[{c}], v[{p}]), b = max(v[{c}], v[{p}]); v[{lo}] = a; v[{hi}] = b; }}")

    def lane_step(m, lowbit):
        mr, ml, lb = m & (nch - 1), m >> r, lowbit >> r
        assert lb >= 1
        src = lambda c: c ^ mr
        if lb in (1, 2):
            # v_mov_b32_dpp writes every lane (a quad permutation always has a source), so no "old" value to materialise
            for g0 in range(0, nch, 8):
                cs = list(range(g0, min(nch, g0 + 8)))
                nout = len(cs)
                lines = ["s_nop 1"]
                for j, c in enumerate(cs):
                    lines.append(f"v_mov_b32_dpp %{j}, %{2 * nout + j} {quad_perm(ml)} row_mask:0xf bank_mask:0xf")
                for j, c in enumerate(cs):
                    lines.append(f"v_med3_u32 %{j}, %{nout + j}, %{j}, %{3 * nout}")
                body = "\\n\\t".join(lines)
                outs = ", ".join(f'"=&v"(t[{c}])' for c in cs)
                ins = ", ".join([f'"v"(v[{c}])' for c in cs] + [f'"v"(v[{src(c)}])' for c in cs] + [f'"v"(k.bound{lb})'])
                w(f'    asm("{body}" : {outs} : {ins});')
            for c in range(nch):
                w(f"    v[{c}] = t[{c}];")
        elif lb in (4, 8):
            if ml in (4, 8):
                lo_ctrl, hi_ctrl = f"row_shl:{ml}", f"row_shr:{ml}"
            else:
                lo_ctrl = hi_ctrl = {7: "row_half_mirror", 15: "row_mirror"}[ml]
            lo_mask, hi_mask = ("0x5", "0xa") if lb == 4 else ("0x3", "0xc")
            # groups of at most 8 registers per asm statement (operand limit); a DPP source must be two wait states old
            for g0 in range(0, nch, 8):
                cs = list(range(g0, min(nch, g0 + 8)))
                lines = ["s_nop 1"]
                nout = len(cs)
                # operands: %0..%(nout-1) = t[c] (early clobber), then own v[c], then partner v[src(c)]
                for j, c in enumerate(cs):
                    lines.append(f"v_min_u32_dpp %{j}, %{2 * nout + j}, %{nout + j} {lo_ctrl} row_mask:0xf bank_mask:{lo_mask}")
                for j, c in enumerate(cs):
                    lines.append(f"v_max_u32_dpp %{j}, %{2 * nout + j}, %{nout + j} {hi_ctrl} row_mask:0xf bank_mask:{hi_mask}")
                body = "\\n\\t".join(lines)
                outs = ", ".join(f'"=&v"(t[{c}])' for c in cs)
                ins = ", ".join([f'"v"(v[{c}])' for c in cs] + [f'"v"(v[{src(c)}])' for c in cs])
                w(f'    asm("{body}" : {outs} : {ins});')
            for c in range(nch):
                w(f"    v[{c}] = t[{c}];")
        else:
            addr = {16: "k.addr16", 31: "k.addr31", 63: "k.addr63", 32: "k.addr32"}[ml]
            for c in range(nch):
                w(f"    t[{c}] = (uint32_t)__builtin_amdgcn_ds_bpermute({addr}, (int)v[{src(c)}]);")
            for c in range(nch):
                w(f"    t[{c}] = med3u(v[{c}], t[{c}], k.bound{lb});")
            for c in range(nch):
                w(f"    v[{c}] = t[{c}];")

    size = 2
    while size <= N:
        w(f"    // level {size}")
        steps = [(size - 1, size // 2)] + [(d, d) for d in [size >> s for s in range(2, size.bit_length())] if d >= 1]
        for m, lowbit in steps:
            if m < nch:
                reg_step(m, lowbit)
            else:
                lane_step(m, lowbit)
        size *= 2
    w("}")
    w("")


def main():
    out = []
    w = out.append
    w("// tile_sort_net.h -- GENERATED by gen_tile_sort_net.py (see its docstring for the design); do not edit.")
    w("#pragma once")
    w("#include <hip/hip_runtime.h>")
    w("#include <stdint.h>")
    w("")
    w("namespace dgr {")
    w("namespace {")
    w("")
    w("// per-lane constants of the networks: med3 bounds (0 where the lane keeps the minimum of a pair whose lanes differ in that")
    w("// bit, 0xffffffff where it keeps the maximum) and ds_bpermute byte addresses of the cross-row partners")
    w("struct NetLane {")
    w("    uint32_t bound1, bound2, bound16, bound32;")
    w("    int addr16, addr31, addr32, addr63;")
    w("};")
    w("__device__ __forceinline__ NetLane net_lane(int lane) {")
    w("    NetLane k;")
    w("    k.bound1 = (lane & 1) ? 0xffffffffu : 0u;   k.bound2 = (lane & 2) ? 0xffffffffu : 0u;")
    w("    k.bound16 = (lane & 16) ? 0xffffffffu : 0u; k.bound32 = (lane & 32) ? 0xffffffffu : 0u;")
    w("    k.addr16 = (lane ^ 16) << 2; k.addr31 = (lane ^ 31) << 2; k.addr32 = (lane ^ 32) << 2; k.addr63 = (lane ^ 63) << 2;")
    w("    return k;")
    w("}")
    w("// the median of three unsigned words: with c = 0 the minimum of a and b, with c = 0xffffffff their maximum")
    w("__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {")
    w("    uint32_t r;")
    w('    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));')
    w("    return r;")
    w("}")
    w("")
    for nch in (1, 2, 4, 8, 16):
        emit_net(nch, out)
    w("template <int NCH> __device__ __forceinline__ void sort_net(uint32_t (&v)[NCH], const NetLane& k);")
    for nch in (1, 2, 4, 8, 16):
        w(f"template <> __device__ __forceinline__ void sort_net<{nch}>(uint32_t (&v)[{nch}], const NetLane& k) {{ sort_net_{nch}(v, k); }}")
    w("")
    w("}  // namespace")
    w("}  // namespace dgr")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
