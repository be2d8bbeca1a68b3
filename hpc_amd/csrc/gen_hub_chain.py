#!/usr/bin/env python3
"""Writes hub_chain_asm.inc: the chain wave's loop of spmm_hub (spmm_kernels.hpp) as ONE inline-assembly block per slice width.

Why assembly: the chain wave of a hub row is a single wave whose every instruction costs it 5-6 cycles, whatever the
instruction is, and whose LDS reads cost least issued back to back in front of the chain
(scripts/experiments/chain_patterns.hip; profiles/r03_hub_experiments.txt).  hipcc kept rearranging the C++ form of this
loop -- sinking the 32 fmas below the flag traffic, splitting the 16 reads 3 + 13 around a full drain, waiting for one of
a trip's fresh reads because the 4-bit counter cannot say "the 16 older ones" -- and every such rearrangement is a
10-20 % loss on rows that are nothing but this loop.  The block below is the schedule the measurements picked, and
nothing in it moves.  The arithmetic is the same chain: acc = fma(a_k, b_k, acc), k ascending, one fmac per nonzero.

Round 4: the wave is bound by its own instruction issue (round 3: 107 instructions per stage of 64 nonzeros at ~5.8 cycles
= 622), so the loop was cut to 88 instructions per stage:
  * the a values no longer take 16 broadcast ds_read_b128 per stage.  ONE ds_read_b128 per stage -- lane i of every 16-lane
    row reads a[4 i .. 4 i + 3] -- leaves the stage's 64 a values in four registers, and link k = 4 i + r takes its factor
    straight out of them through DPP: v_fmac_f32_dpp acc, vA[r], vB[k] row_newbcast:i (lane i of the row, broadcast to the row;
    A_MODE = "quad": quad_perm:[i,i,i,i] with four reads of 16 a values per stage, for hardware where row_newbcast is
    reserved to the 64-bit ops).  The multiply-add itself is the same v_fmac_f32: same bits.
  * the hand-off words are touched once per PAIR of stages: the loaders publish IN ORDER into one word (`pub` = stages
    published so far, all of them; the last stage's publisher writes INT_MAX), so one poll covers the two stages a pair will
    fetch, and `done` is written once per pair.
The loop is unrolled over the ring's 6 slots = 3 pairs, so that every LDS address is a base register plus an immediate.

    python3 hpc_amd/csrc/gen_hub_chain.py        (the output is committed; re-run after editing this file)

Registers (fixed, all on the clobber list, packed from v16 up: round 5): B set 0 (a stage's first half) v[16:47], B set 1 (second half) v[48:79];
a values of even stages v[80:80+NA-1], of odd stages v[80+NA:80+2NA-1] (NA = 4, or 16 with quad_perm); then the polled count, a scratch,
this lane's B address for slots 0-2 / 3-5, this lane's a address for slots 0-2 / 3-5, the address of flags[] (v88 .. v94 with NA = 4); s80 stages finished, s81 byte offset of the tail stage's slot, s84 the count `pub` must have reached before the next
pair starts fetching, s85 scratch, s86 stages the pair loop covers (nf with bit 0 cleared).
Operands: %[acc] +v the lane's accumulator | %[bb] v LDS byte address of this lane's column in slot 0 | %[ra] v LDS byte
address this lane reads slot 0's a values from (a values + 16 (lane & 15); quad_perm: + 16 (lane & 3)) | %[fl] s LDS byte
address of flags[] | %[nf] s whole stages (>= 1).
Protocol (the C++ around it, spmm_kernels.hpp): stage t lives in slot t % 6; flags[0] = pub: stages 0 .. pub-1 are written
(INT_MAX once the row's last stage is); flags[3] = done: stages 0 .. done-1 are consumed, their slots may be refilled;
stage 0 is known to be published on entry.
"""
import os

ST, LOADERS, NB, CS = 64, 3, 6, 68          # HubCfg: nonzeros per stage, loader waves, ring slots, column stride in floats
A_MODE = "row"                              # "row": DPP row_newbcast, 1 a read per stage | "quad": DPP quad_perm, 4 a reads per stage
# Round 5: the fixed registers are packed downwards (they were v32 .. v134 with gaps, which alone made spmm_hub<16> -- and the small-step kernel that runs
# it as one of its roles -- a 135-VGPR, 3-waves-per-SIMD kernel).  Nothing but the names changes.
B = [16, 48]                                # first VGPR of B set 0 (first half of a stage) / 1 (second half)
NA = 4 if A_MODE == "row" else 16
A = [80, 80 + NA]                           # first VGPR of the a values of even / odd stages
_M = 80 + 2 * NA
V_POLL, V_TMP, V_B, V_A, V_FL = _M, _M + 1, (_M + 2, _M + 3), (_M + 4, _M + 5), _M + 6
V_FIRST = B[0]
DONE_OFF = 4 * LOADERS                      # flags[3]


def slot_bytes(sw):
    return (sw * CS + ST) * 4


def a_offset(sw):
    """byte offset of a stage's a values inside its slot"""
    return sw * CS * 4


def read_b(s, sw, slot, half, base=None):
    """the 8 reads of one half stage of this lane's column into B set s"""
    vb = base if base is not None else f"v{V_B[0] if slot < 3 else V_B[1]}"
    off = (slot % 3) * slot_bytes(sw) + 128 * half if base is None else 128 * half
    return [f"ds_read_b128 v[{B[s] + 4 * q}:{B[s] + 4 * q + 3}], {vb} offset:{off + 16 * q}" for q in range(8)]


def read_a(par, sw, slot):
    """a stage's 64 a values into the set of its parity"""
    va = f"v{V_A[0] if slot < 3 else V_A[1]}"
    off = (slot % 3) * slot_bytes(sw)
    if A_MODE == "row":      # register r of row lane i: a[4 i + r]
        return [f"ds_read_b128 v[{A[par]}:{A[par] + 3}], {va} offset:{off}"]
    # quad: register 4 g + r of quad lane i: a[16 g + 4 i + r]
    return [f"ds_read_b128 v[{A[par] + 4 * g}:{A[par] + 4 * g + 3}], {va} offset:{off + 64 * g}" for g in range(4)]


def chain32(s, par, half):
    """links 32 half .. 32 half + 31 of a stage: B set s, the stage's a values (parity par)"""
    out = []
    for j in range(32):
        k = 32 * half + j
        if A_MODE == "row":
            out.append(f"v_fmac_f32_dpp %[acc], v{A[par] + k % 4}, v{B[s] + j} row_newbcast:{k // 4} row_mask:0xf bank_mask:0xf")
        else:
            i = (k % 16) // 4
            out.append(f"v_fmac_f32_dpp %[acc], v{A[par] + 4 * (k // 16) + k % 4}, v{B[s] + j} quad_perm:[{i},{i},{i},{i}] row_mask:0xf bank_mask:0xf")
    return out


def text(sw):
    sb = slot_bytes(sw)
    assert 2 * sb + a_offset(sw) + 256 < 65536          # every immediate offset fits 16 bits
    L = []
    # ---- entry: base addresses; set 0 and the even a set <- first half of stage 0 (slot 0)
    L += [f"v_mov_b32 v{V_B[0]}, %[bb]", f"v_add_u32 v{V_B[1]}, {3 * sb}, v{V_B[0]}",
          f"v_mov_b32 v{V_A[0]}, %[ra]", f"v_add_u32 v{V_A[1]}, {3 * sb}, v{V_A[0]}",
          f"v_mov_b32 v{V_FL}, %[fl]"]
    L += read_b(0, sw, 0, 0) + read_a(0, sw, 0)
    L += ["s_mov_b32 s80, 0", "s_mov_b32 s81, 0", "s_mov_b32 s84, 3", "s_and_b32 s86, %[nf], -2",
          "s_waitcnt lgkmcnt(0)",
          "s_cmp_eq_u32 s86, 0", "s_cbranch_scc1 80f"]
    # ---- pairs of whole stages (t, t + 1), t even; copy k serves the pairs with t % 6 == k
    for k in (0, 2, 4):
        k1, k2 = k + 1, (k + 2) % NB
        L += [f"1{k}:",
              f"ds_read_b32 v{V_POLL}, v{V_FL}"]                               # pub, looked at 32 links from now
        L += read_b(1, sw, k, 1)                                               # second half of stage t
        L += chain32(0, 0, 0)
        L += ["s_waitcnt lgkmcnt(0)",                                          # 32 links after the reads: they have landed
              f"v_readfirstlane_b32 s85, v{V_POLL}", "s_cmp_ge_i32 s85, s84", f"s_cbranch_scc0 2{k}f",
              f"3{k}:",                                                        # stages t + 1 and t + 2 are published
              "s_add_u32 s84, s84, 2"]
        L += read_b(0, sw, k1, 0) + read_a(1, sw, k1)                          # first half of stage t + 1, its a values
        L += chain32(1, 0, 1)
        L += ["s_waitcnt lgkmcnt(0)"]
        L += read_b(1, sw, k1, 1)                                              # second half of stage t + 1
        L += ["s_add_u32 s80, s80, 2", f"v_mov_b32 v{V_TMP}, s80",
              f"ds_write_b32 v{V_FL}, v{V_TMP} offset:{DONE_OFF}"]             # behind those reads in this wave's LDS queue: both slots may be refilled
        L += chain32(0, 1, 0)
        L += ["s_waitcnt lgkmcnt(0)"]
        L += read_b(0, sw, k2, 0) + read_a(0, sw, k2)                          # first half of stage t + 2 (whole or not: unused if not)
        L += chain32(1, 1, 1)
        L += ["s_waitcnt lgkmcnt(0)",
              "s_cmp_lt_u32 s80, s86", f"s_cbranch_scc0 4{k}f"]
        if k == 4:
            L += ["s_branch 10b"]
    # ---- out of line: the poll came back short (the loaders normally run ahead); the exits, which know their slot
    for k in (0, 2, 4):
        L += [f"2{k}:", "s_sleep 1", f"ds_read_b32 v{V_POLL}, v{V_FL}", "s_waitcnt lgkmcnt(0)",
              f"v_readfirstlane_b32 s85, v{V_POLL}", "s_cmp_ge_i32 s85, s84", f"s_cbranch_scc0 2{k}b", f"s_branch 3{k}b"]
    for k in (0, 2, 4):
        L += [f"4{k}:", f"s_mov_b32 s81, {((k + 2) % NB) * sb}", "s_branch 80f"]
    # ---- an odd number of whole stages: one more (slot offset in s81); its first half and a values are in set 0 / the even set
    L += ["80:",
          "s_and_b32 s85, %[nf], 1", "s_cbranch_scc0 99f",
          f"v_add_u32 v{V_B[0]}, s81, %[bb]"]
    L += read_b(1, sw, 0, 1, base=f"v{V_B[0]}")
    L += chain32(0, 0, 0)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += chain32(1, 0, 1)
    L += ["99:"]
    return L


def render():
    clob = [f"v{i}" for i in range(V_FIRST, V_FL + 1)] + ["s80", "s81", "s84", "s85", "s86", "scc", "memory"]
    out = ["// GENERATED by gen_hub_chain.py -- do not edit; see that file for the register map and the protocol.\n"]
    for sw in (16, 32, 64):
        out.append(f"#define MI_HUB_CHAIN_SLOT_BYTES_{sw} {slot_bytes(sw)}\n")
        out.append(f"#define MI_HUB_CHAIN_ASM_{sw} \\\n")
        out += [f'    "{line}\\n\\t" \\\n' for line in text(sw)]
        out.append('    ""\n')
    out.append(f"#define MI_HUB_CHAIN_A_LANES {16 if A_MODE == 'row' else 4}\n")
    out.append("#define MI_HUB_CHAIN_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    return "".join(out)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "hub_chain_asm.inc"), "w") as f:
        f.write(render())
    for sw in (16, 32, 64):
        t = text(sw)
        print(f"SW = {sw}: {len(t)} lines of assembly, slot {slot_bytes(sw)} bytes, a values by DPP '{A_MODE}'")
