#!/usr/bin/env python3
"""Writes hub_chain_asm.inc: the chain wave's loop of spmm_hub (spmm_kernels.hpp) as ONE inline-assembly block per slice width.

Why assembly: the chain wave of a hub row is a single wave whose every instruction costs it 5-6 cycles, whatever the
instruction is, and whose LDS reads cost least issued back to back in front of the chain
(scripts/experiments/chain_patterns.hip; profiles/r03_hub_experiments.txt).  hipcc kept rearranging the C++ form of this
loop -- sinking the 32 fmas below the flag traffic, splitting the 16 reads 3 + 13 around a full drain, waiting for one of
a trip's fresh reads because the 4-bit counter cannot say "the 16 older ones" -- and every such rearrangement is a
10-20 % loss on rows that are nothing but this loop.  The block below is the schedule the measurements picked, and
nothing in it moves.  The arithmetic is the same chain: acc = fma(a_k, b_k, acc), k ascending, one v_fmac_f32 per nonzero.

The loop is unrolled over the ring's 6 slots, so that every LDS address is a base register plus an immediate and a stage
costs 11 instructions beside its 64 fmas and 32 reads (rolled, with the addresses computed per stage: 29).

    python3 hpc_amd/csrc/gen_hub_chain.py        (the output is committed; re-run after editing this file)

Registers (fixed, all on the clobber list): B set 0 v[32:63], A set 0 v[64:95], B set 1 v[96:127], A set 1 v[128:159];
v160 polled count, v161 scratch, v162/v163 this lane's B address for slots 0-2 / 3-5, v164/v165 the a values' address for
slots 0-2 / 3-5, v166 address of flags[]; s80 stages finished, s81 byte offset of the last whole stage's slot, s84 the
count the next stage's loader must have published, s85 scratch, s86 n_full - 1.
Operands: %[acc] +v the lane's accumulator | %[bb] v LDS byte address of this lane's column in slot 0 | %[rba] s LDS byte
address of slot 0's a values | %[fl] s LDS byte address of flags[] | %[nf] s whole stages (>= 1).
Protocol (the C++ around it, spmm_kernels.hpp): stage t lives in slot t % 6 and is loader t % 3's stage number t / 3;
loader w publishes the number of stages it has written in flags[w]; flags[3] = stages whose slot may be refilled;
stage 0 is known to be published on entry.
"""
import os

ST, LOADERS, NB, CS = 64, 3, 6, 68          # HubCfg: nonzeros per stage, loader waves, ring slots, column stride in floats
B = [32, 96]                                # first VGPR of B set 0 / 1
A = [64, 128]


def slot_bytes(sw):
    return (sw * CS + ST) * 4


def read16(s, sw, slot, half):
    """the 16 reads of one half stage into set s: 8 quads of this lane's column, 8 broadcast quads of a values"""
    vb, va = ("v162", "v164") if slot < 3 else ("v163", "v165")
    off = (slot % 3) * slot_bytes(sw) + 128 * half
    out = [f"ds_read_b128 v[{B[s] + 4 * q}:{B[s] + 4 * q + 3}], {vb} offset:{off + 16 * q}" for q in range(8)]
    out += [f"ds_read_b128 v[{A[s] + 4 * q}:{A[s] + 4 * q + 3}], {va} offset:{off + 16 * q}" for q in range(8)]
    return out


def chain32(s):
    return [f"v_fmac_f32 %[acc], v{A[s] + j}, v{B[s] + j}" for j in range(32)]


def text(sw):
    sb = slot_bytes(sw)
    assert 2 * sb + 128 + 112 < 65536
    L = []
    # ---- entry: base addresses; set 0 <- first half of stage 0 (slot 0)
    L += ["v_mov_b32 v162, %[bb]", f"v_add_u32 v163, {3 * sb}, v162",
          "v_mov_b32 v164, %[rba]", f"v_add_u32 v165, {3 * sb}, v164",
          "v_mov_b32 v166, %[fl]"]
    L += read16(0, sw, 0, 0)
    L += ["s_mov_b32 s80, 0", "s_mov_b32 s81, 0", "s_mov_b32 s84, 1", "s_sub_u32 s86, %[nf], 1",
          "s_waitcnt lgkmcnt(0)",
          "s_cmp_eq_u32 s86, 0", "s_cbranch_scc1 90f"]
    # ---- every whole stage but the last; copy k serves the stages with t % 6 == k
    for k in range(NB):
        nxt = (k + 1) % NB
        L += [f"1{k}:",
              f"ds_read_b32 v160, v166 offset:{4 * ((k + 1) % LOADERS)}"]      # the next stage's published count, a trip ahead of its use
        L += read16(1, sw, k, 1)                                               # second half of this stage
        L += ["s_add_u32 s80, s80, 1", "v_mov_b32 v161, s80",
              f"ds_write_b32 v166, v161 offset:{4 * LOADERS}"]                 # behind the reads in this wave's LDS queue: the slot may be refilled
        L += chain32(0)
        L += ["s_waitcnt lgkmcnt(0)",                                          # 32 links after the reads: they have landed
              "v_readfirstlane_b32 s85, v160", "s_cmp_ge_i32 s85, s84", f"s_cbranch_scc0 2{k}f",
              f"3{k}:"]
        if (k + 2) % LOADERS == 0:
            L += ["s_add_u32 s84, s84, 1"]                                     # stage t + 2 is the next one of loader 0: one more from now on
        L += read16(0, sw, nxt, 0)                                             # first half of the next stage
        L += chain32(1)
        L += ["s_waitcnt lgkmcnt(0)",
              "s_cmp_eq_u32 s80, s86", f"s_cbranch_scc1 4{k}f"]
        if k == NB - 1:
            L += ["s_branch 10b"]
    # ---- out of line: the poll came back short (the loaders normally run ahead); the exits, which know their slot
    for k in range(NB):
        L += [f"2{k}:", "s_sleep 1", f"ds_read_b32 v160, v166 offset:{4 * ((k + 1) % LOADERS)}", "s_waitcnt lgkmcnt(0)",
              "v_readfirstlane_b32 s85, v160", "s_cmp_ge_i32 s85, s84", f"s_cbranch_scc0 2{k}b", f"s_branch 3{k}b"]
    for k in range(NB):
        nxt = (k + 1) % NB
        L += [f"4{k}:", f"s_mov_b32 s81, {nxt * sb}", "s_branch 90f"]
    # ---- the last whole stage (slot offset in s81): nothing to fetch behind it
    L += ["90:",
          "v_add_u32 v162, s81, %[bb]", "s_add_u32 s85, s81, %[rba]", "v_mov_b32 v164, s85"]
    L += read16(1, sw, 0, 1)                                                   # v162 / v164 now point at the slot itself
    L += ["s_add_u32 s80, s80, 1", "v_mov_b32 v161, s80", f"ds_write_b32 v166, v161 offset:{4 * LOADERS}"]
    L += chain32(0)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += chain32(1)
    return L


def render():
    clob = [f"v{i}" for i in range(32, 167)] + ["s80", "s81", "s84", "s85", "s86", "scc", "memory"]
    out = ["// GENERATED by gen_hub_chain.py -- do not edit; see that file for the register map and the protocol.\n"]
    for sw in (16, 32, 64):
        out.append(f"#define MI_HUB_CHAIN_ASM_{sw} \\\n")
        out += [f'    "{line}\\n\\t" \\\n' for line in text(sw)]
        out.append('    ""\n')
    out.append("#define MI_HUB_CHAIN_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    return "".join(out)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "hub_chain_asm.inc"), "w") as f:
        f.write(render())
    for sw in (16, 32, 64):
        print(f"SW = {sw}: {len(text(sw))} lines of assembly, slot {slot_bytes(sw)} bytes")
