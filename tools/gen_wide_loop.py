#!/usr/bin/env python3
"""Generates vame_amd/csrc/gru_wide_loop.inc: the inline-assembly contraction loop of gru_wide.hip (one GRU step of one wave:
h_{t-1} (LDS) x W_hh fragments (L2) -> six 32x32 gate accumulators).  Written as assembly because the loop keeps W_hh fragment loads
in flight across its back edge: hipcc either drains the memory queue at every back edge (plain loads) or moves / copies registers that
an inline-asm load is still going to write (asm loads with C++ around them) -- see gru_wide.hip.  Everything between the first load and
the last MFMA is therefore one asm statement on FIXED scratch registers v192..v252 (declared as clobbers).

    python tools/gen_wide_loop.py > vame_amd/csrc/gru_wide_loop.inc
"""
A = [192, 196]                      # a-fragment slots (4 registers each): h rows, k = 8c + 4*(lane>>5) + e
R = [200, 224]                      # weight slots (24 registers each): [block][gate] x float4
P0, P1, LDS = 248, 250, 252         # 64-bit fragment pointers of the two column blocks, LDS address
ACC = [["%[ar0]", "%[au0]", "%[nh0]"], ["%[ar1]", "%[au1]", "%[nh1]"]]      # [block][gate]
CHUNK = 3 * 64 * 16                 # bytes of one chunk of one block: 3 gates x 64 lanes x float4


def preamble(q0, q1, lds, t0, t1):
    """Everything per-lane is derived inside the statement from the lane id, so that the C++ side keeps no vector register alive for the
    loop between steps (hipcc spilled exactly those and reloaded them behind a vmcnt(0)): fragment pointers v[q:q+1] = uniform 64-bit block
    base (two scalar operands each) + lane * 16; A-operand row address = scalar LDS base + (lane & 31) * row bytes + (lane >> 5) * 16.
    t0, t1: scratch registers that the loop overwrites later."""
    out = ["s_nop 4", f"v_mbcnt_lo_u32_b32 v{t0}, -1, 0", f"v_mbcnt_hi_u32_b32 v{t0}, -1, v{t0}", f"v_lshlrev_b32_e32 v{t1}, 4, v{t0}"]
    for q, name in ((q0, "b0"), (q1, "b1")):
        out += [f"v_mov_b32_e32 v{q + 1}, %[{name}hi]", f"v_add_co_u32_e32 v{q}, vcc, %[{name}lo], v{t1}",
                f"v_addc_co_u32_e32 v{q + 1}, vcc, 0, v{q + 1}, vcc"]
    out += [f"v_and_b32_e32 v{t1}, 31, v{t0}", f"v_lshrrev_b32_e32 v{t0}, 5, v{t0}", f"v_mul_u32_u24_e32 v{t1}, %[rowb], v{t1}",
            f"v_lshl_add_u32 v{t0}, v{t0}, 4, v{t1}", f"v_add_u32_e32 v{lds}, %[ldsb], v{t0}", "s_nop 1"]
    return out


def loads(slot):
    out = []
    for cb, p in enumerate((P0, P1)):
        for g in range(3):
            r = R[slot] + (cb * 3 + g) * 4
            off = f" offset:{g * 1024}" if g else ""
            out.append(f"global_load_dwordx4 v[{r}:{r + 3}], v[{p}:{p + 1}], off{off}")
    for p in (P0, P1):
        out += [f"v_add_co_u32_e32 v{p}, vcc, {CHUNK}, v{p}", f"v_addc_co_u32_e32 v{p + 1}, vcc, 0, v{p + 1}, vcc"]
    return out


def mfmas(slot):
    out = []
    for e in range(4):
        for g in range(3):
            for cb in range(2):
                r = R[slot] + (cb * 3 + g) * 4 + e
                out.append(f"v_mfma_f32_32x32x2_f32 {ACC[cb][g]}, v{A[slot] + e}, v{r}, {ACC[cb][g]}")
    return out


def body(refill):
    out = [f"ds_read_b128 v[{A[0]}:{A[0] + 3}], v{LDS}", f"ds_read_b128 v[{A[1]}:{A[1] + 3}], v{LDS} offset:32",
           f"v_add_u32_e32 v{LDS}, 64, v{LDS}"]
    out += [f"s_waitcnt vmcnt({6 if refill else 6}) lgkmcnt(1)"] + mfmas(0)
    if refill:
        out += loads(0)
        out += ["s_waitcnt vmcnt(6) lgkmcnt(0)"] + mfmas(1) + loads(1)
    else:
        out += ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + mfmas(1)
    return out


def bwd_macro():
    """BPTT contraction of one gate phase: acc[cb] += A (32 x H tile in LDS) x W_hh^T fragments of the wave's two column blocks.
    Four chunks per iteration, four weight slots of 2 x dwordx4 (one per block), vmcnt(6) before a slot's 8 MFMAs."""
    NS = 4
    Aq = [192 + 4 * s_ for s_ in range(NS)]
    Rq = [208 + 8 * s_ for s_ in range(NS)]
    Q0, Q1, L = 240, 242, 244
    acc = ["%[a0]", "%[a1]"]

    def refill(s_):
        return [f"global_load_dwordx4 v[{Rq[s_]}:{Rq[s_] + 3}], v[{Q0}:{Q0 + 1}], off" + (f" offset:{s_ * 1024}" if s_ else ""),
                f"global_load_dwordx4 v[{Rq[s_] + 4}:{Rq[s_] + 7}], v[{Q1}:{Q1 + 1}], off" + (f" offset:{s_ * 1024}" if s_ else "")]

    def advance():
        out = []
        for q in (Q0, Q1):
            out += [f"v_add_co_u32_e32 v{q}, vcc, {NS * 1024}, v{q}", f"v_addc_co_u32_e32 v{q + 1}, vcc, 0, v{q + 1}, vcc"]
        return out

    def mf(s_):
        out = []
        for e in range(4):
            for cb in range(2):
                out.append(f"v_mfma_f32_32x32x2_f32 {acc[cb]}, v{Aq[s_] + e}, v{Rq[s_] + 4 * cb + e}, {acc[cb]}")
        return out

    def it(with_refill):
        out = [f"ds_read_b128 v[{Aq[s_]}:{Aq[s_] + 3}], v{L}" + (f" offset:{32 * s_}" if s_ else "") for s_ in range(NS)]
        out.append(f"v_add_u32_e32 v{L}, {32 * NS}, v{L}")
        for s_ in range(NS):
            vm = 2 * (NS - 1) if with_refill else 2 * (NS - 1 - s_)
            out.append(f"s_waitcnt vmcnt({vm}) lgkmcnt({NS - 1 - s_})")
            out += mf(s_)
            if with_refill:
                out += refill(s_)
        if with_refill:
            out += advance()
        return out

    ls = preamble(Q0, Q1, L, Aq[0], Aq[0] + 1)
    for s_ in range(NS):
        ls += refill(s_)
    ls += advance()
    ls += ["1:"] + it(True) + ["s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 1b"] + it(False) + ["s_nop 7", "s_nop 7", "s_nop 7"]
    return ls, list(range(192, 245))


lines = preamble(P0, P1, LDS, A[0], A[0] + 1)
lines += loads(0) + loads(1)
lines += ["1:"] + body(True) + ["s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 1b"]
lines += body(False)
lines += ["s_nop 7", "s_nop 7", "s_nop 7"]
print("// GENERATED by tools/gen_wide_loop.py -- do not edit.  One GRU step's recurrent contraction of one wave (see gru_wide.hip).")
print("#define GRU_WIDE_LOOP_ASM \\")
for l in lines:
    print(f'    "{l}\\n\\t" \\')
print('    ""')
clob = ", ".join(f'"v{r}"' for r in range(192, 253))
print(f"#define GRU_WIDE_LOOP_CLOBBERS {clob}")
bl, bc = bwd_macro()
print("// BPTT: one gate phase (K = H) of dh_{t-1} = [da_r | da_z | dgh_n] W_hh for the wave's two column blocks (gru_wide.hip)")
print("#define GRU_WIDE_BWD_LOOP_ASM \\")
for l in bl:
    print(f'    "{l}\\n\\t" \\')
print('    ""')
print("#define GRU_WIDE_BWD_LOOP_CLOBBERS " + ", ".join(f'"v{r}"' for r in bc))
