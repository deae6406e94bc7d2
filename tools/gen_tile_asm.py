#!/usr/bin/env python3
"""Generates azdopt_amd/csrc/tile_task_asm.inc: the k loop of an evaluator tile task (dfdx.rs:69-84, one 16-column tile of one
layer for a batch of <= 16 rows) as ONE inline-asm block per storage type.

Why asm: the loop wants the weight quads of the NEXT group of 8 k-steps requested before the MFMAs of the current group, and
"the current group has landed" is then `s_waitcnt vmcnt(8)`.  hipcc's own wait insertion cannot be talked into that for a loop
with a runtime trip count: at every merge of control flow it drains to vmcnt(0) and it throttles requests into registers it
believes in flight (tools/probes/group_exchange.hip: k_classic2, the ISA in profiles/r05_tile_pipeline.txt), and requests issued
by inline asm into compiler-allocated registers get spilled before they land.  Inside one asm block with registers of its own
(clobbers) nothing of that can happen.  The MFMA sequence per output element is exactly mlp_tile_task's (async_step.inc): k-steps
in ascending order, per step the quads' x, y, z, w -- results are bit-identical.

Register plan (VGPR): a0 v[56:59], a1 v[60:63] (the rows' quads, one step ahead), ring A v[64:95], ring B v[96:127] (8 steps x 4
dwords each; bf16 storage: 8 x 2 dwords, v[64:79] / v[80:95], quads v[56:57] / v[58:59]).  Requests use the SGPR-base form: the
tile's wave-uniform address (and that + 4 KB: the immediate offset reaches 4095) in SGPR pairs, the lane's offset in one VGPR that
moves on by a group per group.
Inputs: the wave-uniform address of k-step 0 of the tile's fragment-major weights, the lane's byte offset inside a k-step, the LDS
byte address of its row quad at k-step 0 (64 B per step; bf16 32), the number of k-steps.  A ragged last group still issues 8 requests (the fragment-major copy is
padded by 8 KB at its end: mlp_kernels.hip), its MFMAs stop at the last step.
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "azdopt_amd", "csrc", "tile_task_asm.inc")
A0, RA = 56, 64


def ring_b(bf16):
    return RA + (16 if bf16 else 32)


def v(lo, n=1):
    return "v%d" % lo if n == 1 else "v[%d:%d]" % (lo, lo + n - 1)


def loads(ring, bf16):
    """8 requests: the tile's uniform base in an SGPR pair (b0 for steps 0-3 of the group, b1 for 4-7: the immediate offset reaches
    4095), the lane's offset (16 B, or 8 B per lane) in one VGPR"""
    out = []
    for j in range(8):
        base = "%[b0]" if j < 4 else "%[b1]"
        if bf16:   # 512 B per k-step, 8 B per lane
            out.append("global_load_dwordx2 %s, %%[lo], %s offset:%d" % (v(ring + 2 * j, 2), base, 512 * (j & 3)))
        else:      # 1 KB per k-step, 16 B per lane
            out.append("global_load_dwordx4 %s, %%[lo], %s offset:%d" % (v(ring + 4 * j, 4), base, 1024 * (j & 3)))
    return out


def advance(bf16):
    step = 4096 if bf16 else 8192  # one group of 8 k-steps
    return ["v_add_u32 %%[lo], 0x%x, %%[lo]" % step]


def consume(ring, bf16, tag):
    """8 k-steps out of `ring`; step j is skipped -- with everything behind it -- once done + j >= steps"""
    rd = "ds_read_b64" if bf16 else "ds_read_b128"
    an = 2 if bf16 else 4
    abytes = 32 if bf16 else 64
    A1 = A0 + an
    out = ["%s %s, %%[ap]" % (rd, v(A0, an)), "%s %s, %%[ap] offset:%d" % (rd, v(A1, an), abytes)]
    for j in range(8):
        a = A0 if (j & 1) == 0 else A1
        if j:
            out += ["s_add_i32 %[t0], %[done], " + str(j), "s_cmp_ge_i32 %[t0], %[steps]", "s_cbranch_scc1 9f"]
        out.append("s_waitcnt lgkmcnt(1)" if j < 7 else "s_waitcnt lgkmcnt(0)")
        if bf16:
            out.append("v_mfma_f32_16x16x16_bf16 %%[acc], %s, %s, %%[acc]" % (v(a, 2), v(ring + 2 * j, 2)))
        else:
            for i in range(4):
                out.append("v_mfma_f32_16x16x4_f32 %%[acc], %s, %s, %%[acc]" % (v(a + i), v(ring + 4 * j + i)))
        if j + 2 < 8:  # the quad two steps on, into the register pair just used
            out.append("%s %s, %%[ap] offset:%d" % (rd, v(a, an), abytes * (j + 2)))
        elif j == 6:
            pass
    return out


def loads_from(ring, bf16, b0, b1, lo):
    out = []
    for j in range(8):
        base = b0 if j < 4 else b1
        if bf16:
            out.append("global_load_dwordx2 %s, %s, %s offset:%d" % (v(ring + 2 * j, 2), lo, base, 512 * (j & 3)))
        else:
            out.append("global_load_dwordx4 %s, %s, %s offset:%d" % (v(ring + 4 * j, 4), lo, base, 1024 * (j & 3)))
    return out


def body(bf16, chained):
    """chained: the block may start with group 0 already requested into ring A (state 1) or ring B (state 2) by the block
    before it, and during its own last group requests group 0 of the NEXT tile task (base nb0 / nb1, 0 = none) into the ring that
    is free then, leaving through `state` which ring that was; the rings are operands pinned to their registers, so that what is
    in flight between two blocks (the epilogue, the layer barrier) is the compiler's to leave alone."""
    lines = []
    RB = ring_b(bf16)
    if chained:
        lines += ["s_cmp_eq_u32 %[state], 0", "s_cbranch_scc0 4f"]
        lines += loads(RA, bf16)
        lines += ["4:", "s_mov_b32 %[done], 0", "s_cmp_eq_u32 %[state], 2", "s_mov_b32 %[state], 0", "s_cbranch_scc1 5f"]
    else:
        lines += loads(RA, bf16)
        lines += ["s_mov_b32 %[done], 0"]
    lines += ["1:"]
    for ring, other, tag in ((RA, RB, "a"), (RB, RA, "b")):
        if tag == "b" and chained:
            lines += ["5:"]
        # request the group after this one into the other ring, if there is one; then this ring has landed
        lines += ["s_add_i32 %[t0], %[done], 8", "s_cmp_lt_i32 %[t0], %[steps]", "s_cbranch_scc0 2f"]
        lines += advance(bf16)
        lines += loads(other, bf16)
        lines += ["s_waitcnt vmcnt(8)", "s_branch 3f", "2:"]
        if chained:  # the tile's last group: the next task's first group into the free ring
            lines += ["s_cmp_eq_u64 %[nb0], 0", "s_cbranch_scc1 6f"]
            lines += loads_from(other, bf16, "%[nb0]", "%[nb1]", "%[lo0]")
            lines += ["s_mov_b32 %%[state], %d" % (2 if other == RB else 1), "s_waitcnt vmcnt(8)", "s_branch 3f", "6:"]
        lines += ["s_waitcnt vmcnt(0)", "3:"]
        lines += consume(ring, bf16, tag)
        lines += ["s_add_i32 %[done], %[done], 8", "v_add_u32 %%[ap], 0x%x, %%[ap]" % (256 if bf16 else 512),
                  "s_cmp_ge_i32 %[done], %[steps]", "s_cbranch_scc1 9f"]
    lines += ["s_branch 1b", "9:", "s_waitcnt lgkmcnt(0)" if chained else "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return lines


def emit(name, bf16):
    text = "\\n\\t\"\n        \"".join(body(bf16, False))
    clob = ['"memory"', '"scc"'] + ['"v%d"' % r for r in list(range(A0, A0 + (4 if bf16 else 8))) + list(range(RA, RA + (32 if bf16 else 64)))]
    return ('''// ---- generated by tools/gen_tile_asm.py: do not edit
__device__ __forceinline__ void %s(azd_tile_acc &acc, const void *tile_base, uint32_t lane_off, uint32_t ap, const int steps) {
    // tile_base: wave-uniform address of the tile's k-step 0; lane_off: this lane's byte offset inside a k-step
    const unsigned long long b0 = azd_uniform_u64((unsigned long long)(uintptr_t)tile_base), b1 = b0 + %d;
    int done, t0;
    asm volatile(
        "%s\\n\\t"
        : [acc] "+v"(acc), [ap] "+v"(ap), [lo] "+v"(lane_off), [done] "=&s"(done), [t0] "=&s"(t0)
        : [b0] "s"(b0), [b1] "s"(b1), [steps] "s"(__builtin_amdgcn_readfirstlane(steps))
        : %s);
}
''' % (name, 2048 if bf16 else 4096, text, ", ".join(clob)))


def emit_chained(name, bf16):
    text = "\\n\\t\"\n        \"".join(body(bf16, True))
    n = 2 if bf16 else 4
    rt = "azd_tile_ring16" if bf16 else "azd_tile_ring32"
    ring_ops = ", ".join('"+{v[%d:%d]}"(ring.r[%d])' % (RA + n * k, RA + n * k + n - 1, k) for k in range(16))
    clob = ['"memory"', '"scc"'] + ['"v%d"' % r for r in range(A0, A0 + (4 if bf16 else 8))]
    return ('''// ---- generated by tools/gen_tile_asm.py: do not edit
// Chained form: `state` says whether group 0 of THIS task is already on its way (1: in ring.r[0..7], 2: in ring.r[8..15]; 0: no),
// `next_base` (0: none) is the task whose group 0 this block requests during its own last group; `state` leaves as what the next
// block must be told.  Nothing of the ring may be touched between two blocks of a chain: the operands are pinned to their
// registers, and tools/check_kernels.py audits the code between a chain's blocks for moves or spills of them.
__device__ __forceinline__ void %s(azd_tile_acc &acc, %s &ring, int &state, const void *tile_base, const void *next_base, uint32_t lane_off,
                                   uint32_t ap, const int steps) {
    const unsigned long long b0 = azd_uniform_u64((unsigned long long)(uintptr_t)tile_base), b1 = b0 + %d;
    const unsigned long long nb0 = azd_uniform_u64((unsigned long long)(uintptr_t)next_base), nb1 = nb0 + %d;
    int done, t0, st = __builtin_amdgcn_readfirstlane(state);
    const uint32_t lo0 = lane_off;
    asm volatile(
        "%s\\n\\t"
        : [acc] "+v"(acc), [ap] "+v"(ap), [lo] "+v"(lane_off), [done] "=&s"(done), [t0] "=&s"(t0), [state] "+s"(st), %s
        : [b0] "s"(b0), [b1] "s"(b1), [nb0] "s"(nb0), [nb1] "s"(nb1), [lo0] "v"(lo0), [steps] "s"(__builtin_amdgcn_readfirstlane(steps))
        : %s);
    state = st;
}
''' % (name, rt, 2048 if bf16 else 4096, 2048 if bf16 else 4096, text, ring_ops, ", ".join(clob)))


def body_ga(stride):
    """The group evaluator's k loop (pool_step.inc: pool_eval_group): the ROWS' quads come from device memory -- the agents' state
    vectors (64 B per k-step and row) or the exchange buffer of the layer before (fragment-major: 1 KB per k-step) -- by sc1
    buffer loads, two rings of 8 k-steps as above; the WEIGHT quads come from the member's LDS (1 KB per k-step), one step ahead."""
    RB = RA + 32
    lines = []

    def loads(ring):
        out = []
        for j in range(8):
            if stride == 64:
                out.append("buffer_load_dwordx4 %s, %%[vo], %%[rs], 0 offen offset:%d sc1" % (v(ring + 4 * j, 4), 64 * j))
            else:
                out.append("buffer_load_dwordx4 %s, %%[vo], %%[rs], %s offen offset:%d sc1" % (v(ring + 4 * j, 4), "0" if j < 4 else "%[k4]", 1024 * (j & 3)))
        return out

    def consume(ring):
        out = ["ds_read_b128 %s, %%[bp]" % v(A0, 4), "ds_read_b128 %s, %%[bp] offset:1024" % v(A0 + 4, 4)]
        for j in range(8):
            b = A0 if (j & 1) == 0 else A0 + 4
            if j:
                out += ["s_add_i32 %[t0], %[done], " + str(j), "s_cmp_ge_i32 %[t0], %[steps]", "s_cbranch_scc1 9f"]
            out.append("s_waitcnt lgkmcnt(1)" if j < 7 else "s_waitcnt lgkmcnt(0)")
            for i in range(4):
                out.append("v_mfma_f32_16x16x4_f32 %%[acc], %s, %s, %%[acc]" % (v(ring + 4 * j + i), v(b + i)))
            if j + 2 < 8:
                out.append("ds_read_b128 %s, %%[bp] offset:%d" % (v(b, 4), 1024 * (j + 2)))
        return out

    lines += loads(RA)
    lines += ["s_mov_b32 %[done], 0", "1:"]
    for ring, other in ((RA, RB), (RB, RA)):
        lines += ["s_add_i32 %[t0], %[done], 8", "s_cmp_lt_i32 %[t0], %[steps]", "s_cbranch_scc0 2f"]
        lines += ["v_add_u32 %%[vo], 0x%x, %%[vo]" % (8 * stride)]
        lines += loads(other)
        lines += ["s_waitcnt vmcnt(8)", "s_branch 3f", "2:", "s_waitcnt vmcnt(0)", "3:"]
        lines += consume(ring)
        lines += ["s_add_i32 %[done], %[done], 8", "v_add_u32 %[bp], 0x2000, %[bp]", "s_cmp_ge_i32 %[done], %[steps]", "s_cbranch_scc1 9f"]
    lines += ["s_branch 1b", "9:", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return lines


def emit_ga(name, stride):
    text = "\\n\\t\"\n        \"".join(body_ga(stride))
    clob = ['"memory"', '"scc"'] + ['"v%d"' % r for r in list(range(A0, A0 + 8)) + list(range(RA, RA + 64))]
    return ('''// ---- generated by tools/gen_tile_asm.py: do not edit
// The group evaluator's k loop: rows' quads from device memory (buffer resource `rs`, this lane's byte offset of k-step 0 in `vo`,
// %d B per k-step; sc1: L1 bypassed), weight quads from LDS (`bp`: this lane's quad of k-step 0, 1 KB per k-step).  A ragged last
// group still issues 8 requests: they stay inside the buffer resource or are dropped by its range check.
__device__ __forceinline__ void %s(azd_tile_acc &acc, const azd_tile_u4 rs_in, uint32_t vo, uint32_t bp, const int steps) {
    azd_tile_u4 rs; // (made uniform for the compiler's eyes: an "s" operand)
    rs[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rs_in[0]);
    rs[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rs_in[1]);
    rs[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rs_in[2]);
    rs[3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rs_in[3]);
    int done, t0;
    const int k4 = __builtin_amdgcn_readfirstlane(4096);
    asm volatile(
        "%s\\n\\t"
        : [acc] "+v"(acc), [vo] "+v"(vo), [bp] "+v"(bp), [done] "=&s"(done), [t0] "=&s"(t0)
        : [rs] "s"(rs), [k4] "s"(k4), [steps] "s"(__builtin_amdgcn_readfirstlane(steps))
        : %s);
}
''' % (stride, name, text, ", ".join(clob)))


def main():
    with open(OUT, "w") as f:
        f.write("// tile_task_asm.inc -- the software-pipelined k loop of an evaluator tile task (see tools/gen_tile_asm.py for the why and the\n"
                "// register plan).  azd_tile_acc = float x 4 (the 16 x 16 accumulator tile: rows 4 (lane >> 4) + i, column lane & 15).\n")
        f.write("typedef float azd_tile_acc __attribute__((ext_vector_type(4)));\n")
        f.write("typedef unsigned int azd_tile_u2 __attribute__((ext_vector_type(2)));\n")
        f.write("struct azd_tile_ring32 { azd_tile_acc r[16]; }; // two rings of 8 k-steps x 16 B per lane (f32 weights): v[64:127]\n")
        f.write("struct azd_tile_ring16 { azd_tile_u2 r[16]; };  // ... x 8 B per lane (bf16 weights): v[64:95]\n")
        f.write("__device__ __forceinline__ unsigned long long azd_uniform_u64(const unsigned long long x) { // an \"s\" operand must be uniform in the compiler's eyes\n"
                "    return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |\n"
                "           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32);\n}\n")
        f.write(emit("tile_k_loop_f32", False))
        f.write(emit("tile_k_loop_bf16", True))
        f.write("typedef unsigned int azd_tile_u4 __attribute__((ext_vector_type(4)));\n")
        f.write(emit_ga("tile_k_loop_f32_ga64", 64))
        f.write(emit_ga("tile_k_loop_f32_ga1024", 1024))
        f.write(emit_chained("tile_k_chain_f32", False))
        f.write(emit_chained("tile_k_chain_bf16", True))
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
