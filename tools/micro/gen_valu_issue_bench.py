#!/usr/bin/env python
"""Generator of tools/micro/valu_issue_bench.hip: what does a wave64 instruction of the classes the blend kernels are built
from cost on gfx950, measured in SHADER CYCLES (s_memtime inside the kernel; s_memrealtime next to it gives the clock)?

Round 2's table (removed in round 4) divided wall time by an assumed 2.4 GHz and let the compiler pick the
registers; VERDICT r02 asked for a reconciliation with the guide's 2 cycles per v_fma_f32.  Here every loop body is ONE asm
block with fixed registers (so operand banks, encodings and dependencies are exactly what the variant's name says), every wave
stamps s_memtime / s_memrealtime around its loop and reports the SIMD it ran on, and the host prints, per variant and per
resident-waves-per-SIMD setting: cycles per instruction per SIMD (= wave cycles / instructions / waves on that SIMD), the clock
the waves saw, and the spread.

    python tools/micro/gen_valu_issue_bench.py            # writes tools/micro/valu_issue_bench.hip
    hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue_bench.hip -o tools/micro/valu_issue_bench
"""
import os

ACC = list(range(16, 32))        # accumulators v16..v31
V = lambda i: "v%d" % i
# sources: v8..v15 (banks 0..3 twice), entry operands v32..v47, temporaries v48..v63, SGPR operands s60..s75

VARIANTS = []   # (name, n_valu_counted, body_text)


def rep_acc(fmt, n=32):
    """n instructions cycling over the 16 accumulators: two consecutive uses of one accumulator are 16 instructions apart"""
    return "\n".join(fmt.format(a=V(ACC[i % 16])) for i in range(n))


def chain(fmt, chains, n=32):
    return "\n".join(fmt.format(a=V(ACC[i % chains])) for i in range(n))


def add(name, body, n=32):
    VARIANTS.append((name, n, body))


# ---- group A: operand kinds, 16 independent accumulators ---------------------------------------------------------------
add("v_fma_f32 a, v8, v9, a      (3 VGPR, sources in banks 0,1)", rep_acc("v_fma_f32 {a}, v8, v9, {a}"))
add("v_fma_f32 a, v8, v12, a     (3 VGPR, sources both bank 0)", rep_acc("v_fma_f32 {a}, v8, v12, {a}"))
add("v_fma_f32 a, s60, v9, a     (1 SGPR + 2 VGPR)", rep_acc("v_fma_f32 {a}, s60, v9, {a}"))
add("v_fma_f32 a, 2.0, v9, a     (inline constant + 2 VGPR)", rep_acc("v_fma_f32 {a}, 2.0, v9, {a}"))
add("v_fma_f32 a, v8, v9, v10    (3 VGPR sources, accumulator only written)", rep_acc("v_fma_f32 {a}, v8, v9, v10"))
add("v_fma_f32 a, s60, v9, v10   (SGPR, accumulator only written)", rep_acc("v_fma_f32 {a}, s60, v9, v10"))
add("v_fmac_f32 a, v8, v9        (VOP2 encoding, 4 bytes)", rep_acc("v_fmac_f32 {a}, v8, v9"))
add("v_fmac_f32 a, s60, v9       (VOP2, SGPR)", rep_acc("v_fmac_f32 {a}, s60, v9"))
add("v_mul_f32 a, v8, v9         (VOP2, 2 VGPR reads, no accumulator read)", rep_acc("v_mul_f32 {a}, v8, v9"))
add("v_mul_f32 a, s60, v9        (VOP2, SGPR + VGPR)", rep_acc("v_mul_f32 {a}, s60, v9"))
add("v_mul_f32 a, v8, a          (VOP2, reads its accumulator)", rep_acc("v_mul_f32 {a}, v8, {a}"))
add("v_add_f32 a, v8, a", rep_acc("v_add_f32 {a}, v8, {a}"))
add("v_mov_b32 a, v8", rep_acc("v_mov_b32 {a}, v8"))
add("v_mov_b32 a, s60", rep_acc("v_mov_b32 {a}, s60"))
add("v_min_f32 a, v8, a", rep_acc("v_min_f32 {a}, v8, {a}"))
add("v_and_b32 a, v8, a", rep_acc("v_and_b32 {a}, v8, {a}"))
# ---- group B: dependency distance ------------------------------------------------------------------------------------------
for c in (1, 2, 4, 8):
    add("v_fma_f32 a, v8, v9, a      dependent chains: %d" % c, chain("v_fma_f32 {a}, v8, v9, {a}", c))
add("v_fmac_f32 a, s60, v9       dependent chains: 1", chain("v_fmac_f32 {a}, s60, v9", 1))
add("v_mul_f32 a, v8, a          dependent chains: 2", chain("v_mul_f32 {a}, v8, {a}", 2))
# ---- group C: the other instruction classes of the inner loops ---------------------------------------------------------------
add("v_exp_f32 a, v8", rep_acc("v_exp_f32 {a}, v8"))
add("v_rcp_f32 a, v8", rep_acc("v_rcp_f32 {a}, v8"))
add("v_exp_f32 / v_fma_f32 alternating", "\n".join(("v_exp_f32 %s, v8" if i % 2 == 0 else "v_fma_f32 %s, v8, v9, %s") % ((V(ACC[i % 16]),) * (1 if i % 2 == 0 else 2)) for i in range(32)))
add("1 v_rcp_f32 per 7 v_fma_f32", "\n".join(("v_rcp_f32 %s, v8" % V(ACC[i % 16])) if i % 8 == 0 else ("v_fma_f32 {a}, v8, v9, {a}".format(a=V(ACC[i % 16]))) for i in range(32)))
add("v_cndmask_b32 a, v8, a, vcc             (VOP2)", rep_acc("v_cndmask_b32 {a}, v8, {a}, vcc"))
add("v_cndmask_b32_e64 a, v8, a, s[88:89]    (VOP3, SGPR-pair mask)", rep_acc("v_cndmask_b32_e64 {a}, v8, {a}, s[88:89]"))
add("v_cmp_le_f32 vcc, v8, a", rep_acc("v_cmp_le_f32 vcc, v8, {a}"))
add("v_cmp_le_f32_e64 s[64:65], v8, a", rep_acc("v_cmp_le_f32_e64 s[64:65], v8, {a}"))
add("v_cmp_le_f32 vcc + v_cndmask vcc (dependent pair)", "\n".join("v_cmp_le_f32 vcc, v8, {a}\nv_cndmask_b32 {a}, v9, {a}, vcc".format(a=V(ACC[i % 16])) for i in range(16)))
add("v_cmp vcc ; s_and_b64 s[64:65], s[64:65], vcc  (VALU + SALU pair, 16 VALU)", "\n".join("v_cmp_le_f32 vcc, v8, {a}\ns_and_b64 s[64:65], s[64:65], vcc".format(a=V(ACC[i % 16])) for i in range(16)), n=16)
add("v_fma_f32 ; s_add_u32 alternating (32 VALU + 32 SALU)", "\n".join("v_fma_f32 {a}, v8, v9, {a}\ns_add_u32 s66, s66, 1".format(a=V(ACC[i % 16])) for i in range(32)))
add("s_add_u32 only (32 SALU, counted as 32)", "\n".join("s_add_u32 s%d, s%d, 1" % (66 + i % 4, 66 + i % 4) for i in range(32)))
add("v_readlane_b32 s66, a, 3", rep_acc("v_readlane_b32 s66, {a}, 3"))
add("v_readfirstlane_b32 s66, a", rep_acc("v_readfirstlane_b32 s66, {a}"))
add("v_permlane32_swap_b32 a, a'", "\n".join("v_permlane32_swap_b32 %s, %s" % (V(ACC[(2 * i) % 16]), V(ACC[(2 * i + 1) % 16])) for i in range(32)))
add("v_add_f32_dpp a, a, a row_mirror", rep_acc("v_add_f32_dpp {a}, {a}, {a} row_mirror row_mask:0xf bank_mask:0xf"))
add("v_add_f32_dpp a, v8, a quad_perm (source not the accumulator)", rep_acc("v_add_f32_dpp {a}, v8, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
add("v_pk_fma_f32 (2 flops/lane/instr), 8 independent pairs", "\n".join("v_pk_fma_f32 v[%d:%d], v[8:9], v[10:11], v[%d:%d]" % ((16 + 2 * (i % 8), 17 + 2 * (i % 8)) * 2) for i in range(32)))
add("v_pk_mul_f32, 8 independent pairs", "\n".join("v_pk_mul_f32 v[%d:%d], v[8:9], v[%d:%d]" % ((16 + 2 * (i % 8), 17 + 2 * (i % 8)) * 2) for i in range(32)))
# LDS broadcast reads next to VALU work: 3 ds_read_b128 (wave-uniform address) per 30 VALU, the blend kernels' ratio
add("3 ds_read_b128 (uniform address) + 29 v_fma_f32  [32 counted]",
    "v_mov_b32 v48, 0\nds_read_b128 v[32:35], v48\nds_read_b128 v[36:39], v48 offset:4096\nds_read_b128 v[40:43], v48 offset:8192\n" +
    rep_acc("v_fma_f32 {a}, v8, v9, {a}", 28) + "\ns_waitcnt lgkmcnt(0)")
# scalar loads of a 48-byte record next to VALU work (prefetched: waited for at the end of the body)
add("s_load_dwordx8 + s_load_dwordx4 + 32 v_fma_f32  [32 counted]",
    "s_load_dwordx8 s[76:83], s[58:59], 0x0\ns_load_dwordx4 s[84:87], s[58:59], 0x20\n" + rep_acc("v_fma_f32 {a}, v8, v9, {a}", 32) + "\ns_waitcnt lgkmcnt(0)")

# ---- group E: which operand pairs collide in the VGPR banks?  (accumulators restricted to ONE residue class mod 4 where noted)
def rep_regs(fmt, regs, n=32):
    return "\n".join(fmt.format(a=V(regs[i % len(regs)])) for i in range(n))
B0 = [16, 20, 24, 28]; B1 = [17, 21, 25, 29]; B2 = [18, 22, 26, 30]; B3 = [19, 23, 27, 31]
for name, s0, s1 in (("v8, v10 (banks 0,2)", "v8", "v10"), ("v8, v11 (banks 0,3)", "v8", "v11"), ("v8, v40 (0 and 0, 32 apart)", "v8", "v40"),
                     ("v8, v8 (same register twice)", "v8", "v8")):
    add("E v_fma_f32 a, %s, a" % name, rep_acc("v_fma_f32 {a}, %s, %s, {a}" % (s0, s1)))
add("E v_fma_f32 a, v9, v10, a   a in bank 0 only (src2/dst bank differs from both sources)", rep_regs("v_fma_f32 {a}, v9, v10, {a}", B0))
add("E v_fma_f32 a, v8, v9, a    a in bank 0 only (src2 = dst in src0's bank)", rep_regs("v_fma_f32 {a}, v8, v9, {a}", B0))
add("E v_fma_f32 a, v9, v8, a    a in bank 0 only (src2 = dst in src1's bank)", rep_regs("v_fma_f32 {a}, v9, v8, {a}", B0))
add("E v_fma_f32 a, v9, v10, v12 a in bank 0 only (dst in src2's bank, src2 != dst)", rep_regs("v_fma_f32 {a}, v9, v10, v12", B0))
add("E v_fma_f32 a, v9, v10, v12 a in bank 3 only (src2 bank 0, others 1,2, dst 3)", rep_regs("v_fma_f32 {a}, v9, v10, v12", B3))
add("E v_fma_f32 a, v9, v10, v13 a in bank 3 only (src2 in src0's bank)", rep_regs("v_fma_f32 {a}, v9, v10, v13", B3))
add("E v_mul_f32 a, v8, v12      (VOP2, both sources bank 0)", rep_acc("v_mul_f32 {a}, v8, v12"))
add("E v_mul_f32 a, v9, v10      a in bank 1 only (dst in src0's bank)", rep_regs("v_mul_f32 {a}, v9, v10", B1))
add("E v_fmac_f32 a, v9, v10     a in bank 1 only", rep_regs("v_fmac_f32 {a}, v9, v10", B1))
add("E v_fmac_f32 a, v9, v10     a in bank 0 only", rep_regs("v_fmac_f32 {a}, v9, v10", B0))
# ---- group F: min / max / med3, and the IEEE mode bit
add("F v_max_f32 a, v8, a", rep_acc("v_max_f32 {a}, v8, {a}"))
add("F v_min_f32 a, v9, v10      (accumulator only written)", rep_acc("v_min_f32 {a}, v9, v10"))
add("F v_med3_f32 a, v9, v10, a", rep_acc("v_med3_f32 {a}, v9, v10, {a}"))
add("F v_min_f32 a, v8, a        with MODE.IEEE = 0", "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0\n" + rep_acc("v_min_f32 {a}, v8, {a}"))
add("F v_cmp_le_f32 vcc, v8, a   with MODE.IEEE = 0", "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0\n" + rep_acc("v_cmp_le_f32 vcc, v8, {a}"))
add("F v_sub_f32 a, v8, a", rep_acc("v_sub_f32 {a}, v8, {a}"))
add("F v_cndmask_b32 a, v8, a, vcc   right after ONE v_cmp vcc per 8", "\n".join(("v_cmp_le_f32 vcc, v8, v9\n" if i % 8 == 0 else "") + "v_cndmask_b32 {a}, v8, {a}, vcc".format(a=V(ACC[i % 16])) for i in range(32)), n=36)
add("F v_mov_b32 a, v8 under a half-full exec mask", "s_mov_b32 exec_lo, 0x55555555\ns_mov_b32 exec_hi, 0x55555555\n" + rep_acc("v_mov_b32 {a}, v8") + "\ns_mov_b64 exec, -1")
add("F v_exp_f32 then 7 independent v_fma_f32 (4 groups)", "\n".join(("v_exp_f32 %s, v8" % V(ACC[i % 16])) if i % 8 == 0 else ("v_fma_f32 {a}, v9, v10, {a}".format(a=V(ACC[i % 16]))) for i in range(32)))
add("F v_exp_f32 ; v_rcp_f32 back to back + 6 v_fma_f32 (4 groups)", "\n".join(("v_exp_f32 %s, v8" % V(ACC[i % 16])) if i % 8 == 0 else ("v_rcp_f32 %s, v8" % V(ACC[i % 16])) if i % 8 == 1 else ("v_fma_f32 {a}, v9, v10, {a}".format(a=V(ACC[i % 16]))) for i in range(32)))
add("F s_and_b64 + s_cbranch (not taken) per 4 v_fma_f32  [32 VALU counted]", "\n".join("v_fma_f32 {a}, v9, v10, {a}".format(a=V(ACC[i % 16])) + ("\ns_and_b64 s[66:67], s[64:65], exec\ns_cbranch_scc0 2f" if i % 4 == 3 else "") for i in range(32)) + "\n2:")

# ---- group G: when is a select through VCC slow?
def pat(seq, reps, n):
    out = []
    for i in range(reps):
        for j, t in enumerate(seq):
            out.append(t.format(a=V(ACC[(i * len(seq) + j) % 16]), b=V(ACC[(i * len(seq) + j + 5) % 16])))
    return "\n".join(out), n
b, n = pat(["v_cmp_le_f32 vcc, v8, {a}", "v_cndmask_b32 {a}, v9, {a}, vcc", "v_cndmask_b32 {b}, v9, {b}, vcc"], 10, 30); add("G v_cmp vcc ; 2 x v_cndmask vcc", b, n)
b, n = pat(["v_cmp_le_f32 vcc, v8, {a}", "v_fma_f32 {b}, v9, v10, {b}", "v_fma_f32 {a}, v9, v10, {a}", "v_cndmask_b32 {b}, v9, {b}, vcc"], 8, 32); add("G v_cmp vcc ; 2 v_fma ; v_cndmask vcc", b, n)
b, n = pat(["v_cmp_le_f32_e64 s[64:65], v8, {a}", "v_cndmask_b32_e64 {a}, v9, {a}, s[64:65]"], 16, 32); add("G v_cmp_e64 s[64:65] ; v_cndmask_e64 s[64:65]", b, n)
b, n = pat(["v_cmp_le_f32_e64 s[64:65], v8, {a}", "v_fma_f32 {b}, v9, v10, {b}", "v_fma_f32 {a}, v9, v10, {a}", "v_cndmask_b32_e64 {b}, v9, {b}, s[64:65]"], 8, 32); add("G v_cmp_e64 s[64:65] ; 2 v_fma ; v_cndmask_e64 s[64:65]", b, n)
b, n = pat(["s_mov_b64 vcc, s[88:89]"] + ["v_cndmask_b32 {a}, v9, {a}, vcc"] * 7, 4, 28); add("G s_mov_b64 vcc (SALU write) ; 7 x v_cndmask vcc  [28 VALU]", b, n)
add("G v_cndmask_b32_e64 a, v9, a, vcc  (VOP3 encoding, mask is vcc, vcc set once)", rep_acc("v_cndmask_b32_e64 {a}, v9, {a}, vcc"))
b, n = pat(["v_cmp_le_f32 vcc, v8, {a}", "s_nop 7", "v_cndmask_b32 {a}, v9, {a}, vcc"], 16, 32); add("G v_cmp vcc ; s_nop 7 ; v_cndmask vcc", b, n)
b, n = pat(["v_cmp_le_f32_e64 s[64:65], v8, {a}", "s_and_b64 s[66:67], s[64:65], s[88:89]", "v_cndmask_b32_e64 {a}, v9, {a}, s[66:67]"], 16, 32); add("G v_cmp_e64 ; s_and_b64 ; v_cndmask_e64 on the SALU result", b, n)
b, n = pat(["v_cmp_le_f32_e64 s[64:65], v8, {a}", "s_and_b64 exec, exec, s[64:65]", "v_mov_b32 {a}, v9", "s_mov_b64 exec, -1"], 8, 16); add("G v_cmp_e64 ; s_and exec ; v_mov under exec ; restore  [16 VALU]", b, n)

# ---- group D: the alpha evaluation of one (pixel, entry) pair as straight-line code -------------------------------------------
# register map: v8 = pixel x (or u), v9 = pixel y (or v), v10/v11 = sqrt2-scaled u, v; entry operands q0..q2 in v32..v43 or s60..s71
# layout q: 32 Tux 33 Tuy 34 Tuz 35 Tvx 36 Tvy 37 Tvz 38 Twx 39 Twy 40 Twz 41 cx 42 cy 43 opacity
D1 = """
v_fma_f32 v48, v8, v39, -v33
v_fma_f32 v49, v9, v38, -v35
v_fma_f32 v50, v8, v38, -v32
v_fma_f32 v51, v8, v40, -v34
v_fma_f32 v52, v9, v39, -v36
v_mul_f32 v53, v48, v49
v_mul_f32 v54, v52, v51
v_fma_f32 v55, v50, v52, -v53
v_rcp_f32 v56, v55
v_fma_f32 v57, v9, v40, -v37
v_mul_f32 v58, v50, v57
v_fma_f32 v58, v49, v51, -v58
v_sub_f32 v59, v41, v8
v_fma_f32 v48, v48, v57, -v54
v_mul_f32 v58, v56, v58
v_sub_f32 v60, v42, v9
v_mul_f32 v59, v59, v59
v_mul_f32 v48, v56, v48
v_mul_f32 v61, v58, v58
v_fmac_f32 v59, v60, v60
v_fmac_f32 v61, v48, v48
v_add_f32 v59, v59, v59
v_cmp_le_f32 vcc, v61, v59
v_mul_f32 v62, v39, v58
v_fmac_f32 v62, v38, v48
v_cndmask_b32 v61, v59, v61, vcc
v_mul_f32 v61, -0.5, v61
v_mul_f32 v59, 0x3fb8aa3b, v61
v_exp_f32 v59, v59
v_add_f32 v62, v40, v62
v_cndmask_b32 v62, v40, v62, vcc
v_cmp_neq_f32 vcc, 0, v55
v_mul_f32 v63, v43, v59
v_cmp_le_f32_e64 s[64:65], s72, v62
v_min_f32 v63, 0x3f7d70a4, v63
s_and_b64 s[64:65], vcc, s[64:65]
v_cmp_nlt_f32 vcc, 0, v61
s_and_b64 s[64:65], s[64:65], vcc
v_cmp_le_f32 vcc, s73, v63
v_and_b32 v49, 1, v12
s_and_b64 s[64:65], s[64:65], vcc
v_cmp_eq_u32 vcc, 1, v49
s_xor_b64 s[66:67], vcc, -1
s_and_b64 s[66:67], s[64:65], s[66:67]
v_cndmask_b32_e64 v49, 0, 1, s[66:67]
v_cmp_ne_u32 vcc, 0, v49
v_add_f32 v16, v63, v16
v_add_f32 v17, v62, v17
"""
add("D1 alpha evaluation as compiled today (45 VALU + 2 sinks, entry operands in VGPRs)", D1.strip(), n=47)
# affine form: operands A (v32..34) B (35..37) C (38..40) | Tw (41..43) | cxs cys opacity (44..46); or the same in s60..s74
D2 = """
v_fma_f32 v48, v8, v35, v32
v_fma_f32 v49, v8, v36, v33
v_fma_f32 v50, v8, v37, v34
v_fmac_f32 v48, v9, v38
v_fmac_f32 v49, v9, v39
v_fmac_f32 v50, v9, v40
v_rcp_f32 v51, v50
v_sub_f32 v52, v44, v10
v_sub_f32 v53, v45, v11
v_mul_f32 v48, v48, v51
v_mul_f32 v49, v49, v51
v_mul_f32 v52, v52, v52
v_mul_f32 v54, v48, v48
v_fmac_f32 v52, v53, v53
v_fmac_f32 v54, v49, v49
v_fma_f32 v55, v48, v41, v43
v_cmp_le_f32 vcc, v54, v52
v_min_f32 v54, v54, v52
v_fmac_f32 v55, v49, v42
v_mul_f32 v54, 0xbf38aa3b, v54
v_cndmask_b32 v55, v43, v55, vcc
v_exp_f32 v54, v54
v_cmp_neq_f32 vcc, 0, v50
v_cmp_le_f32_e64 s[64:65], s72, v55
v_mul_f32 v54, v46, v54
s_and_b64 s[64:65], vcc, s[64:65]
v_cmp_le_f32 vcc, s73, v54
v_min_f32 v56, 0x3f7d70a4, v54
s_and_b64 s[64:65], s[64:65], vcc
v_add_f32 v16, v56, v16
v_add_f32 v17, v55, v17
"""
add("D2 affine alpha evaluation, trimmed (26 VALU + 2 sinks, entry operands in VGPRs)", D2.strip(), n=28)
D3 = D2
for vg, sg in ((35, 60), (36, 61), (37, 62), (38, 63), (39, 64 + 4), (40, 69), (41, 70), (42, 71), (44, 74), (45, 75), (46, 76)):
    D3 = D3.replace(", v%d," % vg, ", s%d," % sg).replace(", v%d\n" % vg, ", s%d\n" % sg).replace(" v%d, v10" % vg, " s%d, v10" % sg).replace(" v%d, v11" % vg, " s%d, v11" % sg)
# a VOP2/VOP3 instruction may read one SGPR: A (v32..34) and Twz (v43) stay in VGPRs, the fmacs / muls take B, C, Tw.xy, centre, opacity as SGPRs
D3 = D3.replace("v_fmac_f32 v48, v9, s63", "v_fmac_f32 v48, s63, v9").replace("v_fmac_f32 v49, v9, s68", "v_fmac_f32 v49, s68, v9").replace("v_fmac_f32 v50, v9, s69", "v_fmac_f32 v50, s69, v9")
D3 = D3.replace("v_fmac_f32 v55, v49, s71", "v_fmac_f32 v55, s71, v49")
add("D3 affine alpha evaluation, B, C, Tw.xy, centre, opacity as SGPR operands (A, Tw.z in VGPRs)", D3.strip(), n=28)

HEADER = r"""// GENERATED by tools/micro/gen_valu_issue_bench.py -- do not edit.  See that file for what is measured and why.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define CLOBBERS "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
    "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
    "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","vcc","scc","memory"

// prologue: plausible operand values (no denormals, no NaN), stamps; epilogue: stamps, deltas -> out
#define PROLOGUE \
    "v_cvt_f32_u32 v8, %4\n v_mul_f32 v8, 0x3e000000, v8\n v_add_f32 v8, 400.5, v8\n v_add_f32 v9, 1.0, v8\n v_mul_f32 v10, 0x3fb504f3, v8\n v_mul_f32 v11, 0x3fb504f3, v9\n" \
    "v_mov_b32 v12, 0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 0.5\n v_mov_b32 v15, 2.0\n" \
    "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n" \
    "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" \
    "v_mov_b32 v32, 0x44480000\n v_mov_b32 v33, 0x41200000\n v_mov_b32 v34, 0x44fa0000\n v_mov_b32 v35, 0x41a00000\n v_mov_b32 v36, 0x44480000\n v_mov_b32 v37, 0x44fa0000\n" \
    "v_mov_b32 v38, 0x3c23d70a\n v_mov_b32 v39, 0x3ca3d70a\n v_mov_b32 v40, 4.0\n v_mov_b32 v41, 0x43c80000\n v_mov_b32 v42, 0x43c90000\n v_mov_b32 v43, 0.5\n" \
    "v_mov_b32 v44, 0x44100000\n v_mov_b32 v45, 0x44110000\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 1.0\n" \
    "s_mov_b32 s60, 0x3f8ccccd\n s_mov_b32 s61, 0x3c23d70a\n s_mov_b32 s62, 0x3ca3d70a\n s_mov_b32 s63, 0x3c23d70a\n s_mov_b32 s68, 0x3ca3d70a\n s_mov_b32 s69, 0x3c23d70a\n" \
    "s_mov_b32 s70, 0x3c23d70a\n s_mov_b32 s71, 0x3ca3d70a\n s_mov_b32 s72, 0x3e4ccccd\n s_mov_b32 s73, 0x3b808081\n s_mov_b32 s74, 0x44100000\n s_mov_b32 s75, 0x44110000\n s_mov_b32 s76, 0.5\n" \
    "s_mov_b64 s[64:65], -1\n s_mov_b64 s[66:67], 0\n s_mov_b64 s[58:59], %5\n s_mov_b64 s[88:89], 0x5555\n v_cmp_lt_f32 vcc, v14, v13\n" \
    "s_mov_b32 s40, %6\n" \
    "s_getreg_b32 s50, hwreg(HW_REG_HW_ID)\n s_getreg_b32 s51, hwreg(HW_REG_XCC_ID)\n" \
    "s_barrier\n s_memtime s[42:43]\n s_memrealtime s[44:45]\n s_waitcnt lgkmcnt(0)\n" \
    "1:\n"
#define EPILOGUE \
    "\n s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n" \
    "s_memtime s[46:47]\n s_memrealtime s[48:49]\n s_waitcnt lgkmcnt(0)\n" \
    "s_sub_u32 s46, s46, s42\n s_sub_u32 s48, s48, s44\n" \
    "v_mov_b32 %0, s46\n v_mov_b32 %1, s48\n v_mov_b32 %2, s50\n v_mov_b32 %3, s51\n"

"""

KERNEL = r"""
__global__ void __launch_bounds__(256) k{idx}(unsigned* out, const float* mem, int iters)
{{
    __shared__ float4 s_buf[1024];   // 16 KB: the ds_read variants read offsets 0, 4096, 8192
    for (int i = threadIdx.x; i < 1024; i += 256) s_buf[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    unsigned dc, dr, hw, xcc;
    asm volatile(PROLOGUE
{body}
                 EPILOGUE
                 : "=v"(dc), "=v"(dr), "=v"(hw), "=v"(xcc) : "v"(threadIdx.x), "s"(mem), "s"(iters) : CLOBBERS);
    if ((threadIdx.x & 63) == 0) {{
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[4 * w] = dc; out[4 * w + 1] = dr; out[4 * w + 2] = hw; out[4 * w + 3] = xcc;
    }}
    if (s_buf[threadIdx.x].x == 123.f) out[0] = 0;
}}
"""

MAIN = r"""
struct Variant {{ const char* name; int n; void (*fn)(unsigned*, const float*, int); }};
static Variant kVariants[] = {{
{table}
}};

int main(int argc, char** argv)
{{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const int iters = 2000;
    const char* only = argc > 1 ? argv[1] : nullptr;
    printf("%s: %d CUs, nominal %d MHz.  cyc = SIMD cycles per counted instruction = (wave cycles / instructions) / (waves resident on that SIMD);\n"
           "waves/SIMD is what the launch asks for (blocks of 4 waves, N blocks per CU); [min..max] over SIMDs; wall = the same from the kernel's HIP-event duration; clk = s_memtime ticks per s_memrealtime tick x 100 MHz\n",
           prop.gcnArchName, n_cu, prop.clockRate / 1000);
    unsigned* out; float* mem;
    const int max_blocks = n_cu * 8;
    hipMalloc(&out, (size_t)max_blocks * 4 * 16);
    hipMalloc(&mem, 4096);
    hipMemset(mem, 0, 4096);
    std::vector<unsigned> h((size_t)max_blocks * 16);
    const int wps_list[] = {{1, 2, 4, 5, 6, 8}};
    for (const Variant& v : kVariants) {{
        if (only && !strstr(v.name, only)) continue;
        printf("%-96s\n   ", v.name);
        for (int wps : wps_list) {{
            const int grid = n_cu * wps;
            hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, out, mem, 16);   // warm-up (code in the instruction cache)
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, out, mem, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float wall_ms = 0.f;
            hipEventElapsedTime(&wall_ms, e0, e1);
            hipEventDestroy(e0); hipEventDestroy(e1);
            hipMemcpy(h.data(), out, (size_t)grid * 4 * 16, hipMemcpyDeviceToHost);
            std::map<unsigned, std::vector<unsigned>> by_simd;   // (xcc, se, sh, cu, simd) -> wave cycle counts
            double clk = 0;
            for (int w = 0; w < grid * 4; w++) {{
                const unsigned dc = h[4 * w], dr = h[4 * w + 1], hw = h[4 * w + 2], xcc = h[4 * w + 3];
                by_simd[(xcc << 20) | (hw & 0xfff0u)].push_back(dc);   // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
                clk += dr ? (double)dc / dr * 100.0 : 0.0;
            }}
            double lo = 1e30, hi = 0, sum = 0; size_t nw_max = 0;
            for (auto& kv : by_simd) {{
                double c = 0;
                for (unsigned d : kv.second) c = std::max(c, (double)d);
                const double per = c / ((double)iters * v.n) / (double)kv.second.size();
                lo = std::min(lo, per); hi = std::max(hi, per); sum += per;
                nw_max = std::max(nw_max, kv.second.size());
            }}
            // cross-check from the kernel's wall time (HIP events, includes ~10 us of launch): cycles per instruction per SIMD if every SIMD ran wps waves
            const double mhz = clk / (grid * 4);
            const double wall_cyc = (double)wall_ms * 1e-3 * mhz * 1e6 / ((double)iters * v.n * wps);
            printf(" %dw: %5.2f [%4.2f..%4.2f] wall %5.2f (<=%zu w) %4.0f MHz |", wps, sum / by_simd.size(), lo, hi, wall_cyc, nw_max, mhz);
        }}
        printf("\n");
    }}
    return 0;
}}
"""


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    parts = [HEADER]
    table = []
    for i, (name, n, body) in enumerate(VARIANTS):
        lines = "\n".join('                 "%s\\n"' % l.strip() for l in body.split("\n") if l.strip())
        parts.append(KERNEL.format(idx=i, body=lines))
        table.append('    {"%s", %d, k%d},' % (name, n, i))
    parts.append(MAIN.format(table="\n".join(table)))
    with open(os.path.join(here, "valu_issue_bench.hip"), "w") as f:
        f.write("".join(parts))
    print("wrote %d variants" % len(VARIANTS))


if __name__ == "__main__":
    main()
