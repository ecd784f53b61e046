"""Generates pi3_slam_amd/csrc/attn64a_loop.inc: the hand-placed main loop of the global attention (bounded-score path,
eight waves x 64 query rows, d = 64) as ONE inline-asm block (round 5).

Why: hipcc's loop runs QK^T (16 MFMAs, vector unit idle) -> exp / pack (vector bound, matrix pipe half idle) -> row sums
as phases that the two waves of a SIMD walk through together: 3 540 cycles per 64-key tile and SIMD against 2 048 of
matrix pipe (tools/dev_attn_ab.py ablations: no exp -12 %, no row sums -7 %, no barrier 0).  Here every wave's stream
is software-pipelined over QUARTER tiles (16 keys) and every instruction is placed:

  slot q = (tile t, 32-key half kt, 16-key step s2), eight MFMA gaps:
     M1..M4   O^T += V^T(q-1) . P^T(q-1)            (P packed in slot q-1; V fragments read in slot q-1)
     M5..M8   S^T(next half) += K . Q^T, d-steps 2 s2, 2 s2 + 1   (two K fragments read in this slot, both query blocks)
     fillers  16 v_exp_f32 + 8 v_cvt_pk_bf16_f32 on S^T(q) -> P(q);  4 row-sum MFMAs (4x4x4) on P(q-1);
              2 ds_read_b128 (K), 4 ds_read_b64_tr_b16 (V of quarter q), counted lgkmcnt waits
  per tile: 2 LDS-DMA pieces per wave (K two tiles ahead into a 3-slot ring, V one ahead into a 2-slot ring), address
  updates (6 VALU), one vmcnt(0) + barrier.
Per MFMA gap: 2 exp + 1 cvt + <= 1 more single-issue instruction (the guide's budget is 5 fillers, one of them an exp;
d = 64 needs two exps per gap, which is why this loop can only approach the matrix pipe, not reach it).

Registers: everything the loop touches besides its C++ operands is a FIXED vector register (clobber list): two score
half-tiles (ping-pong by kt), two packed-P quarters (ping-pong by s2), one K and one V fragment set, addresses.
The C++ side (attn64a.hip) passes O (4 x 16), Q fragments (8 x 4), row-sum accumulators (2 x 4) as operands.

    python tools/gen_attn_asm.py        # rewrites the .inc (committed; the Makefile does not run this)
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pi3_slam_amd", "csrc", "attn64a_loop.inc")

# ---- fixed vector registers (v150 .. v255) ------------------------------------------------------------------------
V0 = 150
SBUF = [[V0 + 0, V0 + 16], [V0 + 32, V0 + 48]]      # SBUF[kt][blk] -> first of 16 registers (scores, exp in place)
PBUF = [[V0 + 64, V0 + 68], [V0 + 72, V0 + 76]]     # PBUF[parity][blk] -> first of 4 registers (packed bf16 P)
KF = [V0 + 80, V0 + 84]                             # two K fragments (4 registers each)
VF = [V0 + 88, V0 + 92]                             # V^T fragments dt = 0, 1 (lo pair, hi pair)
KADDR = [V0 + 96 + i for i in range(4)]             # LDS byte address of this lane's K fragment, d-step s (current K slot)
VADDR = [V0 + 100, V0 + 101]                        # LDS byte address of this lane's V^T fragment, dt (current V slot)
KSRC, VSRC = V0 + 102, V0 + 103                     # per-lane global byte offsets of the DMA pieces
ONES = V0 + 104                                     # 2 registers: packed bf16 ones (A operand of the row-sum MFMA)
LAST_V = V0 + 105
assert LAST_V <= 255
# ---- fixed scalar registers ------------------------------------------------------------------------------------------
S_CNT, S_KG, S_VG, S_TB, S_KDST, S_VDST, S_KD0, S_KD1, S_KD2, S_VDELTA, S_KDELTA, S_TMP = (
    "s60", "s[62:63]", "s[64:65]", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74")
S_KG_LO, S_KG_HI, S_VG_LO, S_VG_HI = "s62", "s63", "s64", "s65"
S_KA0, S_KA1, S_KA2 = "s75", "s76", "s77"           # LDS byte offsets of the three K slots (relative; for the address delta)
S_KCUR = "s78"                                      # current K slot offset of the fragment addresses
SCALARS = list(range(60, 79))

lines = []
ABL = set(os.environ.get("A64A_ABL", "").split(","))       # timing-only ablations (WRONG results): noexp, norowsum, nopv, noqk, nobarrier
OPT = dict(kv.split("=") for kv in os.environ.get("A64A_OPT", "").split(",") if "=" in kv)      # schedule options (correct results)
OUT = os.environ.get("A64A_OUT", OUT)


def emit(s):
    lines.append(s)


def vr(first, n):
    return f"v[{first}:{first + n - 1}]" if n > 1 else f"v{first}"


def slot(kt, s2, *, pv=True, qk=True, ex=True, rowsum=True, kaddr_update=False, vread=True):
    """One quarter-tile slot.  kt, s2: the quarter whose scores are exponentiated here (and whose V fragments are read)."""
    par = s2                                        # P parity of this quarter; PV / row sums use the other one
    prev = PBUF[par ^ 1]
    cur = PBUF[par]
    S = SBUF[kt]
    tgt = SBUF[kt ^ 1]                              # QK^T target: the other half's score buffer
    tgt_off = 4096 if kt == 0 else 0                # (t, kt1) lives in the current K tile's second 32 rows; (t+1, kt0) at +0
    dsteps = (2 * s2, 2 * s2 + 1)
    # exp / cvt work list: blk A then B, 8 values each -> per gap 2 exps + 1 cvt, cvt two gaps behind its exps
    exps = [(blk, i) for blk in (0, 1) for i in range(8)] if ex else []
    fill = [[] for _ in range(8)]
    for g in range(8):
        for (blk, i) in exps[2 * g: 2 * g + 2]:
            r = S[blk] + 8 * s2 + i
            fill[g].append(f"v_mov_b32_e32 v{r}, v{r}" if "noexp" in ABL else f"v_exp_f32_e32 v{r}, v{r}")
    if ex:
        for j in range(8):                          # cvt j packs exps 2j, 2j+1 (issued in gap j); placed in gap j+1 (last: gap 7 tail)
            blk, i = exps[2 * j]
            r = S[blk] + 8 * s2 + i
            dst = cur[blk] + (i // 2)
            g = min(j + 1, 7)
            fill[g].append(f"v_cvt_pk_bf16_f32 v{dst}, v{r}, v{r + 1}")
    if rowsum and pv and "norowsum" not in ABL:     # 4 row-sum MFMAs on P(q-1): gaps 1, 3, 5, 7
        for n, (blk, half) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            acc = "%[lA]" if blk == 0 else "%[lB]"
            fill[2 * n + 1].append(f"v_mfma_f32_4x4x4_16b_bf16 {acc}, {vr(ONES, 2)}, {vr(prev[blk] + 2 * half, 2)}, {acc}")
    # LDS reads: K fragments for this slot's QK^T right after M1, V fragments of quarter q after M4
    if qk:
        if kaddr_update:
            for s in range(4):
                fill[0].insert(0, f"v_add_u32_e32 v{KADDR[s]}, {S_KDELTA}, v{KADDR[s]}")
        fill[0].append(f"ds_read_b128 {vr(KF[0], 4)}, v{KADDR[dsteps[0]]} offset:{tgt_off}")
        fill[0].append(f"ds_read_b128 {vr(KF[1], 4)}, v{KADDR[dsteps[1]]} offset:{tgt_off}")
    vbase = (32 * kt + 16 * s2) * 128
    vreads = [f"ds_read_b64_tr_b16 {vr(VF[0], 2)}, v{VADDR[0]} offset:{vbase}",
              f"ds_read_b64_tr_b16 {vr(VF[0] + 2, 2)}, v{VADDR[0]} offset:{vbase + 1024}",
              f"ds_read_b64_tr_b16 {vr(VF[1], 2)}, v{VADDR[1]} offset:{vbase}",
              f"ds_read_b64_tr_b16 {vr(VF[1] + 2, 2)}, v{VADDR[1]} offset:{vbase + 1024}"]
    if vread:
        fill[3] += vreads[:2]
        fill[4] += vreads[2:]
    # the eight MFMAs
    O = [["%[oa0]", "%[oa1]"], ["%[ob0]", "%[ob1]"]]
    Q = [["%[qa0]", "%[qa1]", "%[qa2]", "%[qa3]"], ["%[qb0]", "%[qb1]", "%[qb2]", "%[qb3]"]]
    mf = []
    for dt in (0, 1):
        for blk in (0, 1):
            mf.append(f"v_mfma_f32_32x32x16_bf16 {O[blk][dt]}, {vr(VF[dt], 4)}, {vr(prev[blk], 4)}, {O[blk][dt]}" if pv else None)
    for n, s in enumerate(dsteps):
        for blk in (0, 1):
            acc = vr(tgt[blk], 16)
            c = "0" if s == 0 else acc
            mf.append(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(KF[n], 4)}, {Q[blk][s]}, {c}" if qk else None)
    if "nopv" in ABL:
        mf[:4] = [None] * 4
    if "noqk" in ABL and ex:
        mf[4:] = [None] * 4
    emit(f"; ---- slot kt={kt} s2={s2} pv={int(pv)} qk={int(qk)} ex={int(ex)}")
    # dependent row-sum MFMAs must not follow each other directly (the second would read its accumulator before the first
    # has written it): where a slot carries no big MFMA between them (the drain), pad
    if not qk:
        for g in range(len(fill)):
            fill[g] = [y for x in fill[g] for y in ((["s_nop 7"] if x.startswith("v_mfma_f32_4x4x4") else []) + [x])]
    for g in range(8):
        if g == 0 and pv:
            emit("s_waitcnt lgkmcnt(0)")             # V fragments of quarter q-1 (read half a slot ago)
        if g == 4 and qk:
            emit("s_waitcnt lgkmcnt(2)" if vread else "s_waitcnt lgkmcnt(0)")   # both K fragments (the two V reads of gap 3 may be in flight)
        if mf[g] is not None:
            emit(mf[g])
        for f in fill[g]:
            emit(f)


def tile_head():
    emit("; ---- tile head: LDS-DMA of K(t+2) and V(t+1), one piece each per wave; scalar bookkeeping")
    emit(f"s_mov_b32 m0, {S_KDST}")
    emit("s_nop 0")
    emit(f"global_load_lds_dwordx4 v{KSRC}, {S_KG}")
    emit(f"s_mov_b32 m0, {S_VDST}")
    emit("s_nop 0")
    emit(f"global_load_lds_dwordx4 v{VSRC}, {S_VG}")
    emit(f"s_add_u32 {S_KG_LO}, {S_KG_LO}, {S_TB}")
    emit(f"s_addc_u32 {S_KG_HI}, {S_KG_HI}, 0")
    emit(f"s_add_u32 {S_VG_LO}, {S_VG_LO}, {S_TB}")
    emit(f"s_addc_u32 {S_VG_HI}, {S_VG_HI}, 0")
    # rotate the K destination slot 0 -> 1 -> 2 -> 0 and toggle the V destination
    emit(f"s_cmp_eq_u32 {S_KDST}, {S_KD0}")
    emit(f"s_cselect_b32 {S_TMP}, {S_KD1}, {S_KD0}")        # from slot 0 -> 1 ; (from slot 2 -> 0 handled next)
    emit(f"s_cmp_eq_u32 {S_KDST}, {S_KD1}")
    emit(f"s_cselect_b32 {S_KDST}, {S_KD2}, {S_TMP}")       # from slot 1 -> 2 ; else what the first select chose
    emit(f"s_add_u32 {S_VDST}, {S_VDST}, {S_VDELTA}")
    # V fragment addresses follow the tile's V slot (the slot being READ toggles like the one being written)
    emit(f"v_add_u32_e32 v{VADDR[0]}, {S_VDELTA}, v{VADDR[0]}")
    emit(f"v_add_u32_e32 v{VADDR[1]}, {S_VDELTA}, v{VADDR[1]}")
    emit(f"s_sub_u32 {S_VDELTA}, 0, {S_VDELTA}")
    # K fragment address delta for the mid-tile switch to the next tile's slot: next(cur) - cur
    emit(f"s_cmp_eq_u32 {S_KCUR}, {S_KA0}")
    emit(f"s_cselect_b32 {S_TMP}, {S_KA1}, {S_KA0}")
    emit(f"s_cmp_eq_u32 {S_KCUR}, {S_KA1}")
    emit(f"s_cselect_b32 {S_TMP}, {S_KA2}, {S_TMP}")        # S_TMP = next slot offset
    emit(f"s_sub_u32 {S_KDELTA}, {S_TMP}, {S_KCUR}")
    emit(f"s_mov_b32 {S_KCUR}, {S_TMP}")


def tile_tail():
    emit("; ---- tile tail: every DMA piece landed, every fragment read returned, then the workgroup barrier")
    if "nobarrier" not in ABL:
        emit("s_waitcnt vmcnt(0)")
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")


def entry():
    emit("; ==== entry: C++ operands into the fixed registers")
    for i in range(4):
        emit(f"v_mov_b32_e32 v{KADDR[i]}, %[ka{i}]")
    emit(f"v_mov_b32_e32 v{VADDR[0]}, %[va0]")
    emit(f"v_mov_b32_e32 v{VADDR[1]}, %[va1]")
    emit(f"v_mov_b32_e32 v{KSRC}, %[ksrc]")
    emit(f"v_mov_b32_e32 v{VSRC}, %[vsrc]")
    emit(f"v_mov_b32_e32 v{ONES}, 0x3f803f80")
    emit(f"v_mov_b32_e32 v{ONES + 1}, 0x3f803f80")
    emit(f"s_mov_b32 {S_CNT}, %[cnt]")
    emit(f"s_mov_b64 {S_KG}, %[kg]")
    emit(f"s_mov_b64 {S_VG}, %[vg]")
    emit(f"s_mov_b32 {S_TB}, %[tb]")
    # entry is tile t0 = 1: K(3) goes to slot 0, V(2) to V slot 0; fragment addresses: K slot 1, V slot 0 (stepped at the head)
    emit(f"s_mov_b32 {S_KD0}, %[kd0]")                     # LDS address of this wave's 1 KiB piece in K slot 0
    emit(f"s_add_u32 {S_KD1}, {S_KD0}, 8192")
    emit(f"s_add_u32 {S_KD2}, {S_KD0}, 32768")
    emit(f"s_mov_b32 {S_KDST}, {S_KD0}")
    emit(f"s_add_u32 {S_VDST}, {S_KD0}, 16384")
    emit(f"s_mov_b32 {S_VDELTA}, 8192")
    emit(f"s_mov_b32 {S_KA0}, 0")
    emit(f"s_mov_b32 {S_KA1}, 8192")
    emit(f"s_mov_b32 {S_KA2}, 32768")
    emit(f"s_mov_b32 {S_KCUR}, 8192")


def build():
    entry()
    emit("; ==== ramp: scores of (t0, kt0) into SBUF[0]; P(parity 1) = 0 and V fragments of any landed tile for the first, empty P.V")
    for r in range(4):
        emit(f"v_mov_b32_e32 v{PBUF[1][0] + r}, 0")
        emit(f"v_mov_b32_e32 v{PBUF[1][1] + r}, 0")
    # two QK-only pseudo slots filling SBUF[0] from the current K slot at +0 (kt argument 1 -> target SBUF[0], offset 0)
    slot(1, 0, pv=False, qk=True, ex=False, rowsum=False)
    emit("s_waitcnt lgkmcnt(0)")
    slot(1, 1, pv=False, qk=True, ex=False, rowsum=False)
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 7")
    emit("s_nop 7")                                  # the last QK^T MFMA's result is read by the first exp below (16 passes)
    emit("LOOP_%=:")
    tile_head()
    slot(0, 0)
    slot(0, 1)
    slot(1, 0, kaddr_update=True)
    slot(1, 1)
    tile_tail()
    emit(f"s_sub_u32 {S_CNT}, {S_CNT}, 1")
    emit(f"s_cmp_lg_u32 {S_CNT}, 0")
    emit("s_cbranch_scc1 LOOP_%=")
    emit("; ==== drain: P.V and row sums of the last quarter (its V fragments were read in the last slot, before the barrier)")
    slot(0, 0, pv=True, qk=False, ex=False, rowsum=True, vread=False)
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 7")


build()
clob_v = ", ".join(f'"v{i}"' for i in range(V0, LAST_V + 1))
clob_s = ", ".join(f'"s{i}"' for i in SCALARS)
with open(OUT, "w") as f:
    f.write("// GENERATED by tools/gen_attn_asm.py - do not edit.  The hand-placed main loop of attn_fwd64a_kernel (attn64a.hip).\n")
    f.write("#define A64A_LOOP_ASM \\\n")
    for ln in lines:
        f.write('  "' + ln.replace('"', '\\"') + '\\n\\t" \\\n')
    f.write("  \"\"\n")
    f.write(f"#define A64A_CLOBBER_V {clob_v}\n")
    f.write(f"#define A64A_CLOBBER_S {clob_s}\n")
    names = dict(V0=V0, SBUF0=SBUF[0][0], PBUF0=PBUF[0][0], KADDR0=KADDR[0], VADDR0=VADDR[0], KSRC=KSRC, VSRC=VSRC, ONES=ONES)
    for k, v in names.items():
        f.write(f"#define A64A_{k} {v}\n")
print("wrote", OUT, len(lines), "instructions + directives")
