"""Generator of the hand-placed attention main loop, generalised (round 5 experiment): NBLK query blocks of 32 rows per
wave (2: eight waves per workgroup, two per SIMD - the shape of tools/gen_attn_asm.py; 4: FOUR waves per workgroup, ONE per
SIMD with the whole 512-register file, the structure the guide's d = 128 kernel uses), O / Q / row-sum accumulators in
FIXED accumulator registers (no operand limit), Q loaded inside the block.

    python tools/gen_attn_asm2.py NBLK OUT.inc [bench]
    python tools/gen_attn_asm2.py 4 pi3_slam_amd/csrc/attn64b_loop.inc product      # the product loop of attn_fwd64b_kernel

`product` (NBLK = 4): O and the row-sum accumulators are "+a" operands (%[o<blk><dt>], %[l<blk>]) carried in from tile 0 and out
to the last tiles of the C++ side; Q is loaded inside the block into fixed accumulator registers at the top of the file;
the K / V fragment addresses of d-steps 1..3 / dt 1 are derived from %[ka0] / %[va0] (the swizzles are XORs below the
128-byte row: LDS base 128-aligned); the loop runs tiles 1 .. nt-4 (count operand) like tools/gen_attn_asm.py's.

`bench`: a timing harness variant (tools/micro/attn_loop_bench.hip): runs the loop on whatever the buffers hold.
Slot = (tile t, 32-key half kt, 16-key step s2) with G = 4 NBLK MFMA gaps:
   gaps 0 .. 2N-1   O^T[blk][dt] += V^T(q-1)[dt] . P^T(q-1)[blk]      (dt-major)
   gaps 2N .. 4N-1  S^T(next half)[blk] += K[s] . Q^T[blk][s]          (d-step-major: every K fragment feeds N MFMAs)
   fillers per gap: 2 v_exp_f32 + 1 v_cvt_pk_bf16_f32 of quarter q; every second gap a 4x4x4 row-sum MFMA on P(q-1);
   LDS: both K fragments in gap 0, the V^T fragments of quarter q right behind the last MFMA that reads the old ones.
"""
import os
import sys

NBLK = int(sys.argv[1])
OUT = sys.argv[2]
N = NBLK
PRODUCT = len(sys.argv) > 3 and sys.argv[3] == "product"      # the loop of attn_fwd64b_kernel (attn64.hip): O / row sums are operands
ABL = set(os.environ.get("A64A_ABL", "").split(","))
NW = 8 if N == 2 else 4                     # waves per workgroup
PIECES = 16 // NW                           # 1 KiB LDS-DMA pieces per wave and tile (K and V each: 8 KiB)

# ---- fixed arch VGPRs -------------------------------------------------------------------------------------------------
VSUM = "rowsum_vadd" in ABL or "rowsum_pkadd" in ABL or "rowsum_dot2" in ABL     # row sums on the VALU: two f32 accumulators per block
nfix = 32 * N + 8 * N + 8 + 8 + 4 + 2 + 2 * (PIECES // 2) + 2 + (2 * N if VSUM else 0)
nacc = 32 * N + 16 * N + 4 * N
ARCH = 512 // (2 if N == 2 else 1) - nacc            # arch VGPRs available beside the accumulator registers
ARCH = min(ARCH, 256)
V0 = (ARCH - nfix - 2) & ~1          # even: v_pk_add_f32 wants aligned register pairs
cur = [V0]


def alloc(n):
    r = cur[0]
    cur[0] += n
    return r


SBUF = [[alloc(16) for _ in range(N)] for _ in range(2)]
PBUF = [[alloc(4) for _ in range(N)] for _ in range(2)]
KF = [alloc(4), alloc(4)]
VF = [alloc(4), alloc(4)]
KADDR = [alloc(1) for _ in range(4)]
VADDR = [alloc(1), alloc(1)]
KSRC = [alloc(1) for _ in range(PIECES // 2)]
VSRC = [alloc(1) for _ in range(PIECES // 2)]
ONES = alloc(2)
LSUM = [alloc(2) for _ in range(N)] if VSUM else None
LASTV = cur[0] - 1
assert LASTV < ARCH and V0 >= 24, (V0, LASTV, ARCH)
# ---- fixed accumulator registers ------------------------------------------------------------------------------------------
O = [[(blk * 2 + dt) * 16 for dt in range(2)] for blk in range(N)]
Q = [[32 * N + (blk * 4 + s) * 4 for s in range(4)] for blk in range(N)]
LACC = [32 * N + 16 * N + 4 * blk for blk in range(N)]
LASTA = 32 * N + 16 * N + 4 * N - 1
FIRSTA = 0
if PRODUCT:
    FIRSTA = 256 - 16 * N
    Q = [[FIRSTA + (blk * 4 + s) * 4 for s in range(4)] for blk in range(N)]
    LASTA = 255


def o_reg(blk, dt):
    return f"%[o{blk}{dt}]" if PRODUCT else ar(O[blk][dt], 16)


def l_reg(blk):
    return f"%[l{blk}]" if PRODUCT else ar(LACC[blk], 4)

S_CNT, S_KG, S_VG, S_TB, S_KDST, S_VDST, S_KD0, S_KD1, S_KD2, S_VDELTA, S_KDELTA, S_TMP = (
    "s60", "s[62:63]", "s[64:65]", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74")
S_KG_LO, S_KG_HI, S_VG_LO, S_VG_HI = "s62", "s63", "s64", "s65"
S_KA0, S_KA1, S_KA2, S_KCUR = "s75", "s76", "s77", "s78"
SCALARS = list(range(60, 80))
lines = []
emit = lines.append


def vr(first, n):
    return f"v[{first}:{first + n - 1}]" if n > 1 else f"v{first}"


def ar(first, n):
    return f"a[{first}:{first + n - 1}]" if n > 1 else f"a{first}"


def slot(kt, s2, *, pv=True, qk=True, ex=True, rowsum=True, kaddr_update=False, vread=True):
    G = 4 * N
    par = s2
    prev, curp = PBUF[par ^ 1], PBUF[par]
    S, tgt = SBUF[kt], SBUF[kt ^ 1]
    tgt_off = 4096 if kt == 0 else 0
    dsteps = (2 * s2, 2 * s2 + 1)
    exps = [(blk, i) for blk in range(N) for i in range(8)] if ex else []
    fill = [[] for _ in range(G)]
    for g in range(G):
        for (blk, i) in exps[2 * g: 2 * g + 2]:
            r = S[blk] + 8 * s2 + i
            fill[g].append(f"v_mov_b32_e32 v{r}, v{r}" if "noexp" in ABL else f"v_exp_f32_e32 v{r}, v{r}")
    if ex:
        for j in range(4 * N):
            blk, i = exps[2 * j]
            r = S[blk] + 8 * s2 + i
            if "nocvt" not in ABL:
                fill[min(j + 1, G - 1)].append(f"v_cvt_pk_bf16_f32 v{curp[blk] + i // 2}, v{r}, v{r + 1}")
    if ex and "rowsum_vadd" in ABL:            # row sums as f32 adds on the exps (two accumulators per block), one gap behind the exp
        for j, (blk, i) in enumerate(exps):
            r = S[blk] + 8 * s2 + i
            g = min(j // 2 + 1, G - 1)
            fill[g].insert(0, f"v_add_f32_e32 v{LSUM[blk] + (i & 1)}, v{LSUM[blk] + (i & 1)}, v{r}")
    if ex and "rowsum_pkadd" in ABL:           # the same as ONE packed add per pair of exps (v_pk_add_f32 on aligned pairs)
        for j in range(4 * N):
            blk, i = exps[2 * j]
            r = S[blk] + 8 * s2 + i
            assert r % 2 == 0 and LSUM[blk] % 2 == 0
            ins = f"v_pk_add_f32 {vr(LSUM[blk], 2)}, {vr(LSUM[blk], 2)}, {vr(r, 2)}"
            if j + 1 < G:
                fill[j + 1].insert(0, ins)
            else:
                fill[G - 1].append(ins)          # behind the last gap's converts: two instructions after its exps
    if rowsum and pv and "rowsum_dot2" in ABL:     # v_dot2c_f32_bf16 with packed ones on the packed P of quarter q-1: one per MFMA gap
        n = 0
        for blk in range(N):
            for r in range(4):
                fill[n % G].append(f"v_dot2c_f32_bf16_e32 v{LSUM[blk] + (r & 1)}, v{prev[blk] + r}, v{ONES}")
                n += 1
    if rowsum and pv and "norowsum" not in ABL and not VSUM:
        n = 0
        for blk in range(N):
            for half in range(2):
                acc = l_reg(blk)
                fill[2 * n + 1].append(f"v_mfma_f32_4x4x4_16b_bf16 {acc}, {vr(ONES, 2)}, {vr(prev[blk] + 2 * half, 2)}, {acc}")
                n += 1
    if qk:
        if kaddr_update:
            for s in range(4):
                fill[0].insert(0, f"v_add_u32_e32 v{KADDR[s]}, {S_KDELTA}, v{KADDR[s]}")
        fill[0].append(f"ds_read_b128 {vr(KF[0], 4)}, v{KADDR[dsteps[0]]} offset:{tgt_off}")
        fill[0].append(f"ds_read_b128 {vr(KF[1], 4)}, v{KADDR[dsteps[1]]} offset:{tgt_off}")
    vbase = (32 * kt + 16 * s2) * 128
    if vread:
        fill[N - 1 if pv else 1] += [f"ds_read_b64_tr_b16 {vr(VF[0], 2)}, v{VADDR[0]} offset:{vbase}",
                                     f"ds_read_b64_tr_b16 {vr(VF[0] + 2, 2)}, v{VADDR[0]} offset:{vbase + 1024}"]
        fill[2 * N - 1 if pv else 2] += [f"ds_read_b64_tr_b16 {vr(VF[1], 2)}, v{VADDR[1]} offset:{vbase}",
                                         f"ds_read_b64_tr_b16 {vr(VF[1] + 2, 2)}, v{VADDR[1]} offset:{vbase + 1024}"]
    mf = []
    for dt in range(2):
        for blk in range(N):
            o = o_reg(blk, dt)
            mf.append(f"v_mfma_f32_32x32x16_bf16 {o}, {vr(VF[dt], 4)}, {vr(prev[blk], 4)}, {o}" if pv and "nopv" not in ABL else None)
    for n, s in enumerate(dsteps):
        for blk in range(N):
            acc = vr(tgt[blk], 16)
            c = "0" if s == 0 else acc
            mf.append(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(KF[n], 4)}, {ar(Q[blk][s], 4)}, {c}"
                      if qk and not ("noqk" in ABL and ex) else None)
    if "nolds" in ABL:
        fill = [[x for x in f if not x.startswith("ds_read")] for f in fill]
    if "noldsk" in ABL:
        fill = [[x for x in f if not x.startswith("ds_read_b128")] for f in fill]
    if "noldsv" in ABL:
        fill = [[x for x in f if not x.startswith("ds_read_b64_tr")] for f in fill]
    if "cvtdummy" in ABL:     # the cvts issue but write a scratch register: no P dependency into the MFMAs
        fill = [[(f"v_cvt_pk_bf16_f32 v{ONES}, " + x.split(", ", 1)[1]) if x.startswith("v_cvt_pk") else x for x in f] for f in fill]
    if "noexp2" in ABL:
        fill = [[x for x in f if not x.startswith(("v_exp", "v_mov_b32_e32"))] for f in fill]
    emit(f"; ---- slot kt={kt} s2={s2} pv={int(pv)} qk={int(qk)} ex={int(ex)}")
    # dependent row-sum MFMAs must not follow each other directly (the second would read its accumulator before the first
    # has written it): where a slot carries no big MFMA between them (the drain), pad
    if not qk:
        for g in range(len(fill)):
            fill[g] = [y for x in fill[g] for y in ((["s_nop 7"] if x.startswith("v_mfma_f32_4x4x4") else []) + [x])]
    for g in range(G):
        if g == 0 and pv and "nowait" not in ABL:
            emit("s_waitcnt lgkmcnt(0)")
        if g == 2 * N and qk and "nowait" not in ABL:
            emit("s_waitcnt lgkmcnt(4)" if vread else "s_waitcnt lgkmcnt(0)")
        if mf[g] is not None:
            emit(mf[g])
        for f in fill[g]:
            emit(f)


def tile_head():
    emit("; ---- tile head")
    for i in range(0 if "nodma" in ABL else PIECES // 2):
        emit(f"s_mov_b32 m0, {S_KDST}" if i == 0 else f"s_add_u32 m0, {S_KDST}, {1024 * i}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 v{KSRC[i]}, {S_KG}")
    for i in range(0 if "nodma" in ABL else PIECES // 2):
        emit(f"s_mov_b32 m0, {S_VDST}" if i == 0 else f"s_add_u32 m0, {S_VDST}, {1024 * i}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 v{VSRC[i]}, {S_VG}")
    emit(f"s_add_u32 {S_KG_LO}, {S_KG_LO}, {S_TB}")
    emit(f"s_addc_u32 {S_KG_HI}, {S_KG_HI}, 0")
    emit(f"s_add_u32 {S_VG_LO}, {S_VG_LO}, {S_TB}")
    emit(f"s_addc_u32 {S_VG_HI}, {S_VG_HI}, 0")
    emit(f"s_cmp_eq_u32 {S_KDST}, {S_KD0}")
    emit(f"s_cselect_b32 {S_TMP}, {S_KD1}, {S_KD0}")
    emit(f"s_cmp_eq_u32 {S_KDST}, {S_KD1}")
    emit(f"s_cselect_b32 {S_KDST}, {S_KD2}, {S_TMP}")
    emit(f"s_add_u32 {S_VDST}, {S_VDST}, {S_VDELTA}")
    emit(f"v_add_u32_e32 v{VADDR[0]}, {S_VDELTA}, v{VADDR[0]}")
    emit(f"v_add_u32_e32 v{VADDR[1]}, {S_VDELTA}, v{VADDR[1]}")
    emit(f"s_sub_u32 {S_VDELTA}, 0, {S_VDELTA}")
    emit(f"s_cmp_eq_u32 {S_KCUR}, {S_KA0}")
    emit(f"s_cselect_b32 {S_TMP}, {S_KA1}, {S_KA0}")
    emit(f"s_cmp_eq_u32 {S_KCUR}, {S_KA1}")
    emit(f"s_cselect_b32 {S_TMP}, {S_KA2}, {S_TMP}")
    emit(f"s_sub_u32 {S_KDELTA}, {S_TMP}, {S_KCUR}")
    emit(f"s_mov_b32 {S_KCUR}, {S_TMP}")


def tile_tail():
    if "nobarrier" not in ABL:
        emit("s_waitcnt vmcnt(0)")
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")


def entry():
    emit("; ==== entry")
    if PRODUCT:
        for i in range(4):
            emit(f"v_xor_b32_e32 v{KADDR[i]}, {32 * i}, %[ka0]")
        emit(f"v_mov_b32_e32 v{VADDR[0]}, %[va0]")
        emit(f"v_xor_b32_e32 v{VADDR[1]}, 64, %[va0]")
    else:
        for i in range(4):
            emit(f"v_mov_b32_e32 v{KADDR[i]}, %[ka{i}]")
        emit(f"v_mov_b32_e32 v{VADDR[0]}, %[va0]")
        emit(f"v_mov_b32_e32 v{VADDR[1]}, %[va1]")
    for i in range(PIECES // 2):
        emit(f"v_add_u32_e32 v{KSRC[i]}, {i}*%[rowstep], %[ksrc]" if False else f"v_mov_b32_e32 v{KSRC[i]}, %[ksrc{i}]")
        emit(f"v_mov_b32_e32 v{VSRC[i]}, %[vsrc{i}]")
    emit(f"v_mov_b32_e32 v{ONES}, 0x3f803f80")
    emit(f"v_mov_b32_e32 v{ONES + 1}, 0x3f803f80")
    emit(f"s_mov_b32 {S_CNT}, %[cnt]")
    emit(f"s_mov_b64 {S_KG}, %[kg]")
    emit(f"s_mov_b64 {S_VG}, %[vg]")
    emit(f"s_mov_b32 {S_TB}, %[tb]")
    emit(f"s_mov_b32 {S_KD0}, %[kd0]")
    emit(f"s_add_u32 {S_KD1}, {S_KD0}, 8192")
    emit(f"s_add_u32 {S_KD2}, {S_KD0}, 32768")
    emit(f"s_mov_b32 {S_KDST}, {S_KD0}")
    emit(f"s_add_u32 {S_VDST}, {S_KD0}, 16384")
    emit(f"s_mov_b32 {S_VDELTA}, 8192")
    emit(f"s_mov_b32 {S_KA0}, 0")
    emit(f"s_mov_b32 {S_KA1}, 8192")
    emit(f"s_mov_b32 {S_KA2}, 32768")
    emit(f"s_mov_b32 {S_KCUR}, 8192")
    # Q fragments straight into the accumulator file, O and the row sums zeroed
    for blk in range(N):
        for s in range(4):
            emit(f"global_load_dwordx4 {ar(Q[blk][s], 4)}, %[qp{blk}], off offset:{32 * s}")
    for r in range(0 if PRODUCT else 32 * N):
        emit(f"v_accvgpr_write_b32 a{r}, 0")
    for blk in range(0 if PRODUCT else N):
        for r in range(4):
            emit(f"v_accvgpr_write_b32 a{LACC[blk] + r}, 0")
    for par in range(2):
        for blk in range(N):
            for r in range(4):
                emit(f"v_mov_b32_e32 v{PBUF[par][blk] + r}, 0")
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 4")


def build():
    entry()
    slot(1, 0, pv=False, qk=True, ex=False, rowsum=False)
    emit("s_waitcnt lgkmcnt(0)")
    slot(1, 1, pv=False, qk=True, ex=False, rowsum=False)
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 7")
    emit("s_nop 7")
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
    slot(0, 0, pv=True, qk=False, ex=False, rowsum=True, vread=False)
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 7")
    emit("s_nop 7")
    if PRODUCT:
        return
    # bench: park one register of every accumulator in the output operand so that the result depends on all the work
    emit(f"v_accvgpr_read_b32 %[out], a{O[0][0]}")
    for blk in range(N):
        for dt in range(2):
            emit(f"v_accvgpr_read_b32 v{KF[0]}, a{O[blk][dt] + 3}")
            emit(f"v_add_f32_e32 %[out], %[out], v{KF[0]}")
        emit(f"v_accvgpr_read_b32 v{KF[0]}, a{LACC[blk]}")
        emit(f"v_add_f32_e32 %[out], %[out], v{KF[0]}")


build()
with open(OUT, "w") as f:
    f.write(f"// GENERATED by tools/gen_attn_asm2.py {NBLK} - do not edit.\n")
    f.write(f"#define A64B{N}_LOOP_ASM \\\n")
    for ln in lines:
        f.write('  "' + ln + '\\n\\t" \\\n')
    f.write('  ""\n')
    f.write(f"#define A64B{N}_CLOBBER_V " + ", ".join(f'"v{i}"' for i in range(V0, LASTV + 1)) + "\n")
    f.write(f"#define A64B{N}_CLOBBER_A " + ", ".join(f'"a{i}"' for i in range(FIRSTA, LASTA + 1)) + "\n")
    f.write(f"#define A64B{N}_CLOBBER_S " + ", ".join(f'"s{i}"' for i in SCALARS) + "\n")
print("wrote", OUT, len(lines), "lines; arch v", V0, "..", LASTV, "acc a0 ..", LASTA)
