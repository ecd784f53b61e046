// Attention forward, 64 query rows per wave (two 32-row blocks A/B), 256 query rows per 256-thread workgroup.
//
// Same math and data flow as attn.hip (see its header: S^T = K.Q^T with the query on the lane, the bf16-converted
// accumulators are the B operand of O^T += V^T.P^T, -m folded into the first MFMA's C operand, deferred rescale), but
// every K fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16) feeds TWO MFMAs, one per query block:
//   * K/V global->LDS traffic per FLOP halves (256 instead of 128 query rows share a staged tile): the ablation of
//     attn.hip shows 19 % of its time is the staging path (7 TB/s of L2/MALL reads at S = 64 300);
//   * LDS fragment reads per FLOP halve;
//   * the two blocks are independent, so one block's softmax (VALU) overlaps the other block's MFMAs inside a wave.
// Cost: ~250 VGPRs -> 2 waves per SIMD (2 workgroups per CU).
#include "common.h"
#include <stdlib.h>

struct Attn64Params {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  long tok_stride, batch_stride;
  bf16_t* o; long o_tok_stride, o_batch_stride;
  int S, H, B, nqb;
};

#define A64_QB 256
#define A64_KT 64
#define A64_THR 6.0f

__device__ __forceinline__ bf16x8 a64_cat4(bf16x4 a, bf16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// online-softmax step of one 32-row block on its two raw score tiles; m is the per-lane running max (exp2 domain)
template <bool FIRST>
__device__ __forceinline__ void a64_softmax(f32x16 (&sc)[2], float& m, f32x16 (&o)[2], float& l, bf16x8 (&pf)[2][2]) {
  float tmax = sc[0][0];
#pragma unroll
  for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, sc[0][i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, sc[1][i]);
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
  if (FIRST) {
    m = tmax;
  } else if (!__all(tmax <= m + A64_THR)) {
    const float mn = fmaxf(m, tmax);
    float alpha = __builtin_amdgcn_exp2f(m - mn);
    asm volatile("s_nop 1" : "+v"(alpha));  // v_exp -> VALU (trans) hazard is not padded ahead of inline asm
    l *= alpha;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(o[dt][i]) : "v"(alpha));
    m = mn;
  }
  float psum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u32x4 pw;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float p0 = __builtin_amdgcn_exp2f(sc[kt][8 * s2 + 2 * jj] - m);
        const float p1 = __builtin_amdgcn_exp2f(sc[kt][8 * s2 + 2 * jj + 1] - m);
        psum += p0 + p1;
        pw[jj] = pack_bf16x2(p0, p1);
      }
      pf[kt][s2] = __builtin_bit_cast(bf16x8, pw);
    }
  l += psum;
}

// NW = waves per workgroup (4 or 8): NW * 64 query rows share one staged K/V tile.
// GLDS: stage K/V with LDS-DMA (global_load_lds, swizzle on the source address) instead of registers + ds_write.
// ORDER 0: the two query blocks share every fragment read (QK and PV MFMAs interleaved A/B).
// ORDER 1: block-sequential issue  QK_A, QK_B, softmax_A, PV_A, softmax_B, PV_B  (fragments re-read per block): one
//          block's softmax (VALU) is issued while the other block's MFMAs are still in the matrix pipe.
template <int NW, bool GLDS = false, int ORDER = 0>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd64_kernel(Attn64Params p) {
  __shared__ __attribute__((aligned(16))) char lds[32768];  // K ring [2][64][128 B] then V ring [2][64][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nwg = p.nqb * p.H * p.B;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int qb = id % p.nqb;
  const int head = (id / p.nqb) % p.H;
  const int b = id / (p.nqb * p.H);
  const int S = p.S;

  const int q0 = qb * (NW * 64) + wave * 64;
  bf16x8 qfA[4], qfB[4];
  {
    const int ra = min(q0 + r, S - 1), rb = min(q0 + 32 + r, S - 1);
    const bf16_t* pa = p.q + (long)b * p.batch_stride + (long)ra * p.tok_stride + head * 64;
    const bf16_t* pb = p.q + (long)b * p.batch_stride + (long)rb * p.tok_stride + head * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qfA[s] = *(const bf16x8*)(pa + 16 * s + 8 * h);
      qfB[s] = *(const bf16x8*)(pb + 16 * s + 8 * h);
    }
  }
  f32x16 oA[2], oB[2];
  float mA = 0.f, mB = 0.f;
  oA[0] = oA[1] = oB[0] = oB[1] = (f32x16)(0.f);
  float lA = 0.f, lB = 0.f;

  const int nt = (S + A64_KT - 1) / A64_KT;
  const bf16_t* kbase = p.k + (long)b * p.batch_stride + head * 64;
  const bf16_t* vbase = p.v + (long)b * p.batch_stride + head * 64;
  constexpr int CPT = 512 / (NW * 64);   // 16-byte chunks of K (and of V) per thread and tile
  int srow[CPT], sch[CPT], kw[CPT], vw[CPT];
  unsigned goff[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int cid = tid + NW * 64 * i;
    srow[i] = cid >> 3;
    sch[i] = cid & 7;
    kw[i] = srow[i] * 128 + ((sch[i] ^ ((srow[i] >> 1) & 7)) << 4);
    vw[i] = 16384 + srow[i] * 128 + ((sch[i] ^ (((srow[i] >> 1) & 1) << 2)) << 4);
    goff[i] = (unsigned)(srow[i] * p.tok_stride + sch[i] * 8) * 2u;
  }
  u32x4 kr[CPT], vr[CPT];
  auto load_tile = [&](int T, bool clamp) {
    if (clamp) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        int grow = T * A64_KT + srow[i];
        grow = grow < S ? grow : S - 1;
        kr[i] = *(const u32x4*)(kbase + (long)grow * p.tok_stride + sch[i] * 8);
        vr[i] = *(const u32x4*)(vbase + (long)grow * p.tok_stride + sch[i] * 8);
      }
    } else {
      const char* kt_ = (const char*)(kbase + (long)T * A64_KT * p.tok_stride);
      const char* vt_ = (const char*)(vbase + (long)T * A64_KT * p.tok_stride);
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        kr[i] = *(const u32x4*)(kt_ + goff[i]);
        vr[i] = *(const u32x4*)(vt_ + goff[i]);
      }
    }
  };
  // LDS-DMA form: wave w, instruction i writes the 1 KiB segment (rows 8*seg .. 8*seg+7) of the K (V) tile; the lane's
  // slot (row, pos) must receive source chunk pos ^ f(row)
  auto glds_tile = [&](int T, bool clamp, int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int seg = wave * CPT + i;
      const int row = seg * 8 + (lane >> 3), pos = lane & 7;
      int grow = T * A64_KT + row;
      if (clamp) grow = grow < S ? grow : S - 1;
      const char* ks = (const char*)(kbase + (long)grow * p.tok_stride) + ((pos ^ ((row >> 1) & 7)) << 4);
      const char* vs = (const char*)(vbase + (long)grow * p.tok_stride) + ((pos ^ (((row >> 1) & 1) << 2)) << 4);
      __builtin_amdgcn_global_load_lds(GLB_PTR(ks), LDS_PTR(lds + buf * 8192 + seg * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(vs), LDS_PTR(lds + 16384 + buf * 8192 + seg * 1024), 16, 0, 0);
    }
  };
  auto write_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      *(u32x4*)(lds + buf * 8192 + kw[i]) = kr[i];
      *(u32x4*)(lds + buf * 8192 + vw[i]) = vr[i];
    }
  };
  if constexpr (GLDS) {
    glds_tile(0, true, 0);
  } else {
    load_tile(0, true);
    write_tile(0);
  }
  __syncthreads();

  const int kswz = (r >> 1) & 7;
  const int krow_off = r * 128;
  const int gi = lane & 15, gg = (lane >> 4) & 1;
  const int vrow_l = 4 * h + (gi >> 2);
  const int vcol_l = 16 * gg + 4 * (gi & 3);
  const int vch_l = vcol_l >> 3;
  const int vin_l = (vcol_l & 7) * 2;
  const int vswz = ((vrow_l >> 1) & 1) << 2;

#define A64_QK(QF, SC)                                                                                           \
  {                                                                                                               \
    {                                                                                                             \
      const int off = (h ^ kswz) << 4;                                                                            \
      const bf16x8 a0 = *(const bf16x8*)(kl + off);                                                               \
      const bf16x8 a1 = *(const bf16x8*)(kl + 32 * 128 + off);                                                    \
      SC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, QF[0], (f32x16)(0.f), 0, 0, 0);                         \
      SC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, QF[0], (f32x16)(0.f), 0, 0, 0);                         \
    }                                                                                                             \
    _Pragma("unroll") for (int s = 1; s < 4; ++s) {                                                               \
      const int off = ((2 * s + h) ^ kswz) << 4;                                                                  \
      const bf16x8 a0 = *(const bf16x8*)(kl + off);                                                               \
      const bf16x8 a1 = *(const bf16x8*)(kl + 32 * 128 + off);                                                    \
      SC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, QF[s], SC[0], 0, 0, 0);                                 \
      SC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, QF[s], SC[1], 0, 0, 0);                                 \
    }                                                                                                             \
  }
#define A64_PV(PF, O)                                                                                            \
  _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                                \
  _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                              \
    const int row0 = 32 * kt + 16 * s2 + vrow_l;                                                                  \
    _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                            \
      const int ch = (4 * dt + vch_l) ^ vswz;                                                                     \
      const char* a = vl + row0 * 128 + (ch << 4) + vin_l;                                                        \
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                                 \
          (__attribute__((address_space(3))) bf16x4*)LDS_PTR(a));                                                 \
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                                 \
          (__attribute__((address_space(3))) bf16x4*)LDS_PTR(a + 8 * 128));                                       \
      O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a64_cat4(lo, hi), PF[kt][s2], O[dt], 0, 0, 0);              \
    }                                                                                                             \
  }
#define A64_MASK(T, SC)                                                                                          \
  {                                                                                                               \
    const int kb = (T) * A64_KT + 4 * h;                                                                          \
    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                              \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                              \
      const int key = kb + 32 * kt + (i & 3) + 8 * (i >> 2);                                                      \
      if (key >= S) SC[kt][i] = -INFINITY;                                                                        \
    }                                                                                                             \
  }

#define A64_TILE(T, BUF, FIRST, LAST, CLAMPNEXT)                                                                  \
  {                                                                                                               \
    const int buf = (BUF);                                                                                        \
    if (!(LAST)) { if constexpr (GLDS) glds_tile((T) + 1, CLAMPNEXT, buf ^ 1); else load_tile((T) + 1, CLAMPNEXT); } \
    f32x16 scA[2], scB[2];                                                                                        \
    bf16x8 pfA[2][2], pfB[2][2];                                                                                  \
    const char* kl = lds + buf * 8192 + krow_off;                                                                 \
    const char* vl = lds + 16384 + buf * 8192;                                                                    \
    const bool masktail = (LAST) && (S & (A64_KT - 1));                                                           \
    if constexpr (ORDER == 1) {                                                                                   \
      A64_QK(qfA, scA)                                                                                            \
      A64_QK(qfB, scB)                                                                                            \
      if (masktail) { A64_MASK(T, scA) A64_MASK(T, scB) }                                                         \
      a64_softmax<FIRST>(scA, mA, oA, lA, pfA);                                                                   \
      A64_PV(pfA, oA)                                                                                             \
      a64_softmax<FIRST>(scB, mB, oB, lB, pfB);                                                                   \
      A64_PV(pfB, oB)                                                                                             \
    } else {                                                                                                      \
      {                                                                                                           \
        const int off = (h ^ kswz) << 4;                                                                          \
        const bf16x8 a0 = *(const bf16x8*)(kl + off);                                                             \
        const bf16x8 a1 = *(const bf16x8*)(kl + 32 * 128 + off);                                                  \
        scA[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qfA[0], (f32x16)(0.f), 0, 0, 0);                     \
        scB[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qfB[0], (f32x16)(0.f), 0, 0, 0);                     \
        scA[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qfA[0], (f32x16)(0.f), 0, 0, 0);                     \
        scB[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qfB[0], (f32x16)(0.f), 0, 0, 0);                     \
      }                                                                                                           \
      _Pragma("unroll") for (int s = 1; s < 4; ++s) {                                                             \
        const int off = ((2 * s + h) ^ kswz) << 4;                                                                \
        const bf16x8 a0 = *(const bf16x8*)(kl + off);                                                             \
        const bf16x8 a1 = *(const bf16x8*)(kl + 32 * 128 + off);                                                  \
        scA[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qfA[s], scA[0], 0, 0, 0);                            \
        scB[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qfB[s], scB[0], 0, 0, 0);                            \
        scA[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qfA[s], scA[1], 0, 0, 0);                            \
        scB[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qfB[s], scB[1], 0, 0, 0);                            \
      }                                                                                                           \
      if (masktail) { A64_MASK(T, scA) A64_MASK(T, scB) }                                                         \
      a64_softmax<FIRST>(scA, mA, oA, lA, pfA);                                                                   \
      a64_softmax<FIRST>(scB, mB, oB, lB, pfB);                                                                   \
      _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                            \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                          \
        const int row0 = 32 * kt + 16 * s2 + vrow_l;                                                              \
        _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                        \
          const int ch = (4 * dt + vch_l) ^ vswz;                                                                 \
          const char* a = vl + row0 * 128 + (ch << 4) + vin_l;                                                    \
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                             \
              (__attribute__((address_space(3))) bf16x4*)LDS_PTR(a));                                             \
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                             \
              (__attribute__((address_space(3))) bf16x4*)LDS_PTR(a + 8 * 128));                                   \
          const bf16x8 vf = a64_cat4(lo, hi);                                                                     \
          oA[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfA[kt][s2], oA[dt], 0, 0, 0);                     \
          oB[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfB[kt][s2], oB[dt], 0, 0, 0);                     \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    if (!(LAST)) {                                                                                                \
      if constexpr (!GLDS) write_tile(buf ^ 1);                                                                   \
      __syncthreads();                                                                                            \
    }                                                                                                             \
  }

  const bool tail = (S & (A64_KT - 1)) != 0;
  if (nt == 1) {
    A64_TILE(0, 0, true, true, false)
  } else {
    A64_TILE(0, 0, true, false, (tail && nt == 2))
    for (int t = 1; t < nt - 1; ++t) A64_TILE(t, (t & 1), false, false, (tail && t == nt - 2))
    A64_TILE(nt - 1, ((nt - 1) & 1), false, true, false)
  }

  // finalize both blocks
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const float lsum = blk ? lB : lA;
    const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32, 64));
    const int row = q0 + 32 * blk + r;
    if (row < S) {
      bf16_t* optr = p.o + (long)b * p.o_batch_stride + (long)row * p.o_tok_stride + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x16& ov = blk ? oB[dt] : oA[dt];
          u32x2 w;
          w[0] = pack_bf16x2(ov[4 * g + 0] * inv, ov[4 * g + 1] * inv);
          w[1] = pack_bf16x2(ov[4 * g + 2] * inv, ov[4 * g + 3] * inv);
          *(u32x2*)(optr + 32 * dt + 8 * g + 4 * h) = w;
        }
    }
  }
}

// Called by pi3_attention (attn.hip) for long sequences; same argument meaning.
int pi3_attention64_launch(const void* q, const void* k, const void* v, long tok_stride, long batch_stride, void* o,
                           long o_tok_stride, long o_batch_stride, int B, int S, int H, hipStream_t stream) {
  Attn64Params p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v;
  p.tok_stride = tok_stride; p.batch_stride = batch_stride;
  p.o = (bf16_t*)o; p.o_tok_stride = o_tok_stride; p.o_batch_stride = o_batch_stride;
  static int nw = -1;   // PI3_ATTN_NW: 4 or 8 waves per workgroup (A/B knob)
  if (nw < 0) {
    const char* e = getenv("PI3_ATTN_NW");
    nw = e ? atoi(e) : 8;
  }
  const int qrows = nw == 8 ? 512 : 256;
  p.S = S; p.H = H; p.B = B; p.nqb = (S + qrows - 1) / qrows;
  const long nwg = (long)p.nqb * H * B;
  static int glds = -1;   // PI3_ATTN_GLDS: 1 = LDS-DMA staging (A/B knob)
  if (glds < 0) {
    const char* e = getenv("PI3_ATTN_GLDS");
    glds = e ? atoi(e) : 1;
  }
  static int order = -1;   // PI3_ATTN_ORDER: 0 interleaved blocks, 1 block-sequential issue (A/B knob)
  if (order < 0) {
    const char* e = getenv("PI3_ATTN_ORDER");
    order = e ? atoi(e) : 0;
  }
  if (nw == 8 && glds && order == 1)
    hipLaunchKernelGGL((attn_fwd64_kernel<8, true, 1>), dim3((unsigned)nwg), dim3(512), 0, stream, p);
  else if (nw == 8 && glds)
    hipLaunchKernelGGL((attn_fwd64_kernel<8, true>), dim3((unsigned)nwg), dim3(512), 0, stream, p);
  else if (nw == 8)
    hipLaunchKernelGGL(attn_fwd64_kernel<8>, dim3((unsigned)nwg), dim3(512), 0, stream, p);
  else
    hipLaunchKernelGGL(attn_fwd64_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, stream, p);
  return pi3_check_launch("attn_fwd64");
}
