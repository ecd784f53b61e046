// Shared by gemm.hip (128x128 tile) and gemm256.hip (256x256 pipelined tile): parameter block and the fused epilogue.
#pragma once
#include "common.h"

struct GemmParams {
  const void* A; long lda;   // [M][K], element stride
  const void* W; long ldw;   // [N][K]
  int M, N, K;
  const float* bias;         // [N] or null
  const float* gamma;        // [N] or null  (LayerScale)
  const float* resid; long ldr;  // fp32 [*][ldr] residual source (indexed by the remapped row) or null
  void* out; long ldo;       // bf16 or f32
  int rpg, gstride, goff;    // row remap: orow = (m / rpg) * gstride + goff + (m % rpg); rpg == 0 -> orow = m
  const float* addtab; long ldadd;  // optional f32 table [rpg][ldadd], row (m % rpg), added after everything else
  float qscale; int qcols;   // columns n < qcols are multiplied by qscale (softmax scale folded into q)
  // implicit-GEMM 3x3 convolution, replicate padding (moge/model/modules.py:47-60): A is an NHWC image
  // [B][cH][cW][cC] (cC % 64 == 0), row m = pixel, K = 9 * cC with k = (ky*3 + kx) * cC + ci; cW == 0 -> plain GEMM
  int cH, cW, cC;
  int f16;                   // gemm.hip kernels only: the 16-bit operands (and a 16-bit output) are IEEE half, not bf16:
                             // v_mfma_f32_16x16x32_f16, same rate - MoGe, which the reference runs under fp16 autocast
  int tile_gm;               // gemm256: m-tiles per scheduling group (0 = default 8); consecutive ids walk a group's m-tiles
  int tile_order;            // gemm256: 0 = m-tiles first inside a group, 1 = column tiles first
  int stagger_ns;            // gemm256 (persistent form): workgroup b starts b * stagger_ns later (0 = all at once)
  int ilv_prio;              // interleaved K loop (knob gemm_ilv = 2 / 3): 1 = waves 4-7 at priority 1, 2 = the two waves of a SIMD take turns
  int rpref;                 // gemm256, f32 output with a residual: touch the tile's residual lines during the last K tiles
#ifdef PI3_DEV_ABLATIONS
  int abl;                   // timing-only ablation bits (development builds; results are wrong)
#endif
  // fused q/k epilogue of the packed qkv projection (gemm256 only; FlashAttentionRope.forward,
  // pi3/models/layers/attention.py:323-334): columns [0, H*64) = q, [H*64, 2*H*64) = k, rest = v.  For q and k heads:
  // per-head LayerNorm(64) (optional: qk_w != null), RoPE-2D (optional), softmax scale folded into q, and max_s |k|^2
  // per (attention batch, head) for the bounded-score attention path (optional: k2max != null).
  int qk_mode;               // 0 = off
  int qk_H, qk_T;            // heads; tokens per frame (row % T indexes pos)
  const int* qk_pos;         // [T][2] (y, x) or null (no RoPE)
  const float* qk_cs;        // [npos][16][2] (cos, sin)
  const float* qk_qw; const float* qk_qb; const float* qk_kw; const float* qk_kb;   // LayerNorm(64) affine or null
  float qk_eps, qk_qscale;
  float* qk_k2max; int qk_attnS;   // k2max[(row / attnS) * H + head]
};


// Fused epilogue of one wave, swapped orientation (MFMA tile rows = n, cols = m): the lane owns output row m(mi) and,
// per n-tile ni, the 4 consecutive columns n0(ni) .. n0(ni)+3.  Per-column vectors (bias, LayerScale) are loaded once
// for all NI tiles; per-row operands (residual, table) once per row, all NI at a time, so each optional operand costs
// one uniform branch and one wait per row instead of one per tile.
// RT16: the 16-bit output type is a run-time choice (p.f16: IEEE half instead of bf16) - the kernels of gemm.hip; the
// 256x256 kernel of the pi3 hot path instantiates RT16 = false and pays nothing for it.
template <bool OUT_BF16, int ACT, int NI, int MI, bool RT16 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[NI][MI], int m_base, int n_base,
                                              int lane) {
  const int frow = lane & 15;
  const int nq = (lane >> 4) * 4;
  f32x4 bias4[NI], gamma4[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    bias4[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gamma4[ni] = (f32x4){1.f, 1.f, 1.f, 1.f};
  }
  if (p.bias) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bias4[ni] = *(const f32x4*)(p.bias + n_base + ni * 16 + nq);
  }
  if (p.gamma) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) gamma4[ni] = *(const f32x4*)(p.gamma + n_base + ni * 16 + nq);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m_base + mi * 16 + frow;
    if (m >= p.M) continue;
    long orow = m;
    int trow = 0;
    if (p.rpg > 0) {
      const int gq = m / p.rpg;
      trow = m - gq * p.rpg;
      orow = (long)gq * p.gstride + p.goff + trow;
    }
    f32x4 extra[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) extra[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.resid) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) extra[ni] = *(const f32x4*)(p.resid + orow * p.ldr + n_base + ni * 16 + nq);
    }
    if (p.addtab) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        extra[ni] += *(const f32x4*)(p.addtab + (long)trow * p.ldadd + n_base + ni * 16 + nq);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n0 = n_base + ni * 16 + nq;
      f32x4 v = acc[ni][mi] + bias4[ni];
      if (n0 < p.qcols) v *= p.qscale;
      if constexpr (ACT == 1) {
        v = gelu_erf4(v);
      } else if constexpr (ACT == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_as(v[r]);
      } else if constexpr (ACT == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      v = v * gamma4[ni] + extra[ni];
      if constexpr (OUT_BF16) {
        u32x2 o;
        if (RT16 && p.f16) {
          o[0] = pack_f16x2(v[0], v[1]);
          o[1] = pack_f16x2(v[2], v[3]);
        } else {
          o[0] = pack_bf16x2(v[0], v[1]);
          o[1] = pack_bf16x2(v[2], v[3]);
        }
        *(u32x2*)((bf16_t*)p.out + orow * p.ldo + n0) = o;
      } else {
        *(f32x4*)((float*)p.out + orow * p.ldo + n0) = v;
      }
    }
  }
}
