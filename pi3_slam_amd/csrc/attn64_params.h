// Launch parameters shared by the 64-rows-per-wave attention kernels (attn64.hip, attn64p.hip): one definition, so the
// two translation units cannot drift apart (pi3_attention64p_launch receives the struct attn64.hip filled).
#pragma once
#include "common.h"

struct Attn64Params {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  long tok_stride, batch_stride;
  bf16_t* o; long o_tok_stride, o_batch_stride;
  int S, H, B, nqb;
  const float* k2max;   // [B][H] max over keys of |k|^2, or null (no a-priori test: optimistic pass, or online-max loop under knob 1)
  int prio;             // 1: waves NW/2 .. NW-1 (the later-dispatched wave of every SIMD) run at s_setprio 1 (A/B knob)
  int tailopt;          // 1: short-sequence waves skip query blocks / key halves that do not exist (A/B knob, default 1)
  unsigned long long* dbg;   // -DPI3_ATTN_STAMPS builds only: s_memtime stamps of workgroup 0
  unsigned* stats;      // optional caller-owned path counters (pi3_attention_path_counters), else null
  int optim;            // 1: optimistic bounded-score loop + acceptance test (a64_reject); 0: a-priori test on k2max
  int redo;             // 1: the follow-up launch of the optimistic form: only workgroups that left the mark run, on the online-max loop
};
