// Spatial blocks of the F16X3 flow, fused: the LayerNorm-folded qkv GEMM of a frame group with the 17-key GRAND attention of its
// frames run from LDS (S2S:67 + 73-83 for the per-frame groups of S2S:119).  What it removes from the unfused flow
// (k_linear_x3q_persist<qkv form> + k_attn_temporal_x3p<1,8,3>): the q / k / v planes never exist in HBM -- 1.62 GB written and
// 1.62 GB read back per launch pair at the bench shape -- and one kernel launch per spatial block.
//
// Tile = 15 whole frames (255 token rows; 256 are staged and multiplied) x ONE head's q, k, v (192 output columns: the folded weight
// is stored HEAD-MAJOR at commit, rows [192 h, 192 h + 192) = q_h, k_h, v_h, so an N-tile is contiguous).  Eight waves (2 x 4), a
// wave owns 128 rows x 48 columns = 8 x 3 accumulator tiles of 16x16 (96 VGPRs).  The k-loop is the two-phase persistent loop of
// kernels_gemm_x3p.hip (LDS-DMA staging, counted vmcnt waits, W fragments read a phase ahead) on a 256 x 192 x 32 stage of
// 56 KiB; per output element the same MFMAs in the same order as every other F16X3 GEMM shape, and the epilogue arithmetic is that
// of x3q_epilogue8<LN-folded, planes>: the q / k / v values are bit for bit those the unfused qkv GEMM writes to HBM, so the whole
// block is bit-identical to the unfused flow ("fused_spatial" engine option, tests/test_gpu_round4.py).
//
// After the k-loop the tile's 255 x 192 values (196 KB as hi / lo fp16) do not fit the LDS at once; two passes:
//   pass 0: frames 0-3 and 8-11 (rows 0..67 of wave-row 0, 136..203 of wave-row 1: BOTH wave rows write half their accumulators)
//           -> q / k / v hi / lo planes of 8 frame slots (6 x 17 rows x 128 B each, the swizzles of kernels_attn_x3.hip) -> barrier
//           -> wave w runs one frame (scores, softmax, (P - I) V as in k_attn_temporal_x3p) and stores its 17 x 64 outputs as
//           whole 128-byte lines of the pair layout
//   pass 1: frames 4-7 and 12-14 likewise, 7 slots; the next tile's first k-tile is requested in front of its attention step
// (first version: pass 0 = rows 0..135, pass 1 = the rest -- each pass written by ONE wave row while the other waited: 3.8 + 4.4 us
// of a 40.5 us tile, in-kernel stamps; the epilogue steps are VALU-issue-bound with two waves per SIMD)
// LDS map (160 KiB): [0, 56 K) stage 0 | [56 K, 158 K) stage 1, then the frame slots (+ the raw row-statistics block during the
// k-loop) | 2 KiB of per-row LayerNorm statistics.
#include "d3d_kernels.h"
#include "qkv_fused_kloop.h"

#include <math.h>
#include <stdio.h>
#include <vector>

namespace d3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_alias __attribute__((ext_vector_type(4), may_alias));
typedef unsigned u32x2_alias __attribute__((ext_vector_type(2), may_alias));

constexpr int QS_BM = 256, QS_BN = 192, QS_TM = 8, QS_NJ = 3;
constexpr int QS_AREG = QS_BM * 128, QS_STAGE = (QS_BM + QS_BN) * 128;   // 57344
constexpr int QS_AIT = 4, QS_BIT = 3;                                    // 1-KiB DMA pieces per wave per k-tile
constexpr int QS_J = 17, QS_FPT = 15, QS_ROWS = QS_J * QS_FPT;           // 255 token rows per tile
constexpr int QS_PLANE = QS_J * 128, QS_SLOT = 6 * QS_PLANE;             // 2176, 13056
constexpr int QS_QKV = QS_STAGE;                                         // frame slots start behind stage 0
constexpr int QS_STX = QS_QKV + 8 * QS_SLOT;                             // 161792: (rstd', -mean rstd) of the tile's 256 rows, 2 KiB
constexpr int QS_RAW = 2 * QS_STAGE;                                     // raw statistics partials while the k-loop runs (16 KiB)
constexpr int QS_RAW_MAX = 16384;
constexpr int QS_LDS = QS_STX + QS_BM * 8;                               // 163840
static_assert(QS_LDS <= 160 * 1024 && QS_RAW + QS_RAW_MAX <= QS_STX, "LDS map");
// planes of a slot: V hi, V lo, K hi, K lo, Q hi, Q lo.  The fragment reads of the 15 pad rows of a plane run on into what follows
// it: for V that must be FINITE (0 x NaN in the second product) -- V hi runs into V lo, V lo into K hi, both written in the same pass;
// pad keys of K are overwritten with -inf scores and pad queries of Q are never stored: any bits will do there.
constexpr int QS_PV = 0, QS_PK = 2 * QS_PLANE, QS_PQ = 4 * QS_PLANE;

// swizzles of kernels_attn_x3.hip (K / Q rows: fragment reads of 16 consecutive rows at one logical chunk; V rows: transpose reads)
__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int vkey(int row) { return (((row >> 1) & 1) << 2) ^ ((row >> 2) & 3); }
__device__ __forceinline__ int vswz(int row, int chunk) { return row * 128 + ((chunk ^ vkey(row)) << 4); }

__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// (v0, v1) -> packed fp16 pairs hi = fp16(k v), lo = fp16(k v - hi): the split of split8_x3 / split_pair_f16 (same single roundings)
__device__ __forceinline__ void split_pair(float v0, float v1, float k, unsigned& hi, unsigned& lo) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(k));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(v0), "v"(k), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(k), "v"(hi));
}
__device__ __forceinline__ void split_pair_s(float e0, float e1, float k, unsigned& hi, unsigned& lo) {   // (scalar k: the E split)
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(e0), "s"(k));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(e1), "s"(k));
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(e0), "s"(k), "v"(hi));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(e1), "s"(k), "v"(hi));
}
__device__ __forceinline__ void split8_e(const float (&e)[8], h8& eh, h8& el) {
  u32x4 hv, lv;
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    unsigned a, b;
    split_pair_s(e[2 * pr], e[2 * pr + 1], 1024.0f, a, b);
    hv[pr] = a; lv[pr] = b;
  }
  eh = __builtin_bit_cast(h8, hv);
  el = __builtin_bit_cast(h8, lv);
}
// output patch (kernels_attn_x3.hip): lanes < 32 end up owning the whole hi chunk of their row, lanes >= 32 the whole lo chunk
__device__ __forceinline__ void patch_wr(unsigned char* patch, int r, int h, int g, h4 oh, h4 ol) {
  const uint2 a = __builtin_bit_cast(uint2, oh), b = __builtin_bit_cast(uint2, ol);
  const auto s0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
  const auto s1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
  u32x4_alias v;
  v[0] = s0[0]; v[1] = s1[0]; v[2] = s0[1]; v[3] = s1[1];
  *reinterpret_cast<u32x4_alias*>(patch + r * 128 + ((((h << 2) + g) ^ (r & 7)) << 4)) = v;
}
__device__ __forceinline__ u32x4 patch_rd(const unsigned char* patch, int row, int chunk) {
  return *reinterpret_cast<const u32x4_alias*>(patch + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// Diagnostic builds only (-DQS_ABL=n, wrong results; experiments/lds_conflict_attribution.sh): 1 no slot writes, 2 no attention units,
// 4 no output patches -- which LDS accesses the bank-conflict counter belongs to.
#ifndef QS_ABL
#define QS_ABL 0
#endif

struct QsArgs {
  const _Float16* Ap;      // residual stream, pair layout [>= 255 mtiles + 1 rows][2 K] of 8 x
  const _Float16* Wp;      // folded qkv weight W diag(gamma), pair layout, 2^k w, rows in TILE order: row 192 h + 48 wn + 16 part + x
                           // = original row 512 part + 64 h + 16 wn + x (wave wn of a tile holds q, k, v columns 16 wn .. + 15 of head h)
  const float* bias;       // b + W beta, head-major
  const float* csum;       // sum_k W[n, k] gamma[k], head-major
  const float* st_in;      // (sum, sum of squares) partials of the rows: [rows][st_np][2]
  int st_np;
  float eps, out_scale;    // LayerNorm eps; 2^-(3 + k)
  _Float16* out;           // attention output, pair layout [M][2 D] of 8 o
  int M, K, F, mtiles, D;  // tokens, GEMM depth, frames (M / 17), M-tiles (ceil(F / 15)), model width (8 heads x 64)
  unsigned* range;         // the engine's range-guard word
  unsigned long long* diag;   // diagnostic launches only ("qs_diag"): per workgroup 8 words -- cycles of wave 0 in the k-loop, the
                              // statistics step, write 0, attention 0, write 1 (+ prefetch), attention 1, tiles, 100 MHz ticks
};

#define QS_GLDS(SRC, DSTOFF)                                                                                            \
  __builtin_amdgcn_global_load_lds((SRC), (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (DSTOFF)), 16, 0, 0)

__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt vmcnt(n), n wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
  }
}

// One frame of one head from its LDS slot: the arithmetic of k_attn_temporal_x3p<1, 8, 3> (same MFMAs in the same order, same
// softmax, same conversions), outputs through the wave-private patch (aliasing the slot's Q planes, dead once the query fragments
// are in registers) as whole 128-byte lines.
__device__ __forceinline__ void qs_attention(unsigned char* slot, int lane, _Float16* out_row0, int D) {
  constexpr int T = QS_J;
  unsigned char* const sVh = slot + QS_PV;
  unsigned char* const sVl = slot + QS_PV + QS_PLANE;
  unsigned char* const sKh = slot + QS_PK;
  unsigned char* const sKl = slot + QS_PK + QS_PLANE;
  unsigned char* const sQh = slot + QS_PQ;
  unsigned char* const sQl = slot + QS_PQ + QS_PLANE;
  unsigned char* const patch = slot + QS_PQ;      // 4 KiB over the Q planes (4352 B)
  const int r = lane & 31, h = lane >> 5;
  h8 qh[4], ql[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {                // rows >= 17 read the planes behind: their query columns are never stored
    const int qo = kswz(r, 2 * ks + h);
    qh[ks] = *reinterpret_cast<const h8*>(sQh + qo);
    ql[ks] = *reinterpret_cast<const h8*>(sQl + qo);
  }
  f32x16 sacc;
#pragma unroll
  for (int q = 0; q < 16; ++q) sacc[q] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int ko = kswz(r, 2 * ks + h);
    const h8 kh = *reinterpret_cast<const h8*>(sKh + ko);
    const h8 kl = *reinterpret_cast<const h8*>(sKl + ko);
    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], sacc, 0, 0, 0);
  }
  // keys of accumulator register q in lane half h: (q & 3) + 8 (q >> 2) + 4 h.  With 17 keys, registers 9..15 hold pad keys in both
  // halves and register 8 (key 16 / 20) a real one in half 0 only: their numerators are exact zeros -- no exponentials, no sums
  float m = -INFINITY;
  if (h != 0) sacc[8] = -INFINITY;
#pragma unroll
  for (int q = 0; q < 9; ++q) m = fmaxf(m, sacc[q]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  constexpr float C_EXP = 1.4426950408889634f / 64.0f;
  const float mb = m * C_EXP;
  float l = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float e = 0.0f;
    if (q < 9) {
      e = __builtin_amdgcn_exp2f(fmaf(sacc[q], C_EXP, -mb));
      l += e;
    }
    sacc[q] = e;
  }
  l += __shfl_xor(l, 32, 64);
  f32x16 oacc[2];
#pragma unroll
  for (int q = 0; q < 16; ++q) { oacc[0][q] = 0.f; oacc[1][q] = 0.f; }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    h8 eh, el;
    {
      if (s2 == 0) {
        float e8[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) e8[jj] = sacc[jj];
        split8_e(e8, eh, el);
      } else {   // registers 9..15 are exact zeros: only the first pair carries a numerator
        unsigned a, b;
        split_pair_s(sacc[8], 0.0f, 1024.0f, a, b);
        u32x4 hv = {a, 0u, 0u, 0u}, lv = {b, 0u, 0u, 0u};
        eh = __builtin_bit_cast(h8, hv);
        el = __builtin_bit_cast(h8, lv);
      }
    }
    const int k0 = 16 * s2 + 4 * h;
    const int gi = lane & 15, tq_ = gi >> 2, tp_ = gi & 3;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int d0 = dt * 32 + 16 * ((lane >> 4) & 1);
      const int ch = (d0 >> 3) + (tp_ >> 1), sub = (tp_ & 1) * 8;
      const int o0 = vswz(k0 + tq_, ch) + sub, o1 = vswz(k0 + 8 + tq_, ch) + sub;
      const s4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)(sVh + o0));
      const s4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)(sVh + o1));
      const s4v c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)(sVl + o0));
      const s4v c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)(sVl + o1));
      h8 vh, vl;
      {
        const h4 a0h = __builtin_bit_cast(h4, a0), a1h = __builtin_bit_cast(h4, a1);
        const h4 c0h = __builtin_bit_cast(h4, c0), c1h = __builtin_bit_cast(h4, c1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vh[e] = a0h[e]; vh[4 + e] = a1h[e]; vl[e] = c0h[e]; vl[4 + e] = c1h[e]; }
      }
      oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, eh, oacc[dt], 0, 0, 0);
      oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, el, oacc[dt], 0, 0, 0);
      oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, eh, oacc[dt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // O = O^T / (2^13 l) - v_query, packed as hi / lo of 8 o; whole lines out through the patch
  const float inv = 1.0f / (8192.0f * l);
  const int tqc = r < T ? r : 0;
  // (no range tracking here: |o| = |(P - I) V| <= 2.0001 max |v|, and the slot writer raises the guard when a |v| exceeds 4090 --
  // half the plane range --, so an un-flagged run cannot overflow the output planes; 28 VALU instructions per unit less)
  u32x4 pw[6];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int vo = vswz(tqc, dt * 4 + g4) + 8 * h;
      // -v_query = -(hi + lo) / 8 by two v_fma_mix_f32 on the packed fp16 halves (exact: the pair sums to <= 22 bits), o = O^T inv - v_q
      // in one fma (the rounding every form of this kernel makes), then hi = fp16(8 o), lo = fp16(8 o - hi) by v_fma_mixlo / mixhi:
      // 6 VALU instructions per value where convert / add / scale / fma / clamp / convert / convert back / subtract / convert took 12
      // (same bits whenever |8 o| is inside the fp16 range; beyond it the range guard fires either way)
      const uint2 vqh = *reinterpret_cast<const uint2*>(sVh + vo);
      const uint2 vql = *reinterpret_cast<const uint2*>(sVl + vo);
      float o4[4];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const unsigned ph_ = pr ? vqh.y : vqh.x, pl_ = pr ? vql.y : vql.x;
        float n0, n1;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(n0) : "v"(pl_), "v"(-0.125f));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(n0) : "v"(ph_), "v"(-0.125f), "v"(n0));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(n1) : "v"(pl_), "v"(-0.125f));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(n1) : "v"(ph_), "v"(-0.125f), "v"(n1));
        o4[2 * pr] = __builtin_fmaf(oacc[dt][4 * g4 + 2 * pr], inv, n0);
        o4[2 * pr + 1] = __builtin_fmaf(oacc[dt][4 * g4 + 2 * pr + 1], inv, n1);
      }
      unsigned h0, l0, h1, l1;
      split_pair(o4[0], o4[1], 8.0f, h0, l0);
      split_pair(o4[2], o4[3], 8.0f, h1, l1);
      const h4 oh = __builtin_bit_cast(h4, make_uint2(h0, h1)), ol = __builtin_bit_cast(h4, make_uint2(l0, l1));
      if (!(QS_ABL & 4)) patch_wr(patch, r, h, g4, oh, ol);
      else asm volatile("" ::"v"(oh), "v"(ol));
    }
    asm volatile("" ::: "memory");     // (the rows read back were written by other lanes)
#pragma unroll
    for (int it = 0; it < 3; ++it) pw[dt * 3 + it] = (QS_ABL & 4) ? u32x4{0u, 0u, 0u, 0u} : patch_rd(patch, 8 * it + (lane >> 3), lane & 7);
    asm volatile("" ::: "memory");
  }
  _Float16* const pw_ptr = out_row0 + (size_t)(lane >> 3) * 2 * D + 8 * (lane & 7);
  const size_t pw_stride = (size_t)8 * 2 * D;
#pragma unroll
  for (int it = 0; it < 3; ++it)
    if (8 * it + (lane >> 3) < T) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<u32x4*>(pw_ptr + it * pw_stride + dt * 64) = pw[dt * 3 + it];
    }
}

__global__ __launch_bounds__(512) void k_qkv_sattn(QsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int tiles = a.mtiles * 8;
  if (b >= tiles) return;
  const int nitems = (tiles - b + G - 1) / G;
  const int vfull = (a.mtiles / 8) * 64, mrem = a.mtiles % 8;
  // tile ordinal -> (M-tile, head): all heads of an M-tile on one XCD, as the GEMM walks (kernels_gemm_x3p.hip)
  auto tile_of = [&](int o, int& mt, int& hd) {
    if (o < vfull) {
      const int xcd = o & 7, slot = o >> 3;
      mt = (slot >> 3) * 8 + xcd;
      hd = slot & 7;
    } else {
      const int o2 = o - vfull;
      mt = (a.mtiles / 8) * 8 + o2 % mrem;
      hd = o2 / mrem;
    }
  };

  const int K = a.K;
  const size_t K2 = 2 * (size_t)K;
  const int nk = K / 32;
  int mt = 0, hd = 0;
  tile_of(b, mt, hd);
  {   // first k-tile of the first tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lr = lane >> 3, csrc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
    const unsigned lofs = (unsigned)(lr * (int)K2 + csrc * 8) * 2u;
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(mt * QS_ROWS + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(hd * QS_BN + wave * 8) * K2 * 2;
    const size_t it_stride = (size_t)64 * K2 * 2;
#pragma unroll
    for (int it = 0; it < QS_AIT; ++it) QS_GLDS(sgpr_ptr(ubA + it * it_stride) + lofs, wave * 1024 + lane * 16 + it * 8192);
#pragma unroll
    for (int it = 0; it < QS_BIT; ++it) QS_GLDS(sgpr_ptr(ubB + it * it_stride) + lofs, QS_AREG + wave * 1024 + lane * 16 + it * 8192);
  }
  int tid_o = (int)threadIdx.x;
  unsigned long long dg[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long dg_r0 = a.diag ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define QS_STAMP(I)                                                         \
  if (a.diag) {                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
    dg[I] += now_ - dg_t;                                                   \
    dg_t = now_;                                                            \
  }
  for (int item = 0; item < nitems; ++item) {
    unsigned long long dg_t = a.diag ? __builtin_amdgcn_s_memtime() : 0ull;
    asm volatile("" : "+v"(tid_o));   // per-lane offsets are re-derived in every tile instead of being hoisted (and spilled)
    const int tid = tid_o;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, q = lane >> 4;
    const bool has_next = item + 1 < nitems;
    int mtn = 0, hdn = 0;
    if (has_next) tile_of((item + 1) * G + b, mtn, hdn);
    const int m0 = mt * QS_ROWS, n0 = hd * QS_BN;

    // ---- row statistics of the folded LayerNorm: raw partials by LDS-DMA under the k-loop (16-byte aligned blocks), else read later
    const int st_bytes = QS_BM * a.st_np * 8;
    const bool st_dma = st_bytes <= QS_RAW_MAX && (((size_t)m0 * a.st_np * 8) & 15) == 0;   // (uniform)
    int st_issued = 0;
    if (st_dma) {
      const char* src = reinterpret_cast<const char*>(a.st_in + (size_t)m0 * a.st_np * 2);
#pragma unroll
      for (int it = 0; it < QS_RAW_MAX / 1024 / 8; ++it) {
        const int pc = wave + it * 8;
        if (pc * 1024 < st_bytes) {
          QS_GLDS(sgpr_ptr(src + pc * 1024) + lane * 16, QS_RAW + pc * 1024);
          ++st_issued;
        }
      }
    }

    // ---- DMA plan (kernels_gemm_x3p.hip D3D_DMA_PLAN)
    const int lr_ = lane >> 3;
    const int csrc_ = (lane & 7) ^ (((wave & 1) << 2) | (lr_ >> 1));
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(m0 + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(n0 + wave * 8) * K2 * 2;
    unsigned lofs_ = (unsigned)(lr_ * (int)K2 + csrc_ * 8) * 2u;
    const size_t it_stride = (size_t)64 * K2 * 2;
    const int dstA = wave * 1024 + lane * 16, dstB = QS_AREG + wave * 1024 + lane * 16;
    // next tile's operand bases: its first k-tile is issued by the last two phases of this tile's k-loop (stage 0 is free then)
    const char* ubAn = reinterpret_cast<const char*>(a.Ap) + (size_t)(mtn * QS_ROWS + wave * 8) * K2 * 2;
    const char* ubBn = reinterpret_cast<const char*>(a.Wp) + (size_t)(hdn * QS_BN + wave * 8) * K2 * 2;
    // piece IT (A: 0..3, W: 4..6) of k-tile KTT of this tile, or (KTT == nk) of k-tile 0 of the next one
#define QS_PIECE(KTT, IT)                                                                                               \
    do {                                                                                                                \
      const bool nxt_ = (KTT) >= nk;                                                                                    \
      const int st_ = ((KTT) & 1) * QS_STAGE;                                                                           \
      if ((IT) < QS_AIT) {                                                                                              \
        const char* b_ = nxt_ ? ubAn + (IT) * it_stride : ubA + ((size_t)(KTT) * 128 + (IT) * it_stride);               \
        QS_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstA + (IT) * 8192);                                                        \
      } else {                                                                                                          \
        const char* b_ = nxt_ ? ubBn + ((IT) - QS_AIT) * it_stride : ubB + ((size_t)(KTT) * 128 + ((IT) - QS_AIT) * it_stride); \
        QS_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstB + ((IT) - QS_AIT) * 8192);                                             \
      }                                                                                                                 \
    } while (0)

    f32x4 acc[QS_TM][QS_NJ];
#pragma unroll
    for (int i = 0; i < QS_TM; ++i)
#pragma unroll
      for (int j = 0; j < QS_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    const int foff = (q ^ (r16 >> 1)) << 4;
    const int aoff = (wm * 128 + r16) * 128 + foff, boff = QS_AREG + (wn * 48 + r16) * 128 + foff;
    h8 bh[QS_NJ], bl[QS_NJ], ah[2], al[2];
    int issued_prev = st_issued;
    // the k-loop: qkv_fused_kloop.h (shared with the other fused kernel), this kernel's constants and DMA pieces behind the QF_ names
#define QF_STAGE QS_STAGE
#define QF_NJ QS_NJ
#define QF_TM QS_TM
#define QF_AIT QS_AIT
#define QF_BIT QS_BIT
#define QF_PIECE(KTT, IT) QS_PIECE(KTT, IT)
    QF_KLOOP_HEAD
    // ---- row statistics -> (rstd * out_scale, -mean rstd) per tile row, in front of the last k-tile: every wave reduces 32 rows (lanes
    // 0-31) in the shadow of its SIMD partner's MFMAs -- behind the k-loop this step was 1.1 us of a 38 us tile with half the waves
    // idle.  The raw partials landed long ago (the first counted wait of the tile retired them; phase barriers since).
    {
      float2* const srow = reinterpret_cast<float2*>(lds + QS_STX);
      if (lane < 32) {
        const int t = wave * 32 + lane, row = m0 + t;
        float sm = 0.f, sq = 0.f;
        if (row < a.M) {
          const float2* raw = st_dma ? reinterpret_cast<const float2*>(lds + QS_RAW) + t * a.st_np
                                     : reinterpret_cast<const float2*>(a.st_in) + (size_t)row * a.st_np;
          for (int p = 0; p < a.st_np; ++p) { sm += raw[p].x; sq += raw[p].y; }
        }
        if (sq >= (X3_HALF_MAX * 0.125f) * (X3_HALF_MAX * 0.125f)) range_raise(a.range, RANGE_BIT_ACT);   // (producer's planes, as x3q_tile)
        const float mean = sm / (float)K;
        const float var = fmaxf(sq / (float)K - mean * mean, 0.0f);
        if (row < a.M && mean * mean > 256.0f * var) range_raise(a.range, RANGE_BIT_STATS);
        const float rstd = 1.0f / sqrtf(var + a.eps);
        srow[t] = make_float2(rstd * a.out_scale, -mean * rstd);
      }
    }
    QF_KLOOP_TAIL
#undef QF_STAGE
#undef QF_NJ
#undef QF_TM
#undef QF_AIT
#undef QF_BIT
#undef QF_PIECE
#undef QS_PIECE
    __builtin_amdgcn_s_setprio(0);
    QS_STAMP(0);

    __syncthreads();   // statistics visible; every wave is out of the k-loop: stage 1 and the LDS behind it become the frame slots
    QS_STAMP(1);

    float2 st[QS_TM];
#pragma unroll
    for (int i = 0; i < QS_TM; ++i) st[i] = reinterpret_cast<const float2*>(lds + QS_STX)[wm * 128 + 16 * i + r16];
    float4 cs4[QS_NJ], b4[QS_NJ];
#pragma unroll
    for (int j = 0; j < QS_NJ; ++j) {
      const int n = n0 + wn * 48 + 16 * j + 4 * q;
      cs4[j] = *reinterpret_cast<const float4*>(a.csum + n);
      b4[j] = *reinterpret_cast<const float4*>(a.bias + n);
    }
    float amaxj[QS_NJ] = {0.0f, 0.0f, 0.0f};   // max |value| per accumulator column tile j = q / k / v (plane scale applied at the end)
    // q / k / v of this pass's frames -> frame slots (LayerNorm fold and hi / lo split of x3q_epilogue8).  Frame fr of the tile:
    // pass (fr >> 2) & 1, slot (fr & 3) + 4 (fr >> 3).  Column tile j of a wave IS part j (the weight rows are ordered that way at
    // commit): q columns 16 wn .. + 15 of the head in j = 0, the same k columns in j = 1, v in j = 2 -- plane and scale are
    // compile-time per j, and one address serves q and k (same row swizzle), one v.  Rows beyond the matrix hold finite values
    // (the engine zeroes the pad rows of the stream): nothing non-finite can reach a slot.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int chunk = 2 * wn + (q >> 1), half8 = (q & 1) << 3;            // this lane's 8 bytes: 16-byte chunk d / 8, half (d & 4)
    auto write_pass = [&](int pass) {
#pragma unroll
      for (int i = 0; i < QS_TM; ++i) {
        if (pass == 0 ? (i > 4) : (i < 4 && !(wm == 1 && i == 0))) continue;   // (wave-uniform: m-tiles without rows of this pass)
        const int R = wm * 128 + 16 * i + r16;
        const int fr = (R * 241) >> 12, jr = R - fr * QS_J;               // R / 17 for R < 256
        if (((fr >> 2) & 1) != pass || fr >= QS_FPT) continue;
        unsigned char* const row = lds + QS_QKV + ((fr & 3) + 4 * (fr >> 3)) * QS_SLOT + jr * 128 + half8;
        unsigned char* const pkq = row + ((chunk ^ ((jr >> 1) & 7)) << 4);
        unsigned char* const pv = row + ((chunk ^ vkey(jr)) << 4);
        const f2 sx = (f2)(st[i].x), sy = (f2)(st[i].y);
#pragma unroll
        for (int j = 0; j < QS_NJ; ++j) {
          const float osc = j == 0 ? 1.0f : 8.0f;
          f2 a01, a23, c01, c23, b01, b23;
          a01.x = acc[i][j][0]; a01.y = acc[i][j][1]; a23.x = acc[i][j][2]; a23.y = acc[i][j][3];
          c01.x = cs4[j].x; c01.y = cs4[j].y; c23.x = cs4[j].z; c23.y = cs4[j].w;
          b01.x = b4[j].x; b01.y = b4[j].y; b23.x = b4[j].z; b23.y = b4[j].w;
          const f2 v01 = __builtin_elementwise_fma(sx, a01, __builtin_elementwise_fma(sy, c01, b01));
          const f2 v23 = __builtin_elementwise_fma(sx, a23, __builtin_elementwise_fma(sy, c23, b23));
          amaxj[j] = fmaxf(fmaxf(amaxj[j], fabsf(v01.x)), fabsf(v01.y));
          amaxj[j] = fmaxf(fmaxf(amaxj[j], fabsf(v23.x)), fabsf(v23.y));
          unsigned h0, l0, h1, l1;
          split_pair(v01.x, v01.y, osc, h0, l0);
          split_pair(v23.x, v23.y, osc, h1, l1);
          unsigned char* const ph = j == 0 ? pkq + QS_PQ : (j == 1 ? pkq + QS_PK : pv + QS_PV);
          u32x2_alias hv, lv;
          hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
          if (!(QS_ABL & 1)) {
            *reinterpret_cast<u32x2_alias*>(ph) = hv;
            *reinterpret_cast<u32x2_alias*>(ph + QS_PLANE) = lv;
          }
        }
      }
    };
    auto attend = [&](int pass) {
      const int fr = (wave & 3) + 4 * pass + 8 * (wave >> 2);             // frame of the tile this wave takes: slot = wave
      const long long gf = (long long)mt * QS_FPT + fr;
      if (!(QS_ABL & 2) && fr < QS_FPT && gf < a.F)
        qs_attention(lds + QS_QKV + wave * QS_SLOT, lane, a.out + ((size_t)gf * QS_J) * 2 * a.D + hd * 128, a.D);
    };
    write_pass(0);
    __syncthreads();
    QS_STAMP(2);
    attend(0);
    __syncthreads();
    QS_STAMP(3);
    write_pass(1);
    {
      float amax = 0.0f;
#pragma unroll
      for (int j = 0; j < QS_NJ; ++j) amax = fmaxf(amax, amaxj[j] * (j == 0 ? 1.0f : (j == 1 ? 8.0f : 16.015f)));   // v: flagged from |v| > 4090 on
      if (amax > X3_HALF_MAX) range_raise(a.range, RANGE_BIT_ACT);
    }
    __syncthreads();
    QS_STAMP(4);
    attend(1);
    mt = mtn; hd = hdn;
    __syncthreads();   // the slots are read before the next tile's statistics block and second k-tile are staged over them
    QS_STAMP(5);
  }
#undef QS_STAMP
  if (a.diag && threadIdx.x == 0) {
    for (int i = 0; i < 6; ++i) a.diag[8 * b + i] = dg[i];
    a.diag[8 * b + 6] = (unsigned long long)nitems;
    a.diag[8 * b + 7] = __builtin_amdgcn_s_memrealtime() - dg_r0;
  }
}

}  // namespace

static std::atomic<int> g_qs_diag{0};
void set_qkv_sattn_diag(int on) { g_qs_diag = on; }

bool qkv_sattn_ok(int J, int D, int H, int K) { return J == QS_J && H == 8 && D == 512 && K % 64 == 0 && K >= 128; }

// Tokens M = frames * 17; A / st_in must span 255 * ceil(frames / 15) + 1 rows (the engine's workspace does).
hipError_t launch_qkv_sattn(const void* Apair, const void* Wpair_headmajor, const float* bias_hm, const float* csum_hm, const float* st_in,
                            int st_np, float eps, int w_exp, void* out_x3, int M, int K, int J, int D, int H, hipStream_t s) {
  if (!qkv_sattn_ok(J, D, H, K) || M <= 0 || M % J != 0 || st_np < 1 || !Apair || !Wpair_headmajor || !bias_hm || !csum_hm || !st_in || !out_x3)
    return hipErrorInvalidValue;
  if (w_exp < -14 || w_exp > 12) return hipErrorInvalidValue;
  QsArgs a{};
  a.Ap = (const _Float16*)Apair; a.Wp = (const _Float16*)Wpair_headmajor; a.bias = bias_hm; a.csum = csum_hm; a.st_in = st_in;
  a.st_np = st_np; a.eps = eps; a.out_scale = ldexpf(1.0f, -(3 + w_exp));
  a.out = (_Float16*)out_x3; a.M = M; a.K = K; a.F = M / J; a.mtiles = (a.F + QS_FPT - 1) / QS_FPT; a.D = D;
  a.range = launch_range_word();
  static std::atomic<unsigned long long> attr_done{0};   // one bit per device
  if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(k_qkv_sattn), QS_LDS, attr_done)) return ae;
  int n_cu = device_cu_count();
  if (n_cu <= 0) return hipErrorUnknown;
  const int tiles = a.mtiles * 8;
  const int grid = tiles < n_cu ? tiles : n_cu;
  hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;   // the stamp report allocates, synchronises and copies: illegal inside a
  (void)hipStreamIsCapturing(s, &cap_st);                          // hipGraph capture (d3d_engine_set_graph_mode) -- plain launch there
  if (g_qs_diag.load() > 0 && cap_st == hipStreamCaptureStatusNone) {   // "qs_diag" option: every 50th launch with stamps, summarised on stderr (synchronises the stream)
    static std::atomic<int> count{0};
    if (count.fetch_add(1) % 50 == 10) {
      unsigned long long* buf = nullptr;
      if (hipMalloc(&buf, (size_t)grid * 64) != hipSuccess) return hipErrorOutOfMemory;
      (void)hipMemsetAsync(buf, 0, (size_t)grid * 64, s);
      a.diag = buf;
      hipLaunchKernelGGL(k_qkv_sattn, dim3(grid), dim3(512), QS_LDS, s, a);
      (void)hipStreamSynchronize(s);
      std::vector<unsigned long long> h((size_t)grid * 8);
      (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
      (void)hipFree(buf);
      double sum[6] = {0}, tiles_n = 0, cyc = 0, ticks = 0;
      for (int g = 0; g < grid; ++g) {
        for (int i = 0; i < 6; ++i) { sum[i] += (double)h[8 * g + i]; cyc += (double)h[8 * g + i]; }
        tiles_n += (double)h[8 * g + 6]; ticks += (double)h[8 * g + 7];
      }
      const double ghz = ticks > 0 ? cyc / (ticks * 10.0) : 0.0;   // cycles per ns (100 MHz ticks = 10 ns)
      fprintf(stderr, "[qs diag] M=%d tiles %d on %d workgroups, clock %.2f GHz; per tile (us, wave 0): k-loop %.2f  stats %.2f  write0 %.2f  "
              "attn0 %.2f  write1+prefetch %.2f  attn1 %.2f  | total %.2f\n", M, tiles, grid, ghz,
              sum[0] / tiles_n / ghz / 1e3, sum[1] / tiles_n / ghz / 1e3, sum[2] / tiles_n / ghz / 1e3, sum[3] / tiles_n / ghz / 1e3,
              sum[4] / tiles_n / ghz / 1e3, sum[5] / tiles_n / ghz / 1e3, cyc / tiles_n / ghz / 1e3);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(k_qkv_sattn, dim3(grid), dim3(512), QS_LDS, s, a);
  return hipGetLastError();
}

}  // namespace d3d
