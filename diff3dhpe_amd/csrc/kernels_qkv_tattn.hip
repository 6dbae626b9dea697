// Temporal blocks of the F16X3 flow, fused: the LayerNorm-folded qkv GEMM of one (batch, joint) group with the T-key GRAND attention of
// its frames run from LDS (S2S:67 + 73-83 for the per-joint groups of S2S:125).  The temporal counterpart of kernels_qkv_sattn.hip; what it
// removes from the unfused flow (k_linear_x3q_persist<qkv form> + k_attn_temporal_x3s): the q / k / v planes never exist in HBM -- 1.62 GB
// written and 1.62 GB read back per launch pair at the bench shape -- and one kernel launch per temporal block.
//
// Tile = the T (193 ... 255) frames of ONE joint of one batch element x ONE head's q, k, v: 256 token rows (row t = token (b T + t) J + j:
// the DMA walks the rows by stride, pad rows repeat the last frame) x 192 output columns of the tile-ordered folded weight (as the spatial
// kernel: accumulator column tile j of every wave is q / k / v).  Eight waves (2 x 4), 128 rows x 48 columns each; the k-loop is the
// two-phase persistent loop of kernels_qkv_sattn.hip on the same 256 x 192 x 32 stage.  Per output element the same MFMAs in the same
// order as every other F16X3 GEMM shape and the epilogue arithmetic of x3q_epilogue8<LN-folded, planes>; the attention is the arithmetic
// of k_attn_temporal_x3s (same MFMAs in the same order, same softmax, same conversions): the block is bit-identical to the unfused flow.
//
// After the k-loop the LDS is re-cut to the byte: K hi | K lo | V hi | V lo planes of 256 rows x 128 B (128 KiB; the swizzles of
// kernels_attn_x3.hip) + 32 KiB through which the queries are exchanged -- a wave of the GEMM holds 16 of the 64 head dims of 128 rows,
// a wave of the attention needs all 64 dims of 32 rows -- in two halves of 128 rows, and which then serve as the eight 4 KiB output
// patches.  No step barriers inside the attention (K and V are static, unlike the restaged planes of the stand-alone kernel): the
// two waves of a SIMD drift through their MFMA and VALU steps out of phase by themselves (the second half starts an exchange later); the
// halves meet at two LDS counters in a pad row of the V lo plane (key 255: T <= 255), not at workgroup barriers, which held the first
// half at the second half's pace.  The next tile's first k-tile goes into the dead K planes, every wave's pieces issued by the first
// half after its outputs (it idles there until the second half is through).  In-kernel stamps: "qt_diag" option.
//
// T <= 127 (the T = 81 / 27 configurations): the GROUPED form k_qkv_tattn<true> -- the frames of G = 255 / T joints of ONE batch element per
// tile (tile row R = frame R % T of joint jt G + R / T), the same k-loop / plane writes / exchange; a query sees the keys of its own joint
// (masks in the softmax) and a wave computes only the key tiles that hold keys of its queries' joints.  As accurate as the two-kernel flow,
// not bit-identical to it (the softmax sum and the key-tile products group the keys by tile position); which joints share a tile depends on
// the joint index alone, so a batch element's arithmetic does not depend on its position in the batch.
#include "d3d_kernels.h"
#include "qkv_fused_kloop.h"

#include <math.h>
#include <stdio.h>
#include <utility>
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

constexpr int QT_BM = 256, QT_BN = 192, QT_TM = 8, QT_NJ = 3;
constexpr int QT_AREG = QT_BM * 128, QT_STAGE = (QT_BM + QT_BN) * 128;   // 57344
constexpr int QT_AIT = 4, QT_BIT = 3;                                    // 1-KiB DMA pieces per wave per k-tile
constexpr int QT_PLANE = 256 * 128;                                      // one fp16 plane of 256 key rows x 64 dims
constexpr int QT_K = 0, QT_V = 2 * QT_PLANE, QT_Q = 4 * QT_PLANE;        // K hi | K lo | V hi | V lo | Q exchange, then output patches (32 KiB)
constexpr int QT_RAW = 2 * QT_STAGE;                                     // raw statistics partials while the k-loop runs (16 KiB)
constexpr int QT_RAW_MAX = 16384;
constexpr int QT_STX = QT_Q + 30 * 1024;                                 // (rstd', -mean rstd) of the tile's 256 rows, 2 KiB, until the epilogue has read them
constexpr int QT_LDS = QT_Q + 32 * 1024;                                 // 163840
static_assert(QT_LDS <= 160 * 1024 && QT_RAW + QT_RAW_MAX <= QT_Q && QT_STX + QT_BM * 8 <= QT_LDS, "LDS map");

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

struct QtArgs {
  const _Float16* Ap;      // residual stream, pair layout [M rows][2 K] of 8 x
  const _Float16* Wp;      // folded qkv weight W diag(gamma), pair layout, 2^k w, rows in TILE order (kernels_qkv_sattn.hip)
  const float* bias;       // b + W beta, tile order
  const float* csum;       // sum_k W[n, k] gamma[k], tile order
  const float* st_in;      // (sum, sum of squares) partials of the rows: [rows][st_np][2]
  int st_np;
  float eps, out_scale;    // LayerNorm eps; 2^-(3 + k)
  _Float16* out;           // attention output, pair layout [M][2 D] of 8 o
  int M, K, T, J, BJ, D;   // tokens, GEMM depth, frames per group, joints, tiles per head (B J; grouped form: B TPS), model width (8 heads x 64)
  int G, TPS, Tinv;        // grouped form (k_qkv_tattn<true>, T <= 127): G = 255 / T joints' frames per tile, TPS = ceil(J / G) tiles per batch
                           // element, Tinv = 65536 / T + 1: R / T = (R Tinv) >> 16 for every tile row R < 256 (the fraction of R / T is at most
                           // 126 / 127, the product's excess below 256 / 65536)
  unsigned* range;         // the engine's range-guard word
  unsigned long long* diag;   // diagnostic launches only ("qs_diag"): per workgroup 8 words -- cycles of wave 0 in the k-loop, the statistics
                              // step, the plane writes, the query exchange, scores + softmax, the rest of the attention, tiles, 100 MHz ticks
};

#define QT_GLDS(SRC, DSTOFF)                                                                                            \
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


// LDS fragment reads as inline asm with hand-placed counted waits (kernels_attn_x3.hip: the compiler's own waits came out as
// lgkmcnt(0) behind every pair of reads)
template <int OFF>
__device__ __forceinline__ void lds_read_b128(h8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr16_b64(s4v& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int... Js, class F>
__device__ __forceinline__ void static_for_(std::integer_sequence<int, Js...>, F&& f) { (f(std::integral_constant<int, Js>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_(std::make_integer_sequence<int, N>{}, f); }
__device__ __forceinline__ void split_pair_v(float e0, float e1, float k, unsigned& hi, unsigned& lo) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(e0), "v"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(e1), "v"(k));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(e0), "v"(k), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(e1), "v"(k), "v"(hi));
}

constexpr int QT_NKT = 8;
#ifndef QT_PB1
#define QT_PB1 1   // the second half's steps run one priority level above the first half's: it is the one that is behind (-0.7 % per launch)
#endif
// Scores (qt_scores) and softmax (qt_softmax) of one wave's 32 queries against the T keys in the K planes: the score and softmax steps
// of k_attn_temporal_x3s (same MFMAs in the same order, same arithmetic).  After qt_softmax sacc[kt] holds the packed hi (slots 0-7) /
// lo (8-15) halves of 1024 e, l the row sum.
// Diagnostic builds only (-DQT_ABL=n, wrong results; experiments/lds_conflict_attribution.sh): 1 no plane / exchange writes, 2 no score
// step (K fragment reads), 4 no V fragment reads in the product step, 8 no output patches, 16 no query fragment reads.
#ifndef QT_ABL
#define QT_ABL 0
#endif

template <int PB, bool GRP = false>
__device__ __forceinline__ void qt_scores(unsigned char* const lds, int lane, const h8 (&qh)[4], const h8 (&ql)[4], f32x16 (&sacc)[QT_NKT],
                                          int kt_lo = 0, int kt_hi = QT_NKT - 1) {
  constexpr int NKT = QT_NKT, PLANE = QT_PLANE;
  const int r = lane & 31, h = lane >> 5;
  unsigned char* const sKh = lds + QT_K;
  __builtin_amdgcn_s_setprio(2 + PB);
  if constexpr (GRP) {   // grouped form: only the key tiles [kt_lo, kt_hi] (wave-uniform) that hold keys of this wave's groups
    unsigned kaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = (unsigned)(uintptr_t)(sKh + kswz(r, 2 * ks + h));
    static_for<NKT>([&](auto ktc) {
      constexpr int kt = decltype(ktc)::value;
      if (kt >= kt_lo && kt <= kt_hi) {
        h8 kfh[3], kfl[3];
        lds_read_b128<kt * 4096>(kfh[0], kaddr[0]); lds_read_b128<kt * 4096 + PLANE>(kfl[0], kaddr[0]);
        lds_read_b128<kt * 4096>(kfh[1], kaddr[1]); lds_read_b128<kt * 4096 + PLANE>(kfl[1], kaddr[1]);
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc[kt][q] = 0.f;
        static_for<4>([&](auto ksc) {
          constexpr int ks = decltype(ksc)::value;
          if constexpr (ks + 2 < 4) {
            lds_read_b128<kt * 4096>(kfh[(ks + 2) % 3], kaddr[ks + 2]);
            lds_read_b128<kt * 4096 + PLANE>(kfl[(ks + 2) % 3], kaddr[ks + 2]);
          }
          lgkm_wait<(ks + 2 < 4) ? 4 : (ks + 1 < 4 ? 2 : 0)>();
          __builtin_amdgcn_sched_barrier(0);
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[ks % 3], qh[ks], sacc[kt], 0, 0, 0);
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[ks % 3], ql[ks], sacc[kt], 0, 0, 0);
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[ks % 3], qh[ks], sacc[kt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc[kt][q] = 0.f;
      }
    });
  } else {
    h8 kfh[3], kfl[3];
    unsigned kaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = (unsigned)(uintptr_t)(sKh + kswz(r, 2 * ks + h));   // + 4096 kt; lo plane + PLANE
    static_assert(PLANE + (NKT - 1) * 4096 < 65536, "ds offset field");
    lds_read_b128<0>(kfh[0], kaddr[0]); lds_read_b128<PLANE>(kfl[0], kaddr[0]);
    lds_read_b128<0>(kfh[1], kaddr[1]); lds_read_b128<PLANE>(kfl[1], kaddr[1]);
    static_for<4 * NKT>([&](auto jc) {
      constexpr int j = decltype(jc)::value, kt = j >> 2, ks = j & 3, jn = j + 2;
      if constexpr (jn < 4 * NKT) {
        lds_read_b128<(jn >> 2) * 4096>(kfh[jn % 3], kaddr[jn & 3]);
        lds_read_b128<(jn >> 2) * 4096 + PLANE>(kfl[jn % 3], kaddr[jn & 3]);
      }
      lgkm_wait<(jn < 4 * NKT) ? 4 : (j + 1 < 4 * NKT ? 2 : 0)>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc[kt][q] = 0.f;
      }
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[j % 3], qh[ks], sacc[kt], 0, 0, 0);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[j % 3], ql[ks], sacc[kt], 0, 0, 0);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[j % 3], qh[ks], sacc[kt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  __builtin_amdgcn_s_setprio(PB);
}
// GRP (grouped form): this lane's query sees the keys [klo, khi) of its own group only, all inside the key tiles [kt_lo, kt_hi] (wave-uniform:
// the tiles the score step computed; the others are never touched)
template <int PB, bool GRP = false>
__device__ __forceinline__ void qt_softmax(int lane, int T, f32x16 (&sacc)[QT_NKT], float& l, int kt_lo = 0, int kt_hi = QT_NKT - 1, int klo = 0,
                                           int khi = 0) {
  constexpr int NKT = QT_NKT;
  const int h = lane >> 5;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (GRP && (kt < kt_lo || kt > kt_hi)) continue;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (GRP) {
        const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (key < klo || key >= khi) sacc[kt][q] = -INFINITY;
      } else if (kt >= 6) {   // (T > 192: only the last two key tiles can hold pad keys)
        const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (key >= T) sacc[kt][q] = -INFINITY;
      }
      m = fmaxf(m, sacc[kt][q]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  constexpr float C_EXP = 1.4426950408889634f / 64.0f;
  const float mb = m * C_EXP;
  l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (GRP && (kt < kt_lo || kt > kt_hi)) continue;
    float e[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      e[q] = __builtin_amdgcn_exp2f(fmaf(sacc[kt][q], C_EXP, -mb));
      l += e[q];
    }
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      unsigned hp, lp;
      split_pair_s(e[2 * pr], e[2 * pr + 1], 1024.0f, hp, lp);
      sacc[kt][pr] = __builtin_bit_cast(float, hp);
      sacc[kt][8 + pr] = __builtin_bit_cast(float, lp);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  l += __shfl_xor(l, 32, 64);
}

// Products, output arithmetic and stores of one wave's 32 queries: the product and output steps of k_attn_temporal_x3s.
// GRP (grouped form): key tiles [kt_lo, kt_hi] only; TV = G T valid tile rows; the output row of tile row R is out_unit + orow(R) (a lambda of
// the kernel: the row's token), stored where orow(R) >= 0
template <int PB, bool GRP = false, class ORow = int>
__device__ __forceinline__ void qt_products_outputs(unsigned char* const lds, int wave, int lane, int T, int J, int D, const f32x16 (&sacc)[QT_NKT],
                                                    float l, _Float16* out_unit, unsigned* rw, int kt_lo = 0, int kt_hi = QT_NKT - 1,
                                                    ORow orow = 0) {
  constexpr int NKT = QT_NKT, PLANE = QT_PLANE;
  const int r = lane & 31, h = lane >> 5;
  unsigned char* const sVh = lds + QT_V;
  unsigned char* const sVl = lds + QT_V + PLANE;
  unsigned char* const patch = lds + QT_Q + wave * 4096;
  const int tq = 32 * wave + r;
  __builtin_amdgcn_s_setprio(1 + PB);
  f32x16 oacc[2];
#pragma unroll
  for (int q = 0; q < 16; ++q) { oacc[0][q] = 0.f; oacc[1][q] = 0.f; }
  {
    const int gi = lane & 15, tq_ = gi >> 2, tp_ = gi & 3;
    unsigned vaddr[4];     // [dt][row half]: + 2048 per step; lo plane + PLANE
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int d0 = dt * 32 + 16 * ((lane >> 4) & 1);
      const int ch = (d0 >> 3) + (tp_ >> 1), sub = (tp_ & 1) * 8;
      vaddr[2 * dt] = (unsigned)(uintptr_t)(sVh + vswz(4 * h + tq_, ch) + sub);
      vaddr[2 * dt + 1] = (unsigned)(uintptr_t)(sVh + vswz(4 * h + 8 + tq_, ch) + sub);
    }
    s4v vf[2][8];
    auto vread = [&](auto jc, s4v(&f)[8]) {
      constexpr int off = decltype(jc)::value * 2048;
      static_assert(off + PLANE < 65536, "ds offset field");
      if constexpr ((QT_ABL & 4) != 0) return;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        lds_read_tr16_b64<off>(f[4 * dt], vaddr[2 * dt]);
        lds_read_tr16_b64<off>(f[4 * dt + 1], vaddr[2 * dt + 1]);
        lds_read_tr16_b64<off + PLANE>(f[4 * dt + 2], vaddr[2 * dt]);
        lds_read_tr16_b64<off + PLANE>(f[4 * dt + 3], vaddr[2 * dt + 1]);
      }
    };
    if constexpr (!GRP) vread(std::integral_constant<int, 0>{}, vf[0]);
    static_for<2 * NKT>([&](auto jc) {
      constexpr int j = decltype(jc)::value, kt = j >> 1, s2 = j & 1;
      if (GRP && (kt < kt_lo || kt > kt_hi)) return;
      if constexpr (GRP) {   // (no read ahead across key tiles: which one follows is a run-time matter; both steps of a tile at once)
        if constexpr (s2 == 0) { vread(jc, vf[0]); vread(std::integral_constant<int, j + 1>{}, vf[1]); }
      } else {
        if constexpr (j + 1 < 2 * NKT) vread(std::integral_constant<int, j + 1>{}, vf[(j + 1) & 1]);
      }
      typedef float f32x4_ __attribute__((ext_vector_type(4)));
      const h8 eh = __builtin_bit_cast(h8, (f32x4_)__builtin_shufflevector(sacc[kt], sacc[kt], 4 * s2, 4 * s2 + 1, 4 * s2 + 2, 4 * s2 + 3));
      const h8 el = __builtin_bit_cast(h8, (f32x4_)__builtin_shufflevector(sacc[kt], sacc[kt], 8 + 4 * s2, 9 + 4 * s2, 10 + 4 * s2, 11 + 4 * s2));
      lgkm_wait<(GRP ? s2 == 0 : j + 1 < 2 * NKT) ? 8 : 0>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        h8 vh, vl;
        const h4 a0h = __builtin_bit_cast(h4, vf[j & 1][4 * dt]), a1h = __builtin_bit_cast(h4, vf[j & 1][4 * dt + 1]);
        const h4 c0h = __builtin_bit_cast(h4, vf[j & 1][4 * dt + 2]), c1h = __builtin_bit_cast(h4, vf[j & 1][4 * dt + 3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vh[e] = a0h[e]; vh[4 + e] = a1h[e]; vl[e] = c0h[e]; vl[4 + e] = c1h[e]; }
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, eh, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, el, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, eh, oacc[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  __builtin_amdgcn_s_setprio(PB);
  // v_query (this wave's own rows of V)
  h4 vqh[8], vql[8];
  {
    const int tqc = (GRP || tq < T) ? tq : 0;   // (grouped form: every plane row holds finite values)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int vo = vswz(tqc, c) + 8 * h;
      vqh[c] = *reinterpret_cast<const h4*>(sVh + vo);
      vql[c] = *reinterpret_cast<const h4*>(sVl + vo);
    }
  }
  // O = O^T / (2^13 l) - v_query, packed as hi / lo of 8 o (the output step of k_attn_temporal_x3s), whole lines out through the patch
  u32x4 po[8];
  {
    const float inv8 = 8.0f * (1.0f / (8192.0f * l));
    float amax = 0.0f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const uint2 ph2 = __builtin_bit_cast(uint2, vqh[dt * 4 + g4]), pl2 = __builtin_bit_cast(uint2, vql[dt * 4 + g4]);
        float o8[4];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const unsigned ph_ = pr ? ph2.y : ph2.x, pl_ = pr ? pl2.y : pl2.x;
          float n0, n1;
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(n0) : "v"(pl_), "v"(-1.0f));
          asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(n0) : "v"(ph_), "v"(-1.0f), "v"(n0));
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(n1) : "v"(pl_), "v"(-1.0f));
          asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(n1) : "v"(ph_), "v"(-1.0f), "v"(n1));
          o8[2 * pr] = __builtin_fmaf(oacc[dt][4 * g4 + 2 * pr], inv8, n0);
          o8[2 * pr + 1] = __builtin_fmaf(oacc[dt][4 * g4 + 2 * pr + 1], inv8, n1);
        }
        amax = fmaxf(fmaxf(amax, fabsf(o8[0])), fabsf(o8[1]));
        amax = fmaxf(fmaxf(amax, fabsf(o8[2])), fabsf(o8[3]));
        unsigned h0, l0, h1, l1;
        split_pair_v(o8[0], o8[1], 1.0f, h0, l0);
        split_pair_v(o8[2], o8[3], 1.0f, h1, l1);
        const h4 oh = __builtin_bit_cast(h4, make_uint2(h0, h1)), ol = __builtin_bit_cast(h4, make_uint2(l0, l1));
        if (!(QT_ABL & 8)) patch_wr(patch, r, h, g4, oh, ol);
        else asm volatile("" ::"v"(oh), "v"(ol));
      }
      asm volatile("" ::: "memory");     // (the rows read back were written by other lanes)
#pragma unroll
      for (int it = 0; it < 4; ++it) po[dt * 4 + it] = (QT_ABL & 8) ? u32x4{0u, 0u, 0u, 0u} : patch_rd(patch, 8 * it + (lane >> 3), lane & 7);
      asm volatile("" ::: "memory");
    }
    if constexpr (GRP) {
      if (orow(tq) >= 0 && amax > X3_HALF_MAX) range_raise(rw, RANGE_BIT_ACT);
    } else {
      if (tq < T && amax > X3_HALF_MAX) range_raise(rw, RANGE_BIT_ACT);
    }
  }
  if constexpr (GRP) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int tokr = orow(32 * wave + 8 * it + (lane >> 3));
      if (tokr >= 0) {
        _Float16* const pp = out_unit + (size_t)tokr * 2 * D + 8 * (lane & 7);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<u32x4*>(pp + dt * 64) = po[dt * 4 + it];
      }
    }
  } else {
    _Float16* const po_ptr = out_unit + (size_t)(32 * wave + (lane >> 3)) * J * 2 * D + 8 * (lane & 7);
    const size_t po_stride = (size_t)8 * J * 2 * D;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (32 * wave + 8 * it + (lane >> 3) < T) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<u32x4*>(po_ptr + it * po_stride + dt * 64) = po[dt * 4 + it];
      }
  }
}

// GRP = false: one (batch, joint) group of 193 ... 255 frames per tile.  GRP = true (T <= 127): the frames of G = 255 / T joints of ONE batch
// element per tile -- tile row R = frame R % T of joint jt G + R / T --; a query sees the keys of its own joint only (masks in the softmax,
// and only the key tiles that hold them are computed).  Which joints share a tile depends on the joint index alone: a batch element's
// arithmetic does not depend on its position in the batch.
template <bool GRP>
__global__ __launch_bounds__(512) void k_qkv_tattn(QtArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int tiles = a.BJ * 8;
  if (b >= tiles) return;
  const int nitems = (tiles - b + G - 1) / G;
  const int vfull = (a.BJ / 8) * 64, mrem = a.BJ % 8;
  // tile ordinal -> (group, head): all heads of a group on one XCD, as the GEMM walks (kernels_gemm_x3p.hip)
  auto tile_of = [&](int o, int& bj, int& hd) {
    if (o < vfull) {
      const int xcd = o & 7, slot = o >> 3;
      bj = (slot >> 3) * 8 + xcd;
      hd = slot & 7;
    } else {
      const int o2 = o - vfull;
      bj = (a.BJ / 8) * 8 + o2 % mrem;
      hd = o2 / mrem;
    }
  };
  const int K = a.K, T = a.T, J = a.J;
  const size_t K2 = 2 * (size_t)K;
  const int nk = K / 32;
  int bj = 0, hd = 0;
  tile_of(b, bj, hd);
  // grouped form: token row (from the batch element's first token) of tile row R of joint tile jt, -1 for the pad rows of the tile and for
  // joints beyond J in an element's last tile
  auto grow = [&](int R, int jt) -> int {
    const int g = (R * a.Tinv) >> 16, t = R - g * T, j = jt * a.G + g;
    return (g < a.G && j < J) ? t * J + j : -1;
  };
  // first k-tile of a tile: wave w moves pieces w, w + 8, ... (8 rows x 128 B each) of A, then of W; tile row t is token tok0 + min(t, T - 1) J
  auto stage_first = [&](int bj_, int hd_, int wave) {   // (the pieces of `wave`: the issuing wave may move another wave's share as well)
    const int lane = threadIdx.x & 63;
    const int lr = lane >> 3, csrc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
    const size_t tok0_ = GRP ? (size_t)(bj_ / a.TPS) * T * J : (size_t)(bj_ / J) * T * J + (size_t)(bj_ % J);
    const char* tA = reinterpret_cast<const char*>(a.Ap) + tok0_ * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(hd_ * QT_BN + wave * 8) * K2 * 2;
    const unsigned lofsW = (unsigned)(lr * (int)K2 + csrc * 8) * 2u;
#pragma unroll
    for (int it = 0; it < QT_AIT; ++it) {
      const int row = wave * 8 + 64 * it + lr;
      unsigned lo;
      if constexpr (GRP) {
        const int jt_ = bj_ % a.TPS, tr = grow(row, jt_);
        lo = (unsigned)(((size_t)(tr >= 0 ? tr : jt_ * a.G) * K2 + csrc * 8) * 2);   // (pad rows repeat the tile's first token: finite values)
      } else {
        lo = (unsigned)(((size_t)(row < T ? row : T - 1) * J * K2 + csrc * 8) * 2);
      }
      QT_GLDS(sgpr_ptr(tA) + lo, wave * 1024 + lane * 16 + it * 8192);
    }
    const size_t it_stride = (size_t)64 * K2 * 2;
#pragma unroll
    for (int it = 0; it < QT_BIT; ++it) QT_GLDS(sgpr_ptr(ubB + it * it_stride) + lofsW, QT_AREG + wave * 1024 + lane * 16 + it * 8192);
  };
  stage_first(bj, hd, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));
  int tid_o = (int)threadIdx.x;
  unsigned long long dg[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long dg_r0 = a.diag ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define QT_STAMP(I)                                                         \
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
    const bool more = item + 1 < nitems;
    constexpr bool has_next = false;   // (the k-loop stages nothing of the next tile: the planes need the whole LDS)
    int bjn = 0, hdn = 0;
    if (more) tile_of((item + 1) * G + b, bjn, hdn);
    const size_t tok0 = GRP ? (size_t)(bj / a.TPS) * T * J : (size_t)(bj / J) * T * J + (size_t)(bj % J);   // GRP: the batch element's first token
    const int jt = GRP ? bj % a.TPS : 0;
    const int n0 = hd * QT_BN;

    // ---- row statistics of the folded LayerNorm: raw partials gathered by LDS-DMA under the k-loop -- lane l of piece pc serves row
    // 16 pc + l / 4, 16-byte chunk l % 4 of its 64 bytes (8 partials) --, else read at the reduction
    const bool st_dma = a.st_np == 8;   // (uniform)
    int st_issued = 0;
    if (st_dma) {
      const char* src = reinterpret_cast<const char*>(a.st_in + tok0 * 16);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int pc = wave + it * 8, row = 16 * pc + (lane >> 2);
        unsigned lo;
        if constexpr (GRP) {
          const int tr = grow(row, jt);
          lo = (unsigned)((size_t)(tr >= 0 ? tr : jt * a.G) * 64 + (lane & 3) * 16);
        } else {
          lo = (unsigned)((size_t)(row < T ? row : T - 1) * J * 64 + (lane & 3) * 16);
        }
        QT_GLDS(sgpr_ptr(src) + lo, QT_RAW + pc * 1024 + lane * 16);
        ++st_issued;
      }
    }

    // ---- DMA plan: A rows by stride (pieces 0-2 of a wave never reach the pad rows: T > 192), W rows contiguous
    const int lr_ = lane >> 3;
    const int csrc_ = (lane & 7) ^ (((wave & 1) << 2) | (lr_ >> 1));
    const char* tA = reinterpret_cast<const char*>(a.Ap) + tok0 * K2 * 2;                       // row 0 of the tile
    const char* ubA = tA + (size_t)(wave * 8) * J * K2 * 2;                                     // this wave's first row
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(n0 + wave * 8) * K2 * 2;
    unsigned lofs_ = (unsigned)(lr_ * (int)K2 + csrc_ * 8) * 2u;                                // W pieces
    unsigned lofsA_ = (unsigned)(((size_t)lr_ * J * K2 + csrc_ * 8) * 2);                       // A pieces 0-2, from the piece's first row
    unsigned lofsA3_;                                                                           // A piece 3, from row 0 of the tile (clamped)
    {
      const int row = wave * 8 + 192 + lr_;
      lofsA3_ = (unsigned)(((size_t)(row < T ? row : T - 1) * J * K2 + csrc_ * 8) * 2);
    }
    unsigned lofsAg_[QT_AIT] = {0u, 0u, 0u, 0u};                                                // grouped form: A piece it, from the element's first token
    if constexpr (GRP) {
#pragma unroll
      for (int it = 0; it < QT_AIT; ++it) {
        const int tr = grow(wave * 8 + 64 * it + lr_, jt);
        lofsAg_[it] = (unsigned)(((size_t)(tr >= 0 ? tr : jt * a.G) * K2 + csrc_ * 8) * 2);
      }
    }
    const size_t it_stride = (size_t)64 * K2 * 2, it_strideA = (size_t)64 * J * K2 * 2;
    const int dstA = wave * 1024 + lane * 16, dstB = QT_AREG + wave * 1024 + lane * 16;
    // piece IT (A: 0..3, W: 4..6) of k-tile KTT of this tile
#define QT_PIECE(KTT, IT)                                                                                               \
    do {                                                                                                                \
      const int st_ = ((KTT) & 1) * QT_STAGE;                                                                           \
      if (GRP && (IT) < QT_AIT) {                                                                                       \
        QT_GLDS(sgpr_ptr(tA + (size_t)(KTT) * 128) + lofsAg_[(IT) < QT_AIT ? (IT) : 0], st_ + dstA + (IT) * 8192);      \
      } else if ((IT) < 3) {                                                                                            \
        QT_GLDS(sgpr_ptr(ubA + ((size_t)(KTT) * 128 + (IT) * it_strideA)) + lofsA_, st_ + dstA + (IT) * 8192);          \
      } else if ((IT) == 3) {                                                                                           \
        QT_GLDS(sgpr_ptr(tA + (size_t)(KTT) * 128) + lofsA3_, st_ + dstA + 3 * 8192);                                   \
      } else {                                                                                                          \
        QT_GLDS(sgpr_ptr(ubB + ((size_t)(KTT) * 128 + ((IT) - QT_AIT) * it_stride)) + lofs_, st_ + dstB + ((IT) - QT_AIT) * 8192); \
      }                                                                                                                 \
    } while (0)

    f32x4 acc[QT_TM][QT_NJ];
#pragma unroll
    for (int i = 0; i < QT_TM; ++i)
#pragma unroll
      for (int j = 0; j < QT_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    const int foff = (q ^ (r16 >> 1)) << 4;
    const int aoff = (wm * 128 + r16) * 128 + foff, boff = QT_AREG + (wn * 48 + r16) * 128 + foff;
    h8 bh[QT_NJ], bl[QT_NJ], ah[2], al[2];
    int issued_prev = st_issued;
    // the k-loop: qkv_fused_kloop.h (shared with the other fused kernel), this kernel's constants and DMA pieces behind the QF_ names
#define QF_STAGE QT_STAGE
#define QF_NJ QT_NJ
#define QF_TM QT_TM
#define QF_AIT QT_AIT
#define QF_BIT QT_BIT
#define QF_PIECE(KTT, IT) QT_PIECE(KTT, IT)
    QF_KLOOP_HEAD
    // ---- row statistics -> (rstd * out_scale, -mean rstd) per tile row, in front of the last k-tile: every wave reduces 32 rows (lanes
    // 0-31) in the shadow of its SIMD partner's MFMAs -- behind the k-loop this step was 1.1 us of a 38 us tile with half the waves
    // idle.  The raw partials landed long ago (the first counted wait of the tile retired them; phase barriers since).
    {
      float2* const srow = reinterpret_cast<float2*>(lds + QT_STX);
      if (lane < 32) {
        const int t = wave * 32 + lane;
        size_t row;
        if constexpr (GRP) {
          const int tr = grow(t, jt);
          row = tok0 + (size_t)(tr >= 0 ? tr : jt * a.G);
        } else {
          row = tok0 + (size_t)(t < T ? t : T - 1) * J;          // (pad rows of the tile repeat the last frame: finite values)
        }
        float sm = 0.f, sq = 0.f;
        {
          const float2* raw = st_dma ? reinterpret_cast<const float2*>(lds + QT_RAW) + t * a.st_np
                                     : reinterpret_cast<const float2*>(a.st_in) + row * a.st_np;
          for (int p = 0; p < a.st_np; ++p) { sm += raw[p].x; sq += raw[p].y; }
        }
        if (sq >= (X3_HALF_MAX * 0.125f) * (X3_HALF_MAX * 0.125f)) range_raise(a.range, RANGE_BIT_ACT);   // (producer's planes, as x3q_tile)
        const float mean = sm / (float)K;
        const float var = fmaxf(sq / (float)K - mean * mean, 0.0f);
        if (mean * mean > 256.0f * var) range_raise(a.range, RANGE_BIT_STATS);
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
#undef QT_PIECE
    __builtin_amdgcn_s_setprio(0);
    if constexpr (GRP) asm volatile("" : "+v"(lofsAg_[0]), "+v"(lofsAg_[1]), "+v"(lofsAg_[2]), "+v"(lofsAg_[3]));
    else asm volatile("" : "+v"(lofsA_), "+v"(lofsA3_));
    QT_STAMP(0);

    __syncthreads();   // statistics visible; every wave is out of the k-loop: the LDS becomes planes + exchange
    float2 st[QT_TM];
#pragma unroll
    for (int i = 0; i < QT_TM; ++i) st[i] = reinterpret_cast<const float2*>(lds + QT_STX)[wm * 128 + 16 * i + r16];
    float4 cs4[QT_NJ], b4[QT_NJ];
#pragma unroll
    for (int j = 0; j < QT_NJ; ++j) {
      const int n = n0 + wn * 48 + 16 * j + 4 * q;
      cs4[j] = *reinterpret_cast<const float4*>(a.csum + n);
      b4[j] = *reinterpret_cast<const float4*>(a.bias + n);
    }
    QT_STAMP(1);

    // ---- q / k / v -> LDS (LayerNorm fold and hi / lo split of x3q_epilogue8; column tile j of a wave is q / k / v: plane and scale
    // are compile-time per j).  Lane: rows 128 wm + 16 i + r16, head dims 16 wn + 4 q .. + 3 = 8 bytes of 16-byte chunk 2 wn + (q >> 1).
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int chunk = 2 * wn + (q >> 1), half8 = (q & 1) << 3;
    float amaxj[QT_NJ] = {0.0f, 0.0f, 0.0f};
    auto write_rows = [&](bool want_q, bool want_kv) {
#pragma unroll
      for (int i = 0; i < QT_TM; ++i) {
        const int R = wm * 128 + 16 * i + r16;                             // tile row = frame
        const int Rq = 16 * i + r16;                                       // row inside this half's exchange planes
        const f2 sx = (f2)(st[i].x), sy = (f2)(st[i].y);
#pragma unroll
        for (int j = 0; j < QT_NJ; ++j) {
          if (j == 0 ? !want_q : !want_kv) continue;
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
          unsigned char* ph;
          int pstride;
          if (j == 0) { ph = lds + QT_Q + kswz(Rq, chunk) + half8; pstride = 128 * 128; }
          else if (j == 1) { ph = lds + QT_K + kswz(R, chunk) + half8; pstride = QT_PLANE; }
          else { ph = lds + QT_V + vswz(R, chunk) + half8; pstride = QT_PLANE; }
          u32x2_alias hv, lv;
          hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
          if (!(QT_ABL & 1)) {
            *reinterpret_cast<u32x2_alias*>(ph) = hv;
            *reinterpret_cast<u32x2_alias*>(ph + pstride) = lv;
          }
        }
      }
    };
    // K and V planes first (they do not touch the region the statistics sat in), then the first half's queries behind a barrier that by
    // then costs nothing: every wave read its statistics long before
    write_rows(false, true);
    __syncthreads();                                   // B0: every wave holds its rows' statistics: the exchange planes may be written
    if (wm == 0) write_rows(true, false);
    {
      float amax = fmaxf(fmaxf(amaxj[1], amaxj[2]) * 8.0f, wm == 0 ? amaxj[0] : 0.0f);
      if (amax > X3_HALF_MAX) range_raise(a.range, RANGE_BIT_ACT);
    }
    // the word through which the second half synchronises its query exchange by itself (B3'): in the last pad row of the V lo plane
    // (T <= 255: key 255 is a pad key -- its numerator is an exact zero, any finite bits will do), zeroed by the lane that wrote there
    unsigned* const xsync = reinterpret_cast<unsigned*>(lds + QT_V + QT_PLANE + 255 * 128);
    if (wave == 7 && lane == 63) {   // [0]: second half's queries written; [1]: second half through its scores
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      reinterpret_cast<volatile unsigned*>(xsync)[0] = 0u;
      reinterpret_cast<volatile unsigned*>(xsync)[1] = 0u;
    }
    __syncthreads();                                   // B1: K, V, Q(half 0) written
    QT_STAMP(2);
    // From here the two halves run separate instruction streams that meet at ONE barrier (B2) and two counters (B3', B4'); from then on the two
    // waves of a SIMD (w and w + 4) are about a step apart -- one in an MFMA step, one in a VALU step:
    //   half 0: read Q | B2 | scores, softmax                          | B4' (no wait) | products, outputs, both halves' prefetch pieces
    //   half 1:        | B2 | write Q(half 1) | B3' | read Q, scores | raise B4' | softmax, products, outputs
    _Float16* const out_unit = a.out + tok0 * 2 * a.D + hd * 128;
    // grouped form: the key tiles this wave's 32 queries need (wave-uniform) and this lane's key window
    int kt_lo = 0, kt_hi = QT_NKT - 1, klo = 0, khi = 0;
    if constexpr (GRP) {
      const int TV = a.G * T;
      const int qa = 32 * wave < TV ? 32 * wave : TV - 1, qb = 32 * wave + 31 < TV ? 32 * wave + 31 : TV - 1;
      kt_lo = (((qa * a.Tinv) >> 16) * T) >> 5;
      kt_hi = (((qb * a.Tinv) >> 16) * T + T - 1) >> 5;
      const int tq = 32 * wave + (lane & 31);
      klo = (((tq < TV ? tq : TV - 1) * a.Tinv) >> 16) * T;
      khi = klo + T;
    }
    auto orow = [&](int R) -> int { return grow(R, jt); };
    auto read_q = [&](h8 (&qh)[4], h8 (&ql)[4]) {   // this wave's 32 queries (rows 32 (wave & 3) + r of its half's exchange planes), all 64 dims
      const int r = lane & 31, h = lane >> 5;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (QT_ABL & 16) continue;
        const int qo = kswz(32 * (wave & 3) + r, 2 * ks + h);
        qh[ks] = *reinterpret_cast<const h8*>(lds + QT_Q + qo);
        ql[ks] = *reinterpret_cast<const h8*>(lds + QT_Q + 128 * 128 + qo);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if (wm == 0) {
      f32x16 sacc[QT_NKT];
      float lsum = 0.f;
      {
        h8 qh[4], ql[4];
        read_q(qh, ql);
        __builtin_amdgcn_s_barrier();                  // B2: half 0 holds its queries: the exchange planes may be rewritten
        if (!(QT_ABL & 2)) qt_scores<0, GRP>(lds, lane, qh, ql, sacc, kt_lo, kt_hi);
      }
      qt_softmax<0, GRP>(lane, T, sacc, lsum, kt_lo, kt_hi, klo, khi);
      // B4': the second half is through its scores -- it holds its queries (the exchange planes become patches) and K is dead (the
      // prefetch below); a counter it raised long before this point, not a barrier it would have to wait at
      while (reinterpret_cast<volatile unsigned*>(xsync)[1] < 4u) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      QT_STAMP(3);
      if constexpr (GRP) qt_products_outputs<0, true>(lds, wave, lane, T, J, a.D, sacc, lsum, out_unit, a.range, kt_lo, kt_hi, orow);
      else qt_products_outputs<0>(lds, wave, lane, T, J, a.D, sacc, lsum, out_unit, a.range);
      // the next tile's first k-tile into the dead K planes: both halves' pieces from this half, which is a step ahead and would idle
      if (more) { stage_first(bjn, hdn, wave); stage_first(bjn, hdn, wave + 4); }
      QT_STAMP(4);
    } else {
      __builtin_amdgcn_s_barrier();                    // B2
      write_rows(true, false);
      if (amaxj[0] > X3_HALF_MAX) range_raise(a.range, RANGE_BIT_ACT);
      // B3': the four waves of this half meet at an LDS counter (a workgroup barrier would hold the first half at its scores)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(xsync, 1u);
      while (*reinterpret_cast<volatile unsigned*>(xsync) < 4u) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      f32x16 sacc[QT_NKT];
      float lsum = 0.f;
      {
        h8 qh[4], ql[4];
        read_q(qh, ql);
        if (!(QT_ABL & 2)) qt_scores<QT_PB1, GRP>(lds, lane, qh, ql, sacc, kt_lo, kt_hi);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(xsync + 1, 1u);         // B4'
      QT_STAMP(3);
      qt_softmax<QT_PB1, GRP>(lane, T, sacc, lsum, kt_lo, kt_hi, klo, khi);
      if constexpr (GRP) qt_products_outputs<QT_PB1, true>(lds, wave, lane, T, J, a.D, sacc, lsum, out_unit, a.range, kt_lo, kt_hi, orow);
      else qt_products_outputs<QT_PB1>(lds, wave, lane, T, J, a.D, sacc, lsum, out_unit, a.range);
      QT_STAMP(4);
    }
    bj = bjn; hd = hdn;
    __syncthreads();   // planes and patches are read before the next tile's statistics block and second k-tile are staged over them
    QT_STAMP(5);
  }
#undef QT_STAMP
  if (a.diag && (threadIdx.x == 0 || threadIdx.x == 256)) {   // wave 0 (first half) and wave 4 (second half)
    unsigned long long* d = a.diag + 16 * b + (threadIdx.x ? 8 : 0);
    for (int i = 0; i < 6; ++i) d[i] = dg[i];
    d[6] = (unsigned long long)nitems;
    d[7] = __builtin_amdgcn_s_memrealtime() - dg_r0;
  }
}

}  // namespace

static std::atomic<int> g_qt_diag{0};
void set_qkv_tattn_diag(int on) { g_qt_diag = on; }

// T in 193 ... 255: one joint's frames per tile; T in 2 ... 127: the frames of 255 / T joints per tile (grouped form)
bool qkv_tattn_ok(int T, int J, int D, int H, int K) {
  return ((T > 192 && T <= 255) || (T >= 2 && T <= 127)) && J >= 1 && H == 8 && D == 512 && K % 64 == 0 && K >= 128;
}

// Tokens M = B T J, rows (b T + t) J + j.
hipError_t launch_qkv_tattn(const void* Apair, const void* Wpair_tileorder, const float* bias_to, const float* csum_to, const float* st_in,
                            int st_np, float eps, int w_exp, void* out_x3, int B, int T, int J, int K, int D, int H, hipStream_t s) {
  if (!qkv_tattn_ok(T, J, D, H, K) || B <= 0 || st_np < 1 || !Apair || !Wpair_tileorder || !bias_to || !csum_to || !st_in || !out_x3)
    return hipErrorInvalidValue;
  if (w_exp < -14 || w_exp > 12) return hipErrorInvalidValue;
  if ((long long)B * J * 8 > 0x7fffffffLL / 2 || (size_t)255 * J * 2 * K * 2 > 0xffffffffull) return hipErrorInvalidValue;   // 32-bit lane offsets
  QtArgs a{};
  a.Ap = (const _Float16*)Apair; a.Wp = (const _Float16*)Wpair_tileorder; a.bias = bias_to; a.csum = csum_to; a.st_in = st_in;
  a.st_np = st_np; a.eps = eps; a.out_scale = ldexpf(1.0f, -(3 + w_exp));
  a.out = (_Float16*)out_x3; a.M = B * T * J; a.K = K; a.T = T; a.J = J; a.BJ = B * J; a.D = D;
  const bool grp = T <= 127;
  if (grp) {
    a.G = 255 / T < J ? 255 / T : J;
    a.TPS = (J + a.G - 1) / a.G;
    a.Tinv = 65536 / T + 1;
    a.BJ = B * a.TPS;
    if ((size_t)T * J * 2 * K * 2 > 0xffffffffull) return hipErrorInvalidValue;
  }
  a.range = launch_range_word();
  const void* kfn = grp ? reinterpret_cast<const void*>(k_qkv_tattn<true>) : reinterpret_cast<const void*>(k_qkv_tattn<false>);
  static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};   // one bit per device
  if (hipError_t ae = lds_optin(kfn, QT_LDS, attr_done[grp ? 1 : 0])) return ae;
  int n_cu = device_cu_count();
  if (n_cu <= 0) return hipErrorUnknown;
  const int tiles = a.BJ * 8;
  const int grid = tiles < n_cu ? tiles : n_cu;
  hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;   // the stamp report allocates, synchronises and copies: illegal inside a
  (void)hipStreamIsCapturing(s, &cap_st);                          // hipGraph capture (d3d_engine_set_graph_mode) -- plain launch there
  if (g_qt_diag.load() > 0 && cap_st == hipStreamCaptureStatusNone) {   // "qt_diag" option: every 50th launch with stamps of wave 0, summarised on stderr (synchronises the stream)
    static std::atomic<int> count{0};
    if (count.fetch_add(1) % 50 == 10) {
      unsigned long long* buf = nullptr;
      if (hipMalloc(&buf, (size_t)grid * 128) != hipSuccess) return hipErrorOutOfMemory;
      (void)hipMemsetAsync(buf, 0, (size_t)grid * 128, s);
      a.diag = buf;
      if (grp) hipLaunchKernelGGL(k_qkv_tattn<true>, dim3(grid), dim3(512), QT_LDS, s, a);
      else hipLaunchKernelGGL(k_qkv_tattn<false>, dim3(grid), dim3(512), QT_LDS, s, a);
      (void)hipStreamSynchronize(s);
      std::vector<unsigned long long> h((size_t)grid * 16);
      (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
      (void)hipFree(buf);
      for (int half = 0; half < 2; ++half) {
        double sum[6] = {0}, tiles_n = 0, cyc = 0, ticks = 0;
        for (int g = 0; g < grid; ++g) {
          const unsigned long long* d = &h[16 * (size_t)g + 8 * half];
          for (int i = 0; i < 6; ++i) { sum[i] += (double)d[i]; cyc += (double)d[i]; }
          tiles_n += (double)d[6]; ticks += (double)d[7];
        }
        const double ghz = ticks > 0 ? cyc / (ticks * 10.0) : 0.0;   // cycles per ns (100 MHz ticks = 10 ns)
        fprintf(stderr, "[qt diag] half %d (wave %d): B J = %d, tiles %d on %d workgroups, clock %.2f GHz; per tile (us): k-loop %.2f  statistics -> registers %.2f  "
                "plane writes %.2f  %s %.2f  %s %.2f  end barrier %.2f  | total %.2f\n", half, 4 * half, a.BJ, tiles, grid, ghz,
                sum[0] / tiles_n / ghz / 1e3, sum[1] / tiles_n / ghz / 1e3, sum[2] / tiles_n / ghz / 1e3,
                half ? "wait + query exchange + scores" : "query read + scores + softmax", sum[3] / tiles_n / ghz / 1e3,
                half ? "softmax + products + outputs" : "products + outputs + prefetch issue", sum[4] / tiles_n / ghz / 1e3,
                sum[5] / tiles_n / ghz / 1e3, cyc / tiles_n / ghz / 1e3);
      }
      return hipGetLastError();
    }
  }
  if (grp) hipLaunchKernelGGL(k_qkv_tattn<true>, dim3(grid), dim3(512), QT_LDS, s, a);
  else hipLaunchKernelGGL(k_qkv_tattn<false>, dim3(grid), dim3(512), QT_LDS, s, a);
  return hipGetLastError();
}

}  // namespace d3d
