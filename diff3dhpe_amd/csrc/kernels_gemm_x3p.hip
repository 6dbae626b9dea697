// F16X3 token GEMM, pre-split operands ("x3q"): the production GEMM of the F16X3 precision mode.
//
//   C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] )
//
// Both operands arrive pre-split into fp16 hi/lo (see kernels_gemm_f16x3.hip for the arithmetic and its measured
// accuracy) in the PAIR layout of d3d_kernels.h: a row of K values is 2K fp16, and the 32 hi and 32 lo values of
// k-tile t form ONE 128-byte line at [64 t, 64 t + 64):
//   A pair [M][2K]  = split(8 * a)      written by the PRODUCER of the activation (LayerNorm, attention, GELU epilogue)
//   W pair [N][2K]  = split(4096 * w)   made once at weight-commit time
// so the k-loop is a pure fp16 MFMA loop: per output tile and 32-deep k-tile the three products a_lo b_hi, a_hi b_lo,
// a_hi b_hi go into one fp32 accumulator, result scaled by 2^-15 in the epilogue.
//
// Why the pair layout: a k-tile of a row is exactly one cache line.  With separate hi and lo planes each row
// contributes two half-used 128-byte lines per k-tile, and the other halves (the next k-tile) are long evicted from
// the 32 KiB L1 when they are wanted (a k-tile's footprint is 128 KiB of lines), so the L2->L1 fill traffic doubles.
// Measured on MI355X, staging alone (no MFMA) for the qkv GEMM: 0.58 ms with planes, 0.33 ms with whole lines, against
// 0.63 ms of MFMA work to hide it under.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): each wave-instruction drops 1 KiB = 8 rows x 128 B straight into LDS,
// no staging VGPRs.  The LDS image is lane-linear (a tile row is 128 B: chunks 0-3 hi, 4-7 lo), so the bank swizzle
// (16-byte chunk index XOR (row>>1)&7, which makes every ds_read_b128 16-lane group hit 16 distinct 4-bank slots) is
// applied to the per-lane SOURCE address -- it permutes chunks inside one line -- and again on the fragment read.
// Two LDS stages; the DMA of k-tile t+1 is issued after the barrier that retires k-tile t-1 and flies under the MFMAs
// of k-tile t (one barrier per k-tile; __syncthreads() drains the wave's own DMA with vmcnt(0) before the barrier).
//
// Products are issued as v_mfma_f32_16x16x32_f16 (one MFMA spans the 32-deep k-tile): measured on MI355X with random
// fp16 operands (experiments/mfma_ceiling.hip) the chip sustains 1.97 PFLOP/s on this shape against 1.54 PFLOP/s on
// 32x32x16 (it holds a higher clock).  The weight fragment is the first operand, so an accumulator tile holds C^T:
// lane -> token m = lane&15, registers -> n = 4*(lane>>4) + reg; fragment read: lane (r16 = lane&15, q = lane>>4) takes
// 16-byte chunk q (hi) and 4+q (lo) of row r16.
#include <cstdlib>
#include "d3d_kernels.h"

namespace d3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int PBK = 32;                          // k-tile depth (fp16 elements)
constexpr float P_OUT_SCALE = 1.0f / 32768.0f;   // 2^-(3+12)
constexpr float P_A_SCALE = 8.0f;

// F16X3 range guard (d3d_kernels.h): every plane writer tracks max |scaled value| per lane; a lane whose value left the fp16
// range (the clamp below fired: |x| > 8188) ORs bit 0 into this sticky per-device word once, at the end of its epilogue.
__device__ unsigned g_range_x3p;
__device__ __forceinline__ void range_note(float amax) {
#ifndef D3D_NO_RANGE_GUARD
  if (amax > X3_HALF_MAX) atomicOr(&g_range_x3p, 1u);
#endif
}

// OUTSPLIT: 0 = fp32 C; 1 = hi/lo PLANES of C (two [M][N] fp16 matrices: the temporal attention kernel reads q/k/v
// that way); 2 = PAIR layout (the consumer is another x3 GEMM).  Both carry 8*c (columns < qcols: 1*c, the q third
// of a temporal qkv GEMM, which absorbs the dh^-0.5 = 2^-3 attention scale).
template <int OUTSPLIT, bool GUARD = true>
__device__ __forceinline__ void store_split4(const float (&v)[4], float osc, _Float16* Cht, _Float16* Clt, int off, int poff,
                                             float& amax) {
  h4 hh, ll;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float raw = v[e] * osc;
    if (GUARD) amax = __builtin_fmaxf(amax, __builtin_fabsf(raw));
    const float sc = __builtin_amdgcn_fmed3f(raw, -65504.0f, 65504.0f);
    hh[e] = (_Float16)sc;
    ll[e] = (_Float16)(sc - (float)hh[e]);
  }
  if (OUTSPLIT == 2) {
    *reinterpret_cast<h4*>(Cht + poff) = hh;
    *reinterpret_cast<h4*>(Cht + poff + PAIR_LO) = ll;
  } else {
    *reinterpret_cast<h4*>(Cht + off) = hh;
    *reinterpret_cast<h4*>(Clt + off) = ll;
  }
}

// Operand tile in LDS: rows of 128 B = 8 chunks of 16 B (0-3 hi, 4-7 lo of the k-tile); physical chunk = c ^ ((row>>1)&7).
// A 16-lane ds_read_b128 group reads 16 consecutive rows at one logical chunk: row parity picks the half of the 256-byte
// bank row, (row>>1)&7 permutes the 8 chunks of that half -> 16 distinct 4-bank slots.  The lo chunk of a fragment is
// the hi chunk's offset XOR 64.

// ---- DMA plan (branch-free): the k-tile of a BM x BN tile is (BM + BN)/8 pieces of 8 rows x
// 128 B; wave w moves pieces w, w + NW, ... of A, then of W.  A lane serves row (8 piece + lane/8), LDS slot lane%8,
// and fetches the source chunk the swizzle assigns to that slot (constant per lane: NW is even, so (row>>1)&7 =
// 4 (w&1) + lane/16).  Contract: the A buffer holds >= mtiles*BM rows and the W buffer >= ntiles*BN rows
// (padding rows are staged and multiplied but never stored).
// Source addresses are formed as (wave-uniform byte base: SGPR pair, advanced by scalar adds) + (one 32-bit per-lane byte
// offset, the same for every piece and k-tile), so that the DMA takes the saddr form and needs no per-piece 64-bit VALU
// address arithmetic.
#define D3D_DMA_PLAN(NW_, BM_)                                                                                          \
  const int lr_ = lane >> 3;                                                                                            \
  const int csrc_ = (lane & 7) ^ (((wave & 1) << 2) | (lr_ >> 1));                                                      \
  const size_t K2_ = 2 * (size_t)K;                                                                                     \
  const char* ubA = reinterpret_cast<const char*>(Ap) + (size_t)(m0 + wave * 8) * K2_ * 2;                             \
  const char* ubB = reinterpret_cast<const char*>(Wp) + (size_t)(n0 + wave * 8) * K2_ * 2;                             \
  unsigned lofs_ = (unsigned)(lr_ * (int)K2_ + csrc_ * 8) * 2u;                                                   \
  const size_t it_stride = (size_t)((NW_) * 8) * K2_ * 2;                 /* bytes */                                   \
  const int dstA = wave * 1024 + lane * 16, dstB = (BM_) * 128 + wave * 1024 + lane * 16

#define D3D_GLDS(SRC, DSTOFF)                                                                                           \
  __builtin_amdgcn_global_load_lds((SRC), (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (DSTOFF)), 16, 0, 0)
// wave-uniform pointer pinned into an SGPR pair
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

#ifdef D3D_X3_PHASE_DIAG   // timing experiments: shader-clock stamps inside the two phases of k-tile 5 (workgroup 3, every wave)
__device__ unsigned long long g_x3_phase_diag[8 * 2 * 8];
#endif
void x3_phase_diag_report() {
#ifdef D3D_X3_PHASE_DIAG
  unsigned long long h[8 * 2 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_x3_phase_diag), sizeof(h)) != hipSuccess) return;
  fprintf(stderr, "[phase diag] cycles per wave: even phase wait | barrier->reads | group 0..3   odd phase wait | - | group 4..7   (k-tile 5, workgroup 3)\n");
  for (int w = 0; w < 8; ++w) {
    const unsigned long long* e = &h[(w * 2 + 0) * 8];
    const unsigned long long* o = &h[(w * 2 + 1) * 8];
    fprintf(stderr, "  wave %d: %5llu | %5llu | %5llu %5llu %5llu %5llu      %5llu | %5llu | %5llu %5llu %5llu %5llu    k-tile %llu\n", w, e[1] - e[0],
            e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - e[4], e[6] - e[5], o[1] - o[0], o[2] - o[1], o[3] - o[2], o[4] - o[3], o[5] - o[4],
            o[6] - o[5], o[6] - e[0]);
  }
#endif
}

// Between a wave's writes to its LDS patch and its reads of OTHER lanes' rows of it: nothing orders them for the compiler (one
// thread's load does not alias its own stores), LDS itself executes a wave's operations in order.  A compiler-level barrier.
#ifndef D3D_NO_PATCH_FENCE
#define D3D_PATCH_FENCE() asm volatile("" ::: "memory")
#else
#define D3D_PATCH_FENCE() do { } while (0)
#endif

// LDS fragment read as inline asm: the compiler does not know it as an LDS operation and inserts no s_waitcnt for it -- the
// k-loop places COUNTED lgkmcnt waits itself (LDS operations of one wave return in order).  Left to the compiler, every phase
// opened with its 10-12 fragment reads followed by s_waitcnt lgkmcnt(0): ~350 cycles per phase in which neither wave of the
// SIMD (both just released by the same barrier) had an MFMA to issue.
__device__ __forceinline__ void lds_rd128(h8& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }
// wait until at most N LDS operations of this wave are outstanding; the fragments named become usable (data dependence for
// the scheduler: the MFMAs that read them cannot be moved above the wait)
template <int N>
__device__ __forceinline__ void lgkm_wait4(h8& a, h8& b, h8& c, h8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait2(h8& a, h8& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

// Launch-time dispatch over (epilogue, output form): the five combinations the engine and the op hooks use.
#define D3D_X3_DISPATCH(LAUNCH)                                                                                          \
  do {                                                                                                                   \
    if (outsplit == 0) {                                                                                                 \
      if (epi == EPI_NONE) LAUNCH(EPI_NONE, 0);                                                                          \
      else if (epi == EPI_GELU) LAUNCH(EPI_GELU, 0);                                                                     \
      else if (epi == EPI_RESIDUAL) LAUNCH(EPI_RESIDUAL, 0);                                                             \
      else return hipErrorInvalidValue;                                                                                  \
    } else if (outsplit == 1) {                                                                                          \
      if (epi == EPI_NONE) LAUNCH(EPI_NONE, 1);                                                                          \
      else return hipErrorInvalidValue;                                                                                  \
    } else {                                                                                                             \
      if (epi == EPI_GELU) LAUNCH(EPI_GELU, 2);                                                                          \
      else if (epi == EPI_NONE) LAUNCH(EPI_NONE, 2);                                                                     \
      else return hipErrorInvalidValue;                                                                                  \
    }                                                                                                                    \
  } while (0)

// GELU(x) = x Phi(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2), with erfc from Abramowitz & Stegun 7.1.26
// (erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), |error| <= 1.5e-7): one v_rcp_f32, one v_exp_f32 and a
// handful of FMAs, branch-free -- the library erff costs about three times as much, and the fc1 epilogue is VALU-bound
// (128 outputs per lane).  The absolute error of the result stays below 1e-7 |x|, the rounding level of the fp32 path.
__device__ __forceinline__ float gelu_fast(float x) {
  const float ax = __builtin_fabsf(x);
  const float z = ax * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
  return __builtin_fmaf(-(0.5f * ax * (p * t)), e, __builtin_fmaxf(x, 0.0f));
}

// Two elements at a time: the epilogues are VALU-bound (the fc1 one: ~22 VALU instructions per output element, 13 us of a 52 us
// tile), and gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on register pairs at the rate of the scalar forms.  Every
// multiply-add is written as an explicit fma, in the order of the scalar code above, so that both give the same bits.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float a) { return (f2)(a); }
__device__ __forceinline__ f2 gelu_fast2(f2 x) {
  f2 ax, t, e, m;
  ax.x = __builtin_fabsf(x.x); ax.y = __builtin_fabsf(x.y);
  const f2 z = ax * 0.70710678118654752440f;
  const f2 den = fma2(splat2(0.3275911f), z, splat2(1.0f));
  t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  f2 p = fma2(splat2(1.061405429f), t, splat2(-1.453152027f));
  p = fma2(p, t, splat2(1.421413741f));
  p = fma2(p, t, splat2(-0.284496736f));
  p = fma2(p, t, splat2(0.254829592f));
  const f2 zz = z * z * -1.44269504088896340736f;
  e.x = __builtin_amdgcn_exp2f(zz.x); e.y = __builtin_amdgcn_exp2f(zz.y);
  m.x = __builtin_fmaxf(x.x, 0.0f); m.y = __builtin_fmaxf(x.y, 0.0f);
  return fma2(-(splat2(0.5f) * ax * (p * t)), e, m);
}
// 8 values -> fp16 (hi, lo) of osc * v, clamped to the fp16 range
template <bool GUARD = true>
__device__ __forceinline__ void split8_x3(const f2 (&v)[4], float osc, h8& oh, h8& ol, float& amax) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f2 sc = v[e] * osc;
    if (GUARD) amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(sc.x)), __builtin_fabsf(sc.y));
    sc.x = __builtin_amdgcn_fmed3f(sc.x, -65504.0f, 65504.0f);
    sc.y = __builtin_amdgcn_fmed3f(sc.y, -65504.0f, 65504.0f);
    oh[2 * e] = (_Float16)sc.x;
    oh[2 * e + 1] = (_Float16)sc.y;
    f2 back;
    back.x = (float)oh[2 * e]; back.y = (float)oh[2 * e + 1];
    const f2 d = sc - back;
    ol[2 * e] = (_Float16)d.x;
    ol[2 * e + 1] = (_Float16)d.y;
  }
}
// (sum, sum of squares) of 8 values
__device__ __forceinline__ void sums8(const f2 (&v)[4], float& sm, float& sq) {
  const f2 s2 = (v[0] + v[1]) + (v[2] + v[3]);
  const f2 q2 = fma2(v[0], v[0], v[1] * v[1]) + fma2(v[2], v[2], v[3] * v[3]);
  sm = s2.x + s2.y;
  sq = q2.x + q2.y;
}
__device__ __forceinline__ void load8(const float* p, f2 (&o)[4]) {
  const float4 t0 = *reinterpret_cast<const float4*>(p), t1 = *reinterpret_cast<const float4*>(p + 4);
  o[0].x = t0.x; o[0].y = t0.y; o[1].x = t0.z; o[1].y = t0.w; o[2].x = t1.x; o[2].y = t1.y; o[3].x = t1.z; o[3].y = t1.w;
}
// 8 fp16 (hi) + 8 fp16 (lo) of 8 r -> r
__device__ __forceinline__ void unsplit8(const uint4 rh, const uint4 rl, f2 (&o)[4]) {
  const h8 hh = __builtin_bit_cast(h8, rh), ll = __builtin_bit_cast(h8, rl);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f2 hf, lf;
    hf.x = (float)hh[2 * e]; hf.y = (float)hh[2 * e + 1];
    lf.x = (float)ll[2 * e]; lf.y = (float)ll[2 * e + 1];
    o[e] = hf + lf;
  }
}

// FX flags of the folded forms (X3Fold in d3d_kernels.h)
constexpr int FX_LNF = 1;   // LayerNorm folded into this GEMM: per-row (rstd, -mean rstd) from LDS, csum per column
constexpr int FX_RP = 2;    // residual from pair-layout planes
constexpr int FX_SO = 4;    // per-row (sum, sum of squares) of the output rows -> st_out
constexpr int FX_PN = 8;    // the tile spans whole rows: post-norm of the new rows in the epilogue (X3PostNorm)

// sum over the 16 lanes of a DPP row (all 16 lanes get the total): quad xor 1, xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

struct X3Tail {            // per-launch extras of the folded forms (device copy of X3Fold + derived)
  const float* st_in; int st_np; const float* csum; float eps;
  const _Float16* Rp;
  float* st_out;
  X3PostNorm pn;
};

// lds_x: the workgroup's LDS beyond the operand stages: [BM] float2 row statistics (FX_LNF)
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX, bool CHECK>
__device__ __forceinline__ void x3q_epilogue(f32x4 (&acc)[TM][4], float* patch, unsigned char* lds_x, const float* __restrict__ bias,
                                             const float* Rt, float* Ct, _Float16* Cht, _Float16* Clt, const _Float16* Rpt,
                                             const float* __restrict__ csum, float* st_out, int mt0, int nt0, int rbase, int lane,
                                             int M, int N, int qcols, int gl, int gh) {
  // [gl, gh): the wave's m-tiles that are computed (all of them except in a split tail tile, x3q_tile)
  // patch: two wave-private 16 rows x 64 floats (alternating, so the LDS round trip of one m-tile overlaps the stores
  // of the previous one), 16-byte chunks XOR-swizzled by (row & 7)
    const int m16 = lane & 15, q4 = lane >> 4;
  const int rrow = lane >> 4, rc4 = lane & 15;         // read side: 16 lanes per row, 4 rows per pass
  const int n = nt0 + 4 * rc4;
  const bool ncol_ok = !CHECK || n < N;
  float4 b4 = make_float4(0, 0, 0, 0), cs4 = make_float4(0, 0, 0, 0);
  if (bias && ncol_ok) b4 = *reinterpret_cast<const float4*>(bias + n);
  if ((FX & FX_LNF) && ncol_ok) cs4 = *reinterpret_cast<const float4*>(csum + n);
  const float osc = (n < qcols) ? 1.0f : P_A_SCALE;
  const int pc = (int)pair_col(4 * rc4);
  const float2* srow = reinterpret_cast<const float2*>(lds_x);                 // (rstd, -mean * rstd) per workgroup row
  const int npart = (N + 63) >> 6;
  // Residual rows are fetched PF m-tiles (PF * 4 KiB per wave) ahead of their use: vmcnt retires in order, so a load
  // issued right behind the previous m-tile's stores and consumed at once waits for those stores' acknowledgement as well
  // as its own latency (measured: 22 us per 256x256 tile with load-add-store in sequence, against 3.8 us for the plain
  // store epilogue; 12 us with the window).  Touching the tile's lines from inside the last k-tile to pull them into L2
  // was tried and lost: the 64-line gathers are slower than the window they were meant to shorten.
  constexpr int PF = (EPI == EPI_RESIDUAL) ? (TM < 4 ? TM : 4) : 0;
  // Addressing: wave-uniform tile base (SGPR pair) + 32-bit unsigned byte offset per lane, so that a load/store needs one
  // address VGPR (global_* saddr form) instead of a 64-bit pair -- with 64-bit pairs the 32 row addresses of a wave tile
  // cost more registers than the residual window.
  const unsigned ob = (unsigned)(rrow * N + 4 * rc4) * 4u;          // byte offset of this lane's float4 in row rrow
  const unsigned rstep = (unsigned)N * 16u;                          // 4 rows
  const unsigned obp = (unsigned)(rrow * 2 * N + pc) * 2u;          // same position in a pair-layout buffer (hi; lo 64 B on)
  const char* Rb = reinterpret_cast<const char*>(Rt);
  const char* Rpb = reinterpret_cast<const char*>(Rpt);
  char* Cb = reinterpret_cast<char*>(Ct);
  float4 rr[TM][4];
  float amax = 0.0f;   // range guard
  auto load_res = [&](int i) {
    if (i < gl || i >= gh) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = rrow + 4 * p;
      rr[i][p] = make_float4(0, 0, 0, 0);
      if (!CHECK || (mt0 + 16 * i + row < M && ncol_ok)) {
        if (FX & FX_RP) {   // 4 hi + 4 lo fp16 of 8 r, kept packed (same 4 registers as the fp32 form)
          const float2 hh = *reinterpret_cast<const float2*>(Rpb + (obp + (unsigned)(4 * i + p) * rstep));
          const float2 ll = *reinterpret_cast<const float2*>(Rpb + (obp + (unsigned)(4 * i + p) * rstep) + 64u);
          rr[i][p] = make_float4(hh.x, hh.y, ll.x, ll.y);
        } else {
          rr[i][p] = *reinterpret_cast<const float4*>(Rb + (ob + (unsigned)(4 * i + p) * rstep));
        }
      }
    }
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < PF; ++i) load_res(i);
  __syncthreads();   // every wave is done with the operand stages: reuse LDS for the transpose patches
#pragma unroll
  for (int i = 0; i < TM; ++i) {
   if (i >= gl && i < gh) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(patch + (i & 1) * 1024 + m16 * 64 + (((4 * j + q4) ^ (m16 & 7)) << 2)) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    D3D_PATCH_FENCE();   // the strip is read back transposed: other lanes' rows
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = rrow + 4 * p;
      const float4 a4 = *reinterpret_cast<const float4*>(patch + (i & 1) * 1024 + row * 64 + ((rc4 ^ (row & 7)) << 2));
      const int m = mt0 + 16 * i + row;
      const bool ok = !CHECK || (m < M && ncol_ok);
      if (!(FX & FX_SO) && !ok) continue;
      float v[4];
      if (FX & FX_LNF) {   // LN(x) W^T + b = rstd (x W'^T) - rstd mean csum + b'
        const float2 st = srow[rbase + 16 * i + row];
        v[0] = fmaf(st.x, a4.x * P_OUT_SCALE, fmaf(st.y, cs4.x, b4.x));
        v[1] = fmaf(st.x, a4.y * P_OUT_SCALE, fmaf(st.y, cs4.y, b4.y));
        v[2] = fmaf(st.x, a4.z * P_OUT_SCALE, fmaf(st.y, cs4.z, b4.z));
        v[3] = fmaf(st.x, a4.w * P_OUT_SCALE, fmaf(st.y, cs4.w, b4.w));
      } else {
        v[0] = a4.x * P_OUT_SCALE + b4.x; v[1] = a4.y * P_OUT_SCALE + b4.y;
        v[2] = a4.z * P_OUT_SCALE + b4.z; v[3] = a4.w * P_OUT_SCALE + b4.w;
      }
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
      }
      if (EPI == EPI_RESIDUAL) {
        float4 r4 = rr[i][p];
        if (FX & FX_RP) {
          const h4 hh = __builtin_bit_cast(h4, make_float2(r4.x, r4.y)), ll = __builtin_bit_cast(h4, make_float2(r4.z, r4.w));
          r4 = make_float4(((float)hh[0] + (float)ll[0]) * 0.125f, ((float)hh[1] + (float)ll[1]) * 0.125f,
                           ((float)hh[2] + (float)ll[2]) * 0.125f, ((float)hh[3] + (float)ll[3]) * 0.125f);
        }
        v[0] = r4.x + v[0]; v[1] = r4.y + v[1]; v[2] = r4.z + v[2]; v[3] = r4.w + v[3];
      }
      if (FX & FX_SO) {   // row statistics of the new residual stream for the LayerNorm folded into the next GEMM
        float sm = ok ? (v[0] + v[1]) + (v[2] + v[3]) : 0.0f;
        float sq = ok ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.0f;
        sm = row16_sum(sm);
        sq = row16_sum(sq);
        // one partial per (row, 64-column wave block): the consumer adds the N/64 partials of a row in column order, so the
        // statistics -- like every GEMM element -- do not depend on the tile shape that produced them
        if (rc4 == 0 && m < M) *reinterpret_cast<float2*>(st_out + 2 * ((size_t)m * npart + (nt0 >> 6))) = make_float2(sm, sq);
        if (!ok) continue;
      }
      if (OUTSPLIT) {
        const int off = (16 * i + row) * N + 4 * rc4;
        store_split4<OUTSPLIT, !(FX & FX_SO)>(v, osc, Cht, Clt, off, (16 * i + row) * 2 * N + pc, amax);   // (FX_SO: see x3q_epilogue8)
      } else {
        *reinterpret_cast<float4*>(Cb + (ob + (unsigned)(4 * i + p) * rstep)) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
   }
    if (EPI == EPI_RESIDUAL) {   // this m-tile's accumulators and residual registers are dead: refill the residual window
      if (i + PF < TM) load_res(i + PF);
      __builtin_amdgcn_sched_barrier(0);
    } else if (i & 1) {
      __builtin_amdgcn_sched_barrier(0);   // two m-tiles (two patches) in flight at a time
    }
  }
  if constexpr (OUTSPLIT != 0 && !(FX & FX_SO)) range_note(amax);
}

// The same epilogue for the forms that touch fp16 planes (plane / pair outputs, plane residual): the read-back side gives a
// lane EIGHT consecutive columns (8 lanes per row, 8 rows per pass, 2 passes per m-tile), so that every plane access is a
// 16-byte one (8 fp16): half as many load / store instructions as with 4 columns per lane.
__device__ __forceinline__ float row8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX, bool CHECK>
__device__ __forceinline__ void x3q_epilogue8(f32x4 (&acc)[TM][4], float* patch, unsigned char* lds_x, const float* __restrict__ bias,
                                              float* Ct, _Float16* Cht, _Float16* Clt, const _Float16* Rpt,
                                              const float* __restrict__ csum, float* st_out, int mt0, int nt0, int rbase, int lane,
                                              int M, int N, int qcols, int gl, int gh) {
  static_assert(EPI != EPI_RESIDUAL || (FX & FX_RP), "the 8-column epilogue takes its residual from planes");
  const int m16 = lane & 15, q4 = lane >> 4;          // write side: accumulator layout
  const int rrow = lane >> 3, rc8 = lane & 7;          // read side
  const int n = nt0 + 8 * rc8;
  const bool ncol_ok = !CHECK || n < N;
  f2 bb[4], cs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { bb[e] = splat2(0.f); cs[e] = splat2(0.f); }
  if (bias && ncol_ok) load8(bias + n, bb);
  if ((FX & FX_LNF) && ncol_ok) load8(csum + n, cs);
  const float osc = (n < qcols) ? 1.0f : P_A_SCALE;
  const int pc = (int)pair_col(8 * rc8);
  const float2* srow = reinterpret_cast<const float2*>(lds_x);
  const int npart = (N + 63) >> 6;
#ifndef D3D_X3_PFMAX_SO
#define D3D_X3_PFMAX_SO 3
#endif
  constexpr int PFMAX = (FX & FX_SO) ? D3D_X3_PFMAX_SO : 4;   // (the row-statistics form: 2 and 3 measure alike, 4 spills more)
  constexpr int PF = (EPI == EPI_RESIDUAL) ? (TM < PFMAX ? TM : PFMAX) : 0;   // residual window, see x3q_epilogue
  const unsigned ob = (unsigned)(rrow * N + 8 * rc8) * 4u;            // this lane's 8 floats in row rrow (fp32 buffer)
  const unsigned obh = (unsigned)(rrow * N + 8 * rc8) * 2u;           // ... in an [M][N] fp16 plane
  const unsigned obp = (unsigned)(rrow * 2 * N + pc) * 2u;            // ... in a pair-layout buffer (hi; lo 64 B on)
  const unsigned rstep = (unsigned)N * 32u;                            // 8 rows of an fp32 or pair buffer
  const unsigned rsteph = (unsigned)N * 16u;                           // 8 rows of an fp16 plane
  const char* Rpb = reinterpret_cast<const char*>(Rpt);
  char* Cb = reinterpret_cast<char*>(Ct);
  char* Chb = reinterpret_cast<char*>(Cht);
  char* Clb = reinterpret_cast<char*>(Clt);
  uint4 rh[TM][2], rl[TM][2];
  float amax = 0.0f;   // range guard
  auto load_res = [&](int i) {
    if (i < gl || i >= gh) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = rrow + 8 * p;
      rh[i][p] = make_uint4(0, 0, 0, 0);
      rl[i][p] = make_uint4(0, 0, 0, 0);
      if (!CHECK || (mt0 + 16 * i + row < M && ncol_ok)) {
        rh[i][p] = *reinterpret_cast<const uint4*>(Rpb + (obp + (unsigned)(2 * i + p) * rstep));
        rl[i][p] = *reinterpret_cast<const uint4*>(Rpb + (obp + (unsigned)(2 * i + p) * rstep) + 64u);
      }
    }
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < PF; ++i) load_res(i);
  __syncthreads();   // every wave is done with the operand stages: reuse LDS for the transpose patches
#pragma unroll
  for (int i = 0; i < TM; ++i) {
   if (i >= gl && i < gh) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(patch + (i & 1) * 1024 + m16 * 64 + (((4 * j + q4) ^ (m16 & 7)) << 2)) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    D3D_PATCH_FENCE();   // the strip is read back transposed: other lanes' rows
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = rrow + 8 * p;
      const float* prow = patch + (i & 1) * 1024 + row * 64;
      const float4 a0 = *reinterpret_cast<const float4*>(prow + (((2 * rc8) ^ (row & 7)) << 2));
      const float4 a1 = *reinterpret_cast<const float4*>(prow + (((2 * rc8 + 1) ^ (row & 7)) << 2));
      f2 a[4];
      a[0].x = a0.x; a[0].y = a0.y; a[1].x = a0.z; a[1].y = a0.w; a[2].x = a1.x; a[2].y = a1.y; a[3].x = a1.z; a[3].y = a1.w;
      const int m = mt0 + 16 * i + row;
      const bool ok = !CHECK || (m < M && ncol_ok);
      if (!(FX & FX_SO) && !ok) continue;
      f2 v[4];
      if (FX & FX_LNF) {
        const float2 st = srow[rbase + 16 * i + row];
        const f2 sx = splat2(st.x), sy = splat2(st.y);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fma2(sx, a[e] * P_OUT_SCALE, fma2(sy, cs[e], bb[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fma2(a[e], splat2(P_OUT_SCALE), bb[e]);
      }
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_fast2(v[e]);
      }
      if (EPI == EPI_RESIDUAL) {
        f2 r8[4];
        unsplit8(rh[i][p], rl[i][p], r8);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fma2(r8[e], splat2(0.125f), v[e]);
      }
      if (FX & FX_SO) {
        float sm = 0.f, sq = 0.f;
        if (ok) sums8(v, sm, sq);
        sm = row8_sum(sm);
        sq = row8_sum(sq);
        if (rc8 == 0 && m < M) *reinterpret_cast<float2*>(st_out + 2 * ((size_t)m * npart + (nt0 >> 6))) = make_float2(sm, sq);
        if (!ok) continue;
      }
      if (OUTSPLIT) {
        h8 oh, ol;
        if constexpr ((FX & FX_SO) != 0) {   // range guard of this form: by the consumer of its row statistics (x3q_tile, FX_LNF) -- this
          split8_x3<false>(v, osc, oh, ol, amax);   // epilogue sits at the 256-register limit: one more live register costs it 60 spilled
        } else {                                  // accumulators (+17 % per launch)
          split8_x3<true>(v, osc, oh, ol, amax);
        }
        if (OUTSPLIT == 2) {
          *reinterpret_cast<h8*>(Chb + (obp + (unsigned)(2 * i + p) * rstep)) = oh;
          *reinterpret_cast<h8*>(Chb + (obp + (unsigned)(2 * i + p) * rstep) + 64u) = ol;
        } else {
          *reinterpret_cast<h8*>(Chb + (obh + (unsigned)(2 * i + p) * rsteph)) = oh;
          *reinterpret_cast<h8*>(Clb + (obh + (unsigned)(2 * i + p) * rsteph)) = ol;
        }
      } else {
        *reinterpret_cast<float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep)) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
        *reinterpret_cast<float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep) + 16u) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
      }
    }
   }
    if (EPI == EPI_RESIDUAL) {
      if (i + PF < TM) load_res(i + PF);
      __builtin_amdgcn_sched_barrier(0);
    } else if (i & 1) {
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (OUTSPLIT != 0 && !(FX & FX_SO)) range_note(amax);
}

// GELU + pair output straight from the accumulators (fc1 -> hidden activation).  The hidden activation is only ever the A
// operand of the fc2 GEMM, so its k order inside a 32-column group is free: "accumulator order" (pair_col_acc, d3d_kernels.h;
// the fc2 weight is split in the same order at commit) makes the 8 values a lane holds of a group one 16-byte piece.  No LDS
// transpose, no barrier: lane (m = lane & 15, q = lane >> 4) owns row 16 i + m of m-tile i and columns 16 j + 4 q + r.
template <int TM, int WM, int WN, int FX, bool CHECK>
__device__ __forceinline__ void x3q_epilogue_acc(f32x4 (&acc)[TM][4], unsigned char* lds_x, const float* __restrict__ bias,
                                                 _Float16* Cht, const float* __restrict__ csum, int mt0, int nt0, int rbase, int lane,
                                                 int M, int N, int gl, int gh) {
  const int m16 = lane & 15, q4 = lane >> 4;
  f2 bb[4][2], cs[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nt0 + 16 * j + 4 * q4;
    bb[j][0] = bb[j][1] = cs[j][0] = cs[j][1] = splat2(0.f);
    if (!CHECK || n < N) {
      if (bias) {
        const float4 t = *reinterpret_cast<const float4*>(bias + n);
        bb[j][0].x = t.x; bb[j][0].y = t.y; bb[j][1].x = t.z; bb[j][1].y = t.w;
      }
      if (FX & FX_LNF) {
        const float4 t = *reinterpret_cast<const float4*>(csum + n);
        cs[j][0].x = t.x; cs[j][0].y = t.y; cs[j][1].x = t.z; cs[j][1].y = t.w;
      }
    }
  }
  const float2* srow = reinterpret_cast<const float2*>(lds_x);
  char* Chb = reinterpret_cast<char*>(Cht);
  const unsigned ob = (unsigned)(m16 * 2 * N + 8 * q4) * 2u;        // row m16, piece q4 of the wave's first group (hi; lo 64 B on)
  const unsigned rstep = (unsigned)N * 64u;                          // 16 rows of the pair buffer
  float amax = 0.0f;   // range guard
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i < gl || i >= gh) continue;
    const int row = 16 * i + m16;
    if (CHECK && mt0 + row >= M) continue;
    f2 sx = splat2(1.f), sy = splat2(0.f);
    if (FX & FX_LNF) {
      const float2 st = srow[rbase + row];
      sx = splat2(st.x); sy = splat2(st.y);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (CHECK && nt0 + 32 * c >= N) continue;
      f2 v[4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * c + jj;
        f2 a0, a1;
        a0.x = acc[i][j][0]; a0.y = acc[i][j][1]; a1.x = acc[i][j][2]; a1.y = acc[i][j][3];
        if (FX & FX_LNF) {   // LN(x) W^T + b = rstd (x W'^T) - rstd mean csum + b'
          v[2 * jj] = fma2(sx, a0 * P_OUT_SCALE, fma2(sy, cs[j][0], bb[j][0]));
          v[2 * jj + 1] = fma2(sx, a1 * P_OUT_SCALE, fma2(sy, cs[j][1], bb[j][1]));
        } else {
          v[2 * jj] = fma2(a0, splat2(P_OUT_SCALE), bb[j][0]);
          v[2 * jj + 1] = fma2(a1, splat2(P_OUT_SCALE), bb[j][1]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_fast2(v[e]);
      h8 oh, ol;
      split8_x3(v, P_A_SCALE, oh, ol, amax);
      *reinterpret_cast<h8*>(Chb + (ob + (unsigned)i * rstep + (unsigned)c * 128u)) = oh;
      *reinterpret_cast<h8*>(Chb + (ob + (unsigned)i * rstep + (unsigned)c * 128u) + 64u) = ol;
    }
    if (i & 1) __builtin_amdgcn_sched_barrier(0);
  }
  range_note(amax);
}

// Post-norm epilogue (FX_PN): the workgroup's tile is BM full rows (WM == 1, N == 64 WN), so the block's post-norm
//   y = LN(r + a W^T + b) [+ pos] [+ tvec]      (launch_layernorm's operations, two-pass variance)
// is applied before the rows leave the chip -- the fp32 round trip through HBM and the row kernel's launch are gone.
// Sweep 1 forms the new rows (8-column read-back as x3q_epilogue8) and keeps them in the registers the accumulators
// vacate; row statistics go through xch ([BM][WN] float2 of LDS beside the patches: one (sum, M2) partial per wave);
// sweep 2 normalises and stores planes + the (sum, sum of squares) partials of y for the next folded GEMM
// (OUTSPLIT 2) or fp32 rows (OUTSPLIT 0, last block).
template <int TM, int WN, int OUTSPLIT, bool CHECK>
__device__ __forceinline__ void x3q_epilogue_pn(f32x4 (&acc)[TM][4], float* patch, float* xch, const float* __restrict__ bias,
                                                float* Ct, _Float16* Cht, const _Float16* Rpt, const X3Tail& fx, int mt0, int nt0,
                                                int wn, int lane, int M, int N, int gl, int gh) {
  static_assert(WN == 8, "row partials are read back as four float4");
  const int m16 = lane & 15, q4 = lane >> 4;          // write side: accumulator layout
  const int rrow = lane >> 3, rc8 = lane & 7;          // read side
  const int n = nt0 + 8 * rc8;
  f2 bb[4];
  load8(bias + n, bb);
  const int pc = (int)pair_col(8 * rc8);
  constexpr int PF = TM < 3 ? TM : 3;                                  // residual window, see x3q_epilogue
  const unsigned ob = (unsigned)(rrow * N + 8 * rc8) * 4u;            // this lane's 8 floats in row rrow (fp32 buffer)
  const unsigned obp = (unsigned)(rrow * 2 * N + pc) * 2u;            // ... in a pair-layout buffer (hi; lo 64 B on)
  const unsigned rstep = (unsigned)N * 32u;                            // 8 rows of an fp32 or pair buffer
  const char* Rpb = reinterpret_cast<const char*>(Rpt);
  char* Cb = reinterpret_cast<char*>(Ct);
  char* Chb = reinterpret_cast<char*>(Cht);
  uint4 rh[TM][2], rl[TM][2];
  auto load_res = [&](int i) {
    if (i < gl || i >= gh) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      rh[i][p] = make_uint4(0, 0, 0, 0);
      rl[i][p] = make_uint4(0, 0, 0, 0);
      if (!CHECK || mt0 + 16 * i + rrow + 8 * p < M) {
        rh[i][p] = *reinterpret_cast<const uint4*>(Rpb + (obp + (unsigned)(2 * i + p) * rstep));
        rl[i][p] = *reinterpret_cast<const uint4*>(Rpb + (obp + (unsigned)(2 * i + p) * rstep) + 64u);
      }
    }
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < PF; ++i) load_res(i);
  __syncthreads();   // every wave is done with the operand stages: reuse LDS for the transpose patches
  f2 vv[TM][2][4];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
   if (i >= gl && i < gh) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(patch + (i & 1) * 1024 + m16 * 64 + (((4 * j + q4) ^ (m16 & 7)) << 2)) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    D3D_PATCH_FENCE();   // the strip is read back transposed: other lanes' rows
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = rrow + 8 * p;
      const float* prow = patch + (i & 1) * 1024 + row * 64;
      const float4 a0 = *reinterpret_cast<const float4*>(prow + (((2 * rc8) ^ (row & 7)) << 2));
      const float4 a1 = *reinterpret_cast<const float4*>(prow + (((2 * rc8 + 1) ^ (row & 7)) << 2));
      f2 a[4], r8[4];
      a[0].x = a0.x; a[0].y = a0.y; a[1].x = a0.z; a[1].y = a0.w; a[2].x = a1.x; a[2].y = a1.y; a[3].x = a1.z; a[3].y = a1.w;
      const bool ok = !CHECK || mt0 + 16 * i + row < M;
      unsplit8(rh[i][p], rl[i][p], r8);
      f2 (&v)[4] = vv[i][p];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fma2(r8[e], splat2(0.125f), fma2(a[e], splat2(P_OUT_SCALE), bb[e]));
        if (CHECK && !ok) v[e] = splat2(0.0f);
      }
      // this wave's 64 columns of the row: sum, and sum of squared deviations from their own mean (combined below by the
      // pairwise update formula -- as accurate as a two-pass variance, with one exchange)
      const f2 s2 = (v[0] + v[1]) + (v[2] + v[3]);
      const float sm = row8_sum(s2.x + s2.y);
      const f2 lm = splat2(sm * (1.0f / 64.0f));
      f2 d[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = v[e] - lm;
      const f2 q2 = fma2(d[0], d[0], d[1] * d[1]) + fma2(d[2], d[2], d[3] * d[3]);
      const float sq = row8_sum(q2.x + q2.y);
      if (rc8 == 0) *reinterpret_cast<float2*>(xch + 2 * ((16 * i + row) * WN + wn)) = make_float2(sm, sq);
    }
   }
    if (i + PF < TM) load_res(i + PF);
    __builtin_amdgcn_sched_barrier(0);
  }
  const float invn = 1.0f / (float)N;
  f2 gg[4], be[4], tv[4];
  load8(fx.pn.g + n, gg);
  load8(fx.pn.b + n, be);
  const bool tv_uniform = fx.pn.tvec != nullptr && fx.pn.tvec_stride == 0;
  const bool tv_rows = fx.pn.tvec != nullptr && fx.pn.tvec_stride != 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) tv[e] = splat2(0.0f);
  if (tv_uniform) load8(fx.pn.tvec + n, tv);
  const int npart = N >> 6;
  float amax = 0.0f;   // range guard
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i < gl || i >= gh) continue;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = 16 * i + rrow + 8 * p;
      const int m = mt0 + r;
      const float4* xr = reinterpret_cast<const float4*>(xch + 2 * r * WN);   // (sum, M2) of the row's 8 column blocks
      const float4 p0 = xr[0], p1 = xr[1], p2 = xr[2], p3 = xr[3];
      const float mean = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * invn;
      const float e0 = p0.x * (1.0f / 64.0f) - mean, e1 = p0.z * (1.0f / 64.0f) - mean, e2 = p1.x * (1.0f / 64.0f) - mean,
                  e3 = p1.z * (1.0f / 64.0f) - mean, e4 = p2.x * (1.0f / 64.0f) - mean, e5 = p2.z * (1.0f / 64.0f) - mean,
                  e6 = p3.x * (1.0f / 64.0f) - mean, e7 = p3.z * (1.0f / 64.0f) - mean;
      const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                       64.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
      const f2 rstd = splat2(1.0f / sqrtf(m2 * invn + fx.pn.eps)), mean2 = splat2(mean);
      f2 (&v)[4] = vv[i][p];
      if (CHECK && m >= M) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fma2((v[e] - mean2) * rstd, gg[e], be[e]);
      if (fx.pn.pos) {
        f2 t[4];
        load8(fx.pn.pos + (size_t)((m / fx.pn.pos_div) % fx.pn.pos_mod) * N + n, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += t[e];
      }
      if (tv_rows) {
        f2 t[4];
        load8(fx.pn.tvec + (size_t)(m / fx.pn.rows_per_batch) * fx.pn.tvec_stride + n, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += t[e];
      } else if (tv_uniform) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += tv[e];
      }
      if (OUTSPLIT == 2) {
        float sm, sq;
        sums8(v, sm, sq);
        sm = row8_sum(sm);
        sq = row8_sum(sq);
        if (rc8 == 0) *reinterpret_cast<float2*>(fx.st_out + 2 * ((size_t)m * npart + (nt0 >> 6))) = make_float2(sm, sq);
        h8 oh, ol;
        split8_x3(v, P_A_SCALE, oh, ol, amax);
        *reinterpret_cast<h8*>(Chb + (obp + (unsigned)(2 * i + p) * rstep)) = oh;
        *reinterpret_cast<h8*>(Chb + (obp + (unsigned)(2 * i + p) * rstep) + 64u) = ol;
      } else {
        *reinterpret_cast<float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep)) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
        *reinterpret_cast<float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep) + 16u) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (OUTSPLIT == 2) range_note(amax);
}

// Tile shapes: BM = 16*TM*WM rows, BN = 64*WN columns, WM x WN waves, each wave (16 TM) x 64 = TM x 4 MFMA tiles.
//   <8,2,4> 256x256, 8 waves of 128x64, 128 KiB LDS  -- large problems
//   <4,4,2> 256x128, 8 waves of  64x64,  96 KiB LDS  -- problems too small to fill the chip with 256x256 tiles
// Both put ONE 8-wave workgroup on a CU.  Every shape adds the same MFMA results in the same order into an output element,
// so an element's value does not depend on the tile shape that produced it (results are batch-size independent, bitwise).
// Shapes with 4-wave workgroups (<4,2,2>, <8,2,2>) or two workgroups per CU (<2,4,2>, <4,2,2>) stay instantiable for
// experiments/gemm_bench.py but are NOT used: they are slower, and at one stage of this round they gave run-to-run
// different results in about 1 of 1000 launches while a second process shared the GPU (a k-tile
// barrier without a vmcnt wait -- see D3D_QKTILE -- would produce exactly that, but the ISA of that stage was not kept).
// One output tile (device function: the launch wrapper below maps blockIdx to tiles).
// PERSIST (k_linear_x3q_persist): the workgroup walks several tiles.  Then (i) the first k-tile of a tile has already been
// staged (by the caller for the first tile, by the previous tile otherwise), (ii) the LAST k-tile of this tile -- which
// reads stage 1 when K/32 is even -- stages the first k-tile of the NEXT tile (m0n, n0n) into stage 0, so that it lands
// under the last MFMAs and the epilogue, (iii) the epilogue's transpose patches live in stage 1.
// FULL: the caller guarantees a whole tile inside the matrix (m0 + BM <= M, n0 + BN <= N): only the unchecked epilogue is
// instantiated -- the persistent walk sends ragged tiles through the SUB instantiation, so that its whole-tile path carries ONE
// branch-free epilogue (the row-statistics form used to run the checked copy -- an exec-mask branch around every residual load
// and store -- for every tile, because two copies under a run-time branch spilled accumulators).
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX, bool PERSIST = false, bool SUB = false, bool FULL = false>
__device__ __forceinline__ void x3q_tile(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                         const float* __restrict__ bias, const float* R, float* C, _Float16* Ch, _Float16* Cl,
                                         int M, int N, int K, int m0, int n0, int nt, int ntiles, int qcols,
                                         unsigned long long* diag, const X3Tail& fx, bool has_next = false, int m0n = 0,
                                         int n0n = 0, int tid_in = -1, int sub_wm = -1, int g_lo = 0, int g_hi = TM) {
  // SUB -- split tail tile (k_linear_x3q_persist): only the m-tiles [g_lo, g_hi) (g_lo even) of the waves in wave-row sub_wm (-1:
  // every wave-row) are computed and stored; the other waves still stage W pieces and meet the barriers.  A pieces outside
  // the computed rows are not staged (after the first k-tile, which the previous tile staged in full).
  // PERSIST passes the thread index behind an opaque barrier so that the per-lane offsets are re-derived in every tile
  // instead of being hoisted out of the tile loop and held in (spilled) registers
  const int tidx = PERSIST ? tid_in : (int)threadIdx.x;
  constexpr int NW = WM * WN, BM = 16 * TM * WM, BN = 64 * WN;
  constexpr int A_REG = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW, N_IT = A_IT + B_IT;   // 1-KiB DMA pieces per wave per k-tile
  constexpr int PPG = (N_IT + TM - 1) / TM;                                   // pieces issued per MFMA group
  static_assert(NW % 2 == 0 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must split evenly over the waves");
  static_assert(2 * STAGE >= NW * 2 * 16 * 64 * 4, "epilogue patches must fit in the operand stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int bid = blockIdx.x;
  // diag (diagnostic launches only, experiments/gemm_bench.py): shader-clock and 100 MHz stamps around the k-loop and the
  // epilogue of every workgroup, into a buffer nothing else reads
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (diag) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  unsigned char* const lds_x = lds + 2 * STAGE;   // beyond the operand stages (allocated only for the folded forms)
  if (FX & FX_LNF) {   // row statistics of the LayerNorm folded into this GEMM; visible after the first k-tile barrier
    if (tidx < BM) {
      const int row = m0 + tidx;
      float sm = 0.f, sq = 0.f;
      if (row < M)
        for (int p = 0; p < fx.st_np; ++p) {
          const float2 t = *reinterpret_cast<const float2*>(fx.st_in + 2 * ((size_t)row * fx.st_np + p));
          sm += t.x; sq += t.y;
        }
      // range guard for the producer of these rows (the proj / fc2 epilogues write the planes of x and these statistics of the
      // unclamped values): a clamped element |x| > 8188 implies sum x^2 > 8188^2 -- never missed; rows of ~512 values above ~360
      // would raise it falsely, a LayerNorm-ed stream is orders of magnitude below
      if (sq >= (X3_HALF_MAX * 0.125f) * (X3_HALF_MAX * 0.125f)) range_note(2.0f * X3_HALF_MAX);
      const float mean = sm / (float)K;
      const float var = fmaxf(sq / (float)K - mean * mean, 0.0f);
      const float rstd = 1.0f / sqrtf(var + fx.eps);
      reinterpret_cast<float2*>(lds_x)[tidx] = make_float2(rstd, -mean * rstd);
    }
  }

  const int tid = tidx;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, q = lane >> 4;
#if D3D_X3_YOUNG_PRIO
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);   // (experiment: static priority for the later-dispatched SIMD partners)
#endif
  // (a compile-time switch: with run-time ranges in the whole-tile path too, the whole GEMM ran 3.5 % slower)
  const bool w_act = !SUB || sub_wm < 0 || wm == sub_wm;          // wave-uniform
  const int gl = SUB ? (w_act ? g_lo : 0) : 0, gh = SUB ? (w_act ? g_hi : 0) : TM;
  unsigned amask = SUB ? 0u : ~0u;                                // A pieces of this wave that carry computed rows
  if (SUB) {
    const int a_lo = (sub_wm < 0 ? 0 : sub_wm) * 16 * TM + g_lo * 16, a_hi = (sub_wm < 0 ? WM - 1 : sub_wm) * 16 * TM + g_hi * 16;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int r0 = (it * NW + wave) * 8;
      if (r0 + 8 > a_lo && r0 < a_hi) amask |= 1u << it;
    }
  }

  D3D_DMA_PLAN(NW, BM);
#define D3D_QSTAGE_ONE(ST, KT, IT)                                                                                      \
  do {                                                                                                                  \
    if ((IT) < A_IT) {                                                                                                  \
      if ((amask >> (IT)) & 1u)                                                                                         \
        D3D_GLDS(sgpr_ptr(ubA + ((size_t)(KT) * 128 + (IT) * it_stride)) + lofs_, (ST) * STAGE + dstA + (IT) * NW * 1024); \
    } else                                                                                                              \
      D3D_GLDS(sgpr_ptr(ubB + ((size_t)(KT) * 128 + ((IT) - A_IT) * it_stride)) + lofs_,                               \
               (ST) * STAGE + dstB + ((IT) - A_IT) * NW * 1024);                                                        \
  } while (0)

  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

  // fragment offsets: rows r16 + 16 i all share the swizzle key r16>>1
  const int foff = (q ^ (r16 >> 1)) << 4;
  const int aoff = (wm * 16 * TM + r16) * 128 + foff, boff = A_REG + (wn * 64 + r16) * 128 + foff;
  const int nk = K / PBK;
  if (!PERSIST) {
#pragma unroll
    for (int it = 0; it < N_IT; ++it) D3D_QSTAGE_ONE(0, 0, it);
  }
  // next tile's operand bases (PERSIST): same per-lane offset, other uniform bases
  const char* ubAn = reinterpret_cast<const char*>(Ap) + (size_t)(m0n + wave * 8) * K2_ * 2;
  const char* ubBn = reinterpret_cast<const char*>(Wp) + (size_t)(n0n + wave * 8) * K2_ * 2;
#define D3D_QSTAGE_NEXT(IT)                                                                                             \
  do {                                                                                                                  \
    if ((IT) < A_IT) D3D_GLDS(sgpr_ptr(ubAn + (IT) * it_stride) + lofs_, dstA + (IT) * NW * 1024);                       \
    else D3D_GLDS(sgpr_ptr(ubBn + ((IT) - A_IT) * it_stride) + lofs_, dstB + ((IT) - A_IT) * NW * 1024);                 \
  } while (0)

  // one k-tile = TM groups (one 16-row m-tile each): the A fragments of group g+1 are read, and PPG DMA pieces of the
  // next k-tile are issued, before the 12 MFMAs of group g; the 8 W fragments are read once at the top of the k-tile.
#define D3D_QKTILE(KT, PREFETCH)                                                                                         \
  do {                                                                                                                   \
    /* this wave's DMA pieces of k-tile KT have landed before it meets the barrier: written out, not left to the fence */\
    /* of __syncthreads() -- in the SUB instantiation (uniform branches around the DMA issues) the compiler emitted no  */\
    /* vmcnt wait in the k-loop at all, and the slices read stale W rows (the last pieces issued)                        */\
    __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0) */                                                                 \
    __syncthreads();                                                                                                     \
    asm volatile("" : "+v"(lofs_)); /* keeps the lane offset out of the loop's pointer induction (saddr form) */         \
    const int nst = ((KT) + 1) & 1;                                                                                      \
    const unsigned char* sb = lds + ((KT) & 1) * STAGE;                                                                  \
    h8 bh[4], bl[4], ah[2], al[2];                                                                                       \
    ah[0] = *reinterpret_cast<const h8*>(sb + aoff + gl * 2048);                                                         \
    al[0] = *reinterpret_cast<const h8*>(sb + ((aoff + gl * 2048) ^ 64));                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
      bh[j] = *reinterpret_cast<const h8*>(sb + boff + j * 2048);                                                        \
      bl[j] = *reinterpret_cast<const h8*>(sb + ((boff + j * 2048) ^ 64));                                               \
    }                                                                                                                    \
    _Pragma("unroll") for (int g = 0; g < TM; ++g) {                                                                     \
      const bool g_act = g >= gl && g < gh;                                                                              \
      if (g_act && g + 1 < gh) {                                                                                         \
        ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + aoff + (g + 1) * 2048);                                      \
        al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + (g + 1) * 2048) ^ 64));                             \
      }                                                                                                                  \
      if (PREFETCH) {                                                                                                    \
        _Pragma("unroll") for (int pp = 0; pp < PPG; ++pp)                                                               \
          if (g * PPG + pp < N_IT) D3D_QSTAGE_ONE(nst, (KT) + 1, g * PPG + pp);                                          \
      } else if (PERSIST) {                                                                                              \
        if (has_next) {                                                                                                  \
          _Pragma("unroll") for (int pp = 0; pp < PPG; ++pp)                                                             \
            if (g * PPG + pp < N_IT) D3D_QSTAGE_NEXT(g * PPG + pp);                                                      \
        }                                                                                                                \
      }                                                                                                                  \
      if (g_act) {                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);                      \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);                      \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);                      \
        }                                                                                                                \
      }                                                                                                                  \
      if (D3D_X3_SGBQ == 2 && !SUB && g == 0) {   /* the k-tile opening (see D3D_X3_SGB 14) */                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                               \
      } else if (D3D_X3_SGBQ && !SUB) {                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                               \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
  } while (0)

#ifndef D3D_X3_PIPE2
#define D3D_X3_PIPE2 1
#endif
  // Measured (same box, experiments/ab_libs.sh, two alternations): qkv 1.122 -> 1.095 ms per launch, fc1 0.829 -> 0.830; proj 0.491 ->
  // 0.502 before the group order was stated (D3D_X3_SGB), 0.497 -> 0.492 with it; whole-row fc2 0.826 -> 0.889 (8 W pieces per wave
  // and phase): on for the 256 x 256 forms (D3D_X3_PIPE2=2 forces it everywhere, =0 nowhere -- experiments/build_variant.sh).
#ifndef D3D_X3_YOUNG_PRIO
#define D3D_X3_YOUNG_PRIO 0
#endif
// Instruction order inside an m-tile group (12 MFMAs, the next group's two A-fragment reads, 1-4 staging pieces), stated with
// sched_group_barrier: 2 MFMAs, a read, 2 MFMAs, a read, 2 MFMAs, a piece, 2 MFMAs, the other pieces, 4 MFMAs.  Left alone the
// scheduler puts the reads and the pieces at the top of the group and the 12 MFMAs behind them.  Same-box A/B (two alternations
// each): qkv 1.110 -> 1.084 ms, fc1 0.849 -> 0.830, fc2 + post-norm 0.851 -> 0.827, proj unchanged; patterns with the reads at
// the top and only the pieces spread (10), coarser ones (1, 3, 4, 7) or single MFMAs between the reads (5) gain less or nothing.
#ifndef D3D_X3_SGB
#define D3D_X3_SGB 14     // two-phase k-loop (qkv, proj, fc1): 2 = the pattern above in every group; 14 = also the k-tile opening group stated
                          // (A pair + first W pair, then a W pair ahead of each MFMA triple): proj 0.483 -> 0.479, fc1 0.823 -> 0.818
#endif
#ifndef D3D_X3_SGBQ
#define D3D_X3_SGBQ 2     // one-barrier k-loop (fc2 + post-norm; tail slices keep the scheduler's order): 1 = the pattern in every group, 2 = the
                          // k-tile opening group stated as well (fc2 0.838 -> 0.816 ms)
#endif
#ifndef D3D_X3_HPSTAG
#define D3D_X3_HPSTAG 0          // measured: qkv 1.10 -> 1.23 ms, fc1 0.84 -> 0.94 (+11 %): four barriers per k-tile cost more than the
#endif                           // hidden k-tile openings give back (the MFMA work per barrier interval is the same for both rows)
#ifndef D3D_X3_HPSTAG_ALL
#define D3D_X3_HPSTAG_ALL 0      // 1: the forms with a residual read (proj) too
#endif
  if constexpr (D3D_X3_HPSTAG != 0 && PERSIST && !SUB && TM == 8 && WM == 2 && WN == 4 && (D3D_X3_HPSTAG_ALL || EPI != EPI_RESIDUAL)) {
    // ---- Four steps per k-tile, the two wave rows ONE STEP APART (whole tiles of the persistent walk).  In-kernel stamps of the
    // two-phase form (experiments/gemm_bench.py with a -DD3D_X3_PHASE_DIAG library) show where a phase loses its time: the two
    // waves of a SIMD (w, w + 4) leave the barrier together, wait ~300 cycles for their first fragments together, then the OLDER
    // wave wins every MFMA arbitration, finishes its 48 MFMAs in ~1200 cycles and sits at the barrier for ~700 while the younger
    // one runs its last ~24 alone at ~20 cycles per MFMA (16 when the pipe is shared).  Here a k-tile is four steps of two m-tile
    // groups (24 MFMAs per wave), every step opens with a counted vmcnt wait + workgroup barrier, and wave row 1 (waves 4-7, the
    // SIMD partners of 0-3) runs one step BEHIND wave row 0: a wave opens its k-tile (8 W fragment reads + the first A pair, the
    // only reads that cannot be requested across a barrier) while its partner is in the middle of one.
    //   A operand: each wave row stages the 128 rows IT reads (pieces (w & 3) + 4 it of its band): A(t+1) in its own steps (t, 0),
    //              (t, 1) -- the band's previous contents, A(t-1), were last read in its own step (t-1, 3);
    //   W operand: all eight waves, W(t+2) in steps (t, 2), (t, 3) -- W(t) went to registers in step (t, 0) of either row;
    //   waits:     vmcnt(pieces of this wave's last two steps): a piece has two to three steps (~2000 cycles) to land; in the
    //              first k-tile of a tile (whose W(1) is staged late: its stage held the epilogue's patches) one step.
    // Same MFMAs in the same order per output element as every other form.
    auto wait_vm_ = [](int n) {
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      }
    };
    static_assert(A_IT == 4 && B_IT == 4, "four A and four W pieces per wave and k-tile");
    const int half = wm;                                                        // wave row = SIMD-partner half
    const char* ubAh = reinterpret_cast<const char*>(Ap) + (size_t)(m0 + 128 * half + 8 * (wave & 3)) * K2_ * 2;
    const char* ubAhn = reinterpret_cast<const char*>(Ap) + (size_t)(m0n + 128 * half + 8 * (wave & 3)) * K2_ * 2;
    const size_t itA_ = (size_t)32 * K2_ * 2;                                   // bytes between this wave's A pieces (32 rows)
    const int dstAh = (128 * half + 8 * (wave & 3)) * 128 + lane * 16;          // (+ 4096 per piece; the swizzle of lofs_ fits: piece parity = wave parity)
#define D3D_HP_A(KTT, IT)                                                                                                 \
    do {                                                                                                                  \
      const char* b_ = ((KTT) >= nk) ? ubAhn + (IT) * itA_ : ubAh + ((size_t)(KTT) * 128 + (IT) * itA_);                   \
      D3D_GLDS(sgpr_ptr(b_) + lofs_, ((KTT) & 1) * STAGE + dstAh + (IT) * 4096);                                          \
    } while (0)
#define D3D_HP_W(KTT, IT)                                                                                                 \
    do {                                                                                                                  \
      const char* b_ = ((KTT) >= nk) ? ubBn + (IT) * it_stride : ubB + ((size_t)(KTT) * 128 + (IT) * it_stride);           \
      D3D_GLDS(sgpr_ptr(b_) + lofs_, ((KTT) & 1) * STAGE + dstB + (IT) * NW * 1024);                                      \
    } while (0)
    h8 bh[4], bl[4], ah[2], al[2];
    int n1 = 0, n2 = 0;                                                          // pieces this wave issued in its last / last but one step
    // one step: Q = 0..3, m-tile groups 2Q, 2Q+1
#define D3D_HP(KT, Q, DO_A, DO_W1, DO_W, STRICT)                                                                          \
    do {                                                                                                                  \
      wait_vm_((STRICT) ? n1 : n1 + n2);                                                                                  \
      __builtin_amdgcn_s_barrier();                                                                                       \
      asm volatile("" : "+v"(lofs_) : : "memory");                                                                        \
      const unsigned char* sb = lds + ((KT) & 1) * STAGE;                                                                 \
      if ((Q) == 0) {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
          bh[j] = *reinterpret_cast<const h8*>(sb + boff + j * 2048);                                                     \
          bl[j] = *reinterpret_cast<const h8*>(sb + ((boff + j * 2048) ^ 64));                                            \
        }                                                                                                                 \
        ah[0] = *reinterpret_cast<const h8*>(sb + aoff);                                                                  \
        al[0] = *reinterpret_cast<const h8*>(sb + (aoff ^ 64));                                                           \
      }                                                                                                                   \
      int issued_ = 0;                                                                                                    \
      _Pragma("unroll") for (int g = 2 * (Q); g < 2 * (Q) + 2; ++g) {                                                     \
        if (g + 1 < TM) {     /* the next group's A pair: across the step barrier too (A(KT) landed before step 0) */     \
          ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + aoff + (g + 1) * 2048);                                     \
          al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + (g + 1) * 2048) ^ 64));                            \
        }                                                                                                                 \
        if ((Q) < 2) {                                                                                                    \
          if (DO_A) { D3D_HP_A((KT) + 1, 2 * (Q) + (g & 1)); ++issued_; }                                                 \
          if (DO_W1) { D3D_HP_W(1, 2 * (Q) + (g & 1)); ++issued_; }                                                       \
        } else if (DO_W) {                                                                                                \
          D3D_HP_W((KT) + 2, 2 * ((Q) - 2) + (g & 1));                                                                    \
          ++issued_;                                                                                                      \
        }                                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);                       \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);                       \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);                       \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
      }                                                                                                                   \
      n2 = n1;                                                                                                            \
      n1 = issued_;                                                                                                       \
    } while (0)
    if (half == 1) {      // wave row 1 sits out global step 0
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
      const bool do_a = (t + 1 < nk) || has_next;                       // A(t+1): of this tile, or k-tile 0 of the next one
      const bool do_w = (t + 2 < nk) || (t + 2 == nk && has_next);      // W(t+2): of this tile, or W(0) of the next one (W(1) waits)
      const bool first = t == 0;
      D3D_HP(t, 0, do_a, first, false, first);
      D3D_HP(t, 1, do_a, first, false, first);
      D3D_HP(t, 2, false, false, do_w, first);
      D3D_HP(t, 3, false, false, do_w, first);
    }
    if (half == 0) {      // wave row 0 sits out the last global step
      wait_vm_(n1 + n2);
      __builtin_amdgcn_s_barrier();
    }
#undef D3D_HP
#undef D3D_HP_A
#undef D3D_HP_W
  } else
  if constexpr (PERSIST && TM % 2 == 0 && (D3D_X3_PIPE2 > 1 || (D3D_X3_PIPE2 == 1 && WM == 2))) {
    // ---- Two phases per k-tile, staging two phases ahead (persistent walk).  A k-tile is split where its buffers die: the W
    // fragments go to registers at the top of the first phase, the A rows of m-tiles 0..TM/2-1 are read in the first phase, those
    // of TM/2..TM-1 in the second.  So the pieces of a LATER k-tile can be issued into a stage while its other half is still read:
    //   phase 2t   (m-tiles 0..TM/2-1 of k-tile t) issues A(t+1)  [its stage last held A(t-1), read through phase 2t-1]
    //   phase 2t+1 (m-tiles TM/2..TM-1)           issues W(t+2)  [its stage half held W(t), in registers since phase 2t]
    // and every piece has at least one whole phase to land: the wait before the barrier of a phase is a COUNTED vmcnt that leaves
    // the pieces of the phase just finished in flight (the one-barrier form waited vmcnt(0) for pieces issued a sixth of a k-tile
    // earlier).  The stream runs on across tiles: W(0) / A(0) of the next tile are issued by phases 2nk-3 / 2nk-2 of this one, its
    // W(1) -- whose stage holds the epilogue's patches -- with A(1) in its own phase 0.  Same MFMAs in the same order per element
    // as the one-barrier form (values unchanged).
#ifndef D3D_X3_DMASPREAD
#define D3D_X3_DMASPREAD 0   // 1: the waves issue their staging pieces at different points of an MFMA group (two waves per n-tile slot,
                             // SIMD partners in different slots) instead of all eight at its top: qkv 1.066 -> 1.094 ms, fc1 0.818 -> 0.831 (slower)
#endif
    const int jw_ = (wave + (wave >> 2)) & 3;      // SIMD partners (w, w + 4) get different slots
    auto wait_vm = [](int n) {   // s_waitcnt vmcnt(n), n wave-uniform (counts differ per wave only in tail slices)
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      }
    };
    const int nA = SUB ? __builtin_popcount(amask & ((1u << A_IT) - 1u)) : A_IT;   // A pieces this wave issues per k-tile
    int issued_prev = 0;                                                            // pieces this wave issued in the previous phase
    // piece `it` (A: 0..A_IT-1, W: A_IT..N_IT-1) of k-tile KTT of this tile (KTT < nk) or of k-tile 0 of the next one (KTT == nk)
#ifndef D3D_X3_DMA_OLD4
#define D3D_X3_DMA_OLD4 0   // 1 (whole tiles): waves 0-3 issue the staging pieces of their SIMD partners 4-7 as well (experiment)
#endif
    constexpr bool OLD4 = D3D_X3_DMA_OLD4 != 0 && !SUB && NW == 8;
    const size_t half_rows_ = (size_t)32 * K2_ * 2;      // bytes between the pieces of waves w and w + 4 (32 rows)
#define D3D_PIECE(KTT, IT)                                                                                               \
    do {                                                                                                                 \
      const bool nxt_ = (KTT) >= nk;                                                                                     \
      const int st_ = ((KTT) & 1) * STAGE;                                                                               \
      if (!OLD4 || wave < 4) {                                                                                           \
        if ((IT) < A_IT) {                                                                                               \
          if ((amask >> (IT)) & 1u) {                                                                                    \
            const char* b_ = nxt_ ? ubAn + (IT) * it_stride : ubA + ((size_t)(KTT) * 128 + (IT) * it_stride);            \
            D3D_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstA + (IT) * NW * 1024);                                               \
            if (OLD4) D3D_GLDS(sgpr_ptr(b_ + half_rows_) + lofs_, st_ + dstA + (IT) * NW * 1024 + 4096);                 \
          }                                                                                                              \
        } else {                                                                                                         \
          const char* b_ = nxt_ ? ubBn + ((IT) - A_IT) * it_stride : ubB + ((size_t)(KTT) * 128 + ((IT) - A_IT) * it_stride); \
          D3D_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstB + ((IT) - A_IT) * NW * 1024);                                        \
          if (OLD4) D3D_GLDS(sgpr_ptr(b_ + half_rows_) + lofs_, st_ + dstB + ((IT) - A_IT) * NW * 1024 + 4096);          \
        }                                                                                                                \
      }                                                                                                                  \
    } while (0)
    // one phase: H = 0 / 1.  Even phase (H = 0) of k-tile KT: A(KT+1) and the W pieces W_ODD..B_IT-1 of W(KT+1) if DO_A (all of
    // W(1) in a tile's first phase, W_FULL1); odd phase: the W pieces 0..W_ODD-1 of W(KT+2) if DO_W.  W_ODD = B_IT where A and W
    // are the same size (256 x 256 tiles: 4 + 4 pieces per wave and k-tile), B_IT / 2 for the whole-row tiles (2 + 8).
    constexpr int W_ODD = (WM == 2) ? B_IT : B_IT / 2;
#ifdef D3D_X3_PHASE_DIAG
#define D3D_PSTAMP(KT, H, I) do { if (bid == 3 && (KT) == 5) pst_[H][I] = __builtin_amdgcn_s_memtime(); } while (0)
    unsigned long long pst_[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
#else
#define D3D_PSTAMP(KT, H, I) do { } while (0)
#endif
#define D3D_PHASE(KT, H, DO_A, W_FULL1, DO_W)                                                                             \
    do {                                                                                                                  \
      D3D_PSTAMP(KT, H, 0);                                                                                               \
      wait_vm(issued_prev);                                                                                               \
      __builtin_amdgcn_s_barrier();                                                                                       \
      D3D_PSTAMP(KT, H, 1);                                                                                               \
      asm volatile("" : "+v"(lofs_) : : "memory");                                                                        \
      const unsigned char* sb = lds + ((KT) & 1) * STAGE;                                                                 \
      constexpr int G0 = (H) * (TM / 2), G1 = G0 + TM / 2;                                                                \
      const int gf = gl > G0 ? gl : G0;                     /* first active m-tile of this phase */                       \
      if (gf < gh && gf < G1) {                                                                                           \
        ah[gf & 1] = *reinterpret_cast<const h8*>(sb + aoff + gf * 2048);                                                 \
        al[gf & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + gf * 2048) ^ 64));                                        \
      }                                                                                                                   \
      if ((H) == 0) {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
          bh[j] = *reinterpret_cast<const h8*>(sb + boff + j * 2048);                                                     \
          bl[j] = *reinterpret_cast<const h8*>(sb + ((boff + j * 2048) ^ 64));                                            \
        }                                                                                                                 \
      }                                                                                                                   \
      D3D_PSTAMP(KT, H, 2);                                                                                               \
      _Pragma("unroll") for (int g = G0; g < G1; ++g) {                                                                   \
        const bool g_act = g >= gl && g < gh;                                                                             \
        if (g_act && g + 1 < gh && g + 1 < G1) {                                                                          \
          ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + aoff + (g + 1) * 2048);                                     \
          al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + (g + 1) * 2048) ^ 64));                            \
        }                                                                                                                 \
        constexpr int NP_ = ((H) == 0) ? A_IT + B_IT : W_ODD;      /* piece slots of this phase kind, spread over TM/2 groups */ \
        constexpr int PPG_ = (NP_ + TM / 2 - 1) / (TM / 2);                                                               \
        auto pieces_ = [&]() {                                                                                            \
          _Pragma("unroll") for (int pp = 0; pp < PPG_; ++pp) {                                                           \
            const int sl = (g - G0) * PPG_ + pp;                                                                          \
            if ((H) == 0) {                                                                                               \
              if (sl < A_IT) { if (DO_A) D3D_PIECE((KT) + 1, sl); }                                                       \
              else if (sl < A_IT + B_IT) {                                                                                \
                if ((W_FULL1) || ((DO_A) && sl - A_IT >= W_ODD)) D3D_PIECE((KT) + 1, sl);                                 \
              }                                                                                                           \
            } else if (sl < W_ODD) {                                                                                      \
              if (DO_W) D3D_PIECE((KT) + 2, A_IT + sl);                                                                   \
            }                                                                                                             \
          }                                                                                                               \
        };                                                                                                                \
        if (!D3D_X3_DMASPREAD || SUB) pieces_();                                                                          \
        if (g_act) {                                                                                                      \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                 \
            if (D3D_X3_DMASPREAD && !SUB) {                                                                               \
              if (j == jw_) pieces_();                                                                                    \
              __builtin_amdgcn_sched_barrier(0);                                                                          \
            }                                                                                                             \
            acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);                     \
            acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);                     \
            acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);                     \
          }                                                                                                               \
        }                                                                                                                 \
        if (D3D_X3_SGB == 14 && !SUB && (H) == 0 && g == G0) {   /* the k-tile opening: fragments just ahead of their MFMAs */\
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                              \
        } else if ((D3D_X3_SGB == 2 || D3D_X3_SGB == 14) && !SUB) {                                                       \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
        } else if (D3D_X3_SGB == 1 && !SUB) {   /* 4 MFMAs, the A-pair reads, 4 MFMAs, the staging pieces, 4 MFMAs */     \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
        } else if (D3D_X3_SGB == 2 && !SUB) {   /* finer: 2 MFMAs between single reads / pieces */                         \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
        } else if (D3D_X3_SGB == 5 && !SUB) {                                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                                              \
        } else if (D3D_X3_SGB == 6 && !SUB) {                                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
        } else if (D3D_X3_SGB == 7 && !SUB) {                                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
        } else if (D3D_X3_SGB == 8 && !SUB) {                                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
        } else if (D3D_X3_SGB == 9 && !SUB) {                                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
        } else if (D3D_X3_SGB == 10 && !SUB) {                                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                                              \
        } else if (D3D_X3_SGB == 11 && !SUB) {                                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
        } else if (D3D_X3_SGB == 12 && !SUB) {                                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
        } else if (D3D_X3_SGB == 13 && !SUB) {                                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
        } else if (D3D_X3_SGB == 3 && !SUB) {   /* 6 MFMAs, reads and pieces, 6 MFMAs */                                   \
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                                              \
        } else if (D3D_X3_SGB == 4 && !SUB) {   /* 8 MFMAs first, then reads, pieces, 4 MFMAs */                           \
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        D3D_PSTAMP(KT, H, 3 + (g - G0));                                                                                  \
      }                                                                                                                   \
      if ((H) == 0) issued_prev = ((DO_A) ? nA : 0) + ((W_FULL1) ? B_IT : ((DO_A) ? B_IT - W_ODD : 0));                   \
      else issued_prev = (DO_W) ? W_ODD : 0;                                                                              \
      if (OLD4) issued_prev = wave < 4 ? 2 * issued_prev : 0;                                                             \
    } while (0)
#ifndef D3D_X3_ASMREAD
#define D3D_X3_ASMREAD 0   // measured (same box, 3 alternations): qkv 1.136 -> 1.145 ms, fc1 0.875 -> 0.878: no gain -- the ~350 cycles behind
#endif                     // a phase's barrier are not what the k-loop loses (the staging instructions are: MI355X notes in DESIGN.md)
    // The same phase for whole tiles (!SUB) with the fragment reads as inline asm and counted waits:
    //   even phase: A(G0) pair and the 8 W fragments are requested behind the barrier; the first group's MFMAs for n-tile j wait
    //               for THEIR fragments only (lgkmcnt 8, 6, 4, 2 with the next A pair already requested behind them);
    //   every group requests the A pair of the next group first -- the last group of the even phase that of the ODD phase's first
    //   group (those rows landed with the rest of A(KT) before the even phase's barrier), so the odd phase opens with MFMAs.
    const unsigned lds_u = (unsigned)(uintptr_t)lds;
#define D3D_PHASE_A(KT, H, DO_A, W_FULL1, DO_W)                                                                           \
    do {                                                                                                                  \
      D3D_PSTAMP(KT, H, 0);                                                                                               \
      wait_vm(issued_prev);                                                                                               \
      __builtin_amdgcn_s_barrier();                                                                                       \
      D3D_PSTAMP(KT, H, 1);                                                                                               \
      asm volatile("" : "+v"(lofs_) : : "memory");                                                                        \
      const unsigned sb_ = lds_u + ((KT) & 1) * STAGE;                                                                    \
      const unsigned aH_ = sb_ + aoff, aL_ = sb_ + (aoff ^ 64), bH_ = sb_ + boff, bL_ = sb_ + (boff ^ 64);                \
      constexpr int G0 = (H) * (TM / 2), G1 = G0 + TM / 2;                                                                \
      if ((H) == 0) {                                                                                                     \
        lds_rd128(ah[G0 & 1], aH_ + G0 * 2048);                                                                           \
        lds_rd128(al[G0 & 1], aL_ + G0 * 2048);                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
          lds_rd128(bh[j], bH_ + j * 2048);                                                                               \
          lds_rd128(bl[j], bL_ + j * 2048);                                                                               \
        }                                                                                                                 \
      }                                                                                                                   \
      D3D_PSTAMP(KT, H, 2);                                                                                               \
      _Pragma("unroll") for (int g = G0; g < G1; ++g) {                                                                   \
        const bool pre_ = (g + 1 < G1) || ((H) == 0);       /* an A pair is requested for the next group */               \
        if (pre_) {                                                                                                       \
          lds_rd128(ah[(g + 1) & 1], aH_ + (g + 1) * 2048);                                                               \
          lds_rd128(al[(g + 1) & 1], aL_ + (g + 1) * 2048);                                                               \
        }                                                                                                                 \
        constexpr int NP_ = ((H) == 0) ? A_IT + B_IT : W_ODD;                                                             \
        constexpr int PPG_ = (NP_ + TM / 2 - 1) / (TM / 2);                                                               \
        _Pragma("unroll") for (int pp = 0; pp < PPG_; ++pp) {                                                             \
          const int sl = (g - G0) * PPG_ + pp;                                                                            \
          if ((H) == 0) {                                                                                                 \
            if (sl < A_IT) { if (DO_A) D3D_PIECE((KT) + 1, sl); }                                                         \
            else if (sl < A_IT + B_IT) {                                                                                  \
              if ((W_FULL1) || ((DO_A) && sl - A_IT >= W_ODD)) D3D_PIECE((KT) + 1, sl);                                   \
            }                                                                                                             \
          } else if (sl < W_ODD) {                                                                                        \
            if (DO_W) D3D_PIECE((KT) + 2, A_IT + sl);                                                                     \
          }                                                                                                               \
        }                                                                                                                 \
        if ((H) == 0 && g == G0) {                                                                                        \
          lgkm_wait4<8>(bh[0], bl[0], ah[g & 1], al[g & 1]);                                                              \
        } else if (pre_) {                                                                                                \
          lgkm_wait2<2>(ah[g & 1], al[g & 1]);                                                                            \
        } else {                                                                                                          \
          lgkm_wait2<0>(ah[g & 1], al[g & 1]);                                                                            \
        }                                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
          if ((H) == 0 && g == G0) {                                                                                      \
            if (j == 1) lgkm_wait2<6>(bh[1], bl[1]);                                                                      \
            if (j == 2) lgkm_wait2<4>(bh[2], bl[2]);                                                                      \
            if (j == 3) lgkm_wait2<2>(bh[3], bl[3]);                                                                      \
          }                                                                                                               \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);                       \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);                       \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);                       \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        D3D_PSTAMP(KT, H, 3 + (g - G0));                                                                                  \
      }                                                                                                                   \
      if ((H) == 0) issued_prev = ((DO_A) ? nA : 0) + ((W_FULL1) ? B_IT : ((DO_A) ? B_IT - W_ODD : 0));                   \
      else issued_prev = (DO_W) ? W_ODD : 0;                                                                              \
      if (OLD4) issued_prev = wave < 4 ? 2 * issued_prev : 0;                                                             \
    } while (0)
    h8 bh[4], bl[4], ah[2], al[2];
#ifndef D3D_X3_STAGGER
#define D3D_X3_STAGGER 0
#endif
    if constexpr (D3D_X3_STAGGER != 0 && !SUB && WM == 2 && TM == 8) {
      // ---- Staggered form (experiment, MI355X_MICROARCH "Two waves per SIMD" item 9): the waves of wave-row 1 (waves 4..7, the
      // SIMD partners of 0..3) run ONE PHASE behind those of wave-row 0, so that partners are never in the same kind of phase
      // start (fragment burst, staging issue) together.  Wave-row 1 takes its m-tiles in the order 4..7, 0..3: then in an even
      // global phase every wave runs the m-tile group 0..3 (row 0 on k-tile t, row 1 on k-tile t-1) and in an odd phase the
      // group 4..7 of k-tile t -- one instruction stream, the stage differs.  A tile takes 2 nk + 1 phases (row 1 idles in the
      // first, row 0 in the last).  Staging by global phase: even 2t: W(t+1) and the A rows of row 0's first half of k-tile t+1;
      // odd 2t+1: the other three A row bands of k-tile t+1.
      const int lag = wm;                                            // wave-uniform: 0 / 1
      const int nA0 = 1, nA1 = A_IT - 1;
      static_assert(A_IT == 4, "row bands of 64 rows");
#define D3D_SGROUPS(G0_, SB_)                                                                                             \
      do {                                                                                                                \
        ah[(G0_) & 1] = *reinterpret_cast<const h8*>((SB_) + aoff + (G0_) * 2048);                                        \
        al[(G0_) & 1] = *reinterpret_cast<const h8*>((SB_) + ((aoff + (G0_) * 2048) ^ 64));                               \
        _Pragma("unroll") for (int g = (G0_); g < (G0_) + 4; ++g) {                                                       \
          if (g + 1 < (G0_) + 4) {                                                                                        \
            ah[(g + 1) & 1] = *reinterpret_cast<const h8*>((SB_) + aoff + (g + 1) * 2048);                                \
            al[(g + 1) & 1] = *reinterpret_cast<const h8*>((SB_) + ((aoff + (g + 1) * 2048) ^ 64));                       \
          }                                                                                                               \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                 \
            acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);                     \
            acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);                     \
            acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);                     \
          }                                                                                                               \
          __builtin_amdgcn_sched_barrier(0);                                                                              \
        }                                                                                                                 \
      } while (0)
#define D3D_SWFRAGS(SB_)                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                     \
        bh[j] = *reinterpret_cast<const h8*>((SB_) + boff + j * 2048);                                                    \
        bl[j] = *reinterpret_cast<const h8*>((SB_) + ((boff + j * 2048) ^ 64));                                           \
      }
      issued_prev = 0;
      for (int t = 0; t <= nk; ++t) {
        {   // ---- even global phase 2t: m-tiles 0..3; wave-row 0 on k-tile t, wave-row 1 on k-tile t-1
          wait_vm(issued_prev);
          __builtin_amdgcn_s_barrier();
          asm volatile("" : "+v"(lofs_) : : "memory");
          const int my_t = t - lag;
          const bool issue = t < nk && (t + 1 < nk || has_next);
          if (issue) {   // W(t+1) and row band 0 of A(t+1)
#pragma unroll
            for (int it = 0; it < B_IT; ++it) D3D_PIECE(t + 1, A_IT + it);
            D3D_PIECE(t + 1, 0);
          }
          if (my_t >= 0 && my_t < nk) {
            const unsigned char* sb = lds + (my_t & 1) * STAGE;
            if (lag == 0) { D3D_SWFRAGS(sb); }
            D3D_SGROUPS(0, sb);
          }
          issued_prev = issue ? B_IT + nA0 : 0;
        }
        if (t < nk) {   // ---- odd global phase 2t+1: m-tiles 4..7 of k-tile t, every wave
          wait_vm(issued_prev);
          __builtin_amdgcn_s_barrier();
          asm volatile("" : "+v"(lofs_) : : "memory");
          const bool issue = t + 1 < nk || has_next;
          if (issue) {
#pragma unroll
            for (int it = 1; it < A_IT; ++it) D3D_PIECE(t + 1, it);
          }
          const unsigned char* sb = lds + (t & 1) * STAGE;
          if (lag == 1) { D3D_SWFRAGS(sb); }
          D3D_SGROUPS(4, sb);
          issued_prev = issue ? nA1 : 0;
        }
      }
#undef D3D_SGROUPS
#undef D3D_SWFRAGS
    } else {
    // first k-tile: everything issued before this tile (stores of the previous epilogue included) has landed: vmcnt(0)
#define D3D_SCHEDULE(PH)                                                                                                  \
    do {                                                                                                                  \
      issued_prev = 0;                                                                                                    \
      PH(0, 0, true, true, false);                                                                                        \
      PH(0, 1, false, false, nk > 2 || has_next);                                                                         \
      int kt = 1;                                                                                                         \
      for (; kt + 2 < nk; ++kt) {                                                                                         \
        PH(kt, 0, true, false, false);                                                                                    \
        PH(kt, 1, false, false, true);                                                                                    \
      }                                                                                                                   \
      if (nk > 2) { /* k-tile nk-2: A(nk-1) of this tile, then W(0) of the next tile */                                   \
        PH(kt, 0, true, false, false);                                                                                    \
        PH(kt, 1, false, false, has_next);                                                                                \
        ++kt;                                                                                                             \
      }                                                                                                                   \
      /* k-tile nk-1: A(0) of the next tile; W(1) of the next tile waits for its own phase 0 (the epilogue's patches) */   \
      PH(kt, 0, has_next, false, false);                                                                                  \
      PH(kt, 1, false, false, false);                                                                                     \
    } while (0)
    if constexpr (!SUB && D3D_X3_ASMREAD != 0) { D3D_SCHEDULE(D3D_PHASE_A); }
    else { D3D_SCHEDULE(D3D_PHASE); }
#undef D3D_SCHEDULE
#ifdef D3D_X3_PHASE_DIAG
    if (bid == 3 && lane == 0 && nk > 5 && has_next) {
      for (int hh = 0; hh < 2; ++hh)
        for (int i = 0; i < 8; ++i) g_x3_phase_diag[(wave * 2 + hh) * 8 + i] = pst_[hh][i];
    }
#endif
    }
#undef D3D_PHASE_A
#undef D3D_PHASE
#undef D3D_PIECE
  } else {
    int kt = 0;
    for (; kt + 1 < nk; ++kt) D3D_QKTILE(kt, true);
    D3D_QKTILE(kt, false);
  }
#undef D3D_QKTILE
#undef D3D_QSTAGE_ONE
#undef D3D_QSTAGE_NEXT

  const int mt0 = m0 + wm * 16 * TM, nt0 = n0 + wn * 64;          // wave-uniform
  const size_t tbase = (size_t)mt0 * N + nt0;
  const float* Rt = R ? R + tbase : nullptr;
  float* Ct = C ? C + tbase : nullptr;
  _Float16* Cht = Ch ? Ch + (OUTSPLIT == 2 ? 2 * tbase : tbase) : nullptr;
  _Float16* Clt = Cl ? Cl + tbase : nullptr;
  unsigned long long st_c1 = 0, st_r1 = 0;
  if (diag) { st_c1 = __builtin_amdgcn_s_memtime(); st_r1 = __builtin_amdgcn_s_memrealtime(); }
  float* patch = reinterpret_cast<float*>(lds + (PERSIST ? STAGE : 0)) + wave * (2 * 16 * 64);
  const _Float16* Rpt = (FX & FX_RP) ? fx.Rp + 2 * tbase : nullptr;
  constexpr bool PLANES = (OUTSPLIT != 0 || (FX & FX_RP)) && !(EPI == EPI_RESIDUAL && !(FX & FX_RP));
  const bool full = FULL || (m0 + BM <= M && n0 + BN <= N);
  bool done = false;
  if constexpr ((FX & FX_PN) != 0) {   // whole rows in the tile: post-norm here (the launcher guarantees N == BN)
    static_assert(WM == 1 && EPI == EPI_RESIDUAL && (FX & FX_RP) && OUTSPLIT != 1, "post-norm form");
    static_assert(STAGE >= 65536 + 2 * BM * WN * 4, "row-sum exchange beside the patches");
    float* xch = reinterpret_cast<float*>(lds + STAGE + 65536);
    // (one copy per instantiation: with a checked and an unchecked copy under a branch the accumulators spill)
    x3q_epilogue_pn<TM, WN, OUTSPLIT, !FULL>(acc, patch, xch, bias, Ct, Cht, Rpt, fx, mt0, nt0, wn, lane, M, N, gl, gh);
    done = true;
  } else if constexpr (EPI == EPI_GELU && OUTSPLIT == 2) {   // hidden activation, accumulator order: no transpose
    static_assert(!(FX & (FX_RP | FX_SO)), "fc1 form");
    if constexpr (FULL)
      x3q_epilogue_acc<TM, WM, WN, FX, false>(acc, lds_x, bias, Cht, fx.csum, mt0, nt0, mt0 - m0, lane, M, N, gl, gh);
    else if (full)
      x3q_epilogue_acc<TM, WM, WN, FX, false>(acc, lds_x, bias, Cht, fx.csum, mt0, nt0, mt0 - m0, lane, M, N, gl, gh);
    else
      x3q_epilogue_acc<TM, WM, WN, FX, true>(acc, lds_x, bias, Cht, fx.csum, mt0, nt0, mt0 - m0, lane, M, N, gl, gh);
    done = true;
  } else if constexpr (PLANES) {   // 8 columns per lane: 16-byte plane accesses
    if ((N & 7) == 0) {
      // (the row-statistics form keeps one copy per instantiation: with two copies under the branch its accumulators spill)
      if constexpr (FULL)
        x3q_epilogue8<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0, nt0,
                                                           mt0 - m0, lane, M, N, qcols, gl, gh);
      else if (full && !(FX & FX_SO))
        x3q_epilogue8<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0, nt0,
                                                           mt0 - m0, lane, M, N, qcols, gl, gh);
      else
        x3q_epilogue8<TM, WM, WN, EPI, OUTSPLIT, FX, true>(acc, patch, lds_x, bias, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0, nt0,
                                                          mt0 - m0, lane, M, N, qcols, gl, gh);
      done = true;
    }
  }
  if (!done) {
    if constexpr (FULL)
      x3q_epilogue<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Rt, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0,
                                                        nt0, mt0 - m0, lane, M, N, qcols, gl, gh);
    else if (full)
      x3q_epilogue<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Rt, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0,
                                                        nt0, mt0 - m0, lane, M, N, qcols, gl, gh);
    else
      x3q_epilogue<TM, WM, WN, EPI, OUTSPLIT, FX, true>(acc, patch, lds_x, bias, Rt, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0,
                                                       nt0, mt0 - m0, lane, M, N, qcols, gl, gh);
  }
  if (diag) {
    __builtin_amdgcn_s_waitcnt(0);   // the wave's own stores issued and acknowledged
    const unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* d = diag + ((size_t)bid * NW + wave) * 6;
      d[0] = st_c0; d[1] = st_r0; d[2] = st_c1; d[3] = st_r1; d[4] = c2; d[5] = r2;
    }
  }
}

// The SUB instantiation (tail slices, ragged edge tiles) as a real function: one copy per kernel, register-allocated on its
// own, so that what it spills (it carries the checked epilogues) stays out of the whole-tile path of the persistent walk.
// (measured as a real, non-inlined function -- one copy, register-allocated on its own --: the kernels then carry a scratch
// segment and the post-norm fc2 launch took 6 % longer; inlined it stays)
#ifdef D3D_X3_SUB_NOINLINE
#define D3D_SUB_ATTR __attribute__((noinline))
#else
#define D3D_SUB_ATTR __forceinline__
#endif
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX>
__device__ D3D_SUB_ATTR void x3q_tile_sub(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                                       const float* __restrict__ bias, const float* R, float* C, _Float16* Ch,
                                                       _Float16* Cl, int M, int N, int K, int m0, int n0, int nt, int ntiles,
                                                       int qcols, const X3Tail& fx, bool has_next, int m0n, int n0n, int tid_in,
                                                       int sub_wm, int g_lo, int g_hi) {
  x3q_tile<TM, WM, WN, EPI, OUTSPLIT, FX, true, true>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, m0, n0, nt, ntiles, qcols, nullptr, fx,
                                                      has_next, m0n, n0n, tid_in, sub_wm, g_lo, g_hi);
}

// Uniform launch: every workgroup one BM x BN tile; blockIdx -> tile keeps all N-tiles of an M-tile on one XCD.
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX>
__global__ __launch_bounds__(64 * WM * WN) void k_linear_x3q(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                                             const float* __restrict__ bias, const float* R, float* C,
                                                             _Float16* Ch, _Float16* Cl, int M, int N, int K, int mtiles,
                                                             int ntiles, int qcols, unsigned long long* diag, X3Tail fx) {
  constexpr int BM = 16 * TM * WM, BN = 64 * WN;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / ntiles) * 8 + xcd;
  const int nt = slot % ntiles;
  if (mt >= mtiles) return;
  x3q_tile<TM, WM, WN, EPI, OUTSPLIT, FX>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols, diag, fx);
}

// Persistent launch (one 8-wave workgroup per CU): the workgroup walks the tiles blockIdx, blockIdx + gridDim, ... of the
// uniform launch's order (so it stays on its XCD class, gridDim % 8 == 0).  Saves the per-tile workgroup relaunch and hides
// the first-k-tile staging latency of every tile but the first (x3q_tile, PERSIST).
// Tail: tiles = nfull * gridDim + rem.  The rem tiles of the last, partly filled round are cut into `split` (1, 2 or 4) row
// slices handled by split * rem <= gridDim workgroups (x3q_tile's sub_wm / g_lo / g_hi), so the round that would keep rem CUs
// busy for a whole tile time keeps split * rem CUs busy for a fraction of it.  A slice runs the same MFMAs in the same
// order for its rows as the whole tile would: values do not change.
struct X3Walk { int nfull, rem, split; unsigned long long* stamps; };   // stamps: diagnostic (100 MHz start / end per workgroup)

template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX>
__global__ __launch_bounds__(512) void k_linear_x3q_persist(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                                            const float* __restrict__ bias, const float* R, float* C,
                                                            _Float16* Ch, _Float16* Cl, int M, int N, int K, int mtiles,
                                                            int ntiles, int qcols, X3Walk wk, X3Tail fx) {
  constexpr int NW = WM * WN, BM = 16 * TM * WM, BN = 64 * WN;
  constexpr int A_IT = BM / 8 / NW, N_IT = (BM + BN) / 8 / NW;
  static_assert(NW == 8, "one 8-wave workgroup per CU");
  static_assert(TM % 4 == 0 && (WM == 1 || WM == 2), "tail slices");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int vfull = (mtiles / 8) * 8 * ntiles, mrem = mtiles % 8;
  // valid-tile ordinal -> tile: the uniform launch's blockIdx order without its padding slots
  auto tile_of = [&](int o, int& mt, int& nt) {
    if (o < vfull) {
      const int xcd = o & 7, slot = o >> 3;
      mt = (slot / ntiles) * 8 + xcd;
      nt = slot % ntiles;
    } else {
      const int o2 = o - vfull;
      mt = (mtiles / 8) * 8 + o2 % mrem;
      nt = o2 / mrem;
    }
  };
  const int nitems = wk.nfull + (b < wk.split * wk.rem ? 1 : 0);
  if (nitems == 0) return;
  const unsigned long long t_begin = wk.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto stamp_end = [&]() {
    if (wk.stamps && threadIdx.x == 0) {
      __builtin_amdgcn_s_waitcnt(0);
      wk.stamps[2 * b] = t_begin;
      wk.stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
    }
  };
  // item k of this workgroup: a whole tile (k < nfull) or a slice of a tail tile
  auto item_of = [&](int k, int& mt, int& nt, int& sub_wm, int& g_lo, int& g_hi) {
    sub_wm = -1; g_lo = 0; g_hi = TM;
    if (k < wk.nfull) {
      tile_of(k * G + b, mt, nt);
      return;
    }
    tile_of(wk.nfull * G + b / wk.split, mt, nt);
    const int sub = b % wk.split;
    if (wk.split == 2) {
      if (WM == 2) sub_wm = sub;
      else { g_lo = sub * (TM / 2); g_hi = g_lo + TM / 2; }
    } else if (wk.split == 4) {
      if (WM == 2) { sub_wm = sub >> 1; g_lo = (sub & 1) * (TM / 2); g_hi = g_lo + TM / 2; }
      else { g_lo = sub * (TM / 4); g_hi = g_lo + TM / 4; }
    }
  };
  int k = 0, mt = 0, nt = 0, sub_wm = -1, g_lo = 0, g_hi = TM;
  item_of(0, mt, nt, sub_wm, g_lo, g_hi);
  {   // stage the first k-tile of the first item (what x3q_tile does for itself in the uniform launch)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int m0 = mt * BM, n0 = nt * BN;
    D3D_DMA_PLAN(NW, BM);
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      if (it < A_IT) D3D_GLDS(sgpr_ptr(ubA + it * it_stride) + lofs_, dstA + it * NW * 1024);
      else D3D_GLDS(sgpr_ptr(ubB + (it - A_IT) * it_stride) + lofs_, dstB + (it - A_IT) * NW * 1024);
    }
  }
  int tid_o = (int)threadIdx.x;
  const int nwhole = (nitems > wk.nfull && wk.split > 1) ? wk.nfull : nitems;   // items run as whole tiles
  while (k < nwhole) {
    asm volatile("" : "+v"(tid_o));
    const bool has_next = k + 1 < nitems;
    int mtn = 0, ntn = 0, swn = -1, gln = 0, ghn = TM;
    if (has_next) item_of(k + 1, mtn, ntn, swn, gln, ghn);
#ifdef D3D_X3_NO_FULL   // (experiments: the round-1 walk -- one instantiation with run-time edge checks for every whole tile)
    if (true)
      x3q_tile<TM, WM, WN, EPI, OUTSPLIT, FX, true, false, false>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols,
                                                                  nullptr, fx, has_next, mtn * BM, ntn * BN, tid_o);
    else
#endif
    if ((mt + 1) * BM <= M && (nt + 1) * BN <= N)   // (wave-uniform) whole tile inside the matrix: the unchecked instantiation
      x3q_tile<TM, WM, WN, EPI, OUTSPLIT, FX, true, false, true>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols,
                                                                 nullptr, fx, has_next, mtn * BM, ntn * BN, tid_o);
    else                                            // ragged edge tile: the checked epilogue lives in the SUB instantiation
      x3q_tile_sub<TM, WM, WN, EPI, OUTSPLIT, FX>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols, fx, has_next,
                                                  mtn * BM, ntn * BN, tid_o, -1, 0, TM);
    if (!has_next) { stamp_end(); return; }
    ++k; mt = mtn; nt = ntn; sub_wm = swn; g_lo = gln; g_hi = ghn;
    __syncthreads();   // the epilogue's patches (stage 1) are read before the next tile's second k-tile is staged there
  }
  asm volatile("" : "+v"(tid_o));
  x3q_tile_sub<TM, WM, WN, EPI, OUTSPLIT, FX>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols, fx, false, 0, 0,
                                              tid_o, sub_wm, g_lo, g_hi);
  stamp_end();
}

// Diagnostic stamp buffer for the next launches (experiments/gemm_bench.py through d3d_op_linear_bench): variant 13 -> per-wave
// k-loop / epilogue stamps; the persistent walk -> start / end per workgroup.
static unsigned long long* g_x3_diag = nullptr;

// walk of `tiles` tiles over `grid` persistent workgroups
static X3Walk x3q_walk(int tiles, int grid) {
  static const bool split_on = getenv("D3D_X3_NO_TAILSPLIT") == nullptr;   // (switch for experiments/)
  X3Walk w{tiles / grid, tiles % grid, 1, g_x3_diag};
  static const int max_split = getenv("D3D_X3_TAILSPLIT") ? atoi(getenv("D3D_X3_TAILSPLIT")) : 4;   // (experiments/)
  if (split_on && w.rem > 0) w.split = (max_split >= 4 && 4 * w.rem <= grid) ? 4 : ((max_split >= 2 && 2 * w.rem <= grid) ? 2 : 1);
  return w;
}

template <int TM, int WM, int WN>
static hipError_t launch_x3q(const _Float16* Ap, const _Float16* Wp, const float* bias, const float* R, float* C, _Float16* Ch,
                             _Float16* Cl, int M, int N, int K, int epi, int outsplit, int qcols, hipStream_t s,
                             size_t lds_extra = 0, unsigned long long* diag = nullptr, const X3Fold* fold = nullptr) {
  constexpr int BM = 16 * TM * WM, BN = 64 * WN;
  const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
  const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
  static const size_t env_extra = getenv("D3D_X3_LDS_EXTRA") ? (size_t)atoi(getenv("D3D_X3_LDS_EXTRA")) : 0;   // (experiments/)
  size_t lds_bytes = 2 * (size_t)((BM + BN) * 128) + lds_extra + env_extra;   // lds_extra: occupancy experiments only
  X3Tail tail{};
  int fx = 0;
  if (fold) {
    if (fold->st_in) fx |= FX_LNF;
    if (fold->Rp) fx |= FX_RP;
    if (fold->st_out) fx |= FX_SO;
    tail.st_in = fold->st_in; tail.st_np = fold->st_np; tail.csum = fold->csum; tail.eps = fold->eps;
    tail.Rp = (const _Float16*)fold->Rp; tail.st_out = fold->st_out;
    if (fx & FX_LNF) lds_bytes += (size_t)BM * 8;   // row statistics beyond the operand stages
    if ((fx & FX_LNF) && (!fold->csum || fold->st_np < 1)) return hipErrorInvalidValue;
  }
#define D3D_X3Q_LAUNCH_FX(EPI_, OS_, FX_)                                                                                 \
  do {                                                                                                                    \
    auto kfn = k_linear_x3q<TM, WM, WN, EPI_, OS_, FX_>;                                                                  \
    static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                       \
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                   \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * WM * WN), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles,    \
                       ntiles, qcols, diag, tail);                                                                        \
  } while (0)
#define D3D_X3Q_LAUNCH(EPI_, OS_) D3D_X3Q_LAUNCH_FX(EPI_, OS_, 0)
  if (fx == 0) {
    D3D_X3_DISPATCH(D3D_X3Q_LAUNCH);
  } else if constexpr (WM * WN == 8) {   // folded forms exist for the production (8-wave) shapes only
    // the four folded forms of the engine's plane-resident block (engine.hip run_blocks)
    if (fx == FX_LNF && epi == EPI_NONE && outsplit == 1) D3D_X3Q_LAUNCH_FX(EPI_NONE, 1, FX_LNF);                        // qkv
    else if (fx == (FX_RP | FX_SO) && epi == EPI_RESIDUAL && outsplit == 2) D3D_X3Q_LAUNCH_FX(EPI_RESIDUAL, 2, FX_RP | FX_SO);  // proj
    else if (fx == FX_LNF && epi == EPI_GELU && outsplit == 2) D3D_X3Q_LAUNCH_FX(EPI_GELU, 2, FX_LNF);                   // fc1
    else if (fx == FX_RP && epi == EPI_RESIDUAL && outsplit == 0) D3D_X3Q_LAUNCH_FX(EPI_RESIDUAL, 0, FX_RP);             // fc2
    else return hipErrorInvalidValue;
  } else {
    return hipErrorInvalidValue;
  }
#undef D3D_X3Q_LAUNCH
#undef D3D_X3Q_LAUNCH_FX
  return hipGetLastError();
}

// Tile choice: 256x256 -- as a persistent walk, one workgroup per CU (k_linear_x3q_persist: +1 % over one workgroup per tile:
// the next tile's first k-tile lands under the epilogue; the tiles of the partly filled last round are cut into row slices,
// +1 %) -- wherever it fills the chip for a few rounds, else 256x128.
// History of the tail: a second launch of 64x256 tiles for the remainder rows gained nothing (launch gap); a first single
// launch carrying two tile shapes computed wrong, run-to-run different values.  Its cause was found later with the slices:
// in an instantiation whose DMA issues sit behind uniform branches the compiler emits NO vmcnt wait before the k-tile
// barrier (the fence of __syncthreads() normally provides it), so fragments were read before the last-issued pieces had
// landed.  D3D_QKTILE now states the wait itself.
static hipError_t launch_x3q_persist(const _Float16* Ap, const _Float16* Wp, const float* bias, const float* R, float* C,
                                     _Float16* Ch, _Float16* Cl, int M, int N, int K, int epi, int outsplit, int qcols,
                                     hipStream_t s, const X3Fold* fold) {
  const int mtiles = (M + 255) / 256, ntiles = (N + 255) / 256;
  const int tiles = mtiles * ntiles;
  int n_cu = device_cu_count() / 8 * 8;   // (per device)
  if (n_cu < 8) n_cu = 8;
  const int grid = tiles < n_cu ? tiles / 8 * 8 : n_cu;
  if (grid < 8) return hipErrorInvalidValue;
  const X3Walk wk = x3q_walk(tiles, grid);
  size_t lds_bytes = 2 * (size_t)(512 * 128);
  X3Tail tail{};
  int fx = 0;
  if (fold) {
    if (fold->st_in) fx |= FX_LNF;
    if (fold->Rp) fx |= FX_RP;
    if (fold->st_out) fx |= FX_SO;
    tail.st_in = fold->st_in; tail.st_np = fold->st_np; tail.csum = fold->csum; tail.eps = fold->eps;
    tail.Rp = (const _Float16*)fold->Rp; tail.st_out = fold->st_out;
    if (fx & FX_LNF) lds_bytes += (size_t)256 * 8;
    if ((fx & FX_LNF) && (!fold->csum || fold->st_np < 1)) return hipErrorInvalidValue;
  }
#define D3D_X3P_LAUNCH_FX(EPI_, OS_, FX_)                                                                                 \
  do {                                                                                                                    \
    auto kfn = k_linear_x3q_persist<8, 2, 4, EPI_, OS_, FX_>;                                                                      \
    static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                       \
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                   \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles, ntiles,     \
                       qcols, wk, tail);                                                                                  \
  } while (0)
#define D3D_X3P_LAUNCH(EPI_, OS_) D3D_X3P_LAUNCH_FX(EPI_, OS_, 0)
  if (fx == 0) {
    D3D_X3_DISPATCH(D3D_X3P_LAUNCH);
  } else {
    if (fx == FX_LNF && epi == EPI_NONE && outsplit == 1) D3D_X3P_LAUNCH_FX(EPI_NONE, 1, FX_LNF);
    else if (fx == (FX_RP | FX_SO) && epi == EPI_RESIDUAL && outsplit == 2) D3D_X3P_LAUNCH_FX(EPI_RESIDUAL, 2, FX_RP | FX_SO);
    else if (fx == FX_LNF && epi == EPI_GELU && outsplit == 2) D3D_X3P_LAUNCH_FX(EPI_GELU, 2, FX_LNF);
    else if (fx == FX_RP && epi == EPI_RESIDUAL && outsplit == 0) D3D_X3P_LAUNCH_FX(EPI_RESIDUAL, 0, FX_RP);
    else return hipErrorInvalidValue;
  }
#undef D3D_X3P_LAUNCH
#undef D3D_X3P_LAUNCH_FX
  return hipGetLastError();
}

// Post-norm form (X3Fold::pn): 128 x 512 tiles (<8,1,8>: eight waves of 128 x 64 side by side, 2 x 80 KiB of LDS -- the whole
// CU), so that a workgroup owns whole rows.  Per k-tile it stages 80 KiB for the MFMA work a 256x256 tile does on 64 KiB;
// what it saves is the fp32 round trip of the stream through HBM and the row kernel behind the fc2 GEMM.
// The same tile function serves every M (persistent walk when the chip is filled for a few rounds, one workgroup per tile
// otherwise), so the result stays batch-size independent bitwise.
bool x3q_postnorm_ok(int N, int K) { return N == 512 && K % PBK == 0; }

static hipError_t launch_x3q_pn(const _Float16* Ap, const _Float16* Wp, const float* bias, float* C, _Float16* Ch, int M, int N,
                                int K, int outsplit, hipStream_t s, const X3Fold* fold) {
  if (!x3q_postnorm_ok(N, K) || !fold->Rp || !fold->pn.b || !bias || (outsplit == 2 ? (!Ch || !fold->st_out) : !C))
    return hipErrorInvalidValue;
  if (outsplit != 0 && outsplit != 2) return hipErrorInvalidValue;
  if (fold->pn.pos && (fold->pn.pos_div < 1 || fold->pn.pos_mod < 1)) return hipErrorInvalidValue;
  if (fold->pn.tvec && fold->pn.tvec_stride != 0 && fold->pn.rows_per_batch < 1) return hipErrorInvalidValue;
  const int mtiles = (M + 127) / 128, ntiles = 1;
  const int vtiles = ((mtiles + 7) / 8) * 8;
  int n_cu = device_cu_count() / 8 * 8;   // (per device)
  if (n_cu < 8) n_cu = 8;
  static const bool persist_on = getenv("D3D_X3_NO_PERSIST") == nullptr;
  const bool persist = persist_on && mtiles >= 4 * n_cu && (K / PBK) % 2 == 0;
  const X3Walk wk = x3q_walk(mtiles * ntiles, n_cu);
  const size_t lds_bytes = 2 * (size_t)((128 + 512) * 128);
  static const bool small_on = getenv("D3D_PN_NO_SMALL") == nullptr;   // (switch for experiments/)
  const bool small = small_on && mtiles < n_cu;
  const int mtiles64 = (M + 63) / 64, vtiles64 = ((mtiles64 + 7) / 8) * 8;
  const size_t lds_small = 2 * (size_t)((64 + 512) * 128);
  X3Tail tail{};
  tail.Rp = (const _Float16*)fold->Rp; tail.st_out = fold->st_out; tail.pn = fold->pn;
  const int qcols = 0;
  unsigned long long* diag = nullptr;
  _Float16* Cl = nullptr;
  const float* R = nullptr;
#define D3D_X3PN_LAUNCH(OS_)                                                                                              \
  do {                                                                                                                    \
    if (persist) {                                                                                                        \
      auto kfn = k_linear_x3q_persist<8, 1, 8, EPI_RESIDUAL, OS_, FX_RP | FX_PN>;                                         \
      static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                     \
      if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                 \
      hipLaunchKernelGGL(kfn, dim3(n_cu), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles, ntiles,   \
                         qcols, wk, tail);                                                                                \
    } else if (small) {   /* fewer 128-row tiles than CUs: 64-row tiles (same values: rows are independent) */            \
      auto kfn = k_linear_x3q<4, 1, 8, EPI_RESIDUAL, OS_, FX_RP | FX_PN>;                                                 \
      static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                     \
      if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_small, attr_done)) return ae;                 \
      hipLaunchKernelGGL(kfn, dim3(vtiles64), dim3(512), lds_small, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles64, ntiles, \
                         qcols, diag, tail);                                                                              \
    } else {                                                                                                              \
      auto kfn = k_linear_x3q<8, 1, 8, EPI_RESIDUAL, OS_, FX_RP | FX_PN>;                                                 \
      static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                     \
      if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                 \
      hipLaunchKernelGGL(kfn, dim3(vtiles), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles, ntiles, \
                         qcols, diag, tail);                                                                              \
    }                                                                                                                     \
  } while (0)
  if (outsplit == 2) D3D_X3PN_LAUNCH(2);
  else D3D_X3PN_LAUNCH(0);
#undef D3D_X3PN_LAUNCH
  return hipGetLastError();
}

static bool x3q_big(int M, int N) {
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
  return N % 256 == 0 && tiles >= 4 * 256;
}
int x3q_ntiles(int M, int N) { (void)M; return (N + 63) / 64; }   // statistics partials per row: one per 64 columns

static hipError_t launch_x3q_auto(const _Float16* ap, const _Float16* wp, const float* bias, const float* R, float* C,
                                  _Float16* ch, _Float16* cl, int M, int N, int K, int epi, int outsplit, int qcols,
                                  hipStream_t s, const X3Fold* fold) {
  static const bool persist = getenv("D3D_X3_NO_PERSIST") == nullptr;   // (switch for experiments/)
  if (x3q_big(M, N) && persist && (K / PBK) % 2 == 0)
    return launch_x3q_persist(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, fold);
  if (x3q_big(M, N)) return launch_x3q<8, 2, 4>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, 0, nullptr, fold);
  return launch_x3q<4, 4, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, 0, nullptr, fold);
}

void set_linear_x3_diag(unsigned long long* dev_buf) { g_x3_diag = dev_buf; }

hipError_t range_flags_gemm(unsigned* flags, bool clear) {
  hipError_t e = hipMemcpyFromSymbol(flags, HIP_SYMBOL(g_range_x3p), sizeof(unsigned));
  const unsigned zero = 0;
  if (e == hipSuccess && clear && *flags) e = hipMemcpyToSymbol(HIP_SYMBOL(g_range_x3p), &zero, sizeof(unsigned));
  return e;
}

// variant: 0 = auto (launch_x3q_auto).  experiments only (gemm_bench.py, two_rank_repeat.sh via D3D_X3_VARIANT):
// 13 = 256x256, 4 = 256x128 (8 waves), 5 = 128x128 (8 waves, 2/CU), 7 = 256x128 (4 waves), 10 = 128x128 (4 waves, 2/CU),
// 8 = 10 at one workgroup per CU
hipError_t launch_linear_x3p(const void* Ap_, const void* Wp_, const float* bias, const float* R, float* C, void* Ch, void* Cl,
                             int M, int N, int K, int epi, int outsplit, int qcols, int variant, hipStream_t s,
                             const X3Fold* fold) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % PBK) != 0 || (N % 4) != 0) return hipErrorInvalidValue;
  if (epi == EPI_RESIDUAL && R == nullptr && !(fold && fold->Rp)) return hipErrorInvalidValue;
  if (fold && variant != 0) return hipErrorInvalidValue;
  if (outsplit == 0 ? !C : outsplit == 1 ? (!Ch || !Cl) : (!Ch || (N % 32) != 0)) return hipErrorInvalidValue;
  const _Float16 *ap = (const _Float16*)Ap_, *wp = (const _Float16*)Wp_;
  _Float16 *ch = (_Float16*)Ch, *cl = (_Float16*)Cl;
  if (variant == 0) {
    static const char* ov = getenv("D3D_X3_VARIANT");   // experiments only
    if (ov) variant = atoi(ov);
  }
  if (fold && fold->pn.g) {
    if (epi != EPI_RESIDUAL) return hipErrorInvalidValue;
    return launch_x3q_pn(ap, wp, bias, C, ch, M, N, K, outsplit, s, fold);
  }
  switch (variant) {
    case 0: return launch_x3q_auto(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, fold);
    case 13: return launch_x3q<8, 2, 4>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, 0, g_x3_diag);
    case 4: return launch_x3q<4, 4, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s);
    case 5: return launch_x3q<2, 4, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s);
    case 7: return launch_x3q<8, 2, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s);
    case 8: return launch_x3q<4, 2, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, 32 * 1024);
    case 10: return launch_x3q<4, 2, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s);
    default: return hipErrorInvalidValue;
  }
}

// fp32 [rows, cols] -> pair layout of 8*x (stand-alone converter: tests, and any activation whose producer is not ours)
__global__ __launch_bounds__(256) void k_split_x3(const float* __restrict__ x, _Float16* __restrict__ pair, size_t n4, int cols) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const size_t row = (4 * i) / cols;
  const int c = (int)(4 * i - row * cols);
  const float f[4] = {v.x, v.y, v.z, v.w};
  h4 a, b;
  float amax = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    amax = __builtin_fmaxf(amax, __builtin_fabsf(f[j] * P_A_SCALE));
    const float s = __builtin_amdgcn_fmed3f(f[j] * P_A_SCALE, -65504.0f, 65504.0f);
    a[j] = (_Float16)s;
    b[j] = (_Float16)(s - (float)a[j]);
  }
  range_note(amax);
  _Float16* p = pair + row * 2 * cols + pair_col(c);
  *reinterpret_cast<h4*>(p) = a;
  *reinterpret_cast<h4*>(p + PAIR_LO) = b;
}

hipError_t launch_split_x3(const float* x, void* pair, size_t rows, int cols, hipStream_t s) {
  if (cols <= 0 || cols % 32) return hipErrorInvalidValue;
  const size_t n4 = rows * cols / 4;
  if (n4 == 0) return hipSuccess;
  hipLaunchKernelGGL(k_split_x3, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, (_Float16*)pair, n4, cols);
  return hipGetLastError();
}

// pair layout of 8*x -> fp32 [rows, cols], and the row totals of (sum, sum of squares) partials: read-back side of the
// plane-form op hooks (d3d_op_linear_postnorm); not on the engine's path
__global__ __launch_bounds__(256) void k_unsplit_x3(const _Float16* __restrict__ pair, float* __restrict__ x, size_t n, int cols,
                                                    const float* __restrict__ part, int np, float* __restrict__ stats,
                                                    size_t rows) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const size_t row = i / cols;
    const int c = (int)(i - row * cols);
    const _Float16* p = pair + row * 2 * cols + pair_col(c);
    x[i] = ((float)p[0] + (float)p[PAIR_LO]) * 0.125f;
  }
  if (stats && i < rows) {
    float sm = 0.f, sq = 0.f;
    for (int k = 0; k < np; ++k) { sm += part[2 * (i * np + k)]; sq += part[2 * (i * np + k) + 1]; }
    stats[2 * i] = sm; stats[2 * i + 1] = sq;
  }
}

hipError_t launch_unsplit_x3(const void* pair, float* x, size_t rows, int cols, const float* part, int np, float* stats,
                             hipStream_t s) {
  if (cols <= 0 || cols % 32) return hipErrorInvalidValue;
  const size_t n = rows * cols;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_unsplit_x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const _Float16*)pair, x, n, cols, part, np,
                     stats, rows);
  return hipGetLastError();
}

}  // namespace d3d
