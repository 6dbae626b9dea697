// Epilogues of the F16X3 token GEMM (kernels_gemm_x3p.hip includes this file inside namespace d3d, after the operand typedefs,
// the range guard and D3D_PATCH_FENCE): bias / LayerNorm fold / GELU / residual / row statistics / post-norm on the accumulator
// tile of one wave, and the hi/lo split of what goes out as planes.  Reference ops: S2S:46-54 (Mlp), 67/84 (qkv, proj),
// 113-135 (Block), 236/245 (post-norms).
#pragma once

#include "x3q_epilogue_acc.h"   // f2 / fma2 / gelu_fast2 / split8_x3, the FX flags, x3q_epilogue_acc (fc1)

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


// 8 values -> bf16 (round to nearest even: v_cvt_pk_bf16_f32) of osc * v
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint4 cvt8_bf16(const f2 (&v)[4], float osc) {
  bf8v o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f2 sc = v[e] * osc;
    o[2 * e] = (__bf16)sc.x;
    o[2 * e + 1] = (__bf16)sc.y;
  }
  return __builtin_bit_cast(uint4, o);
}

// sum over the 16 lanes of a DPP row (all 16 lanes get the total): quad xor 1, xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

struct X3Tail {            // per-launch extras (device copy of X3Fold + derived)
  float out_scale;         // 2^-(3 + k): un-scales the accumulators (A planes hold 8 a, W planes 2^k w; k = 12 unless |w| > 15.99)
  const float* st_in; int st_np; const float* csum; float eps;
  const _Float16* Rp;
  float* st_out;
  X3PostNorm pn;
  unsigned* range;         // the launching engine's range-guard word (d3d_kernels.h)
};

// lds_x: the workgroup's LDS beyond the operand stages: [BM] float2 row statistics (FX_LNF)
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX, bool CHECK>
__device__ __forceinline__ void x3q_epilogue(f32x4 (&acc)[TM][4], float* patch, unsigned char* lds_x, const float* __restrict__ bias,
                                             const float* Rt, float* Ct, _Float16* Cht, _Float16* Clt, const _Float16* Rpt,
                                             const float* __restrict__ csum, float* st_out, int mt0, int nt0, int rbase, int lane,
                                             int M, int N, int qcols, int gl, int gh, const float P_OUT_SCALE, unsigned* rw) {
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
        v[0] = fmaf(st.x, a4.x, fmaf(st.y, cs4.x, b4.x));   // (st.x = rstd * out_scale: x3_row_stats)
        v[1] = fmaf(st.x, a4.y, fmaf(st.y, cs4.y, b4.y));
        v[2] = fmaf(st.x, a4.z, fmaf(st.y, cs4.z, b4.z));
        v[3] = fmaf(st.x, a4.w, fmaf(st.y, cs4.w, b4.w));
      } else {
        v[0] = a4.x * P_OUT_SCALE + b4.x; v[1] = a4.y * P_OUT_SCALE + b4.y;
        v[2] = a4.z * P_OUT_SCALE + b4.z; v[3] = a4.w * P_OUT_SCALE + b4.w;
      }
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {   // (the one GELU of all epilogue forms: an element has the same bits whichever it goes through)
          f2 g;
          g.x = v[e]; g.y = v[e + 1];
          g = gelu_fast2(g);
          v[e] = g.x; v[e + 1] = g.y;
        }
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
  if constexpr (OUTSPLIT != 0 && !(FX & FX_SO)) range_note(rw, amax);
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
                                              int M, int N, int qcols, int gl, int gh, const float P_OUT_SCALE, unsigned* rw,
                                              float2* statp = nullptr) {
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
  const float osc = (OUTSPLIT == 3) ? ((n < qcols) ? 0.125f : 1.0f) : ((n < qcols) ? 1.0f : P_A_SCALE);   // (bf16 planes are unscaled)
  const int pc = (int)pair_col(8 * rc8);
  const float2* srow = reinterpret_cast<const float2*>(lds_x);
  const int npart = (N + 63) >> 6;
  // (the row-statistics form at 256 rows: 2 and 3 measure alike, 4 spills more; at 192 rows -- 232 VGPRs -- 4 fits: proj -0.5 %, 6 spills: +2.3 %)
  constexpr int PFMAX = (FX & FX_SO) ? (TM == 6 ? 4 : 3) : 4;
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
        for (int e = 0; e < 4; ++e) v[e] = fma2(sx, a[e], fma2(sy, cs[e], bb[e]));   // (sx = rstd * out_scale)
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
        // the row's partial goes to a wave-private LDS row first: twelve 8-byte stores of 8 lanes each per wave and tile cost the wave as
        // many vector-memory issue slots as its 16-byte plane stores do; they leave below as two instructions of 64 / 32 rows
        if (statp) { if (rc8 == 0) statp[16 * i + row] = make_float2(sm, sq); }
        else if (rc8 == 0 && m < M) *reinterpret_cast<float2*>(st_out + 2 * ((size_t)m * npart + (nt0 >> 6))) = make_float2(sm, sq);
        if (!ok) continue;
      }
      if constexpr (OUTSPLIT == 3) {
        *reinterpret_cast<uint4*>(Chb + (obh + (unsigned)(2 * i + p) * rsteph)) = cvt8_bf16(v, osc);
      } else if (OUTSPLIT) {
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
  if constexpr ((FX & FX_SO) != 0) {
    if (statp) {
      D3D_PATCH_FENCE();   // (rows written by other lanes)
#pragma unroll
      for (int r0 = 0; r0 < 16 * TM; r0 += 64) {
        const int rr = r0 + lane;
        if (rr < 16 * TM && (rr >> 4) >= gl && (rr >> 4) < gh && mt0 + rr < M)
          *reinterpret_cast<float2*>(st_out + 2 * ((size_t)(mt0 + rr) * npart + (nt0 >> 6))) = statp[rr];
      }
    }
  }
  if constexpr (OUTSPLIT != 0 && OUTSPLIT != 3 && !(FX & FX_SO)) range_note(rw, amax * osc);
}

// Post-norm epilogue (FX_PN): the workgroup's tile is BM full rows (WM == 1, N == 64 WN), so the block's post-norm
//   y = LN(r + a W^T + b) [+ pos] [+ tvec]      (launch_layernorm's operations, two-pass variance)
// is applied before the rows leave the chip -- the fp32 round trip through HBM and the row kernel's launch are gone.
// Sweep 1 forms the new rows (8-column read-back as x3q_epilogue8) and keeps them in the registers the accumulators
// vacate; row statistics go through xch ([BM][WN] float2 of LDS beside the patches: one (sum, M2) partial per wave);
// sweep 2 normalises and stores planes + the (sum, sum of squares) partials of y for the next folded GEMM
// (OUTSPLIT 2) or fp32 rows (OUTSPLIT 0, last block).
// RING (kernels_fc2_ring.hip): the patches are the calling wave's OWN operand slot and the k-loop has no workgroup barrier, so (i) there
// is no entry barrier, (ii) the barrier between the sweeps waits for LDS operations only (the caller keeps staging DMA in flight across
// it), (iii) `hook` runs between sweep 1 and that barrier -- the patches are dead from there on (the caller re-fills its slot).
struct X3NoHook { __device__ __forceinline__ void operator()() const {} };
template <int TM, int WN, int OUTSPLIT, bool CHECK, bool RING = false, class Hook = X3NoHook>
__device__ __forceinline__ void x3q_epilogue_pn(f32x4 (&acc)[TM][4], float* patch, float* xch, const float* __restrict__ bias,
                                                float* Ct, _Float16* Cht, const _Float16* Rpt, const X3Tail& fx, int mt0, int nt0,
                                                int wn, int lane, int M, int N, int gl, int gh, Hook hook = Hook()) {
  static_assert(WN == 8, "row partials are read back as four float4");
  const float P_OUT_SCALE = fx.out_scale;
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
  if constexpr (!RING) __syncthreads();   // every wave is done with the operand stages: reuse LDS for the transpose patches
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
      const f2 lm = splat2_rt(sm * (1.0f / 64.0f));
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
  if constexpr (RING) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's patch reads and exchange writes are done
    hook();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  } else {
    __syncthreads();
  }
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
      const f2 rstd = splat2_rt(1.0f / sqrtf(m2 * invn + fx.pn.eps)), mean2 = splat2_rt(mean);
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
  if (OUTSPLIT == 2) range_note(fx.range, amax * P_A_SCALE);
}

// Whole-row epilogue of the bf16 operand mode (FX_PN | FX_BF16; x3q_epilogue_pn's structure with an fp32 residual stream):
//   sweep 1  v = R + acc + bias (R = the fp32 stream, read through a 3-m-tile register window), per-wave (sum, M2) -> xch
//   sweep 2  y = pn.g ? LN(v) [+ pos] [+ tvec] : v;  y -> C (= R: in place);  if pn.g and pn.g2: (sum, M2) of y -> xch2
//   sweep 3  if pn.g2: bf16(LN(y; g2, b2)) -> Cb      (statistics from xch2, or from xch when there was no first norm)
// Rows stay in the 128 registers the accumulators vacate; the un-normalised sum never makes a second trip through HBM.
template <int TM, int WN, bool CHECK>
__device__ __forceinline__ void x3q_epilogue_rows_bf16(f32x4 (&acc)[TM][4], float* patch, float* xch, float* xch2,
                                                       const float* __restrict__ bias, float* Ct, _Float16* Cbt, const X3Tail& fx,
                                                       int mt0, int nt0, int wn, int lane, int M, int N, int gl, int gh) {
  static_assert(WN == 8, "row partials are read back as four float4");
  const int m16 = lane & 15, q4 = lane >> 4;          // write side: accumulator layout
  const int rrow = lane >> 3, rc8 = lane & 7;          // read side
  const int n = nt0 + 8 * rc8;
  f2 bb[4];
  load8(bias + n, bb);
  constexpr int PF = TM < 3 ? TM : 3;
  const unsigned ob = (unsigned)(rrow * N + 8 * rc8) * 4u;            // this lane's 8 floats in row rrow (fp32 buffer)
  const unsigned obh = (unsigned)(rrow * N + 8 * rc8) * 2u;           // ... in the bf16 buffer
  const unsigned rstep = (unsigned)N * 32u, rsteph = (unsigned)N * 16u;   // 8 rows
  char* Cb = reinterpret_cast<char*>(Ct);
  char* Hb = reinterpret_cast<char*>(Cbt);
  float4 r0[TM][2], r1[TM][2];
  auto load_res = [&](int i) {
    if (i < gl || i >= gh) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      r0[i][p] = make_float4(0, 0, 0, 0);
      r1[i][p] = make_float4(0, 0, 0, 0);
      if (!CHECK || mt0 + 16 * i + rrow + 8 * p < M) {
        r0[i][p] = *reinterpret_cast<const float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep));
        r1[i][p] = *reinterpret_cast<const float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep) + 16u);
      }
    }
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < PF; ++i) load_res(i);
  __syncthreads();   // every wave is done with the operand stages: reuse LDS for the transpose patches
  f2 vv[TM][2][4];
  auto partial = [&](const f2 (&v)[4], float* dst, int r) {   // this wave's 64 columns of row r: (sum, M2 about their own mean)
    const f2 s2 = (v[0] + v[1]) + (v[2] + v[3]);
    const float sm = row8_sum(s2.x + s2.y);
    const f2 lm = splat2_rt(sm * (1.0f / 64.0f));
    f2 d[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = v[e] - lm;
    const f2 q2 = fma2(d[0], d[0], d[1] * d[1]) + fma2(d[2], d[2], d[3] * d[3]);
    const float sq = row8_sum(q2.x + q2.y);
    if (rc8 == 0) *reinterpret_cast<float2*>(dst + 2 * (r * WN + wn)) = make_float2(sm, sq);
  };
  auto stats = [&](const float* src, int r, float invn, float eps, f2& mean2, f2& rstd2) {   // pairwise combination of the 8 partials
    const float4* xr = reinterpret_cast<const float4*>(src + 2 * r * WN);
    const float4 p0 = xr[0], p1 = xr[1], p2 = xr[2], p3 = xr[3];
    const float mean = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * invn;
    const float e0 = p0.x * (1.0f / 64.0f) - mean, e1 = p0.z * (1.0f / 64.0f) - mean, e2 = p1.x * (1.0f / 64.0f) - mean,
                e3 = p1.z * (1.0f / 64.0f) - mean, e4 = p2.x * (1.0f / 64.0f) - mean, e5 = p2.z * (1.0f / 64.0f) - mean,
                e6 = p3.x * (1.0f / 64.0f) - mean, e7 = p3.z * (1.0f / 64.0f) - mean;
    const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                     64.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
    mean2 = splat2_rt(mean);
    rstd2 = splat2_rt(1.0f / sqrtf(m2 * invn + eps));
  };
#pragma unroll
  for (int i = 0; i < TM; ++i) {
   if (i >= gl && i < gh) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(patch + (i & 1) * 1024 + m16 * 64 + (((4 * j + q4) ^ (m16 & 7)) << 2)) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    D3D_PATCH_FENCE();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = rrow + 8 * p;
      const float* prow = patch + (i & 1) * 1024 + row * 64;
      const float4 a0 = *reinterpret_cast<const float4*>(prow + (((2 * rc8) ^ (row & 7)) << 2));
      const float4 a1 = *reinterpret_cast<const float4*>(prow + (((2 * rc8 + 1) ^ (row & 7)) << 2));
      const bool ok = !CHECK || mt0 + 16 * i + row < M;
      f2 (&v)[4] = vv[i][p];
      v[0].x = r0[i][p].x + (a0.x + bb[0].x); v[0].y = r0[i][p].y + (a0.y + bb[0].y);
      v[1].x = r0[i][p].z + (a0.z + bb[1].x); v[1].y = r0[i][p].w + (a0.w + bb[1].y);
      v[2].x = r1[i][p].x + (a1.x + bb[2].x); v[2].y = r1[i][p].y + (a1.y + bb[2].y);
      v[3].x = r1[i][p].z + (a1.z + bb[3].x); v[3].y = r1[i][p].w + (a1.w + bb[3].y);
      if (CHECK && !ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = splat2(0.0f);
      }
      partial(v, xch, 16 * i + row);
    }
   }
    if (i + PF < TM) load_res(i + PF);
    __builtin_amdgcn_sched_barrier(0);
  }
  const float invn = 1.0f / (float)N;
  const bool ln1 = fx.pn.g != nullptr, ln2 = fx.pn.g2 != nullptr;
  f2 gg[4], be[4], tv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { gg[e] = splat2(1.0f); be[e] = splat2(0.0f); tv[e] = splat2(0.0f); }
  if (ln1) { load8(fx.pn.g + n, gg); load8(fx.pn.b + n, be); }
  const bool tv_uniform = ln1 && fx.pn.tvec != nullptr && fx.pn.tvec_stride == 0;
  const bool tv_rows = ln1 && fx.pn.tvec != nullptr && fx.pn.tvec_stride != 0;
  if (tv_uniform) load8(fx.pn.tvec + n, tv);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i < gl || i >= gh) continue;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = 16 * i + rrow + 8 * p;
      const int m = mt0 + r;
      f2 (&v)[4] = vv[i][p];
      if (ln1) {
        f2 mean2, rstd2;
        stats(xch, r, invn, fx.pn.eps, mean2, rstd2);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fma2((v[e] - mean2) * rstd2, gg[e], be[e]);
        if (fx.pn.pos && (!CHECK || m < M)) {
          f2 t[4];
          load8(fx.pn.pos + (size_t)((m / fx.pn.pos_div) % fx.pn.pos_mod) * N + n, t);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
        if (tv_rows && (!CHECK || m < M)) {
          f2 t[4];
          load8(fx.pn.tvec + (size_t)(m / fx.pn.rows_per_batch) * fx.pn.tvec_stride + n, t);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += t[e];
        } else if (tv_uniform) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += tv[e];
        }
        if (ln2) partial(v, xch2, r);
      }
      if (!CHECK || m < M) {
        *reinterpret_cast<float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep)) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
        *reinterpret_cast<float4*>(Cb + (ob + (unsigned)(2 * i + p) * rstep) + 16u) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!ln2) return;   // (uniform)
  f2 g2[4], b2[4];
  load8(fx.pn.g2 + n, g2);
  load8(fx.pn.b2 + n, b2);
  if (ln1) __syncthreads();
  const float* src = ln1 ? xch2 : xch;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i < gl || i >= gh) continue;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = 16 * i + rrow + 8 * p;
      if (CHECK && mt0 + r >= M) continue;
      f2 mean2, rstd2;
      stats(src, r, invn, fx.pn.eps2, mean2, rstd2);
      f2 h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = fma2((vv[i][p][e] - mean2) * rstd2, g2[e], b2[e]);
      *reinterpret_cast<uint4*>(Hb + (obh + (unsigned)(2 * i + p) * rsteph)) = cvt8_bf16(h, 1.0f);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
