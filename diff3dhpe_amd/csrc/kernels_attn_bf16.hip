// GRAND attention (S2S:75-83) of the bf16 operand mode (D3D_PREC_BF16): both products on v_mfma_f32_32x32x16_bf16, one MFMA per
// product, fp32 accumulation and fp32 softmax.  Groups of `T` tokens with token stride `J` (temporal blocks: the frames of one
// joint; spatial blocks: launched as T = 17 joints, J = 1, B * T_frames "batches"), head width 64.
//
//   S^T = K Q'^T          q' = q / 8 arrives pre-scaled (exact) from the qkv GEMM epilogue; one query column per lane
//   P   = softmax(S) - I  exact two-pass softmax in fp32 registers, normalised BEFORE the second product, the diagonal
//                         subtracted, then rounded to bf16 -- the operand the reference's `(attn - eye) @ v` has, which is what
//                         the oracle's bf16-operand emulation rounds (oracle/d3d_oracle.py operand_rounding)
//   O^T = V^T P^T         the accumulator tile of P is the B operand after a pairwise bf16 conversion; V^T fragments come from
//                         row-major V through ds_read_b64_tr_b16
// q, k, v: one bf16 buffer [rows][3 D] written by the qkv GEMM (launch_linear_bf16); output bf16 [rows][D], the proj GEMM's
// operand.  One workgroup per (batch, joint, head) unit; K and V rows of the unit live in LDS (128 B per row, 16-byte chunks
// XOR-swizzled: conflict-free fragment reads); every further key tile / k-step of a fragment read is a compile-time offset of
// four base addresses.  NKT = ceil(T / 32) key tiles; a wave owns one 32-query tile, or two in turn (QT = 2: T > 160) so that
// the workgroup is four waves and two workgroups share a CU.  MU > 1 (NKT == 1): MU independent units per workgroup, one per
// wave.  Plain load -> compute -> store per workgroup (no persistent walk, no LDS-DMA): overlap comes from the co-resident
// workgroups (registers: 61 .. 250 VGPRs, no scratch).
#include "d3d_kernels.h"

#include <math.h>

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int BDH = 64;
__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// V rows: the four key rows of one transposing read must land in four different 64-byte bank groups (kernels_attn_x3.hip)
__device__ __forceinline__ int vkey(int row) { return (((row >> 1) & 1) << 2) ^ ((row >> 2) & 3); }
__device__ __forceinline__ int vswz(int row, int chunk) { return row * 128 + ((chunk ^ vkey(row)) << 4); }
}  // namespace

// QT: 32-query tiles per wave (1 or 2).  QT = 2 halves the workgroup (NKT / 2 waves, each taking query tiles w and w + NKT / 2 one
// after the other against the K / V rows staged once): a T = 243 unit is then a 4-wave workgroup with 64 KiB of LDS, and TWO of
// them share a CU as independent instruction streams -- one's K / V loads and output stores sit under the other's MFMAs and
// softmax (the 8-wave form has the CU to itself and serialises load -> compute -> store).
template <int NKT, int MU, int QT = 1>
__global__ __launch_bounds__(64 * ((NKT + QT - 1) / QT) * MU) __attribute__((amdgpu_waves_per_eu(QT)))   // (QT = 2: <= 256 VGPRs, two workgroups per CU)
void k_attn_bf16(const __bf16* __restrict__ qkv, __bf16* __restrict__ out, int T, int J,
                                                             int H, int D, int units) {
  static_assert(MU == 1 || NKT == 1, "several units per workgroup only for single-tile groups");
  static_assert(QT == 1 || (MU == 1 && NKT % QT == 0), "query-tile passes only for whole multiples");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
  constexpr int TP = 32 * NKT;
  constexpr int NW = (NKT + QT - 1) / QT;                          // waves of a unit
  const int lane = threadIdx.x & 63;
  const int sub = (MU > 1) ? (int)(threadIdx.x >> 6) : 0;          // unit of this workgroup
  const int wave = (MU > 1) ? 0 : (int)(threadIdx.x >> 6);         // first 32-query tile of this wave
  const int tid = (MU > 1) ? lane : (int)threadIdx.x;              // thread index within the unit
  unsigned char* const sK = lds_all + sub * (2 * TP * 128);        // [TP][128 B]
  unsigned char* const sV = sK + TP * 128;

  const int unit_raw = blockIdx.x * MU + sub;                      // (b * J + j) * H + hd
  const bool unit_ok = unit_raw < units;
  const int unit = unit_ok ? unit_raw : units - 1;                 // surplus waves redo the last unit and store nothing
  const int hd = unit % H;
  const int bj = unit / H;
  const int j = bj % J, b = bj / J;
  const int D3 = 3 * D;
  const int r = lane & 31, h = lane >> 5;
  const size_t tok0 = (size_t)b * T * J + j;                       // token(t) = tok0 + t * J

  // the first pass's query fragments are requested together with the K / V rows, in front of the staging barrier (round 5: behind it
  // their latency was exposed once per unit -- these kernels are latency-bound: 3 TB/s of 8 with the matrix pipe 9 % busy)
  // (QT = 1 forms only: the two-pass forms sit at the register limit)
  bf8 qf0[4];
  if constexpr (QT == 1) {
    const int tq0 = 32 * wave + r;
    const size_t o = (tok0 + (size_t)(tq0 < T ? tq0 : 0) * J) * D3 + hd * BDH + 8 * h;   // rows >= T reuse row 0: never stored
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf0[ks] = *reinterpret_cast<const bf8*>(qkv + o + 16 * ks);
  }
  {  // ---- stage K and V rows of the unit (pad rows zero); all global loads are issued before the LDS writes
    constexpr int NIT = 4 * QT;                                    // TP * 8 chunk slots / (64 * NW threads)
    uint4 kk[NIT], vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 64 * NW;
      const int row = idx >> 3, c8 = idx & 7;
      kk[it] = make_uint4(0, 0, 0, 0); vv[it] = kk[it];
      if (row < T) {
        const size_t o = (tok0 + (size_t)row * J) * D3 + hd * BDH + c8 * 8;
        kk[it] = *reinterpret_cast<const uint4*>(qkv + o + D);
        vv[it] = *reinterpret_cast<const uint4*>(qkv + o + 2 * D);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 64 * NW;
      const int row = idx >> 3, c8 = idx & 7;
      *reinterpret_cast<uint4*>(sK + kswz(row, c8)) = kk[it];
      *reinterpret_cast<uint4*>(sV + vswz(row, c8)) = vv[it];
    }
  }
  __syncthreads();
  // fragment addresses: four bases each for K (one per 16-deep d-step) and V^T (d-half x key-row half); every further key tile /
  // k-step is a compile-time offset (both swizzles have period 16 in the row), i.e. the ds_read offset field, not a register
  const unsigned char* kb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kb[ks] = sK + kswz(r, 2 * ks + h);                       // + 4096 kt
  const unsigned char* vb[2][2];
  {
    const int gi = lane & 15, tq_ = gi >> 2, tp_ = gi & 3;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int d0 = dt * 32 + 16 * ((lane >> 4) & 1);
      const int ch = (d0 >> 3) + (tp_ >> 1), sb = (tp_ & 1) * 8;
      vb[dt][0] = sV + vswz(4 * h + tq_, ch) + sb;                                         // + 2048 (2 kt + s)
      vb[dt][1] = sV + vswz(4 * h + 8 + tq_, ch) + sb;
    }
  }
#pragma unroll 1
  for (int pass = 0; pass < QT; ++pass) {
  // ---- this lane's query row as MFMA B fragments: d = 16 ks + 8 h .. + 7
  const int tq = 32 * (wave + pass * NW) + r;
  if (32 * (wave + pass * NW) >= T) break;                        // (wave-uniform)
  bf8 qf[4];
  if (QT == 1) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qf0[ks];
  } else {
    const size_t o = (tok0 + (size_t)(tq < T ? tq : 0) * J) * D3 + hd * BDH + 8 * h;   // rows >= T reuse row 0: never stored
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf8*>(qkv + o + 16 * ks);
  }

  // ---- S^T tiles: rows = keys kt * 32 + (reg & 3) + 8 (reg >> 2) + 4 h, column = query tq
  f32x16 sacc[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc[kt][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf8 kf = *reinterpret_cast<const bf8*>(kb[ks] + kt * 4096);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt], 0, 0, 0);
    }
  }
  // ---- exact softmax over the keys of this query column (fp32); only the last key tile can hold padding keys
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (kt == NKT - 1) {
        const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (key >= T) sacc[kt][q] = -INFINITY;
      }
      m = fmaxf(m, sacc[kt][q]);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  constexpr float LOG2E = 1.4426950408889634f;
  const float mb = m * LOG2E;
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][q], LOG2E, -mb));
      sacc[kt][q] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  // softmax - I (S2S:82): the diagonal sits in ONE key tile (kt == this wave's query tile: wave-uniform) and, inside it, in the
  // lanes whose half h holds key r -- ((r >> 2) & 1) == h -- at register 8 (r >> 4) + 4 ((r >> 3) & 1) + (r & 3).  Its numerator
  // becomes e - l, so that the normalisation below yields p - 1 for it: sixteen selects in one tile instead of a compare per score.
  {
    const int qt = __builtin_amdgcn_readfirstlane(tq >> 5);
    const int didx = (((r >> 2) & 1) == h) ? (8 * (r >> 4) + 4 * ((r >> 3) & 1) + (r & 3)) : -1;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt == qt) {
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc[kt][q] -= (q == didx) ? l : 0.0f;
      }
  }

  // ---- O^T[d][query] = sum_key V^T[d][key] (P - I)^T[key][query].  k-step (kt, s) takes accumulator registers 8 s .. 8 s + 7:
  // element jj of lane half h is key kt * 32 + 16 s + 8 (jj >> 2) + 4 h + (jj & 3); the V^T fragment is read in that order.
  f32x16 oacc[2];
#pragma unroll
  for (int q = 0; q < 16; ++q) { oacc[0][q] = 0.f; oacc[1][q] = 0.f; }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf8 pf;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) pf[jj] = (__bf16)(sacc[kt][8 * s + jj] * inv);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const s4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)(vb[dt][0] + (2 * kt + s) * 2048));
        const s4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)(vb[dt][1] + (2 * kt + s) * 2048));
        bf8 vf;
        const bf4 a0b = __builtin_bit_cast(bf4, a0), a1b = __builtin_bit_cast(bf4, a1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[e] = a0b[e]; vf[4 + e] = a1b[e]; }
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // keeps the conversions / V^T reads of later k-steps from being hoisted (VGPR pressure)
    }
  }
  // ---- O rows out as bf16: lane (query tq, half h) holds d = dt * 32 + 8 g4 + 4 h + e
  if (tq < T && unit_ok) {
    __bf16* orow = out + (tok0 + (size_t)tq * J) * D + hd * BDH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)oacc[dt][4 * g4 + e];
        *reinterpret_cast<bf4*>(orow + dt * 32 + 8 * g4 + 4 * h) = o;
      }
  }
  }   // pass
}

bool attn_bf16_ok(int T, int D, int H) { return T >= 1 && T <= 256 && H > 0 && D == H * BDH; }

template <int NKT, int MU = 1, int QT = 1>
static hipError_t launch_nkt(const __bf16* qkv, __bf16* out, int B, int T, int J, int D, int H, hipStream_t s) {
  const size_t lds_bytes = (size_t)MU * 2 * 32 * NKT * 128;
  static std::atomic<unsigned long long> attr_set{0};   // one bit per device
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_attn_bf16<NKT, MU, QT>), lds_bytes, attr_set)) return e;
  const long long units = (long long)B * J * H;
  if (units > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_attn_bf16<NKT, MU, QT>), dim3((unsigned)((units + MU - 1) / MU)), dim3(64 * ((NKT + QT - 1) / QT) * MU), lds_bytes, s,
                     qkv, out, T, J, H, D, (int)units);
  return hipGetLastError();
}

// qkv: bf16 [B*T*J][3D] (q third pre-scaled by dh^-0.5); out: bf16 [B*T*J][D].  Spatial blocks: call with (B*T, J, 1).
hipError_t launch_attn_bf16(const void* qkv_bf16, void* out_bf16, int B, int T, int J, int D, int H, hipStream_t s) {
  if (!attn_bf16_ok(T, D, H) || !qkv_bf16 || !out_bf16 || B <= 0 || J <= 0) return hipErrorInvalidValue;
  const __bf16* q = (const __bf16*)qkv_bf16;
  __bf16* o = (__bf16*)out_bf16;
  switch ((T + 31) / 32) {
    case 1: return ((long long)B * J * H >= 4096 ? launch_nkt<1, 8> : launch_nkt<1, 1>)(q, o, B, T, J, D, H, s);
    case 2: return launch_nkt<2>(q, o, B, T, J, D, H, s);
    case 3: return launch_nkt<3>(q, o, B, T, J, D, H, s);
    case 4: return launch_nkt<4>(q, o, B, T, J, D, H, s);
    case 5: return launch_nkt<5>(q, o, B, T, J, D, H, s);
    case 6: return launch_nkt<6, 1, 2>(q, o, B, T, J, D, H, s);
    case 7: return launch_nkt<7>(q, o, B, T, J, D, H, s);
    default: return launch_nkt<8, 1, 2>(q, o, B, T, J, D, H, s);   // two 4-wave workgroups per CU
  }
}

}  // namespace d3d
