// F16X3 token GEMM, pre-split operands ("x3p"): the production GEMM of the F16X3 precision mode.
//
//   C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] )
//
// Both operands arrive as fp16 hi/lo planes (see kernels_gemm_f16x3.hip for the arithmetic and its measured accuracy):
//   A_hi/A_lo [M][K]  = split(8 * a)      written by the PRODUCER of the activation (LayerNorm, attention, GELU epilogue)
//   W_hi/W_lo [N][K]  = split(4096 * w)   made once at weight-commit time
// so the k-loop is a pure fp16 MFMA loop: per 16-deep k-step and 32x32 output tile three v_mfma_f32_32x32x16_f16
// (a_lo b_hi, a_hi b_lo, a_hi b_hi) into one fp32 accumulator, result scaled by 2^-15 in the epilogue.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): each wave-instruction drops 1 KiB = 16 rows x 64 B of one plane
// straight into LDS, no staging VGPRs.  The LDS image is lane-linear, so the bank swizzle (16-byte chunk index XOR
// (row>>2)&3, which makes every ds_read_b128 16-lane group hit 16 distinct 4-bank slots) is applied to the per-lane
// SOURCE address and again on the fragment read -- the same involution on both sides.  Two LDS stages; the DMA of
// k-tile t+1 is issued right after the barrier that retires k-tile t-1 and flies under the MFMAs of k-tile t
// (one barrier per k-tile; __syncthreads() drains the wave's own DMA with vmcnt(0) before the barrier).
//
// Tile shapes (BM x BN x 32, waves WM x WN, each wave (BM/WM) x (BN/WN)):
//   256x256, 2x4 waves of 128x64 (128 accumulator VGPRs, 128 KiB LDS, 1 workgroup/CU)   -- large N
//   256x128, 4x2 waves of  64x64                                                         -- N = 512
//   128x128, 2x2 waves of  64x64 ( 64 KiB LDS, 2 workgroups/CU)                          -- small problems / tails
#include "d3d_kernels.h"

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int PBK = 32;                          // k-tile depth (fp16 elements) = 64 B per plane row
constexpr float P_OUT_SCALE = 1.0f / 32768.0f;   // 2^-(3+12)
constexpr float P_A_SCALE = 8.0f;

__device__ __forceinline__ float gelu_erf_p(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ int swz64(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

// Epilogue of one wave.  The MFMAs are issued with the WEIGHT fragment as the first operand, so an accumulator tile
// holds C^T: column (lane&31) = token row m, registers = output columns n = 8*(reg>>2) + 4*(lane>>5) + (reg&3).
// Each 32-row strip of the wave tile is transposed through a wave-private LDS patch (ds_write_b128 of 4 consecutive
// n per lane, chunk index XOR (row&7) -> conflict-free) and read back row-major, 16 bytes per lane with 8*TNJ lanes per row,
// so every global store / residual load instruction covers whole 128-byte lines (a store tail of partial lines or of
// 4-byte-per-lane stores costs more than the transpose).  Addressing is (wave-uniform tile base) + 32-bit offsets.
template <int TMI, int TNJ, int EPI, int OUTSPLIT, bool CHECK>
__device__ __forceinline__ void x3p_epilogue(f32x16 (&acc)[TMI][TNJ], float* patch, const float* __restrict__ bias,
                                             const float* Rt, float* Ct, _Float16* Cht, _Float16* Clt, int mt0, int nt0,
                                             int lane, int M, int N, int qcols) {
  constexpr int LD = 32 * TNJ;              // floats per patch row; 16-byte chunks XOR-swizzled by (row & 7)
  constexpr int LPR = 8 * TNJ;              // lanes per row on the read side (one float4 each)
  constexpr int RPP = 64 / LPR;             // rows per pass
  constexpr int NPASS = 32 / RPP;
  const int r = lane & 31, h = lane >> 5;
  const int rrow = lane / LPR, rc4 = lane % LPR;
  const int n = nt0 + 4 * rc4;
  const bool ncol_ok = !CHECK || n < N;
  float4 b4 = make_float4(0, 0, 0, 0);
  if (bias && ncol_ok) b4 = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
  for (int i = 0; i < TMI; ++i) {
#pragma unroll
    for (int j = 0; j < TNJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(patch + r * LD + (((8 * j + 2 * g + h) ^ (r & 7)) << 2)) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int row = rrow + RPP * p;
      const float4 a4 = *reinterpret_cast<const float4*>(patch + row * LD + ((rc4 ^ (row & 7)) << 2));
      const int m = mt0 + 32 * i + row;
      if (CHECK && (m >= M || !ncol_ok)) continue;
      const int off = (32 * i + row) * N + 4 * rc4;
      float v[4] = {a4.x * P_OUT_SCALE + b4.x, a4.y * P_OUT_SCALE + b4.y, a4.z * P_OUT_SCALE + b4.z, a4.w * P_OUT_SCALE + b4.w};
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_p(v[e]);
      }
      if (EPI == EPI_RESIDUAL) {
        const float4 r4 = *reinterpret_cast<const float4*>(Rt + off);
        v[0] = r4.x + v[0]; v[1] = r4.y + v[1]; v[2] = r4.z + v[2]; v[3] = r4.w + v[3];
      }
      if (OUTSPLIT) {   // the consumer is an F16X3 kernel: hand it hi/lo planes of 8*v (q columns of a qkv GEMM for the
        h4 hh, ll;      // temporal attention carry the dh^-0.5 = 2^-3 attention scale, i.e. planes of 1*v)
        const float osc = (n < qcols) ? 1.0f : P_A_SCALE;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc = __builtin_amdgcn_fmed3f(v[e] * osc, -65504.0f, 65504.0f);
          hh[e] = (_Float16)sc;
          ll[e] = (_Float16)(sc - (float)hh[e]);
        }
        *reinterpret_cast<h4*>(Cht + off) = hh;
        *reinterpret_cast<h4*>(Clt + off) = ll;
      } else {
        *reinterpret_cast<float4*>(Ct + off) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // one strip at a time
  }
}

// Byte offset of logical 16-byte chunk c of tile row `row` in a plane whose rows are BK fp16 wide.  The XOR spreads the
// rows a ds_read_b128 16-lane group touches over 16 distinct 4-bank slots (64-B rows: 4 rows per 256-B bank row ->
// XOR (row>>2)&3; 32-B rows: 8 rows per bank row -> XOR (row>>3)&1).
template <int BK>
__device__ __forceinline__ int swzk(int row, int c) {
  if (BK == 32) return row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
  return row * 32 + ((c ^ ((row >> 3) & 1)) << 4);
}

template <int BM, int BN, int WM, int WN, int EPI, int OUTSPLIT, int ABL = 0, int BK = 32>
__global__ __launch_bounds__(64 * WM * WN) void k_linear_x3p(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                             const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl,
                                                             const float* __restrict__ bias, const float* R, float* C,
                                                             _Float16* Ch, _Float16* Cl, int M, int N, int K, int mtiles,
                                                             int ntiles, int ablate, int qcols) {
  // ablate (timing experiments only): 4 = no epilogue stores
  constexpr int NW = WM * WN;
  constexpr int TMI = BM / WM / 32, TNJ = BN / WN / 32;
  constexpr int RB = BK * 2;                                       // bytes per plane row in a k-tile
  constexpr int CPR = RB / 16;                                     // 16-byte chunks per row
  constexpr int RPI = 1024 / RB;                                   // rows per 1-KiB DMA instruction
  constexpr int KS = BK / 16;                                      // MFMA k-steps per k-tile
  constexpr int A_PLANE = BM * RB, B_PLANE = BN * RB;              // bytes
  constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  constexpr int A_INSTR = BM / RPI, B_INSTR = BN / RPI;            // DMA instructions per plane
  static_assert(NW % 4 == 0 && A_INSTR % (NW / 4) == 0 && B_INSTR % (NW / 4) == 0, "planes must split evenly over the waves");
  static_assert(2 * STAGE >= NW * 32 * (32 * TNJ) * 4, "epilogue patches must fit in the operand stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / ntiles) * 8 + xcd;   // the N-tiles of one M-tile run back to back on one XCD (A rows shared in L2)
  const int nt = slot % ntiles;
  if (mt >= mtiles) return;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  // ---- DMA plan (branch-free): wave w streams plane (w & 3) in {A_hi, A_lo, W_hi, W_lo}; the NW/4 waves that share
  // a plane split its RPI-row groups.  A lane moves row (RPI g + lane/CPR), LDS slot lane%CPR, and fetches the source
  // chunk given by the swizzle: the XOR lives on the SOURCE address, the LDS image stays lane-linear.
  // Contract: the A planes hold >= mtiles*BM rows and the W planes >= ntiles*BN rows (padding rows are never stored).
  constexpr int WPP = NW / 4;                                      // waves per plane
  const int plane = wave & 3, part = wave >> 2;
  const bool isA = plane < 2;
  const int n_it = (isA ? A_INSTR : B_INSTR) / WPP;                 // wave-uniform trip count
  const int g0 = part * n_it;
  const int lrow = lane / CPR, lslot = lane % CPR;
  const _Float16* src;
  {
    const _Float16* pb = (plane == 0) ? Ah : (plane == 1) ? Al : (plane == 2) ? Wh : Wl;
    const int row0 = (isA ? m0 : n0) + g0 * RPI + lrow;
    src = pb + (size_t)row0 * K + (swzk<BK>(lrow, lslot) - lrow * RB) / 2;   // swizzled chunk of this lane's row, in halfs
  }
  const size_t it_stride = (size_t)RPI * K;                         // elements between successive row groups
  const int dst0 = (isA ? plane * A_PLANE : 2 * A_PLANE + (plane - 2) * B_PLANE) + g0 * 1024 + lane * 16;
  constexpr int MAX_IT = (A_INSTR > B_INSTR ? A_INSTR : B_INSTR) / WPP;

#define D3D_STAGE_ONE(ST, K0, IT)                                                                                       \
  __builtin_amdgcn_global_load_lds(src + (K0) + (IT) * it_stride,                                                        \
                                   (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (ST) * STAGE + dst0 + (IT) * 1024), \
                                   16, 0, 0)
#define D3D_STAGE_ALL(ST, K0)                                                                                           \
  do {                                                                                                                  \
    _Pragma("unroll") for (int it_ = 0; it_ < MAX_IT; ++it_) {                                                          \
      if (it_ < n_it) D3D_STAGE_ONE(ST, K0, it_);                                                                       \
    }                                                                                                                   \
  } while (0)

  f32x16 acc[TMI][TNJ];
#pragma unroll
  for (int i = 0; i < TMI; ++i)
#pragma unroll
    for (int j = 0; j < TNJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

  const int arow0 = wm * (BM / WM) + r, brow0 = wn * (BN / WN) + r;
  const int nk = K / BK;
  constexpr int NG = KS * TMI;                                 // MFMA groups per k-tile: (k-step, m-tile)
  constexpr int DMA_NG = (ABL == 4) ? (NG / 2) : (ABL == 5) ? (NG / 4) : NG;   // groups the next k-tile's DMA is spread over
  constexpr int PPG = (MAX_IT + DMA_NG - 1) / DMA_NG;          // DMA pieces issued per group
  D3D_STAGE_ALL(0, 0);
  // One k-tile: software pipeline over the NG groups -- the fragments of group g+1 are read from LDS, and one slice of
  // the NEXT k-tile's DMA is issued, BEFORE the 3*TNJ MFMAs of group g, so LDS latency and DMA issue hide under MFMAs.
  // The body is branch-free (the last k-tile, which has nothing to prefetch, is peeled) so that it stays one scheduling
  // region and hipcc emits counted lgkmcnt waits instead of lgkmcnt(0) at block boundaries.
#define D3D_FRAG_C(KSI) ((BK == 32) ? (2 * (KSI) + h) : h) /* lane half h feeds k = 16 ks + 8 h .. +7 of the MFMA */
#define D3D_KTILE(KT, PREFETCH)                                                                                          \
  do {                                                                                                                   \
    if (ABL != 3) __syncthreads(); /* own DMA drained (vmcnt(0)) + everyone done reading the other stage */             \
    const int nst = ((KT) + 1) & 1, nk0 = ((KT) + 1) * BK;                                                               \
    const unsigned char* sb = lds + ((KT) & 1) * STAGE;                                                                  \
    const unsigned char* sAh = sb;                                                                                       \
    const unsigned char* sAl = sb + A_PLANE;                                                                             \
    const unsigned char* sBh = sb + 2 * A_PLANE;                                                                         \
    const unsigned char* sBl = sb + 2 * A_PLANE + B_PLANE;                                                               \
    h8 bh[2][TNJ], bl[2][TNJ], ah[2], al[2];                                                                             \
    _Pragma("unroll") for (int j = 0; j < TNJ; ++j) {                                                                    \
      const int ob = swzk<BK>(brow0 + 32 * j, D3D_FRAG_C(0));                                                            \
      bh[0][j] = *reinterpret_cast<const h8*>(sBh + ob);                                                                 \
      bl[0][j] = *reinterpret_cast<const h8*>(sBl + ob);                                                                 \
    }                                                                                                                    \
    {                                                                                                                    \
      const int oa = swzk<BK>(arow0, D3D_FRAG_C(0));                                                                     \
      ah[0] = *reinterpret_cast<const h8*>(sAh + oa);                                                                    \
      al[0] = *reinterpret_cast<const h8*>(sAl + oa);                                                                    \
    }                                                                                                                    \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                                     \
      const int ks = g / TMI, i = g % TMI;                                                                               \
      if (g + 1 < NG && (ABL != 2 || (KT) == 0)) {                                                                       \
        const int ks2 = (g + 1) / TMI, i2 = (g + 1) % TMI;                                                               \
        const int oa = swzk<BK>(arow0 + 32 * i2, D3D_FRAG_C(ks2));                                                       \
        ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sAh + oa);                                                        \
        al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sAl + oa);                                                        \
        if (i2 == 0) {                                                                                                   \
          _Pragma("unroll") for (int j = 0; j < TNJ; ++j) {                                                              \
            const int ob = swzk<BK>(brow0 + 32 * j, D3D_FRAG_C(ks2));                                                    \
            bh[ks2 & 1][j] = *reinterpret_cast<const h8*>(sBh + ob);                                                     \
            bl[ks2 & 1][j] = *reinterpret_cast<const h8*>(sBl + ob);                                                     \
          }                                                                                                              \
        }                                                                                                                \
      }                                                                                                                  \
      if (PREFETCH && ABL != 1) {                                                                                        \
        _Pragma("unroll") for (int pp = 0; pp < PPG; ++pp) {                                                             \
          const int it_ = g * PPG + pp;                                                                                  \
          if (it_ < MAX_IT && (UNIFORM_IT || it_ < n_it)) D3D_STAGE_ONE(nst, nk0, it_);                                  \
        }                                                                                                                \
      }                                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < TNJ; ++j) { /* operands swapped: accumulator = C^T tile (x3p_epilogue) */   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks & 1][j], al[g & 1], acc[i][j], 0, 0, 0);                \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks & 1][j], ah[g & 1], acc[i][j], 0, 0, 0);                \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks & 1][j], ah[g & 1], acc[i][j], 0, 0, 0);                \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
  } while (0)

  constexpr bool UNIFORM_IT = (A_INSTR == B_INSTR);
  int kt = 0;
  for (; kt + 1 < nk; ++kt) D3D_KTILE(kt, true);
  D3D_KTILE(kt, false);
#undef D3D_KTILE
#undef D3D_FRAG_C

#undef D3D_STAGE_ONE
#undef D3D_STAGE_ALL
  if ((ablate & 4) && acc[0][0][0] != 12345.678f) return;
  const int mt0 = m0 + wm * (BM / WM), nt0 = n0 + wn * (BN / WN);          // wave-uniform
  const size_t tbase = (size_t)mt0 * N + nt0;
  const float* Rt = R ? R + tbase : nullptr;
  float* Ct = C ? C + tbase : nullptr;
  _Float16* Cht = Ch ? Ch + tbase : nullptr;
  _Float16* Clt = Cl ? Cl + tbase : nullptr;
  __syncthreads();   // every wave is done with the operand stages: reuse LDS for the transpose patches
  float* patch = reinterpret_cast<float*>(lds) + wave * (32 * 32 * TNJ);
  if (m0 + BM <= M && n0 + BN <= N)
    x3p_epilogue<TMI, TNJ, EPI, OUTSPLIT, false>(acc, patch, bias, Rt, Ct, Cht, Clt, mt0, nt0, lane, M, N, qcols);
  else
    x3p_epilogue<TMI, TNJ, EPI, OUTSPLIT, true>(acc, patch, bias, Rt, Ct, Cht, Clt, mt0, nt0, lane, M, N, qcols);
}

template <int BM, int BN, int WM, int WN, int BK = 32>
static hipError_t launch_tile(const _Float16* Ah, const _Float16* Al, const _Float16* Wh, const _Float16* Wl,
                              const float* bias, const float* R, float* C, _Float16* Ch, _Float16* Cl, int M, int N, int K,
                              int epi, int outsplit, int ablate, int qcols, hipStream_t s) {
  const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
  const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
  const size_t lds_bytes = 2 * (size_t)(2 * BM * BK * 2 + 2 * BN * BK * 2);
#define D3D_X3P_LAUNCH(EPI_, OS_)                                                                                         \
  do {                                                                                                                    \
    auto kfn = k_linear_x3p<BM, BN, WM, WN, EPI_, OS_, 0, BK>;                                                                 \
    static bool attr_done = false;                                                                                        \
    if (!attr_done) {                                                                                                     \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)lds_bytes);                                                                \
      if (ae != hipSuccess) return ae;                                                                                    \
      attr_done = true;                                                                                                   \
    }                                                                                                                     \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * WM * WN), lds_bytes, s, Ah, Al, Wh, Wl, bias, R, C, Ch, Cl, M, N, K,    \
                       mtiles, ntiles, ablate, qcols);                                                                    \
  } while (0)
  if (outsplit) {
    if (epi == EPI_GELU) D3D_X3P_LAUNCH(EPI_GELU, 1);
    else if (epi == EPI_NONE) D3D_X3P_LAUNCH(EPI_NONE, 1);
    else return hipErrorInvalidValue;
  } else {
    if (epi == EPI_NONE) D3D_X3P_LAUNCH(EPI_NONE, 0);
    else if (epi == EPI_GELU) D3D_X3P_LAUNCH(EPI_GELU, 0);
    else if (epi == EPI_RESIDUAL) D3D_X3P_LAUNCH(EPI_RESIDUAL, 0);
    else return hipErrorInvalidValue;
  }
#undef D3D_X3P_LAUNCH
  return hipGetLastError();
}

// ---- 16x16x32 form ("x3q") -------------------------------------------------------------------------------------------
// Same algorithm and staging as k_linear_x3p<256,256,2,4,...,32>, but the products are issued as
// v_mfma_f32_16x16x32_f16: one MFMA spans the whole 32-deep k-tile.  Measured on MI355X with random fp16 operands
// (experiments/mfma_ceiling.hip) the chip sustains 1.97 PFLOP/s on this shape against 1.54 PFLOP/s on 32x32x16 (it holds
// a higher clock), so the MFMA-bound part of the GEMM gets ~1.28x faster at identical cycle counts.
// Wave tile 128(m) x 64(n) = 8 x 4 tiles of 16x16 (128 accumulator VGPRs); weight fragment first, so a tile holds C^T:
// lane -> token m = lane&15, registers -> n = 4*(lane>>4) + reg.  Fragment read: lane (r16 = lane&15, q = lane>>4) takes
// 16-byte chunk q of row r16; the chunk XOR [0,3,2,1][(row>>2)&3] makes every ds_read_b128 16-lane group of this map
// (and of the 32x32x16 map) hit 16 distinct 4-bank slots.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swzq(int row, int c) { return row * 64 + ((c ^ ((0 - (row >> 2)) & 3)) << 4); }

template <int EPI, int OUTSPLIT, bool CHECK>
__device__ __forceinline__ void x3q_epilogue(f32x4 (&acc)[8][4], float* patch, const float* __restrict__ bias, const float* Rt,
                                             float* Ct, _Float16* Cht, _Float16* Clt, int mt0, int nt0, int lane, int M, int N,
                                             int qcols) {
  // patch: two wave-private 16 rows x 64 floats (alternating, so the LDS round trip of one m-tile overlaps the stores
  // of the previous one), 16-byte chunks XOR-swizzled by (row & 7)
  const int m16 = lane & 15, q4 = lane >> 4;
  const int rrow = lane >> 4, rc4 = lane & 15;         // read side: 16 lanes per row, 4 rows per pass
  const int n = nt0 + 4 * rc4;
  const bool ncol_ok = !CHECK || n < N;
  float4 b4 = make_float4(0, 0, 0, 0);
  if (bias && ncol_ok) b4 = *reinterpret_cast<const float4*>(bias + n);
  const float osc = (n < qcols) ? 1.0f : P_A_SCALE;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(patch + (i & 1) * 1024 + m16 * 64 + (((4 * j + q4) ^ (m16 & 7)) << 2)) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = rrow + 4 * p;
      const float4 a4 = *reinterpret_cast<const float4*>(patch + (i & 1) * 1024 + row * 64 + ((rc4 ^ (row & 7)) << 2));
      const int m = mt0 + 16 * i + row;
      if (CHECK && (m >= M || !ncol_ok)) continue;
      const int off = (16 * i + row) * N + 4 * rc4;
      float v[4] = {a4.x * P_OUT_SCALE + b4.x, a4.y * P_OUT_SCALE + b4.y, a4.z * P_OUT_SCALE + b4.z, a4.w * P_OUT_SCALE + b4.w};
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_p(v[e]);
      }
      if (EPI == EPI_RESIDUAL) {
        const float4 r4 = *reinterpret_cast<const float4*>(Rt + off);
        v[0] = r4.x + v[0]; v[1] = r4.y + v[1]; v[2] = r4.z + v[2]; v[3] = r4.w + v[3];
      }
      if (OUTSPLIT) {
        h4 hh, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc = __builtin_amdgcn_fmed3f(v[e] * osc, -65504.0f, 65504.0f);
          hh[e] = (_Float16)sc;
          ll[e] = (_Float16)(sc - (float)hh[e]);
        }
        *reinterpret_cast<h4*>(Cht + off) = hh;
        *reinterpret_cast<h4*>(Clt + off) = ll;
      } else {
        *reinterpret_cast<float4*>(Ct + off) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two m-tiles (two patches) in flight at a time
  }
}

template <int EPI, int OUTSPLIT>
__global__ __launch_bounds__(512) void k_linear_x3q(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                    const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl,
                                                    const float* __restrict__ bias, const float* R, float* C, _Float16* Ch,
                                                    _Float16* Cl, int M, int N, int K, int mtiles, int ntiles, int qcols) {
  constexpr int BM = 256, BN = 256, WN = 4, BK = 32;
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, STAGE = 2 * A_PLANE + 2 * B_PLANE;
  constexpr int N_IT = 8;                 // 1-KiB DMA pieces per wave per k-tile (16 per plane, 2 waves per plane)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / ntiles) * 8 + xcd;
  const int nt = slot % ntiles;
  if (mt >= mtiles) return;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, q = lane >> 4;

  const int plane = wave & 3, part = wave >> 2;
  const bool isA = plane < 2;
  const int g0 = part * N_IT;
  const int lrow = lane >> 2, lslot = lane & 3;
  const _Float16* src;
  {
    const _Float16* pb = (plane == 0) ? Ah : (plane == 1) ? Al : (plane == 2) ? Wh : Wl;
    const int row0 = (isA ? m0 : n0) + g0 * 16 + lrow;
    src = pb + (size_t)row0 * K + ((lslot ^ ((0 - (lrow >> 2)) & 3)) << 3);
  }
  const size_t it_stride = (size_t)16 * K;
  const int dst0 = (isA ? plane * A_PLANE : 2 * A_PLANE + (plane - 2) * B_PLANE) + g0 * 1024 + lane * 16;

#define D3D_QSTAGE_ONE(ST, K0, IT)                                                                                      \
  __builtin_amdgcn_global_load_lds(src + (K0) + (IT) * it_stride,                                                        \
                                   (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (ST) * STAGE + dst0 + (IT) * 1024), \
                                   16, 0, 0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

  const int arow0 = wm * 128 + r16, brow0 = wn * 64 + r16;
  const int nk = K / BK;
#pragma unroll
  for (int it = 0; it < N_IT; ++it) D3D_QSTAGE_ONE(0, 0, it);

  // one k-tile = 8 groups (one 16-row m-tile each): the A fragments of group g+1 are read, and one DMA piece of the next
  // k-tile is issued, before the 12 MFMAs of group g; the 8 W fragments are read once at the top of the k-tile.
#define D3D_QKTILE(KT, PREFETCH)                                                                                         \
  do {                                                                                                                   \
    __syncthreads();                                                                                                     \
    const int nst = ((KT) + 1) & 1, nk0 = ((KT) + 1) * BK;                                                               \
    const unsigned char* sb = lds + ((KT) & 1) * STAGE;                                                                  \
    const unsigned char* sAh = sb;                                                                                       \
    const unsigned char* sAl = sb + A_PLANE;                                                                             \
    const unsigned char* sBh = sb + 2 * A_PLANE;                                                                         \
    const unsigned char* sBl = sb + 2 * A_PLANE + B_PLANE;                                                               \
    h8 bh[4], bl[4], ah[2], al[2];                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
      const int ob = swzq(brow0 + 16 * j, q);                                                                            \
      bh[j] = *reinterpret_cast<const h8*>(sBh + ob);                                                                    \
      bl[j] = *reinterpret_cast<const h8*>(sBl + ob);                                                                    \
    }                                                                                                                    \
    {                                                                                                                    \
      const int oa = swzq(arow0, q);                                                                                     \
      ah[0] = *reinterpret_cast<const h8*>(sAh + oa);                                                                    \
      al[0] = *reinterpret_cast<const h8*>(sAl + oa);                                                                    \
    }                                                                                                                    \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                                      \
      if (g + 1 < 8) {                                                                                                   \
        const int oa = swzq(arow0 + 16 * (g + 1), q);                                                                    \
        ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sAh + oa);                                                        \
        al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sAl + oa);                                                        \
      }                                                                                                                  \
      if (PREFETCH) D3D_QSTAGE_ONE(nst, nk0, g);                                                                         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);                        \
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);                        \
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);                        \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
  } while (0)

  int kt = 0;
  for (; kt + 1 < nk; ++kt) D3D_QKTILE(kt, true);
  D3D_QKTILE(kt, false);
#undef D3D_QKTILE
#undef D3D_QSTAGE_ONE

  const int mt0 = m0 + wm * 128, nt0 = n0 + wn * 64;          // wave-uniform
  const size_t tbase = (size_t)mt0 * N + nt0;
  const float* Rt = R ? R + tbase : nullptr;
  float* Ct = C ? C + tbase : nullptr;
  _Float16* Cht = Ch ? Ch + tbase : nullptr;
  _Float16* Clt = Cl ? Cl + tbase : nullptr;
  __syncthreads();   // every wave is done with the operand stages: reuse LDS for the transpose patches
  float* patch = reinterpret_cast<float*>(lds) + wave * (2 * 16 * 64);
  if (m0 + BM <= M && n0 + BN <= N)
    x3q_epilogue<EPI, OUTSPLIT, false>(acc, patch, bias, Rt, Ct, Cht, Clt, mt0, nt0, lane, M, N, qcols);
  else
    x3q_epilogue<EPI, OUTSPLIT, true>(acc, patch, bias, Rt, Ct, Cht, Clt, mt0, nt0, lane, M, N, qcols);
}

static hipError_t launch_x3q(const _Float16* Ah, const _Float16* Al, const _Float16* Wh, const _Float16* Wl, const float* bias,
                             const float* R, float* C, _Float16* Ch, _Float16* Cl, int M, int N, int K, int epi, int outsplit,
                             int qcols, hipStream_t s) {
  const int mtiles = (M + 255) / 256, ntiles = (N + 255) / 256;
  const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
  const size_t lds_bytes = 2 * (size_t)(4 * 256 * 64);
#define D3D_X3Q_LAUNCH(EPI_, OS_)                                                                                         \
  do {                                                                                                                    \
    auto kfn = k_linear_x3q<EPI_, OS_>;                                                                                   \
    static bool attr_done = false;                                                                                        \
    if (!attr_done) {                                                                                                     \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)lds_bytes);                                                                \
      if (ae != hipSuccess) return ae;                                                                                    \
      attr_done = true;                                                                                                   \
    }                                                                                                                     \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds_bytes, s, Ah, Al, Wh, Wl, bias, R, C, Ch, Cl, M, N, K, mtiles,     \
                       ntiles, qcols);                                                                                    \
  } while (0)
  if (outsplit) {
    if (epi == EPI_GELU) D3D_X3Q_LAUNCH(EPI_GELU, 1);
    else if (epi == EPI_NONE) D3D_X3Q_LAUNCH(EPI_NONE, 1);
    else return hipErrorInvalidValue;
  } else {
    if (epi == EPI_NONE) D3D_X3Q_LAUNCH(EPI_NONE, 0);
    else if (epi == EPI_GELU) D3D_X3Q_LAUNCH(EPI_GELU, 0);
    else if (epi == EPI_RESIDUAL) D3D_X3Q_LAUNCH(EPI_RESIDUAL, 0);
    else return hipErrorInvalidValue;
  }
#undef D3D_X3Q_LAUNCH
  return hipGetLastError();
}

template <int ABL>
static hipError_t launch_abl(const _Float16* Ah, const _Float16* Al, const _Float16* Wh, const _Float16* Wl, const float* bias,
                             float* C, int M, int N, int K, int ablate, hipStream_t s) {
  constexpr int BM = 256, BN = 256;
  const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
  const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
  const size_t lds_bytes = 2 * (size_t)(2 * BM * 64 + 2 * BN * 64);
  auto kfn = k_linear_x3p<256, 256, 2, 4, EPI_NONE, 0, ABL>;
  hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (ae != hipSuccess) return ae;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds_bytes, s, Ah, Al, Wh, Wl, bias, nullptr, C, nullptr, nullptr, M, N, K,
                     mtiles, ntiles, ablate, 0);
  return hipGetLastError();
}

// variant: 0 = auto, 1 = 128x128, 2 = 256x128, 3 = 256x256
hipError_t launch_linear_x3p(const void* Ah, const void* Al, const void* Wh, const void* Wl, const float* bias,
                             const float* R, float* C, void* Ch, void* Cl, int M, int N, int K, int epi, int outsplit,
                             int qcols, int variant, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % PBK) != 0 || (N % 4) != 0) return hipErrorInvalidValue;
  if (epi == EPI_RESIDUAL && R == nullptr) return hipErrorInvalidValue;
  if (outsplit ? (!Ch || !Cl) : !C) return hipErrorInvalidValue;
  const _Float16 *ah = (const _Float16*)Ah, *al = (const _Float16*)Al, *wh = (const _Float16*)Wh, *wl = (const _Float16*)Wl;
  _Float16 *ch = (_Float16*)Ch, *cl = (_Float16*)Cl;
  const int ablate = variant >> 4;
  variant &= 15;
  if (variant == 0) {
    // Measured on MI355X (experiments/gemm_bench.py): at M = 264k the 256x256 tile wins for every N in {512,1024,1536}.
    // Small batches need enough workgroups to fill 256 CUs for several rounds, so fall back to smaller tiles there.
    auto wgs = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (N % 256 == 0 && wgs(256, 256) >= 1024) variant = 13;   // 256x256 tile on v_mfma_f32_16x16x32_f16 (3-8 % over the 32x32x16 form)
    else if (N % 128 == 0 && wgs(256, 128) >= 1024) variant = 2;
    else variant = 1;
  }
  switch (variant) {
    case 1: return launch_tile<128, 128, 2, 2>(ah, al, wh, wl, bias, R, C, ch, cl, M, N, K, epi, outsplit, ablate, qcols, s);
    case 2: return launch_tile<256, 128, 4, 2>(ah, al, wh, wl, bias, R, C, ch, cl, M, N, K, epi, outsplit, ablate, qcols, s);
    case 3: return launch_tile<256, 256, 2, 4>(ah, al, wh, wl, bias, R, C, ch, cl, M, N, K, epi, outsplit, ablate, qcols, s);
    case 7: return launch_tile<256, 128, 2, 2, 16>(ah, al, wh, wl, bias, R, C, ch, cl, M, N, K, epi, outsplit, ablate, qcols, s);
    case 8: return launch_tile<256, 256, 2, 4, 16>(ah, al, wh, wl, bias, R, C, ch, cl, M, N, K, epi, outsplit, ablate, qcols, s);
    case 13: return launch_x3q(ah, al, wh, wl, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s);   // 16x16x32 MFMA form
    case 10: return launch_abl<4>(ah, al, wh, wl, bias, C, M, N, K, ablate, s);  // DMA issued in the first half of the groups
    case 11: return launch_abl<5>(ah, al, wh, wl, bias, C, M, N, K, ablate, s);  // ... first quarter
    case 4: return launch_abl<1>(ah, al, wh, wl, bias, C, M, N, K, ablate, s);   // timing experiments (wrong results)
    case 5: return launch_abl<2>(ah, al, wh, wl, bias, C, M, N, K, ablate, s);
    case 6: return launch_abl<3>(ah, al, wh, wl, bias, C, M, N, K, ablate, s);
    default: return hipErrorInvalidValue;
  }
}

// fp32 -> hi/lo planes of 8*x (stand-alone converter: tests, and any activation whose producer is not one of ours)
__global__ __launch_bounds__(256) void k_split_x3(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                  _Float16* __restrict__ lo, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float f[4] = {v.x, v.y, v.z, v.w};
  h4 a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float s = __builtin_amdgcn_fmed3f(f[j] * P_A_SCALE, -65504.0f, 65504.0f);
    a[j] = (_Float16)s;
    b[j] = (_Float16)(s - (float)a[j]);
  }
  reinterpret_cast<h4*>(hi)[i] = a;
  reinterpret_cast<h4*>(lo)[i] = b;
}

hipError_t launch_split_x3(const float* x, void* hi, void* lo, size_t n, hipStream_t s) {
  if (n % 4) return hipErrorInvalidValue;
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(k_split_x3, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, (_Float16*)hi, (_Float16*)lo, n4);
  return hipGetLastError();
}

}  // namespace d3d
