// fp32-accurate token GEMM on the fp16 matrix pipe of gfx950 ("F16X3"):
//
//   C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] ),   A, C fp32 in HBM;  W pre-split into two fp16 planes at load time.
//
// Every fp32 operand x is represented as hi + lo with hi = fp16(s x), lo = fp16(s x - hi) (s a power of two: 2^3 for
// activations, 2^12 for weights, so that lo stays a NORMAL fp16 for every value that matters).  hi+lo carries 22
// significand bits, and the product is assembled from three v_mfma_f32_32x32x16_f16 per k-step into ONE fp32
// accumulator:   acc += a_lo b_hi;  acc += a_hi b_lo;  acc += a_hi b_hi     (a_lo b_lo ~ 2^-22 is dropped),
// un-scaled by 2^-15 in the epilogue.  Measured on MI355X (experiments/f16x3_probe.hip) the result is as close to
// the fp64 product as v_mfma_f32_32x32x2_f32 is (rms error 0.7x, max 0.5-0.8x): the fp16 MFMA sums 16 exact
// products per instruction before it rounds, so there are 8x fewer roundings than in an fp32 fmaf chain.
// Cost: 3 MFMAs at the 2.5 PFLOP/s fp16 rate instead of 1 at the 157 TFLOP/s fp32 rate = 5.3x the fp32-MFMA peak.
//
// Tile 128x128x32, 256 threads = 2x2 waves of 64x64 (64 accumulator VGPRs).  LDS holds four fp16 planes
// (A_hi, A_lo, W_hi, W_lo), 128 rows x 64 B each, 16-byte chunks XOR-swizzled by (row>>2)&3 so that every
// ds_read_b128 16-lane group touches 16 distinct 4-bank slots.  A is split hi/lo on the fly while it is staged
// (fp32 global -> registers -> fp16 LDS); the staging loads of k-tile t+1 are issued before the MFMAs of k-tile t.
#include <math.h>
#include "d3d_kernels.h"

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int XBM = 128, XBN = 128, XBK = 32;
constexpr int PLANE_BYTES = 128 * 64;                 // one fp16 plane of a tile
constexpr float A_SCALE = 8.0f;                       // 2^3
constexpr float OUT_SCALE = 1.0f / 32768.0f;          // 2^-(3+12)

__device__ __forceinline__ float gelu_erf_x(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// byte offset of logical 16-byte chunk `c` (0..3) of tile row `row` inside a plane
__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void split8(const float4 lo4, const float4 hi4, h8& vh, h8& vl) {
  const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s = __builtin_amdgcn_fmed3f(x[j] * A_SCALE, -65504.0f, 65504.0f);
    const _Float16 h = (_Float16)s;
    vh[j] = h;
    vl[j] = (_Float16)(s - (float)h);
  }
}

// Interior tiles take a branch-free epilogue: a per-element bounds branch makes hipcc wait vmcnt(0) (= for the previous
// STORE, stores count in vmcnt on gfx950) before every store.
template <int EPI, bool CHECK>
__device__ __forceinline__ void x3_epilogue(f32x16 (&acc)[2][2], const float* __restrict__ bias, const float* R, float* C, int mw,
                                             int nw, int M, int N) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nw + j * 32;
    if (CHECK && n >= N) continue;
    const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float rv[16];
      if (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = mw + i * 32 + (q & 3) + 8 * (q >> 2);
          rv[q] = (!CHECK || m < M) ? R[(size_t)m * N + n] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = mw + i * 32 + (q & 3) + 8 * (q >> 2);
        if (CHECK && m >= M) continue;
        float v = acc[i][j][q] * OUT_SCALE + bn;
        if (EPI == EPI_GELU) v = gelu_erf_x(v);
        if (EPI == EPI_RESIDUAL) v = rv[q] + v;
        C[(size_t)m * N + n] = v;
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_linear_f16x3(const float* __restrict__ A, const _Float16* __restrict__ Wp,
                                                          const float* __restrict__ bias,
                                                          const float* R, float* C, int M, int N, int K, int mtiles,
                                                          int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * PLANE_BYTES];
  unsigned char* const sAh = lds;
  unsigned char* const sAl = lds + PLANE_BYTES;
  unsigned char* const sBh = lds + 2 * PLANE_BYTES;
  unsigned char* const sBl = lds + 3 * PLANE_BYTES;

  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / ntiles) * 8 + xcd;   // all N-tiles of one M-tile run back to back on one XCD (shared L2)
  const int nt = slot % ntiles;
  if (mt >= mtiles) return;
  const int m0 = mt * XBM, n0 = nt * XBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // staging map: row = tid>>2 (+64), 16-byte fp16 chunk = tid&3  <=> 8 consecutive k
  const int srow = tid >> 2, sc = tid & 3;
  float4 ra[2][2];
  uint4 rwh[2], rwl[2];

  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = srow + 64 * p;
      const int gm = m0 + row, gn = n0 + row;
      if (gm < M) {
        const float* ap = A + (size_t)gm * K + k0 + sc * 8;
        ra[p][0] = *reinterpret_cast<const float4*>(ap);
        ra[p][1] = *reinterpret_cast<const float4*>(ap + 4);
      } else {
        ra[p][0] = make_float4(0, 0, 0, 0);
        ra[p][1] = make_float4(0, 0, 0, 0);
      }
      if (gn < N) {
        const size_t wo = (size_t)gn * 2 * K + 2 * k0 + sc * 8;   // pair layout: k-tile k0/32 is the 64 fp16 at 2*k0
        rwh[p] = *reinterpret_cast<const uint4*>(Wp + wo);
        rwl[p] = *reinterpret_cast<const uint4*>(Wp + wo + PAIR_LO);
      } else {
        rwh[p] = make_uint4(0, 0, 0, 0);
        rwl[p] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = srow + 64 * p;
      const int off = swz(row, sc);
      h8 vh, vl;
      split8(ra[p][0], ra[p][1], vh, vl);
      *reinterpret_cast<h8*>(sAh + off) = vh;
      *reinterpret_cast<h8*>(sAl + off) = vl;
      *reinterpret_cast<uint4*>(sBh + off) = rwh[p];
      *reinterpret_cast<uint4*>(sBl + off) = rwl[p];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

  const int arow0 = wm * 64 + r, brow0 = wn * 64 + r;
  const int nk = K / XBK;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * XBK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + h;   // lane half h feeds k = 16 ks + 8 h .. +7 of the 32x32x16 MFMA
      h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oa = swz(arow0 + 32 * i, c), ob = swz(brow0 + 32 * i, c);
        ah[i] = *reinterpret_cast<const h8*>(sAh + oa);
        al[i] = *reinterpret_cast<const h8*>(sAl + oa);
        bh[i] = *reinterpret_cast<const h8*>(sBh + ob);
        bl[i] = *reinterpret_cast<const h8*>(sBl + ob);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  const int mw = m0 + wm * 64 + 4 * h, nw = n0 + wn * 64 + r;
  if (m0 + 128 <= M && n0 + 128 <= N) x3_epilogue<EPI, false>(acc, bias, R, C, mw, nw, M, N);
  else x3_epilogue<EPI, true>(acc, bias, R, C, mw, nw, M, N);
}

hipError_t launch_linear_f16x3(const float* A, const void* Wpair, const float* bias, const float* R, float* C, int M, int N,
                               int K, int epi, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % XBK) != 0) return hipErrorInvalidValue;
  if (epi == EPI_RESIDUAL && R == nullptr) return hipErrorInvalidValue;
  const int mtiles = (M + XBM - 1) / XBM, ntiles = (N + XBN - 1) / XBN;
  const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
  const _Float16* wp = reinterpret_cast<const _Float16*>(Wpair);
  switch (epi) {
    case EPI_NONE:
      hipLaunchKernelGGL(k_linear_f16x3<EPI_NONE>, dim3(grid), dim3(256), 0, s, A, wp, bias, R, C, M, N, K, mtiles, ntiles);
      break;
    case EPI_GELU:
      hipLaunchKernelGGL(k_linear_f16x3<EPI_GELU>, dim3(grid), dim3(256), 0, s, A, wp, bias, R, C, M, N, K, mtiles, ntiles);
      break;
    case EPI_RESIDUAL:
      hipLaunchKernelGGL(k_linear_f16x3<EPI_RESIDUAL>, dim3(grid), dim3(256), 0, s, A, wp, bias, R, C, M, N, K, mtiles, ntiles);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Host-side weight split: w [rows, cols] -> pair layout (d3d_kernels.h) of hi/lo fp16 of s*w with s = 2^12
// (round-to-nearest-even both times).
// acc_order: the k index of every 32-column group is stored in the order pair_slot_acc() gives it (the layout of the
// hidden activation that the fc1 epilogue writes straight from its accumulators, kernels_gemm_x3p.hip)
int split_weight_f16x3(const float* w, size_t rows, size_t cols, uint16_t* pair, bool acc_order, bool* clamped) {
  // per-matrix scale 2^k: 2^12 whenever the matrix fits (|w| <= 15.99: every matrix of a sane checkpoint -- then the planes are
  // bit for bit what a fixed 4096 gives), smaller when a weight is larger -- LayerNorm-FOLDED weights W diag(gamma) of a
  // checkpoint with big gains get there (gamma 6 x |w| 8 = 48).  The GEMM un-scales by 2^-(3+k) (X3Tail::out_scale); powers of
  // two, so nothing rounds differently.  Small entries whose lo half then falls into fp16 denormals lose bits below
  // 2^-24 2^-k: < 1e-9 absolute at k >= 5.
  float amax = 0.0f;
  bool finite = true;
  for (size_t i = 0; i < rows * cols; ++i) {
    const float a = fabsf(w[i]);
    if (!(a <= 3.0e38f)) finite = false;
    else if (a > amax) amax = a;
  }
  int k = 12;
  while (k > -14 && ldexpf(amax, k) > 65504.0f) --k;
  const float scale = ldexpf(1.0f, k);
  bool in_range = finite;
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < cols; ++c) {
      float s = w[r * cols + c] * scale;
      if (!(s <= 65504.0f && s >= -65504.0f)) in_range = false;
      if (s > 65504.0f) s = 65504.0f;
      if (s < -65504.0f) s = -65504.0f;
      const _Float16 h = (_Float16)s;
      const _Float16 l = (_Float16)(s - (float)h);
      uint16_t* o = pair + r * 2 * cols + (acc_order ? pair_col_acc((int)c) : pair_col((int)c));
      __builtin_memcpy(o, &h, 2);
      __builtin_memcpy(o + PAIR_LO, &l, 2);
    }
  if (clamped && !in_range) *clamped = true;
  return k;
}

}  // namespace d3d
