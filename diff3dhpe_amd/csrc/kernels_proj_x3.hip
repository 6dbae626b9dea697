// proj of the F16X3 block flow as its own kernel: x += attn Wproj^T + b, plane to plane in place, with the (sum, sum of squares) row
// partials of the new x for the LayerNorm folded into fc1 (S2S:84 + 127 behind S2S:101) -- on the hand-specialised two-phase k-loop of the
// fused kernels (qkv_fused_kloop.h) with a 192 x 256 x 32 stage: eight waves (2 x 4) of 96 rows x 64 columns, the shape and the MFMA order of
// k_linear_x3q_persist<6,2,4, EPI_RESIDUAL, pair-out, plane residual + row statistics>, whose epilogue function (x3q_epilogue8) it calls.
// Per element the same MFMAs in the same order and the same epilogue arithmetic: bit for bit the template's result.  Whole tiles only:
// the launcher hands the rows beyond the last whole 192-row tile to the template (their unguarded stores would reach the stream's pad rows).
#include "d3d_kernels.h"
#include "qkv_fused_kloop.h"

#include <math.h>
#include <stdio.h>

namespace d3d {
namespace {

#include "gemm_x3p_prelude.h"
#include "gemm_x3p_epilogue.h"

constexpr int PJ_TM = 6, PJ_NJ = 4;
constexpr int PJ_BM = 32 * PJ_TM, PJ_BN = 256;                           // 192 x 256
constexpr int PJ_AREG = PJ_BM * 128, PJ_STAGE = (PJ_BM + PJ_BN) * 128;   // 57344
constexpr int PJ_AIT = 3, PJ_BIT = 4;                                    // 1-KiB DMA pieces per wave per k-tile
constexpr int PJ_PATCH = PJ_STAGE;                                       // eight 8 KiB transpose patches over stage 1 (and beyond)
constexpr int PJ_STATP = PJ_PATCH + 65536;                               // a kilobyte per wave for the rows' statistics
constexpr int PJ_LDS = PJ_STATP + 8 * 1024;                              // 131072
static_assert(PJ_LDS <= 160 * 1024 && PJ_PATCH + 65536 >= 2 * PJ_STAGE, "LDS map");

__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

struct PjArgs {
  const _Float16* Ap;      // attention output, pair layout [>= 192 mtiles rows][2 K] of 8 o
  const _Float16* Wp;      // proj weight, pair layout, 2^k w, [N padded to 256][2 K]
  const float* bias;
  _Float16* X;             // the stream planes (pair layout [rows][2 N] of 8 x): residual in, new x out, in place
  float* st_out;           // (sum, sum of squares) partials of the new rows: [rows][N / 64][2]
  float out_scale;         // 2^-(3 + k)
  int M, N, K, mtiles, ntiles;   // M = 192 mtiles: whole tiles only
  unsigned* range;
};

#define PJ_GLDS(SRC, DSTOFF)                                                                                            \
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



__global__ __launch_bounds__(512) void k_proj_x3(PjArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int ntiles = a.ntiles, tiles = a.mtiles * ntiles;
  if (b >= tiles) return;
  const int nitems = (tiles - b + G - 1) / G;
  const int vfull = (a.mtiles / 8) * 8 * ntiles, mrem = a.mtiles % 8;
  auto tile_of = [&](int o, int& mt, int& nt) {   // all N-tiles of an M-tile on one XCD, the order of k_linear_x3q_persist
    if (o < vfull) {
      const int xcd = o & 7, slot = o >> 3;
      mt = (slot / ntiles) * 8 + xcd;
      nt = slot % ntiles;
    } else {
      const int o2 = o - vfull;
      mt = (a.mtiles / 8) * 8 + o2 % mrem;
      nt = o2 / mrem;
    }
  };
  const int K = a.K;
  const size_t K2 = 2 * (size_t)K;
  const int nk = K / 32;
  int mt = 0, nt = 0;
  tile_of(b, mt, nt);
  {   // first k-tile of the first tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lr = lane >> 3, csrc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
    const unsigned lofs = (unsigned)(lr * (int)K2 + csrc * 8) * 2u;
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(mt * PJ_BM + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(nt * PJ_BN + wave * 8) * K2 * 2;
    const size_t it_stride = (size_t)64 * K2 * 2;
#pragma unroll
    for (int it = 0; it < PJ_AIT; ++it) PJ_GLDS(sgpr_ptr(ubA + it * it_stride) + lofs, wave * 1024 + lane * 16 + it * 8192);
#pragma unroll
    for (int it = 0; it < PJ_BIT; ++it) PJ_GLDS(sgpr_ptr(ubB + it * it_stride) + lofs, PJ_AREG + wave * 1024 + lane * 16 + it * 8192);
  }
  int tid_o = (int)threadIdx.x;
  for (int item = 0; item < nitems; ++item) {
    asm volatile("" : "+v"(tid_o));   // per-lane offsets are re-derived in every tile instead of being hoisted (and spilled)
    const int tid = tid_o;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, q = lane >> 4;
    const bool has_next = item + 1 < nitems;
    int mtn = 0, ntn = 0;
    if (has_next) tile_of((item + 1) * G + b, mtn, ntn);
    const int m0 = mt * PJ_BM, n0 = nt * PJ_BN;

    // ---- DMA plan (kernels_gemm_x3p.hip D3D_DMA_PLAN)
    const int lr_ = lane >> 3;
    const int csrc_ = (lane & 7) ^ (((wave & 1) << 2) | (lr_ >> 1));
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(m0 + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(n0 + wave * 8) * K2 * 2;
    unsigned lofs_ = (unsigned)(lr_ * (int)K2 + csrc_ * 8) * 2u;
    const size_t it_stride = (size_t)64 * K2 * 2;
    const int dstA = wave * 1024 + lane * 16, dstB = PJ_AREG + wave * 1024 + lane * 16;
    const char* ubAn = reinterpret_cast<const char*>(a.Ap) + (size_t)(mtn * PJ_BM + wave * 8) * K2 * 2;
    const char* ubBn = reinterpret_cast<const char*>(a.Wp) + (size_t)(ntn * PJ_BN + wave * 8) * K2 * 2;
    // piece IT (A: 0..2, W: 3..6) of k-tile KTT of this tile, or (KTT == nk) of k-tile 0 of the next one
#define PJ_PIECE(KTT, IT)                                                                                               \
    do {                                                                                                                \
      const bool nxt_ = (KTT) >= nk;                                                                                    \
      const int st_ = ((KTT) & 1) * PJ_STAGE;                                                                           \
      if ((IT) < PJ_AIT) {                                                                                              \
        const char* b_ = nxt_ ? ubAn + (IT) * it_stride : ubA + ((size_t)(KTT) * 128 + (IT) * it_stride);               \
        PJ_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstA + (IT) * 8192);                                                        \
      } else {                                                                                                          \
        const char* b_ = nxt_ ? ubBn + ((IT) - PJ_AIT) * it_stride : ubB + ((size_t)(KTT) * 128 + ((IT) - PJ_AIT) * it_stride); \
        PJ_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstB + ((IT) - PJ_AIT) * 8192);                                             \
      }                                                                                                                 \
    } while (0)

    f32x4 acc[PJ_TM][PJ_NJ];
#pragma unroll
    for (int i = 0; i < PJ_TM; ++i)
#pragma unroll
      for (int j = 0; j < PJ_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    const int foff = (q ^ (r16 >> 1)) << 4;
    const int aoff = (wm * 16 * PJ_TM + r16) * 128 + foff, boff = PJ_AREG + (wn * 64 + r16) * 128 + foff;
    h8 bh[PJ_NJ], bl[PJ_NJ], ah[2], al[2];
    int issued_prev = 0;
#define QF_STAGE PJ_STAGE
#define QF_NJ PJ_NJ
#define QF_TM PJ_TM
#define QF_AIT PJ_AIT
#define QF_BIT PJ_BIT
#define QF_PIECE(KTT, IT) PJ_PIECE(KTT, IT)
    QF_KLOOP_HEAD
    QF_KLOOP_TAIL
#undef QF_STAGE
#undef QF_NJ
#undef QF_TM
#undef QF_AIT
#undef QF_BIT
#undef QF_PIECE
#undef PJ_PIECE
    __builtin_amdgcn_s_setprio(0);
    {
      const int mt0 = m0 + wm * 16 * PJ_TM, nt0 = n0 + wn * 64;
      const size_t tbase = (size_t)mt0 * a.N + nt0;
      float* const patch = reinterpret_cast<float*>(lds + PJ_PATCH) + wave * (2 * 16 * 64);
      float2* const statp = reinterpret_cast<float2*>(lds + PJ_STATP) + wave * 128;
      x3q_epilogue8<PJ_TM, 2, 4, EPI_RESIDUAL, 2, FX_RP | FX_SO, false>(acc, patch, lds + PJ_STATP, a.bias, nullptr, a.X + 2 * tbase, nullptr,
                                                                         a.X + 2 * tbase, nullptr, a.st_out, mt0, nt0, wm * 16 * PJ_TM, lane,
                                                                         a.M, a.N, 0, 0, PJ_TM, a.out_scale, a.range, statp);
    }
    mt = mtn; nt = ntn;
    __syncthreads();   // the patches (stage 1) are read before the next tile's second k-tile is staged there
  }
}

}  // namespace

bool proj_x3_ok(int N, int K) { return N % 256 == 0 && K % 64 == 0 && K >= 128; }

// X[rows < 192 * (M / 192)] += A W^T + b with the row partials; the caller runs rows beyond the last whole tile through launch_linear_x3p.
hipError_t launch_proj_x3(const void* Apair, const void* Wpair, const float* bias, void* Xpair, float* st_out, int w_exp, int M, int N, int K,
                          hipStream_t s) {
  if (!proj_x3_ok(N, K) || M < PJ_BM || M % PJ_BM != 0 || !Apair || !Wpair || !bias || !Xpair || !st_out) return hipErrorInvalidValue;
  if (w_exp < -14 || w_exp > 12) return hipErrorInvalidValue;
  PjArgs a{};
  a.Ap = (const _Float16*)Apair; a.Wp = (const _Float16*)Wpair; a.bias = bias; a.X = (_Float16*)Xpair; a.st_out = st_out;
  a.out_scale = ldexpf(1.0f, -(3 + w_exp));
  a.M = M; a.N = N; a.K = K; a.mtiles = M / PJ_BM; a.ntiles = N / PJ_BN;
  a.range = launch_range_word();
  static std::atomic<unsigned long long> attr_done{0};   // one bit per device
  if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(k_proj_x3), PJ_LDS, attr_done)) return ae;
  int n_cu = device_cu_count();
  if (n_cu <= 0) return hipErrorUnknown;
  const int tiles = a.mtiles * a.ntiles;
  const int grid = tiles < n_cu ? tiles : n_cu;
  hipLaunchKernelGGL(k_proj_x3, dim3(grid), dim3(512), PJ_LDS, s, a);
  return hipGetLastError();
}

}  // namespace d3d
