// fc2 + post-norm of the F16X3 block flow on a k-loop WITHOUT workgroup barriers: x = post_norm(x + hidden W2^T + b2) [+ pos] [+ time
// vector], whole rows per workgroup (S2S:48-54 + 131-135 behind S2S:236 / 245 and the additions of S2S:238-242 / 113-116).
//
// Shape and arithmetic are those of k_linear_x3q_persist<8,1,8, EPI_RESIDUAL, *, plane residual + post-norm> (128 x 512 tiles, eight waves
// side by side, 128 rows x 64 columns each; per element the same MFMAs in the same order; the template's own epilogue function
// x3q_epilogue_pn): bit for bit the template's result.  What differs is how the k-loop is fed and synchronised:
//
//   * W is WAVE-PRIVATE.  Wave w multiplies columns [64 w, 64 w + 64): the 64 W rows of a k-tile (8 KiB) that it reads are staged by
//     itself (8 LDS-DMA pieces) into its own slot, single-buffered -- the fragments of W(t) are in registers after the k-tile's first
//     instructions, W(t+1) is requested right behind them and has the rest of the k-tile to land.  Ordering W needs the wave's own
//     vmcnt and nothing else.  The slot doubles as the wave's two transpose patches in the epilogue.
//   * A (128 rows x 128 B = 16 KiB per k-tile, read by ALL waves) goes through a ring of four slots, two pieces per wave and k-tile,
//     requested TWO k-tiles ahead.  A wave says "my pieces of A(t+1) have landed" by adding 1 to the slot's counter in LDS in the MIDDLE
//     of its k-tile t (behind a counted vmcnt that leaves this k-tile's own pieces in flight), and opens k-tile t once that slot's
//     counter shows all eight arrivals.  There is no FREE counter: a wave that sees A(t) complete knows that every wave has passed
//     the middle of its k-tile t-1, i.e. is done reading A(t-2), whose slot A(t+2) takes.
//   * So no wave ever waits for another one to REACH a point -- only for data requested a k-tile earlier -- and the two waves of a SIMD
//     (w, w + 4) may drift up to a k-tile apart.  Waves 4-7 start each tile `delay` x 64 cycles late, so that a wave opens its k-tile
//     (counter poll, ten fragment reads) and issues its ten DMA pieces while its SIMD partner is in the MFMA-only half of its own.
//
// LDS: 8 x 8 KiB W slots | 4 x 16 KiB A ring | 8 KiB row-partial exchange | counters = 136 KiB + 64 B.
#include "d3d_kernels.h"

#include <math.h>
#include <stdio.h>

namespace d3d {
namespace {

#include "gemm_x3p_prelude.h"
#include "gemm_x3p_epilogue.h"

constexpr int R2_TM = 8, R2_NJ = 4;
constexpr int R2_BM = 16 * R2_TM, R2_BN = 512;
constexpr int R2_WSLOT = 8192, R2_ASLOT = R2_BM * 128, R2_NA = 4;
constexpr int R2_A = 8 * R2_WSLOT;                     // 65536
constexpr int R2_XCH = R2_A + R2_NA * R2_ASLOT;        // 131072
constexpr int R2_CNT = R2_XCH + R2_BM * 8 * 8;         // 139264
constexpr int R2_LDS = R2_CNT + 64;
static_assert(R2_LDS <= 160 * 1024, "LDS map");
constexpr unsigned RANGE_BIT_RING_TIMEOUT = 16u;       // a counter never filled (a bug, not a precision matter): the spin gave up

__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// Arrival counters in LDS, by address (the low 32 bits of a generic LDS pointer are the LDS offset).  lds_signal: one ds_add_u32 from
// lane 0.  lds_poll_ge: spins (ds_read_b32, s_sleep 1) until the word is >= target or `limit` reads were made; returns the reads made.
__device__ __forceinline__ void lds_signal(unsigned addr, int lane) {
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}
__device__ __forceinline__ unsigned lds_poll_ge(unsigned addr, unsigned target, unsigned limit) {
  unsigned v, n, s;
  asm volatile(
      "s_mov_b32 %1, 0\n"
      "1:\n\t"
      "ds_read_b32 %0, %3\n\t"
      "s_add_u32 %1, %1, 1\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %2, %0\n\t"
      "s_cmp_ge_u32 %2, %4\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_sleep 1\n\t"
      "s_cmp_lt_u32 %1, %5\n\t"
      "s_cbranch_scc1 1b\n"
      "2:\n"
      : "=&v"(v), "=&s"(n), "=&s"(s)
      : "v"(addr), "s"(target), "s"(limit)
      : "memory", "scc");
  return n;
}
constexpr int R2_SIG = 3;                       // the m-tile group behind which a wave signals its A pieces (the middle of its k-tile)
constexpr unsigned R2_POLL_LIMIT = 1u << 13;   // x (64 + ~100) cycles = ~1 ms (a k-tile is ~2 us): then the poll gives up and the launch is flagged

struct R2Args {
  const _Float16* Ap;      // hidden activation, pair layout in accumulator order, [>= 128 mtiles rows][2 K]
  const _Float16* Wp;      // fc2 weight, pair layout (k in accumulator order), 2^k w, [512][2 K]
  const float* bias;
  float* C;                // fp32 rows out (OUTSPLIT 0: last block) or nullptr
  _Float16* Ch;            // stream planes out (OUTSPLIT 2), in place over the residual
  X3Tail tail;             // out_scale, Rp (the stream planes: residual), st_out, pn, range
  int M, N, K, mtiles;
  int delay;               // waves 4-7 start each tile this many x 64 cycles late
  int dbg;                 // debugging: 1 workgroup barrier at every k-tile top, 2 at every tile end, 4 vmcnt(0) + barrier in the hook
  unsigned long long* diag;   // nullable: per workgroup {k-loop cycles, epilogue cycles, tiles, poll spins} of wave 0 / wave 4
};

// The product build carries no debugging path: -DR2_DEBUG builds (experiments/build_variant.sh ringdbg "-DR2_DEBUG" kernels_fc2_ring)
// enable the "fc2_ring_dbg" bits (1 workgroup barrier at every k-tile top, 2 at every tile end, 4 in the epilogue hook, 8 W(0) of the next
// tile issued behind the epilogue, 16 / 32 no W / A pieces, 64 no counters: timing probes with wrong results, 512 static priority for waves
// 4-7) and the "fc2_ring_diag" stamps (experiments/fc2_ring_op.py).
#ifdef R2_DEBUG
#define R2_DBG(BIT) ((a.dbg & (BIT)) != 0)
#define R2_DIAG (a.diag != nullptr)
#else
#define R2_DBG(BIT) false
#define R2_DIAG false
#endif
// -DR2_STAMPS builds (with -DR2_DEBUG): shader-clock stamps inside every k-tile of waves 0 and 4 -- cycles from the k-tile's
// top to: the own pieces landed (vmcnt), the A slot complete (poll), the first MFMA group done, the signal, the end.
#ifdef R2_STAMPS
#define R2_STAMP(I) do { if (R2_DIAG) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[I] += t_ - tprev; tprev = t_; } } while (0)
#else
#define R2_STAMP(I) do { } while (0)
#endif

#define R2_GLDS(SRC, DSTOFF)                                                                                            \
  __builtin_amdgcn_global_load_lds((SRC), (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (DSTOFF)), 16, 0, 0)

template <int OUTSPLIT>
__global__ __launch_bounds__(512) void k_fc2_ring(R2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  if (b >= a.mtiles) return;
  const int nitems = (a.mtiles - b + G - 1) / G;
  const int K = a.K, nk = K / 32;
  const size_t rowb = 4 * (size_t)K;                   // bytes of an operand row (2 K fp16)
  if (threadIdx.x < R2_NA) reinterpret_cast<unsigned*>(lds + R2_CNT)[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned cnt = (unsigned)(uintptr_t)(lds + R2_CNT);   // LDS address of the four arrival counters

  int tid_o = (int)threadIdx.x;
  unsigned T = 0;                                      // k-tiles this workgroup has opened (the ring position)
  unsigned tmo = 0;                                    // a poll gave up: the launch is flagged (its results are wrong)
  unsigned long long dg_k = 0, dg_e = 0;
#ifdef R2_STAMPS
  unsigned long long seg[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
  unsigned dg_s = 0;

  // ---- the first tile's A(0), A(1), W(0)
  {
    const int lane = tid_o & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_o >> 6);
    const int lr = lane >> 3, c8 = lane & 7;
    const unsigned lofs_e = (unsigned)(lr * (int)rowb + ((c8 ^ (lr >> 1)) << 4)), lofs_o = lofs_e ^ 64u;
    const unsigned lofs_a = (wave & 1) ? lofs_o : lofs_e;
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(b * R2_BM + wave * 8) * rowb;
    const char* ubW = reinterpret_cast<const char*>(a.Wp) + (size_t)(wave * 64) * rowb;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int it = 0; it < 2; ++it)
        R2_GLDS(sgpr_ptr(ubA + (size_t)kt * 128 + (size_t)it * 64 * rowb) + lofs_a, R2_A + kt * R2_ASLOT + wave * 1024 + lane * 16 + it * 8192);
#pragma unroll
    for (int p = 0; p < 8; ++p)
      R2_GLDS(sgpr_ptr(ubW + (size_t)p * 8 * rowb) + ((p & 1) ? lofs_o : lofs_e), wave * R2_WSLOT + p * 1024 + lane * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_signal(cnt, lane);                             // A(0); A(1) is signalled at the top of k-tile 0 like every A(t+1)
  }

  for (int item = 0; item < nitems; ++item) {
    asm volatile("" : "+v"(tid_o));   // per-lane offsets are re-derived in every tile instead of being hoisted (and spilled)
    const int tid = tid_o;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const bool has_next = item + 1 < nitems;
    const int mt = b + item * G, mtn = mt + G;
    const int m0 = mt * R2_BM;
    unsigned long long st0 = 0;
    if (R2_DIAG) st0 = __builtin_amdgcn_s_memtime();
#ifdef R2_STAMPS
    tprev = st0;
#endif

    if (wave >= 4) for (int i = 0; i < a.delay; ++i) __builtin_amdgcn_s_sleep(1);
    if (R2_DBG(512) && wave >= 4) __builtin_amdgcn_s_setprio(1);

    const int lr = lane >> 3, c8 = lane & 7;
    unsigned lofs_e = (unsigned)(lr * (int)rowb + ((c8 ^ (lr >> 1)) << 4));
    const char* const ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(m0 + wave * 8) * rowb;
    const char* const ubAn = reinterpret_cast<const char*>(a.Ap) + (size_t)(mtn * R2_BM + wave * 8) * rowb;
    const char* const ubW = reinterpret_cast<const char*>(a.Wp) + (size_t)(wave * 64) * rowb;
    const int dstA = R2_A + wave * 1024, dstW = wave * R2_WSLOT;   // wave-uniform: M0 holds the wave's base, the hardware adds lane * 16
    const size_t a_it = (size_t)64 * rowb, w_it = (size_t)8 * rowb;

    f32x4 acc[R2_TM][R2_NJ];
#pragma unroll
    for (int i = 0; i < R2_TM; ++i)
#pragma unroll
      for (int j = 0; j < R2_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    const int foff = (q ^ (r16 >> 1)) << 4;
    const int aoff = r16 * 128 + foff, boff = wave * R2_WSLOT + r16 * 128 + foff;
    h8 bh[R2_NJ], bl[R2_NJ], ah[2], al[2];

    // A piece IT (0, 1) of ring k-tile T + 2, from SRC (this tile's k-tile kt + 2 or the next tile's kt + 2 - nk)
#define R2_APIECE(SRC, IT)                                                                                              \
    R2_GLDS(sgpr_ptr((SRC) + (IT) * a_it) + (lofs_e ^ ((wave & 1) ? 64u : 0u)), (int)((T + 2u) & 3u) * R2_ASLOT + dstA + (IT) * 8192)
    // W piece P (0 .. 7) of this tile's k-tile KTT
#define R2_WPIECE(KTT, P)                                                                                               \
    R2_GLDS(sgpr_ptr(ubW + ((size_t)(KTT) * 128 + (P) * w_it)) + (lofs_e ^ (((P) & 1) ? 64u : 0u)), dstW + (P) * 1024)

    // The opening of a tile's first k-tile: the wave's own pieces landed, its W(T) fragments, the A slot complete, the first A pair.
    // Every later k-tile finds all of that done by the last group of its predecessor (PREF below).
#define R2_OPEN()                                                                                                       \
    do {                                                                                                                \
      R2_STAMP(5);                                                                                                      \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* my W(T) has landed (and everything older) */               \
      R2_STAMP(0);                                                                                                      \
      _Pragma("unroll") for (int j = 0; j < R2_NJ; ++j) {   /* W: this wave's own slot */                               \
        bh[j] = *reinterpret_cast<const h8*>(lds + boff + j * 2048);                                                    \
        bl[j] = *reinterpret_cast<const h8*>(lds + ((boff + j * 2048) ^ 64));                                           \
      }                                                                                                                 \
      if (!R2_DBG(64)) {                                                                                              \
        const unsigned n_ = (unsigned)__builtin_amdgcn_readfirstlane(                                                  \
            (int)lds_poll_ge(cnt + 4u * (T & 3u), 8u * ((T >> 2) + 1u), R2_POLL_LIMIT));                                \
        tmo |= n_ >= R2_POLL_LIMIT ? 1u : 0u;                                                                           \
        if (R2_DIAG) dg_s += n_;                                                                                         \
      }                                                                                                                 \
      R2_STAMP(1);                                                                                                      \
      asm volatile("" : "+v"(lofs_e) : : "memory");                                                                     \
      const unsigned char* const sa0 = lds + R2_A + (int)(T & 3u) * R2_ASLOT;                                           \
      ah[0] = *reinterpret_cast<const h8*>(sa0 + aoff);                                                                 \
      al[0] = *reinterpret_cast<const h8*>(sa0 + (aoff ^ 64));                                                          \
    } while (0)

    // One k-tile; on entry W(T)'s fragments and the first A pair are in registers and the A slot is known complete.
    // SRCA: wave-uniform source of this wave's first A piece of ring k-tile T + 2; DO_A / DO_W: whether those / the next k-tile's W pieces
    // exist (literal true in the steady state: no branch inside the MFMA stream); STEADY: the signal's wait is counted.
    // Groups: g0 .. g3 2 W pieces each, g4 2 A pieces, g3 the signal, g6 requests the next slot's counter, g7 (PREF) opens the NEXT
    // k-tile: own pieces landed (vmcnt 0), the next A slot complete, its first A pair requested, W(T+1)'s fragments behind the last triples.
#define R2_KTILE(KT, SRCA, DO_A, DO_W, STEADY, PREF)                                                                    \
    do {                                                                                                                \
      const unsigned char* const sa = lds + R2_A + (int)(T & 3u) * R2_ASLOT;                                            \
      unsigned cv_ = 0;                                                                                                 \
      _Pragma("unroll") for (int g = 0; g < R2_TM; ++g) {                                                               \
        if ((PREF) && g == R2_TM - 2 && !R2_DBG(64))   /* the next slot's counter, read a group ahead of its use */    \
          asm volatile("ds_read_b32 %0, %1" : "=v"(cv_) : "v"(cnt + 4u * ((T + 1u) & 3u)) : "memory");                  \
        if ((PREF) && g == R2_TM - 1) {                                                                                 \
          R2_STAMP(4);                                                                                                  \
          if (STEADY) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");   /* my W(T+1) (the two A pieces behind it stay in flight); the counter word; this group's A pair */ \
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                              \
          R2_STAMP(0);                                                                                                  \
          if (!R2_DBG(64)) {                                                                                          \
            const unsigned tgt_ = 8u * (((T + 1u) >> 2) + 1u);                                                          \
            unsigned n_ = 0;                                                                                            \
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)cv_) < tgt_) {                                            \
              n_ = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_poll_ge(cnt + 4u * ((T + 1u) & 3u), tgt_, R2_POLL_LIMIT)); \
              tmo |= n_ >= R2_POLL_LIMIT ? 1u : 0u;                                                                     \
            }                                                                                                           \
            /* the wave that did not have to wait is the one behind: it gets the matrix pipe first */                    \
            if (n_ == 0u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);                             \
            if (R2_DIAG) dg_s += n_;                                                                                     \
          }                                                                                                             \
          R2_STAMP(1);                                                                                                  \
          asm volatile("" : "+v"(lofs_e) : : "memory");                                                                 \
          const unsigned char* const san = lds + R2_A + (int)((T + 1u) & 3u) * R2_ASLOT;                                \
          ah[0] = *reinterpret_cast<const h8*>(san + aoff);                                                             \
          al[0] = *reinterpret_cast<const h8*>(san + (aoff ^ 64));                                                      \
        }                                                                                                               \
        if (g + 1 < R2_TM) {                                                                                            \
          ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sa + aoff + (g + 1) * 2048);                                   \
          al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sa + ((aoff + (g + 1) * 2048) ^ 64));                          \
        }                                                                                                               \
        if (g < 4) {                                                                                                    \
          if (DO_W) { R2_WPIECE((KT) + 1, 2 * g); R2_WPIECE((KT) + 1, 2 * g + 1); }                                     \
        } else if (g == 4) {                                                                                            \
          if (DO_A) { R2_APIECE(SRCA, 0); R2_APIECE(SRCA, 1); }                                                         \
        }                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < R2_NJ; ++j) {                                                             \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);                     \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);                     \
          acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);                     \
          if ((PREF) && g == R2_TM - 1) {   /* W(T+1)'s fragments replace W(T)'s behind their last use */                \
            bh[j] = *reinterpret_cast<const h8*>(lds + boff + j * 2048);                                                \
            bl[j] = *reinterpret_cast<const h8*>(lds + ((boff + j * 2048) ^ 64));                                       \
          }                                                                                                             \
        }                                                                                                               \
        if ((PREF) && g == R2_TM - 1) {   /* the next A pair, then MFMA triple, its W pair's successor, ... */           \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
        } else {        /* 2 MFMAs, a read, 2 MFMAs, a read, 2 MFMAs, a piece, 2 MFMAs, the other piece, 4 MFMAs */     \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                            \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                            \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (g == 0) R2_STAMP(2);                                                                                        \
        if (g == R2_SIG) {   /* "my pieces of A(T+1) have landed": everything older than this k-tile's 8 W pieces */          \
          if (STEADY) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                  \
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
          if (!R2_DBG(64)) lds_signal(cnt + 4u * ((T + 1u) & 3u), lane);                                              \
          __builtin_amdgcn_sched_barrier(0);                                                                            \
          R2_STAMP(3);                                                                                                  \
        }                                                                                                               \
      }                                                                                                                 \
      ++T;                                                                                                              \
    } while (0)

    int kt = 0;
    R2_OPEN();
    if (R2_DBG(48)) {   // timing probes (wrong results): 16 no W pieces, 32 no A pieces in the steady state
      const bool pa = !R2_DBG(32), pw = !R2_DBG(16);
      for (; kt + 2 < nk; ++kt) R2_KTILE(kt, ubA + (size_t)(kt + 2) * 128, pa, pw, false, true);
    }
    for (; kt + 2 < nk; ++kt) R2_KTILE(kt, ubA + (size_t)(kt + 2) * 128, true, true, true, true);
    R2_KTILE(kt, ubAn, has_next, true, false, true);      // k-tile nk - 2: A(0) of the next tile
    ++kt;
    R2_KTILE(kt, ubAn + 128, has_next, false, false, false);   // k-tile nk - 1: A(1) of the next tile; its W(0) waits for the patches
#undef R2_OPEN
#undef R2_KTILE
#undef R2_APIECE

    unsigned long long st1 = 0;
    if (R2_DIAG) { st1 = __builtin_amdgcn_s_memtime(); dg_k += st1 - st0; }
    {
      const int nt0 = wave * 64;
      const size_t tbase = (size_t)m0 * a.N + nt0;
      float* const patch = reinterpret_cast<float*>(lds + wave * R2_WSLOT);   // the wave's W slot: dead until the hook refills it
      float* const xch = reinterpret_cast<float*>(lds + R2_XCH);
      float* const Ct = a.C ? a.C + tbase : nullptr;
      _Float16* const Cht = a.Ch ? a.Ch + 2 * tbase : nullptr;
      const _Float16* const Rpt = a.tail.Rp + 2 * tbase;
      auto refill = [&]() {   // W(0) of the next tile (the same rows again), behind the last read of the patches
#ifdef R2_STAMPS
        if (R2_DIAG) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[6] += t_ - st1; }   // sweep 1 of the epilogue
#endif
        if (R2_DBG(4)) __syncthreads();
        if (has_next && !R2_DBG(8)) {
#pragma unroll
          for (int p = 0; p < 8; ++p) R2_WPIECE(0, p);
        }
      };
      x3q_epilogue_pn<R2_TM, 8, OUTSPLIT, true, true>(acc, patch, xch, a.bias, Ct, Cht, Rpt, a.tail, m0, nt0, wave, lane, a.M, a.N, 0, R2_TM,
                                                      refill);
    }
#undef R2_WPIECE
    if (R2_DIAG) dg_e += __builtin_amdgcn_s_memtime() - st1;
    if (R2_DBG(2)) __syncthreads();
    if (R2_DBG(8) && has_next) {   // the next tile's W(0) only now
      const int lane = tid_o & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid_o >> 6);
      const int lr = lane >> 3, c8 = lane & 7;
      const unsigned lofs_e = (unsigned)(lr * (int)rowb + ((c8 ^ (lr >> 1)) << 4)), lofs_o = lofs_e ^ 64u;
      const char* ubW = reinterpret_cast<const char*>(a.Wp) + (size_t)(wave * 64) * rowb;
#pragma unroll
      for (int p = 0; p < 8; ++p)
        R2_GLDS(sgpr_ptr(ubW + (size_t)p * 8 * rowb) + ((p & 1) ? lofs_o : lofs_e), wave * R2_WSLOT + p * 1024 + lane * 16);
    }
  }
  if (tmo) range_raise(a.tail.range, RANGE_BIT_RING_TIMEOUT);
  if (R2_DIAG && (threadIdx.x == 0 || threadIdx.x == 256)) {
    unsigned long long* d = a.diag + 8 * b + (threadIdx.x ? 4 : 0);
    d[0] = dg_k; d[1] = dg_e; d[2] = (unsigned long long)nitems; d[3] = dg_s;
#ifdef R2_STAMPS
    unsigned long long* e = a.diag + 8 * gridDim.x + 14 * b + (threadIdx.x ? 7 : 0);
    for (int i = 0; i < 7; ++i) e[i] = seg[i];
#endif
  }
}

}  // namespace

bool fc2_ring_ok(int N, int K) { return N == 512 && K % 128 == 0 && K >= 256; }

static std::atomic<int> g_r2_delay{24}, g_r2_dbg{0};
void set_fc2_ring_dbg(int d) { g_r2_dbg = d; }
static std::atomic<unsigned long long*> g_r2_diag{nullptr};
void set_fc2_ring_delay(int d) { g_r2_delay = d < 0 ? 0 : (d > 4096 ? 4096 : d); }
void set_fc2_ring_diag(unsigned long long* dev_buf) { g_r2_diag = dev_buf; }

// The post-norm form of launch_linear_x3p (X3Fold with Rp + pn; outsplit 2: planes + st_out, 0: fp32 rows), every tile -- the ragged last
// one included (checked epilogue) -- by this kernel; A (rows padded to whole 128-row tiles) and W as there.
hipError_t launch_fc2_ring(const void* Apair, const void* Wpair, const float* bias, float* C, void* Ch, int M, int N, int K, int outsplit,
                           const X3Fold* fold, int w_exp, hipStream_t s) {
  if (!fc2_ring_ok(N, K) || M <= 0 || !fold || !fold->Rp || !fold->pn.g || !fold->pn.b || !bias || (outsplit == 2 ? (!Ch || !fold->st_out) : !C))
    return hipErrorInvalidValue;
  if ((outsplit != 0 && outsplit != 2) || w_exp < -14 || w_exp > 12) return hipErrorInvalidValue;
  if (fold->pn.pos && (fold->pn.pos_div < 1 || fold->pn.pos_mod < 1)) return hipErrorInvalidValue;
  if (fold->pn.tvec && fold->pn.tvec_stride != 0 && fold->pn.rows_per_batch < 1) return hipErrorInvalidValue;
  R2Args a{};
  a.Ap = (const _Float16*)Apair; a.Wp = (const _Float16*)Wpair; a.bias = bias; a.C = outsplit == 0 ? C : nullptr;
  a.Ch = outsplit == 2 ? (_Float16*)Ch : nullptr;
  a.tail.out_scale = ldexpf(1.0f, -(3 + w_exp));
  a.tail.range = launch_range_word();
  a.tail.Rp = (const _Float16*)fold->Rp; a.tail.st_out = fold->st_out; a.tail.pn = fold->pn;
  a.M = M; a.N = N; a.K = K; a.mtiles = (M + R2_BM - 1) / R2_BM;
  a.delay = g_r2_delay.load();
  a.dbg = g_r2_dbg.load();
  a.diag = g_r2_diag.load();
  const int n_cu = device_cu_count();
  if (n_cu <= 0) return hipErrorUnknown;
  const int grid = a.mtiles < n_cu ? a.mtiles : n_cu;
  if (outsplit == 2) {
    static std::atomic<unsigned long long> attr_done{0};   // one bit per device
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(k_fc2_ring<2>), R2_LDS, attr_done)) return ae;
    hipLaunchKernelGGL(k_fc2_ring<2>, dim3(grid), dim3(512), R2_LDS, s, a);
  } else {
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(k_fc2_ring<0>), R2_LDS, attr_done)) return ae;
    hipLaunchKernelGGL(k_fc2_ring<0>, dim3(grid), dim3(512), R2_LDS, s, a);
  }
  return hipGetLastError();
}

}  // namespace d3d
