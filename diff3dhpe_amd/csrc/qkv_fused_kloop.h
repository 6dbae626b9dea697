// The k-loop of the two fused qkv + attention kernels (kernels_qkv_sattn.hip, kernels_qkv_tattn.hip): the two-phase persistent loop of
// kernels_gemm_x3p.hip (D3D_PHASE, WPF form) hand-specialised for whole tiles -- eight waves (2 x 4) of 8 QF_TM rows x 16 QF_NJ columns (256 x 192 / 256 x 256 / 192 x 256 stages),
// W fragments a phase ahead, counted vmcnt waits.  Included INSIDE the tile loop of a kernel that has defined, in scope:
//   QF_STAGE, QF_NJ, QF_TM, QF_AIT, QF_BIT (constants), QF_PIECE(KTT, IT) (issues DMA piece IT of k-tile KTT), wait_vm(n),
//   lds, acc[QF_TM][QF_NJ], aoff, boff, ah[2], al[2], bh[QF_NJ], bl[QF_NJ], lofs_, issued_prev, nk, has_next.
// QF_KLOOP_HEAD runs k-tiles 0 .. nk - 2 (declares kt), QF_KLOOP_TAIL the last one: the kernels reduce their row statistics in between,
// in the shadow of the SIMD partner's MFMAs.
#pragma once

// QF_MMA(ACC, BH, BL, AH, AL) (optional): the products of one (m-tile, n-tile) pair for one staged 128-byte line of each operand row.
// Default: F16X3 -- the line holds the 32 hi and the 32 lo halves of a 32-deep k-tile: a_lo b_hi + a_hi b_lo + a_hi b_hi, smallest terms
// first.  kernels_gemm_bf16q.hip defines the bf16 form (the line holds 64 bf16 k values: one MFMA per 16-byte fragment pair).
#ifndef QF_MMA
#define QF_MMA(ACC, BH, BL, AH, AL)                                                                                      \
  do {                                                                                                                   \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(BH, AL, ACC, 0, 0, 0);                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(BL, AH, ACC, 0, 0, 0);                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(BH, AH, ACC, 0, 0, 0);                                                  \
  } while (0)
#endif

    // one phase (kernels_gemm_x3p.hip D3D_PHASE, WPF form): H = 0: m-tiles 0-3 of k-tile KT, issues A(KT+1) (and all of W(1), ahead of
    // A(1), in a tile's first phase); H = 1: m-tiles 4-7, issues W(KT+2); the W fragments of KT+1 replace those of KT behind the
    // last group's MFMA triples (W_AHEAD), the odd phase's first A pair is requested by the last group of the even phase
#define QF_PHASE(KT, H, DO_A, W_FULL1, DO_W, W_AHEAD)                                                                    \
    do {                                                                                                                 \
      wait_vm(issued_prev);                                                                                              \
      __builtin_amdgcn_s_barrier();                                                                                      \
      __builtin_amdgcn_s_setprio(3);                                                                                     \
      asm volatile("" : "+v"(lofs_) : : "memory");                                                                       \
      const unsigned char* sb = lds + ((KT) & 1) * QF_STAGE;                                                             \
      constexpr int G0 = (H) * (QF_TM / 2), G1 = G0 + QF_TM / 2;                                                         \
      constexpr int SPG_ = (QF_AIT + QF_BIT + QF_TM / 2 - 1) / (QF_TM / 2);   /* piece slots per group, even phase (2 at QF_TM 8) */ \
      constexpr int WPG_ = (QF_BIT + QF_TM / 2 - 1) / (QF_TM / 2);             /* W pieces per group, odd phase (1 at QF_TM 8) */    \
      if ((H) == 0) {                                                                                                    \
        ah[0] = *reinterpret_cast<const h8*>(sb + aoff);                                                                 \
        al[0] = *reinterpret_cast<const h8*>(sb + (aoff ^ 64));                                                          \
        if (W_FULL1) {                                                                                                   \
          _Pragma("unroll") for (int j = 0; j < QF_NJ; ++j) {                                                            \
            bh[j] = *reinterpret_cast<const h8*>(sb + boff + j * 2048);                                                  \
            bl[j] = *reinterpret_cast<const h8*>(sb + ((boff + j * 2048) ^ 64));                                         \
          }                                                                                                              \
        }                                                                                                                \
      }                                                                                                                  \
      _Pragma("unroll") for (int g = G0; g < G1; ++g) {                                                                  \
        if (g + 1 < (((H) == 0) ? QF_TM : G1)) {                                                                         \
          ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + aoff + (g + 1) * 2048);                                    \
          al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + (g + 1) * 2048) ^ 64));                           \
        }                                                                                                                \
        if ((H) == 0 && SPG_ == 2) {                                                                                     \
          _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                                             \
            const int sl = (g - G0) * 2 + pp;                                                                            \
            if (W_FULL1) {                                                                                               \
              if (sl < QF_BIT) QF_PIECE((KT) + 1, QF_AIT + sl);                                                          \
              else if (sl < QF_AIT + QF_BIT) { if (DO_A) QF_PIECE((KT) + 1, sl - QF_BIT); }                              \
            } else if (sl < QF_AIT) { if (DO_A) QF_PIECE((KT) + 1, sl); }                                                \
          }                                                                                                              \
        } else if ((H) == 0) {       /* (three m-tile groups per phase: more slots per group) */                          \
          _Pragma("unroll") for (int pp = 0; pp < SPG_; ++pp) {                                                          \
            const int sl = (g - G0) * SPG_ + pp;                                                                         \
            if (W_FULL1) {                                                                                               \
              if (sl < QF_BIT) QF_PIECE((KT) + 1, QF_AIT + sl);                                                          \
              else if (sl < QF_AIT + QF_BIT) { if (DO_A) QF_PIECE((KT) + 1, sl - QF_BIT); }                              \
            } else if (sl < QF_AIT) { if (DO_A) QF_PIECE((KT) + 1, sl); }                                                \
          }                                                                                                              \
        } else if (WPG_ == 1) {                                                                                          \
          if (g - G0 < QF_BIT) {                                                                                         \
            if (DO_W) QF_PIECE((KT) + 2, QF_AIT + (g - G0));                                                             \
          }                                                                                                              \
        } else {                                                                                                         \
          _Pragma("unroll") for (int pp = 0; pp < WPG_; ++pp) {                                                          \
            const int wp = (g - G0) * WPG_ + pp;                                                                         \
            if (wp < QF_BIT) { if (DO_W) QF_PIECE((KT) + 2, QF_AIT + wp); }                                              \
          }                                                                                                              \
        }                                                                                                                \
        const bool w_ahead_ = (H) == 1 && g == G1 - 1 && (W_AHEAD);                                                      \
        _Pragma("unroll") for (int j = 0; j < QF_NJ; ++j) {                                                              \
          QF_MMA(acc[g][j], bh[j], bl[j], ah[g & 1], al[g & 1]);                                                         \
          if (w_ahead_) {                                                                                                \
            const unsigned char* sbn = lds + (((KT) + 1) & 1) * QF_STAGE;                                                \
            bh[j] = *reinterpret_cast<const h8*>(sbn + boff + j * 2048);                                                 \
            bl[j] = *reinterpret_cast<const h8*>(sbn + ((boff + j * 2048) ^ 64));                                        \
          }                                                                                                              \
        }                                                                                                                \
        if (w_ahead_) {              /* MFMA triple, its W pair's successor, ...; the piece in between */                 \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
        } else if ((H) == 0 && g == G0 && (W_FULL1)) {   /* a tile's opening: fragments just ahead of their MFMAs */       \
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                             \
        } else {                     /* 2 MFMAs, a read, 2 MFMAs, a read, 2 MFMAs, a piece, 1 MFMA, the other piece, 2 MFMAs */ \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (g - G0 == 0) __builtin_amdgcn_s_setprio(2);                                                                  \
        else if (g - G0 == 1) __builtin_amdgcn_s_setprio(1);                                                             \
        else __builtin_amdgcn_s_setprio(0);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
      }                                                                                                                  \
      if ((H) == 0) issued_prev = (DO_A) ? QF_AIT : 0;                                                                   \
      else issued_prev = (DO_W) ? QF_BIT : 0;                                                                            \
    } while (0)

#define QF_KLOOP_HEAD                                                                                                    \
    QF_PHASE(0, 0, true, true, false, false);                                                                            \
    QF_PHASE(0, 1, false, false, nk > 2 || has_next, nk > 1);                                                            \
    int kt = 1;                                                                                                          \
    for (; kt + 2 < nk; ++kt) {                                                                                          \
      QF_PHASE(kt, 0, true, false, false, false);                                                                        \
      QF_PHASE(kt, 1, false, false, true, true);                                                                         \
    }                                                                                                                    \
    if (nk > 2) {   /* k-tile nk - 2: A(nk - 1) of this tile, then W(0) of the next tile */                              \
      QF_PHASE(kt, 0, true, false, false, false);                                                                        \
      QF_PHASE(kt, 1, false, false, has_next, true);                                                                     \
      ++kt;                                                                                                              \
    }
/* k-tile nk - 1: A(0) of the next tile */
#define QF_KLOOP_TAIL                                                                                                    \
    QF_PHASE(kt, 0, has_next, false, false, false);                                                                      \
    QF_PHASE(kt, 1, false, false, false, false);
