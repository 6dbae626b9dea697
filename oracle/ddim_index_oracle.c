/*
 * CPU ORACLE (plain C) -- TEST INFRASTRUCTURE ONLY; never linked into libd3d_hip.so.
 *
 * Integer / scalar part of the reference's DDIM loop, restated in C:
 *   oracle_ddim_times   : torch.linspace(-1, N-1, S+1).int() reversed            (DIFF:270-272)
 *   oracle_ddim_update  : one element of the DDIM update with the `alpha * x_start` term, fp32 op by op
 *                         exactly in the order the reference's tensor expression evaluates   (DIFF:287-297)
 *   oracle_extract      : a.gather(-1, t) table lookup used by q_sample / p_losses           (DIFF:21-24)
 * DIFF = /root/reference/common/conditional_diffusion_ddim_normal_directPredict_variableLoss_both_crossFrames.py
 *
 * Parity pin: PINNED -- tests/test_oracle_c.py checks it against tests/golden/ddim_times_N1000.npz (every S in
 * [1,1000], captured from the real reference) and against the per-step trajectories in ddim_small_T81_S5.npz.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see __graft_entry__.build()).
 */
#include <math.h>
#include <stdint.h>

/* ATen's CPU linspace for float: step = (end - start) / (steps - 1) in fp32; element i is start + step*i for
 * i < steps/2 and end - step*(steps-1-i) otherwise (two-sided, so both ends are exact). */
int oracle_ddim_times(int32_t num_timesteps, int32_t sampling_timesteps, int32_t* out) {
  if (num_timesteps < 1 || sampling_timesteps < 1) return -1;
  const int steps = sampling_timesteps + 1;
  const float start = -1.0f, end = (float)(num_timesteps - 1);
  volatile float step = (end - start) / (float)(steps - 1);
  for (int i = 0; i < steps; ++i) {
    volatile float p, v;
    if (i < steps / 2) {
      p = step * (float)i;
      v = start + p;
    } else {
      p = step * (float)(steps - 1 - i);
      v = end - p;
    }
    out[steps - 1 - i] = (int32_t)v; /* .int(): truncation toward zero */
  }
  return 0;
}

/* y_next = x0*sqrt(a_next) + c*((y - a*x0)/sqrt(1-a)[t]) + sigma*noise, every operation rounded to fp32. */
float oracle_ddim_update(float x0, float y, float noise, float alpha, float alpha_next, float somac, float eta) {
  volatile float r = alpha / alpha_next;
  volatile float u = 1.0f - r;
  volatile float w = 1.0f - alpha_next;
  volatile float uw = u * w;
  volatile float d = 1.0f - alpha;
  volatile float q = uw / d;
  volatile float sq = sqrtf(q);
  volatile float sigma = eta * sq;
  volatile float s2 = sigma * sigma;
  volatile float cm = w - s2;
  volatile float c = sqrtf(cm);
  volatile float t1 = x0 * sqrtf(alpha_next);
  volatile float ax = alpha * x0;
  volatile float df = y - ax;
  volatile float t4 = df / somac;
  volatile float t5 = c * t4;
  volatile float t6 = t1 + t5;
  volatile float t7 = sigma * noise;
  return t6 + t7;
}

float oracle_extract(const float* table, int32_t t) { return table[t]; }
