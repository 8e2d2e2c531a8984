// Device-side training schedule: learning-rate decay, per-optimiser Adam bias correction, the
// EMA of t_balance and the tf.cond D-gate -- all on the GPU so that a captured hipGraph of the whole
// training step replays with no host round trip.
//
// Reference: lib/Teco.py:95-99 (exponential_decay, global_step), :415-417 (EMA 0.99 of t_balance),
// :425,439-440 (three AdamOptimizers), :493-494 (tf.cond(tb < Dbalance)).  [TF1] SURVEY A.11.
//
// state (float64[8 + 2*NOPT]):  [0] global_step  [1] tb (EMA shadow)  [2] lr0  [3] decay_steps
//                               [4] decay_rate   [5] staircase        [6] Dbalance  [7] d_lr_factor
//                               [8+2k] t_k (Adam step count of optimiser k)  [9+2k] unused
// hyper (float32[NOPT][8]):     {lr_t, beta1, beta2, eps, gate, lr, 0, 0} consumed by tg_adam_tf.
// Optimiser 0 is the discriminator (gated), the others always step.  t_balance may be NULL (FRVSR).
#include "common.h"

__global__ void schedule_kernel(double* __restrict__ state, float* __restrict__ hyper, int nopt, int gated_opt,
                                const float* __restrict__ t_balance, float beta1, float beta2, float eps) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double step = state[0];
  double pw = step / state[3];
  if (state[5] != 0.0) pw = floor(pw);
  const double lr = state[2] * pow(state[4], pw);
  int gate_open = 1;
  if (t_balance && gated_opt >= 0) {
    gate_open = state[1] < state[6];                           // gate on the OLD average ...
    state[1] = state[1] - (1.0 - 0.99) * (state[1] - (double)t_balance[0]);   // ... then update it
  }
  for (int k = 0; k < nopt; ++k) {
    const int on = (k == gated_opt) ? gate_open : 1;
    double t = state[8 + 2 * k];
    if (on) t += 1.0;
    state[8 + 2 * k] = t;
    const double lrk = (k == gated_opt) ? lr * state[7] : lr;
    const double lr_t = lrk * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    float* h = hyper + 8 * k;
    h[0] = (float)lr_t;
    h[1] = beta1;
    h[2] = beta2;
    h[3] = eps;
    h[4] = on ? 1.f : 0.f;
    h[5] = (float)lrk;
  }
  state[0] = step + 1.0;
}

extern "C" int tg_schedule_step(double* state, float* hyper, int nopt, int gated_opt, const float* t_balance,
                                float beta1, float beta2, float eps, void* stream) {
  TG_CHECK_ARG(state && hyper && nopt > 0 && nopt <= 8, "bad argument");
  hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state, hyper, nopt,
                     gated_opt, t_balance, beta1, beta2, eps);
  TG_CHECK_LAUNCH();
}

// Fade-in factor of the adversarial / layer losses (lib/Teco.py:379-380): dt_ratio = min(max, r0 + add * global_step), from
// the DEVICE-side step counter, so a captured step replays with a changing factor (no launch argument changes).
__global__ void dt_ratio_kernel(const double* __restrict__ state, float r0, float add, float rmax, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double v = (double)r0 + (double)add * state[0];
  out[0] = (float)(v < (double)rmax ? v : (double)rmax);
}

extern "C" int tg_dt_ratio(const double* state, float r0, float add, float rmax, float* out, void* stream) {
  TG_CHECK_ARG(state && out, "bad argument");
  hipLaunchKernelGGL(dt_ratio_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state, r0, add, rmax, out);
  TG_CHECK_LAUNCH();
}
