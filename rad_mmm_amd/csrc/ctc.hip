// CTC loss of the alignment attention (reference loss.py:112-141: torch.nn.CTCLoss with the targets 1, 2, .., L_b -- every
// text position once, in order -- blank 0, zero_infinity) for the whole batch, value and gradient, in TWO launches and
// without a host synchronisation (round 4).  torch's native kernels take 2.3 ms per step at B = 32, T_mel = 800, T_txt = 150
// and copy their length arguments between host and device: six of the step's twelve host synchronisations.
//   alpha_t(s) = lse(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if s odd]) + lp[t][l'_s]      (targets all distinct)
//   nll        = -lse(alpha_{T-1}(S-1), alpha_{T-1}(S-2))
//   beta_t(s)  = lse(beta_{t+1}(s), beta_{t+1}(s+1), [beta_{t+1}(s+2) if s odd]) + lp[t][l'_s]
//   d nll / d lp[t][c] = exp(lp[t][c]) - exp(lse_{s: l'_s = c}(alpha_t(s) + beta_t(s)) + nll - lp[t][c])   for t < T_b, 0 beyond
// -- the formula of torch's ctc_loss_backward (the gradient it defines for log-softmax outputs), so that the module is a
// drop-in for the reference's loop; a non-blank class sits at exactly one state (s = 2 c - 1), the blank at the L + 1 even
// ones.  An utterance without a valid alignment (T_b < L_b) has nll = inf: loss and gradient 0 (zero_infinity).
//
// The two recursions are chains of T_b dependent steps and independent of each other: launch 1 runs them side by side, one
// workgroup per (utterance, direction) with ONE STATE PER THREAD (S = 2 L + 1 <= 1024).  A step is one log-sum-exp of three,
// the neighbours' values come through a double-buffered LDS row (one barrier per step), the thread's own emissions are
// fetched sixteen frames ahead (it only ever needs lp[t][l'_s]), and the rows of alpha / beta go to scratch.  (The first
// version kept an utterance in ONE wave with five states per lane: 5.6 ms -- a lone wave issues one vector instruction
// per ~5 cycles, and the step was ~1000 of them.)  Launch 2 is elementwise: a wave per frame turns the two rows into the
// gradient row (one wave reduction for the blank).
#include "common.h"

namespace {

constexpr int CTC_TCH = 16;          // frames of emissions a thread fetches ahead

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}
// the chain's version on the hardware's exp2 / log2 (v_exp_f32 / v_log_f32, 1 ulp): the arguments are <= 0, the sum lies in
// [1, 3], so the absolute error per step is ~1e-7 like the library functions' -- at 15 instead of ~60 instructions on the
// critical path of a T-step recurrence (0.29 -> 0.2 ms at T = 800).  A term below 2^-126 flushes to 0 as in expf.
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  constexpr float L2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
  const float s = __builtin_amdgcn_exp2f((a - m) * L2E) + __builtin_amdgcn_exp2f((b - m) * L2E) + __builtin_amdgcn_exp2f((c - m) * L2E);
  return fmaf(__builtin_amdgcn_logf(s), LN2, m);
}

// launch 1.  lp [B][T][C] log-probabilities; lens_txt[b] = L_b targets (1 .. L_b), lens_mel[b] = T_b frames;
// tab [2][B][T][SP]: alpha rows (direction 0) and beta rows (direction 1), SP = blockDim.x = S rounded up to 64
__global__ __launch_bounds__(1024) void ctc_chain_kernel(const float* __restrict__ lp, const int* __restrict__ lens_txt,
                                                         const int* __restrict__ lens_mel, float* __restrict__ tab, int B, int T,
                                                         int C) {
  __shared__ float row[2][1024 + 4];                                    // [step parity][2 + state]: two -inf guards on either side
  const int b = blockIdx.x, dir = blockIdx.y, s = threadIdx.x, SP = blockDim.x;
  const int L = min(lens_txt[b], C - 1), Tb = min(lens_mel[b], T);
  const int S = 2 * L + 1;
  if (Tb < L || Tb <= 0) return;                                        // no alignment: launch 2 never reads the tables
  const bool live = s < S, odd = s & 1;
  const float* mine = lp + (long long)b * T * C + (odd ? (s + 1) >> 1 : 0);          // this state's class column
  float* out = tab + ((long long)(dir * B + b) * T) * SP + s;
  if (s < 2) row[0][dir ? 2 + SP + s : s] = row[1][dir ? 2 + SP + s : s] = -INFINITY;
  float cur = -INFINITY;
  float e[CTC_TCH];
  if (dir == 0) {
    for (int t0 = 0; t0 < Tb; t0 += CTC_TCH) {
#pragma unroll
      for (int i = 0; i < CTC_TCH; ++i) e[i] = (live && t0 + i < Tb) ? mine[(long long)(t0 + i) * C] : 0.f;
#pragma unroll
      for (int i = 0; i < CTC_TCH; ++i) {
        const int t = t0 + i;
        if (t < Tb) {                                                   // (uniform)
          float v;
          if (t == 0) {
            v = (s <= 1 && live) ? e[i] : -INFINITY;
          } else {
            const float* r = row[(t - 1) & 1] + 2 + s;
            const float p1 = r[-1], p2 = odd ? r[-2] : -INFINITY;
            v = live ? lse3_fast(cur, p1, p2) + e[i] : -INFINITY;
          }
          cur = v;
          row[t & 1][2 + s] = v;
          out[(long long)t * SP] = v;
          __syncthreads();
        }
      }
    }
  } else {
    for (int t1 = Tb; t1 > 0; t1 -= CTC_TCH) {
#pragma unroll
      for (int i = 0; i < CTC_TCH; ++i) e[i] = (live && t1 - 1 - i >= 0) ? mine[(long long)(t1 - 1 - i) * C] : 0.f;
#pragma unroll
      for (int i = 0; i < CTC_TCH; ++i) {
        const int t = t1 - 1 - i;
        if (t >= 0) {
          float v;
          if (t == Tb - 1) {
            v = (live && s >= S - 2) ? e[i] : -INFINITY;
          } else {
            const float* r = row[(t + 1) & 1] + 2 + s;
            const float n1 = s + 1 < S ? r[1] : -INFINITY, n2 = (odd && s + 2 < S) ? r[2] : -INFINITY;
            v = live ? lse3_fast(cur, n1, n2) + e[i] : -INFINITY;
          }
          cur = v;
          row[t & 1][2 + s] = v;
          out[(long long)t * SP] = v;
          __syncthreads();
        }
      }
    }
  }
}

// launch 2: nll [B] and grad [B][T][C] = d nll_b / d lp (to be multiplied by the upstream gradient of nll_b); one wave per
// frame, 4 frames per workgroup
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const int* __restrict__ lens_txt,
                                                       const int* __restrict__ lens_mel, const float* __restrict__ tab,
                                                       float* __restrict__ nll, float* __restrict__ grad, int B, int T, int C,
                                                       int SP) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int L = min(lens_txt[b], C - 1), Tb = min(lens_mel[b], T);
  const int S = 2 * L + 1;
  const float* al = tab + (long long)b * T * SP;
  const float* be = tab + (long long)(B + b) * T * SP;
  float loss = INFINITY;
  if (Tb >= L && Tb > 0) {
    const float* lastrow = al + (long long)(Tb - 1) * SP;
    loss = -lse2(lastrow[S - 1], S >= 2 ? lastrow[S - 2] : -INFINITY);
  }
  const bool inf = !(loss < INFINITY);                                  // no valid alignment: zero_infinity
  if (t == 0 && lane == 0) nll[b] = inf ? 0.f : loss;
  float* g = grad + ((long long)b * T + t) * C;
  if (inf || t >= Tb) {
    for (int c = lane; c < C; c += 64) g[c] = 0.f;
    return;
  }
  const float* l = lp + ((long long)b * T + t) * C;
  const float* a = al + (long long)t * SP;
  const float* bt = be + (long long)t * SP;
  for (int c = 1 + lane; c < C; c += 64) {
    const float lc = l[c];
    g[c] = c <= L ? expf(lc) - expf(a[2 * c - 1] + bt[2 * c - 1] + loss - lc) : expf(lc);       // classes that are no target
  }
  float mx = -INFINITY;
  for (int i = lane; i <= L; i += 64) mx = fmaxf(mx, a[2 * i] + bt[2 * i]);
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  if (mx > -INFINITY)
    for (int i = lane; i <= L; i += 64) sum += expf(a[2 * i] + bt[2 * i] - mx);
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) {
    const float l0 = l[0];
    const float lab = mx == -INFINITY ? -INFINITY : mx + logf(sum);
    g[0] = expf(l0) - expf(lab + loss - l0);
  }
}

}  // namespace

static inline int ctc_sp(int C) { return (2 * (C - 1) + 1 + 63) / 64 * 64; }

// floats of scratch (alpha and beta rows) radmmm_ctc_monotonic needs
extern "C" int64_t radmmm_ctc_monotonic_scratch_floats(int B, int T, int C) {
  if (B <= 0 || T <= 0 || C <= 1) return 0;
  return (int64_t)2 * B * T * ctc_sp(C);
}

extern "C" int radmmm_ctc_monotonic(const float* lp, const int32_t* lens_txt, const int32_t* lens_mel, float* nll, float* grad,
                                    float* scratch, int B, int T, int C, radmmm_stream_t stream) {
  RADMMM_REQUIRE(lp && lens_txt && lens_mel && nll && grad && scratch, "ctc_monotonic: null pointer");
  RADMMM_REQUIRE(B > 0 && T > 0 && C >= 2 && 2 * (C - 1) + 1 <= 1024, "ctc_monotonic: bad dims (at most 511 text positions)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int SP = ctc_sp(C);
  hipLaunchKernelGGL(ctc_chain_kernel, dim3(B, 2), dim3(SP), 0, st, lp, lens_txt, lens_mel, scratch, B, T, C);
  hipLaunchKernelGGL(ctc_grad_kernel, dim3((T + 3) / 4, B), dim3(256), 0, st, lp, lens_txt, lens_mel, scratch, nll, grad, B, T, C, SP);
  return radmmm::check_launch("ctc_monotonic");
}
