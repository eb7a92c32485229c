// Alignment attention (ConvAttention.forward, common.py:1262-1277) and monotonic alignment
// search (alignment.py:31-59) for gfx950.
//
// attention: one wavefront per (item, mel frame) row; lanes stride the text axis, the row of
// distances lives in LDS, the two softmax reductions are wavefront shuffles.  The reference
// materialises a [B, Ca, T1, T2] difference tensor (1.2 GB at B=32); here nothing larger than
// the [B, T1, T2] outputs touches HBM.
// MAS: one workgroup per item, lanes over the text axis, sequential over mel frames with the
// previous DP row double-buffered in LDS; back-pointers are bytes in a scratch slab; the
// backtrack is a single lane.  Same fp32 additions in the same order as the reference ->
// bit-exact on identical log inputs.
#include "common.h"

namespace {

using radmmm::wave_max;
using radmmm::wave_sum;

constexpr int AT_WAVES = 4;

__global__ __launch_bounds__(256) void attn_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ prior,
    const int* __restrict__ in_lens, float* __restrict__ attn, float* __restrict__ logprob, int B,
    int T1, int T2, int Ca, float temp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* qrow = sm + wave * (Ca + T2);
  float* d = qrow + Ca;
  const long long row = (long long)blockIdx.x * AT_WAVES + wave;
  if (row >= (long long)B * T1) return;
  const int b = (int)(row / T1);
  const float* q = Q + row * Ca;
  for (int c = lane; c < Ca; c += 64) qrow[c] = q[c];
  // single wave owns qrow/d: no block barrier needed, but LDS writes must land before reads
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  float m1 = -INFINITY;
  for (int s = lane; s < T2; s += 64) {
    const float* k = Kx + ((long long)b * T2 + s) * Ca;
    float acc = 0.f;
    for (int c = 0; c < Ca; ++c) {
      const float df = qrow[c] - k[c];
      acc = fmaf(df, df, acc);
    }
    const float v = -temp * acc;
    d[s] = v;
    m1 = fmaxf(m1, v);
  }
  m1 = wave_max(m1);
  float* lp_out = logprob + row * T2;
  float* at_out = attn + row * T2;
  const int len = in_lens ? in_lens[b] : T2;
  float m2 = -INFINITY;
  if (prior) {
    float z1 = 0.f;
    for (int s = lane; s < T2; s += 64) z1 += expf(d[s] - m1);
    z1 = wave_sum(z1);
    const float lse = m1 + logf(z1);
    for (int s = lane; s < T2; s += 64) {
      const float v = (d[s] - lse) + logf(prior[row * T2 + s] + 1e-8f);
      d[s] = v;
      lp_out[s] = v;
      if (s < len) m2 = fmaxf(m2, v);
    }
  } else {
    for (int s = lane; s < T2; s += 64) {
      lp_out[s] = d[s];
      if (s < len) m2 = fmaxf(m2, d[s]);
    }
  }
  m2 = wave_max(m2);
  float z2 = 0.f;
  for (int s = lane; s < T2; s += 64)
    if (s < len) z2 += expf(d[s] - m2);
  z2 = wave_sum(z2);
  for (int s = lane; s < T2; s += 64) at_out[s] = s < len ? expf(d[s] - m2) / z2 : 0.f;
}

// pass 1 of the gradient: per row, g_d (gradient w.r.t. the scaled distances) -> scratch, and gQ
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(
    const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ prior,
    const int* __restrict__ in_lens, const float* __restrict__ attn,
    const float* __restrict__ logprob, const float* __restrict__ gattn,
    const float* __restrict__ glogprob, float* __restrict__ gQ, float* __restrict__ gd, int B,
    int T1, int T2, int Ca, float temp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* qrow = sm + wave * (Ca + T2);
  float* g = qrow + Ca;
  const long long row = (long long)blockIdx.x * AT_WAVES + wave;
  if (row >= (long long)B * T1) return;
  const int b = (int)(row / T1);
  const int len = in_lens ? in_lens[b] : T2;
  const float* q = Q + row * Ca;
  for (int c = lane; c < Ca; c += 64) qrow[c] = q[c];
  // softmax backward: g_lp = glogprob + attn * (gattn - sum(gattn*attn))
  float dot = 0.f;
  if (gattn)
    for (int s = lane; s < T2; s += 64)
      if (s < len) dot = fmaf(gattn[row * T2 + s], attn[row * T2 + s], dot);
  dot = wave_sum(dot);
  float tot = 0.f;
  for (int s = lane; s < T2; s += 64) {
    float v = glogprob ? glogprob[row * T2 + s] : 0.f;
    if (gattn && s < len) v += attn[row * T2 + s] * (gattn[row * T2 + s] - dot);
    g[s] = v;
    tot += v;
  }
  tot = wave_sum(tot);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (prior) {
    // logprob = d - lse(d) + log(prior+1e-8)  ->  softmax(d) = exp(logprob - log(prior+1e-8))
    for (int s = lane; s < T2; s += 64) {
      const float sm_d = expf(logprob[row * T2 + s] - logf(prior[row * T2 + s] + 1e-8f));
      g[s] = g[s] - sm_d * tot;
    }
  }
  for (int s = lane; s < T2; s += 64) gd[row * T2 + s] = g[s];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  // gQ[c] = sum_s g_d[s] * (-2 temp) (q_c - k_sc)
  for (int c = lane; c < Ca; c += 64) {
    float acc = 0.f;
    for (int s = 0; s < T2; ++s) acc = fmaf(g[s], qrow[c] - Kx[((long long)b * T2 + s) * Ca + c], acc);
    gQ[row * Ca + c] = -2.f * temp * acc;
  }
}

// pass 2: gK[b,s,c] = 2 temp * sum_t g_d[b,t,s] (q_tc - k_sc)
__global__ __launch_bounds__(128) void attn_bwd_keys_kernel(
    const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ gd,
    float* __restrict__ gK, int B, int T1, int T2, int Ca, float temp) {
  const long long bs = blockIdx.x;  // b*T2 + s
  const int b = (int)(bs / T2), s = (int)(bs - (long long)b * T2);
  for (int c = threadIdx.x; c < Ca; c += blockDim.x) {
    const float kv = Kx[bs * Ca + c];
    float acc = 0.f;
    for (int t = 0; t < T1; ++t)
      acc = fmaf(gd[((long long)b * T1 + t) * T2 + s], Q[((long long)b * T1 + t) * Ca + c] - kv, acc);
    gK[bs * Ca + c] = 2.f * temp * acc;
  }
}

// the same sums, tiled (round 4): a workgroup owns 4 text positions x all channels of one utterance and walks the frames in
// chunks of 32 through LDS (the kernel above re-read Q once per text position and fetched g_d one broadcast word at a time:
// 0.43 ms at B = 32, 800 x 150, Ca = 80).  Every output keeps the SAME fmaf chain over t in ascending order: identical bits.
constexpr int ABK_S = 4, ABK_T = 32, ABK_MAXJ = 2;       // Ca <= 128; small tiles: several workgroups per CU hide the LDS latency
__global__ __launch_bounds__(256) void attn_bwd_keys_tiled_kernel(
    const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ gd,
    float* __restrict__ gK, int B, int T1, int T2, int Ca, float temp) {
  extern __shared__ float abk_sm[];                      // [ABK_T][ABK_S + 1] of g_d, then [ABK_T][Ca] of Q
  float* sg = abk_sm;
  float* sq = abk_sm + ABK_T * (ABK_S + 1);
  const int b = blockIdx.y, s0 = blockIdx.x * ABK_S, tid = threadIdx.x;
  const int nout = ABK_S * Ca;
  int os[ABK_MAXJ], oc[ABK_MAXJ];
  float kv[ABK_MAXJ], acc[ABK_MAXJ];
#pragma unroll
  for (int j = 0; j < ABK_MAXJ; ++j) {
    const int idx = tid + j * 256;
    const bool on = idx < nout;
    os[j] = on ? idx / Ca : 0;
    oc[j] = on ? idx - os[j] * Ca : 0;
    const bool live = on && s0 + os[j] < T2;
    kv[j] = live ? Kx[((long long)b * T2 + s0 + os[j]) * Ca + oc[j]] : 0.f;
    acc[j] = 0.f;
  }
  for (int t0 = 0; t0 < T1; t0 += ABK_T) {
    for (int i = tid; i < ABK_T * ABK_S; i += 256) {
      const int tt = i / ABK_S, ss = i - tt * ABK_S;
      const bool ok = t0 + tt < T1 && s0 + ss < T2;
      sg[tt * (ABK_S + 1) + ss] = ok ? gd[((long long)b * T1 + t0 + tt) * T2 + s0 + ss] : 0.f;
    }
    const int nq = min(ABK_T, T1 - t0) * Ca;
    const float* qsrc = Q + ((long long)b * T1 + t0) * Ca;
    for (int i = tid; i < ABK_T * Ca; i += 256) sq[i] = i < nq ? qsrc[i] : 0.f;
    __syncthreads();
#pragma unroll 4
    for (int tt = 0; tt < ABK_T; ++tt) {
#pragma unroll
      for (int j = 0; j < ABK_MAXJ; ++j)
        if (j * 256 < nout) acc[j] = fmaf(sg[tt * (ABK_S + 1) + os[j]], sq[tt * Ca + oc[j]] - kv[j], acc[j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < ABK_MAXJ; ++j) {
    const int idx = tid + j * 256;
    if (idx < nout && s0 + os[j] < T2) gK[((long long)b * T2 + s0 + os[j]) * Ca + oc[j]] = 2.f * temp * acc[j];
  }
}

// ------------------------------------------------------------------ MAS
// PROB: the input holds probabilities and the log is taken here, correctly rounded to fp32 (double log, then one
// rounding): a device-independent definition of the reference's `np.log(attn)` (alignment.py:36), whose own float32
// result depends on the host's SIMD log (numpy's vector float32 log is documented to ~4 ulp).
template <bool PROB>
__device__ __forceinline__ float mas_load(const float* p) {
  if constexpr (PROB) return (float)log((double)*p);
  return *p;
}

template <bool PROB>
__global__ __launch_bounds__(256) void mas_width1_kernel(
    const float* __restrict__ logp, const int* __restrict__ in_lens, const int* __restrict__ out_lens,
    float* __restrict__ hard, unsigned char* __restrict__ back, int T1, int T2) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [2][T2]
  const int b = blockIdx.x;
  const int n1 = out_lens[b] < T1 ? out_lens[b] : T1;
  const int n2 = in_lens[b] < T2 ? in_lens[b] : T2;
  const float* lp = logp + (long long)b * T1 * T2;
  float* hd = hard + (long long)b * T1 * T2;
  unsigned char* bk = back + (long long)b * T1 * T2;
  for (long long i = threadIdx.x; i < (long long)T1 * T2; i += blockDim.x) hd[i] = 0.f;
  if (n1 <= 0 || n2 <= 0) return;
  float* prev = sm;
  float* cur = sm + T2;
  for (int j = threadIdx.x; j < n2; j += blockDim.x) {
    prev[j] = j == 0 ? mas_load<PROB>(lp) : -INFINITY;  // first row forced to column 0 (alignment.py:37)
    bk[j] = 0;
  }
  __syncthreads();
  for (int i = 1; i < n1; ++i) {
    for (int j = threadIdx.x; j < n2; j += blockDim.x) {
      float pl = prev[j];
      unsigned char mv = 0;
      if (j >= 1 && prev[j - 1] >= pl) {  // tie -> diagonal (alignment.py:46)
        pl = prev[j - 1];
        mv = 1;
      }
      cur[j] = mas_load<PROB>(lp + (long long)i * T2 + j) + pl;
      bk[(long long)i * T2 + j] = mv;
    }
    __syncthreads();
    float* t = prev;
    prev = cur;
    cur = t;
  }
  // make the back-pointer bytes written by other waves visible to the backtracking lane
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x == 0) {
    int j = n2 - 1;
    for (int i = n1 - 1; i >= 0; --i) {
      hd[(long long)i * T2 + j] = 1.f;
      j -= bk[(long long)i * T2 + j];
    }
    hd[0] = 1.f;  // prev_ind[0, :] is all zeros -> curr_text_idx = 0 -> opt[0, 0] = 1 (alignment.py:58)
  }
}


// ---- the fast path (round 4): T2 <= 1024 and the T1 x T2 back-pointer BITS fit in LDS.
// The round-1 kernel above put a global load, a double-precision log and a global byte store into every one of the T1
// dependent steps, and backtracked through global memory (800 dependent loads): 0.8 ms for 32 utterances of 800 x 150.
// Here a first launch over the whole chip takes the logs (same definition: mas_load) into scratch and zero-fills `hard`;
// the chain launch keeps one text position per thread, fetches its scores sixteen frames ahead, publishes each row through
// a double-buffered LDS row (one barrier per frame), packs the step's moves with one ballot per wave into LDS, and one
// thread walks back through those bits.  Same comparisons on the same values: identical maps.
template <bool PROB>
__global__ __launch_bounds__(256) void mas_prep_kernel(const float* __restrict__ in, float* __restrict__ logs,
                                                       float* __restrict__ hard, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    hard[i] = 0.f;
    if constexpr (PROB) logs[i] = mas_load<true>(in + i);
  }
}

constexpr int MAS_AHEAD = 16;

__global__ __launch_bounds__(1024) void mas_chain_kernel(const float* __restrict__ logp, const int* __restrict__ in_lens,
                                                         const int* __restrict__ out_lens, float* __restrict__ hard, int T1,
                                                         int T2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mas_sm[];
  const int b = blockIdx.x, j = threadIdx.x, nw = blockDim.x >> 6, wave = j >> 6;
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(mas_sm);            // [T1][nw]
  float* rows = reinterpret_cast<float*>(mas_sm + (size_t)T1 * nw * 8);                // [2][1 + blockDim]
  const int n1 = out_lens[b] < T1 ? out_lens[b] : T1;
  const int n2 = in_lens[b] < T2 ? in_lens[b] : T2;
  if (n1 <= 0 || n2 <= 0) return;
  const float* lp = logp + (long long)b * T1 * T2 + j;
  float* hd = hard + (long long)b * T1 * T2;
  const int RW = blockDim.x + 1;
  const bool live = j < n2;
  float cur = (j == 0) ? lp[0] : -INFINITY;      // first row forced to column 0 (alignment.py:37)
  rows[1 + j] = cur;
  if (j == 0) rows[0] = rows[RW] = -INFINITY;
  if ((j & 63) == 0) bits[wave] = 0ull;
  __syncthreads();
  float e[MAS_AHEAD];
  for (int i0 = 1; i0 < n1; i0 += MAS_AHEAD) {
#pragma unroll
    for (int k = 0; k < MAS_AHEAD; ++k) e[k] = (live && i0 + k < n1) ? lp[(long long)(i0 + k) * T2] : 0.f;
#pragma unroll
    for (int k = 0; k < MAS_AHEAD; ++k) {
      const int i = i0 + k;
      if (i < n1) {                                                   // (uniform)
        const float* prev = rows + ((i - 1) & 1) * RW + 1 + j;
        float pl = cur;
        const float pm = prev[-1];
        const bool mv = j >= 1 && pm >= pl;                           // tie -> diagonal (alignment.py:46)
        if (mv) pl = pm;
        cur = e[k] + pl;
        rows[(i & 1) * RW + 1 + j] = cur;
        const unsigned long long m = __ballot(mv && live);
        if ((j & 63) == 0) bits[(size_t)i * nw + wave] = m;
        __syncthreads();
      }
    }
  }
  if (j == 0) {
    int c = n2 - 1;
    for (int i = n1 - 1; i >= 0; --i) {
      hd[(long long)i * T2 + c] = 1.f;
      c -= (int)((bits[(size_t)i * nw + (c >> 6)] >> (c & 63)) & 1ull);
    }
    hd[0] = 1.f;  // prev_ind[0, :] is all zeros -> curr_text_idx = 0 -> opt[0, 0] = 1 (alignment.py:58)
  }
}

template <bool PROB>
int launch_mas(const float* in, const int32_t* in_lens, const int32_t* out_lens, float* hard, void* scratch, int B, int T1, int T2,
               hipStream_t st) {
  const int threads = (T2 + 63) / 64 * 64;
  const size_t fast_smem = (size_t)T1 * (threads / 64) * 8 + (size_t)2 * (threads + 1) * sizeof(float);
  if (T2 <= 1024 && fast_smem <= 160 * 1024) {
    static bool once = false;
    if (!once) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mas_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      once = true;
    }
    const long long n = (long long)B * T1 * T2;
    float* logs = static_cast<float*>(scratch);
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(mas_prep_kernel<PROB>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, in, logs, hard, n);
    hipLaunchKernelGGL(mas_chain_kernel, dim3(B), dim3(threads), fast_smem, st, PROB ? logs : in, in_lens, out_lens, hard, T1, T2);
    return radmmm::check_launch("mas_width1 (chain)");
  }
  const size_t smem = (size_t)2 * T2 * sizeof(float);
  RADMMM_REQUIRE(smem <= 64 * 1024, "mas_width1: T2=%d too long", T2);
  hipLaunchKernelGGL(mas_width1_kernel<PROB>, dim3(B), dim3(256), smem, st, in, in_lens, out_lens, hard,
                     static_cast<unsigned char*>(scratch), T1, T2);
  return radmmm::check_launch("mas_width1");
}

}  // namespace

extern "C" int radmmm_attn_fwd(const float* Q, const float* Kx, const float* prior, const int32_t* in_lens,
                               float* attn, float* logprob, int B, int T1, int T2, int Ca, float temp,
                               radmmm_stream_t stream) {
  RADMMM_REQUIRE(Q && Kx && attn && logprob, "attn_fwd: null pointer");
  RADMMM_REQUIRE(B > 0 && T1 > 0 && T2 > 0 && Ca > 0, "attn_fwd: bad dims");
  const size_t smem = (size_t)AT_WAVES * (Ca + T2) * sizeof(float);
  RADMMM_REQUIRE(smem <= 64 * 1024, "attn_fwd: T2=%d too long for the row buffer", T2);
  const long long rows = (long long)B * T1;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((rows + AT_WAVES - 1) / AT_WAVES)), dim3(256), smem,
                     static_cast<hipStream_t>(stream), Q, Kx, prior, in_lens, attn, logprob, B, T1, T2,
                     Ca, temp);
  return radmmm::check_launch("attn_fwd");
}

extern "C" int radmmm_attn_bwd(const float* Q, const float* Kx, const float* prior, const int32_t* in_lens,
                               const float* attn, const float* logprob, const float* gattn,
                               const float* glogprob, float* gQ, float* gK, float* gd_scratch, int B,
                               int T1, int T2, int Ca, float temp, radmmm_stream_t stream) {
  RADMMM_REQUIRE(Q && Kx && attn && logprob && gQ && gK && gd_scratch, "attn_bwd: null pointer");
  RADMMM_REQUIRE(B > 0 && T1 > 0 && T2 > 0 && Ca > 0, "attn_bwd: bad dims");
  const size_t smem = (size_t)AT_WAVES * (Ca + T2) * sizeof(float);
  RADMMM_REQUIRE(smem <= 64 * 1024, "attn_bwd: T2=%d too long for the row buffer", T2);
  const long long rows = (long long)B * T1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(attn_bwd_rows_kernel, dim3((unsigned)((rows + AT_WAVES - 1) / AT_WAVES)), dim3(256),
                     smem, s, Q, Kx, prior, in_lens, attn, logprob, gattn, glogprob, gQ, gd_scratch, B, T1,
                     T2, Ca, temp);
  if (ABK_S * Ca <= 256 * ABK_MAXJ)       // (ABK_S positions x Ca outputs over 256 threads: at most ABK_MAXJ each)
    hipLaunchKernelGGL(attn_bwd_keys_tiled_kernel, dim3((unsigned)((T2 + ABK_S - 1) / ABK_S), (unsigned)B), dim3(256),
                       (size_t)(ABK_T * (ABK_S + 1) + ABK_T * Ca) * sizeof(float), s, Q, Kx, gd_scratch, gK, B, T1, T2, Ca, temp);
  else
    hipLaunchKernelGGL(attn_bwd_keys_kernel, dim3((unsigned)((long long)B * T2)), dim3(128), 0, s, Q, Kx,
                       gd_scratch, gK, B, T1, T2, Ca, temp);
  return radmmm::check_launch("attn_bwd");
}

extern "C" int64_t radmmm_mas_scratch_bytes(int B, int T1, int T2) { return (int64_t)B * T1 * T2 * 4; }

extern "C" int radmmm_mas_width1(const float* logp, const int32_t* in_lens, const int32_t* out_lens,
                                 float* hard, void* scratch, int B, int T1, int T2,
                                 radmmm_stream_t stream) {
  RADMMM_REQUIRE(logp && in_lens && out_lens && hard && scratch, "mas_width1: null pointer");
  RADMMM_REQUIRE(B > 0 && T1 > 0 && T2 > 0, "mas_width1: bad dims");
  return launch_mas<false>(logp, in_lens, out_lens, hard, scratch, B, T1, T2, static_cast<hipStream_t>(stream));
}

extern "C" int radmmm_mas_width1_prob(const float* attn, const int32_t* in_lens, const int32_t* out_lens,
                                      float* hard, void* scratch, int B, int T1, int T2,
                                      radmmm_stream_t stream) {
  RADMMM_REQUIRE(attn && in_lens && out_lens && hard && scratch, "mas_width1_prob: null pointer");
  RADMMM_REQUIRE(B > 0 && T1 > 0 && T2 > 0, "mas_width1_prob: bad dims");
  return launch_mas<true>(attn, in_lens, out_lens, hard, scratch, B, T1, T2, static_cast<hipStream_t>(stream));
}
