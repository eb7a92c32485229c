// Bidirectional single-layer LSTM (the decoder's context LSTM, reference models/radmmm.py:141-146 /
// torch.nn.LSTM semantics incl. packed variable-length batches), recurrent part.
//
// The time loop is T' strictly sequential steps of a tiny GEMM ([B x H] x [H x 4H], B = 32,
// H = 524) plus gate arithmetic.  MIOpen runs it as 4 launches per step and direction
// (profiles/r01_h3b: 28.6 ms of a 129 ms training step).  Here one launch per step serves BOTH
// directions and fuses GEMM + gates + state update:
//
//  forward step s  (dir 0: t = s, dir 1: t = T-1-s), workgroup = (8 hidden units, direction):
//     a[b][g*8+j] = sum_k h_prev[b][k] * W_hh[g*H + u0 + j][k]        32 x 32 x H split-f16 MFMA, K
//                                                                      split over the 4 waves
//     gates = act(a + Gx[b][t])  (Gx = x W_ih^T + b_ih + b_hh, precomputed by one big GEMM)
//     c = f c_prev + i g,  h = o tanh(c);  frames t >= len[b] give h = c = 0 (packed-sequence
//     semantics: the reverse direction starts at each utterance's own last frame)
//     h is written to y (fp32) and, split hi/lo fp16, to the ping-pong operand buffer of step s+1;
//     the gate activations overwrite Gx in place (saved for backward).
//
//  backward step s (dir 0: t = T-1-s, dir 1: t = s), same workgroup decomposition:
//     dh = dy[b][t] + sum over all slices jj of P_prev[jj][b][u]       (recurrent gradient, see below)
//     gate gradients dG (overwrite the saved gates in place: one [B*T][8H] buffer ends up holding
//     the pre-activation gradients that the batched weight/input-gradient GEMMs consume)
//     P[j][b][u'] = sum_{k' < 32} dG[b][k'] * W_hh[row(k')][u']        this slice's 32 gate rows only
//  i.e. the contraction over the 4H gate rows is split across the workgroups that own them and the
//  partial products are summed by the consumer one step later: every workgroup reads one slice
//  of W_hh (67 KB) + one [B x 8] column block of all partials (67 KB) per step instead of the
//  whole dG row block and a 16-column slab of W_hh^T (400 KB).
//
// All recurrent products use the split-f16 scheme of rowgemm_h3.hip (hi/lo fp16 operands, three
// MFMA products, fp32 accumulate): max rel. error ~2e-6.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int UPW = 8;          // hidden units per workgroup (x 4 gates = 32 gate rows = one MFMA tile)
constexpr int MAXKB = 12;       // k blocks (16 wide) per wave in the forward GEMM: H <= 4*12*16 = 768
constexpr int MAXNS = 128;      // slices whose partials one thread sums: H <= 8 * 128 (the other limits bind first)
constexpr int MAXT = 6;         // 32-column output tiles per wave in the backward GEMM: H <= 4*6*32 = 768

struct LstmArgs {
  // forward
  float* G;                     // [B*T][8H]: in  x W_ih^T + b  /  out gate activations (i, f, g, o per direction)
  const _Float16 *Wh, *Wl;      // [2][NS][ldk/16][64][8] split W_hh slices, fragment-major (zero padded)
  _Float16 *hs_h, *hs_l;        // [2 dirs][2 ping-pong][Bp/32][ldk/16][64][8] split h operand, fragment-major
  float* y;                     // [B*T][2H]
  float* c;                     // [B*T][2H]
  const int* lens;              // [B] or null
  int B, T, H, ldk, Bp;         // Bp = B rounded up to 32
  // backward
  const float* dy;              // [B*T][2H]
  const _Float16 *Wth, *Wtl;    // [2][NS][Hp/32][2][64][8] transposed slices, fragment-major (k' = gate*8 + unit-in-slice)
  float* P;                     // [2 dirs][2 ping-pong][Bp/32][Hp/8 consumer slices][NS producer slices][32][8] partial recurrent gradients
  float* dcbuf;                 // [2][Bp][H] carried cell gradient
  int Hp, NS;                   // Hp = H rounded up to 32, NS = ceil(H / 8)
  const float* gscale;          // device scalar: power-of-two scale applied to dG before the fp16 split
};

__device__ __forceinline__ float sigmoid_f(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
  const float e = __expf(-2.f * fabsf(x));
  const float t = __fdividef(1.f - e, 1.f + e);
  return x < 0.f ? -t : t;
}
__device__ __forceinline__ void split_h(float v, _Float16& h, _Float16& l) {
  v = fminf(fmaxf(v, -60000.f), 60000.f);
  h = (_Float16)v;
  l = (_Float16)(v - (float)h);
}

// ---- weight preparation ---------------------------------------------------------------------------
// Operands are stored FRAGMENT-MAJOR: the 64 lanes of a wave read 64 consecutive 16-byte pieces
// (one fully coalesced 1 KiB request) instead of 32 rows x 32 bytes scattered at the row pitch.
//   fragment (kb, lane): row = lane & 31, k = 16 kb + 8 (lane >> 5) .. + 7
// W [2][4H][H] fp32 -> Wp [2][NS][nkb][64][8]: slice j, gate row n = g*8 + ju <-> W row g*H + 8j + ju
__global__ void lstm_pack_w_kernel(const float* __restrict__ W, _Float16* __restrict__ Wh, _Float16* __restrict__ Wl,
                                   int H, int NS, int nkb) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = 2LL * NS * nkb * 512;
  if (idx >= total) return;
  const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  long long rest = idx >> 9;
  const int kb = (int)(rest % nkb); rest /= nkb;
  const int j = (int)(rest % NS);
  const int d = (int)(rest / NS);
  const int n = lane & 31, k = 16 * kb + 8 * (lane >> 5) + e;
  const int g = n >> 3, unit = 8 * j + (n & 7);
  float v = 0.f;
  if (unit < H && k < H) v = W[((long long)d * 4 * H + (long long)g * H + unit) * H + k];
  _Float16 h, l;
  split_h(v, h, l);
  Wh[idx] = h;
  Wl[idx] = l;
}
// W [2][4H][H] -> Wt [2][NS][ntile][2 kb][64][8]: output unit u' = 32 tile + (lane & 31),
// k' = 16 kb + 8 (lane >> 5) + e = g*8 + ju <-> W row g*H + 8j + ju, column u'
__global__ void lstm_pack_wt_kernel(const float* __restrict__ W, _Float16* __restrict__ Wth, _Float16* __restrict__ Wtl,
                                    int H, int Hp, int NS) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = 2LL * NS * Hp * 32;
  if (idx >= total) return;
  const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), kb = (int)((idx >> 9) & 1);
  long long rest = idx >> 10;
  const int ntile = Hp >> 5;
  const int tile = (int)(rest % ntile); rest /= ntile;
  const int j = (int)(rest % NS);
  const int d = (int)(rest / NS);
  const int u = 32 * tile + (lane & 31), kq = 16 * kb + 8 * (lane >> 5) + e;
  const int g = kq >> 3, unit = 8 * j + (kq & 7);
  float v = 0.f;
  if (unit < H && u < H) v = W[((long long)d * 4 * H + (long long)g * H + unit) * H + u];
  _Float16 h, l;
  split_h(v, h, l);
  Wth[idx] = h;
  Wtl[idx] = l;
}

// ---- grid barrier of the persistent variants ---------------------------------------------------------
// The workgroups of one (direction, batch block) group advance through the time steps together: after
// step s every workgroup has added 1 to the group's counter, step s+1 starts when it reads NS*(s+1).
// L2 is per XCD and not coherent across XCDs.  Writing back / invalidating it around the barrier
// (agent-scope fences: buffer_wbl2 / buffer_inv) works but costs more than a kernel boundary
// (measured: 87-95 vs 78-82 ms per training step), so the EXCHANGED operands (h, partial gradients) are
// stored and loaded with the sc1 bit instead -- agent-coherent accesses that write through / read around
// the non-coherent cache levels, like the counter itself -- and everything else stays cached.
// All workgroups of the grid must be resident at once: the host only takes this path when the grid has
// at most one workgroup per CU.  A barrier that does not complete within ~seconds traps (a loud
// launch failure) instead of hanging the GPU.
constexpr int AUX_SC1 = 16;      // cache-policy bit of the raw buffer builtins: sc1 (agent scope) on gfx94x/gfx950

__device__ __forceinline__ void grid_arrive(unsigned* bar) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's sc1 stores are performed
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_wait(unsigned* bar, unsigned target) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) __builtin_trap();
    }
  }
  __syncthreads();
}
__device__ __forceinline__ f16x8 load_frag(__amdgpu_buffer_rsrc_t r, int voff, int soff, bool sc1) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = sc1 ? __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX_SC1)
                      : __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return __builtin_bit_cast(f16x8, v);
}

// ---- forward ------------------------------------------------------------------------------------------
// PERSIST = false: one launch per step (s0 = the step).  PERSIST = true: one launch runs all T steps,
// W_hh stays in registers and the cell state in a register of the thread that owns (batch row, unit).
template <bool PERSIST>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const LstmArgs a, const int s0, unsigned* __restrict__ bar) {
  __shared__ float red[4][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x, d = blockIdx.y, bb = blockIdx.z * 32;
  const int H = a.H, ldk = a.ldk;
  const int nkb = ldk >> 4, kpw = (nkb + 3) >> 2;
  const int pbl = tid >> 3, pju = tid & 7;
  const int pb = bb + pbl, pu = UPW * j + pju;
  const int pbc = pb < a.B ? pb : a.B - 1, puc = pu < H ? pu : H - 1;
  const int len = a.lens ? a.lens[pbc] : a.T;
  unsigned* gbar = PERSIST ? bar + (d * gridDim.z + blockIdx.z) : nullptr;
  // W slice [dir][slice][kb][64][8]: each wave takes a contiguous range of k blocks; every load is one
  // coalesced 1 KiB request
  const _Float16* Wh = a.Wh + (((long long)d * a.NS + j) * nkb) * 512 + lane * 8;
  const _Float16* Wl = a.Wl + (((long long)d * a.NS + j) * nkb) * 512 + lane * 8;
  f16x8 bh[MAXKB], bl[MAXKB];
  const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const int hs_bytes = 2 * 2 * a.Bp * ldk * 2;
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hs_h, 0, hs_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(a.hs_l, 0, hs_bytes, 0x00020000);
  float c_carry = 0.f;
  const int s_end = PERSIST ? a.T : s0 + 1;
  for (int s = s0; s < s_end; ++s) {
    const int t = d == 0 ? s : a.T - 1 - s;
    // Every launch starts with cold caches (L2 is not coherent across XCDs and is invalidated at kernel
    // boundaries), so a step costs about one memory round trip per DEPENDENT load.  All loads of the
    // gate stage are therefore issued here, before the GEMM operands, with clamped (always valid)
    // addresses; the masks are applied to the values afterwards.
    const long long prow = (long long)pbc * a.T + t;
    float* Gp = a.G + prow * 8 * H + (long long)d * 4 * H + puc;
    const float gx0 = Gp[0], gx1 = Gp[H], gx2 = Gp[2 * H], gx3 = Gp[3 * H];
    const int tp = d == 0 ? t - 1 : t + 1;                       // time index of the previous step
    const int tpc = tp < 0 ? 0 : (tp >= a.T ? a.T - 1 : tp);
    float c_prev_ld = c_carry;
    if (!PERSIST) c_prev_ld = a.c[((long long)pbc * a.T + tpc) * 2 * H + (long long)d * H + puc];
    if (PERSIST && s > 0) grid_wait(gbar, (unsigned)a.NS * (unsigned)s);
    // h operand [dir][ping-pong][batch block][kb][64][8] (byte offsets; out-of-range fragments read as zeros)
    const int hbase = ((d * 2 + (s & 1)) * (a.Bp >> 5) + (int)blockIdx.z) * nkb * 1024;
    f16x8 ah[MAXKB], al[MAXKB];
#pragma unroll
    for (int i = 0; i < MAXKB; ++i) {
      const int kb = wave * kpw + i;
      const bool ok = i < kpw && kb < nkb;
      ah[i] = ok ? load_frag(rh, lane * 16, hbase + kb * 1024, PERSIST) : z8;
      al[i] = ok ? load_frag(rl, lane * 16, hbase + kb * 1024, PERSIST) : z8;
      if (!PERSIST || s == 0) {
        bh[i] = ok ? *reinterpret_cast<const f16x8*>(Wh + kb * 512) : z8;
        bl[i] = ok ? *reinterpret_cast<const f16x8*>(Wl + kb * 512) : z8;
      }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < MAXKB; ++i) {
      if (i < kpw && wave * kpw + i < nkb) {                     // wave-uniform
        f32x16& acc = (i & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[i], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)][lane & 31] = acc0[e] + acc1[e];
    __syncthreads();

    // gates and state update: thread -> (batch row, unit of the slice); operand rows >= B stay zero (memset once)
    const int bl_ = pbl, ju = pju, b = pb, u = pu;
    if (u < H && b < a.B) {
      // element (row bl_, k = u) of the next step's operand: fragment (u >> 4, lane = ((u >> 3) & 1) * 32 + bl_), e = u & 7
      const int ho = ((((d * 2 + ((s + 1) & 1)) * (a.Bp >> 5) + (int)blockIdx.z) * nkb + (u >> 4)) * 64 +
                      ((u >> 3) & 1) * 32 + bl_) * 8 + (u & 7);
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pre[g] = red[0][bl_][g * 8 + ju] + red[1][bl_][g * 8 + ju] + red[2][bl_][g * 8 + ju] + red[3][bl_][g * 8 + ju];
      const bool valid = t < len;
      const float c_prev = (tp >= 0 && tp < a.T) ? c_prev_ld : 0.f;
      float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f, hn = 0.f;
      if (valid) {
        ig = sigmoid_f(pre[0] + gx0);
        fg = sigmoid_f(pre[1] + gx1);
        gg = tanh_f(pre[2] + gx2);
        og = sigmoid_f(pre[3] + gx3);
        cn = fg * c_prev + ig * gg;
        hn = og * tanh_f(cn);
      }
      Gp[0] = ig; Gp[H] = fg; Gp[2 * H] = gg; Gp[3 * H] = og;
      a.c[prow * 2 * H + (long long)d * H + u] = cn;
      a.y[prow * 2 * H + (long long)d * H + u] = hn;
      c_carry = cn;
      _Float16 hh, hl;
      split_h(hn, hh, hl);
      if (PERSIST) {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hh), rh, ho * 2, 0, AUX_SC1);
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hl), rl, ho * 2, 0, AUX_SC1);
      } else {
        a.hs_h[ho] = hh;
        a.hs_l[ho] = hl;
      }
    }
    if (PERSIST) grid_arrive(gbar);                              // (its barrier also protects `red` for the next step)
  }
}

// ---- backward -----------------------------------------------------------------------------------------
// PERSIST as in the forward kernel: one launch for all steps, the W_hh^T fragments stay in registers
// and the carried cell gradient in a register of the owning thread (dcbuf unused).
template <bool PERSIST>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const LstmArgs a, const int s0, unsigned* __restrict__ bar) {
  __shared__ __attribute__((aligned(16))) _Float16 sAh[32][40], sAl[32][40];   // [batch][k' (32) + pad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x, d = blockIdx.y, bz = blockIdx.z, bb = bz * 32;
  const int H = a.H, Hp = a.Hp, NS = a.NS, nbz = a.Bp >> 5;
  const int bl_ = tid >> 3, ju = tid & 7;
  const int b = bb + bl_, u = UPW * j + ju;
  unsigned* gbar = PERSIST ? bar + (d * nbz + bz) : nullptr;

  // this wave's W_hh^T fragments (tiles wave, wave + 4, ...) do not depend on anything computed
  // here: fetch them first so their latency hides behind the partial sums and the gate arithmetic
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  const _Float16* Wth = a.Wth + ((long long)d * NS + j) * Hp * 32 + lane * 8;
  const _Float16* Wtl = a.Wtl + ((long long)d * NS + j) * Hp * 32 + lane * 8;
  const int ntile = Hp >> 5;
  f16x8 wbh[MAXT][2], wbl[MAXT][2];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int tile = wave + 4 * i;
    const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const long long off = ((long long)tile * 2 + kb) * 512;
      wbh[i][kb] = tile < ntile ? *reinterpret_cast<const f16x8*>(Wth + off) : z8;
      wbl[i][kb] = tile < ntile ? *reinterpret_cast<const f16x8*>(Wtl + off) : z8;
    }
  }
  const float gsc = a.gscale[0];
  const float inv = 1.f / gsc;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(a.P, 0, 2 * 2 * nbz * NS * 32 * Hp * 4, 0x00020000);
  const int bc = b < a.B ? b : a.B - 1, uc = u < H ? u : H - 1;
  const int len = a.lens ? a.lens[bc] : a.T;
  float dc_carry = 0.f;
  const int s_end = PERSIST ? a.T : s0 + 1;
  for (int s = s0; s < s_end; ++s) {
    const int t = d == 0 ? a.T - 1 - s : s;
    // all loads of the gate stage, unconditionally and with clamped addresses (cold caches at every
    // launch: one round trip per dependent load, see the forward kernel); masks are applied afterwards
    const long long row = (long long)bc * a.T + t;
    float* Gp = a.G + row * 8 * H + (long long)d * 4 * H + uc;
    const float ig = Gp[0], fg = Gp[H], gg = Gp[2 * H], og = Gp[3 * H];
    const float cn = a.c[row * 2 * H + (long long)d * H + uc];
    const int tp = d == 0 ? t - 1 : t + 1;
    const int tpc = tp < 0 ? 0 : (tp >= a.T ? a.T - 1 : tp);
    const float c_prev_ld = a.c[((long long)bc * a.T + tpc) * 2 * H + (long long)d * H + uc];
    const float dy_ld = a.dy[row * 2 * H + (long long)d * H + uc];
    float dc_ld = dc_carry;
    if (!PERSIST) dc_ld = a.dcbuf[((long long)d * a.Bp + bc) * H + uc];
    if (PERSIST && s > 0) grid_wait(gbar, (unsigned)NS * (unsigned)s);
    float pv[MAXNS];
    {
      // consumer-major layout: the [32 x 8] blocks of all producer slices for this slice's 8 units are
      // contiguous (1 KiB each), so every load of the workgroup is one fully used, coalesced request
      const int pbase = ((((d * 2 + ((s + 1) & 1)) * nbz + bz) * (Hp >> 3) + j) * NS) * 1024;
#pragma unroll
      for (int jj = 0; jj < MAXNS; ++jj)
        pv[jj] = jj < NS ? __builtin_bit_cast(float, PERSIST ? __builtin_amdgcn_raw_buffer_load_b32(rP, tid * 4, pbase + jj * 1024, AUX_SC1)
                                                             : __builtin_amdgcn_raw_buffer_load_b32(rP, tid * 4, pbase + jj * 1024, 0))
                         : 0.f;
    }
#pragma unroll
    for (int w = MAXNS / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int jj = 0; jj < w; ++jj) pv[jj] += pv[jj + w];

    float dG[4] = {0.f, 0.f, 0.f, 0.f};
    if (b < a.B && u < H) {
      float dc_prev = 0.f;
      if (t < len) {
        const float dh = dy_ld + (s > 0 ? pv[0] : 0.f);          // step 0: the partial buffer is uninitialised
        float dc = s > 0 ? dc_ld : 0.f;
        const float c_prev = (tp >= 0 && tp < a.T) ? c_prev_ld : 0.f;
        const float tc = tanh_f(cn);
        dc += dh * og * (1.f - tc * tc);
        dG[0] = dc * gg * ig * (1.f - ig);
        dG[1] = dc * c_prev * fg * (1.f - fg);
        dG[2] = dc * ig * (1.f - gg * gg);
        dG[3] = dh * tc * og * (1.f - og);
        dc_prev = dc * fg;
      }
      Gp[0] = dG[0]; Gp[H] = dG[1]; Gp[2 * H] = dG[2]; Gp[3 * H] = dG[3];
      dc_carry = dc_prev;
      if (!PERSIST) a.dcbuf[((long long)d * a.Bp + b) * H + u] = dc_prev;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      _Float16 h, l;
      split_h(dG[g] * gsc, h, l);
      sAh[bl_][g * 8 + ju] = h;
      sAl[bl_][g * 8 + ju] = l;
    }
    __syncthreads();

    // P[j][b][u'] = (1/gscale) * sum_k' A[b][k'] * Wt[d][j][u'][k']
    f16x8 ah[2], al[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      ah[kb] = *reinterpret_cast<const f16x8*>(&sAh[fr][kb * 16 + fk]);
      al[kb] = *reinterpret_cast<const f16x8*>(&sAl[fr][kb * 16 + fk]);
    }
    const int obase = (((d * 2 + (s & 1)) * nbz + bz) * NS * 32 * Hp + j * 256) * 4;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int tile = wave + 4 * i;
      if (tile < ntile) {                                        // wave-uniform
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb], wbh[i][kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], wbl[i][kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], wbh[i][kb], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {                           // P[consumer slice][producer slice j][row][unit & 7]
          const int po = ((tile * 4 + ((lane & 31) >> 3)) * NS * 256 + ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 8 + (lane & 7)) * 4;
          if (PERSIST) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[e] * inv), rP, po, obase, AUX_SC1);
          else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[e] * inv), rP, po, obase, 0);
        }
      }
    }
    if (PERSIST) grid_arrive(gbar);                              // (its barrier also protects sAh / sAl)
  }
}

constexpr int BAR_BYTES = 256;   // one 32-bit counter per (direction, batch block): Bp / 32 <= 32

// RADMMM_LSTM_PERSISTENT=1: one launch for all time steps with a grid barrier between steps instead of
// one launch per step (the default).  Measured equal on MI355X (79.3 vs 79.0 ms per training step: a
// barrier through agent-coherent memory costs the same ~4 dependent memory round trips as a kernel
// boundary), so the simpler launch-per-step path stays the default; the persistent path needs every
// workgroup resident at once and is only taken when the grid has at most one workgroup per CU.
bool use_persistent(const dim3& grid) {
  const char* e = radmmm::debug_env("RADMMM_LSTM_PERSISTENT");
  if (!e || atoi(e) == 0 || grid.z * 2 * 4 > BAR_BYTES) return false;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return false;
  return (long long)grid.x * grid.y * grid.z <= cus;
}

}  // namespace

extern "C" int64_t radmmm_lstm_scratch_bytes(int B, int H, int which) {
  // which 0: split W_hh (hi + lo), 1: h operand ping-pong (hi + lo), 2: packed transposed slices (hi + lo),
  // 3: partial recurrent gradients P, 4: carried cell gradient
  const int64_t ldk = (H + 15) / 16 * 16, Bp = (B + 31) / 32 * 32, Hp = (H + 31) / 32 * 32, NS = (H + UPW - 1) / UPW;
  switch (which) {
    case 0: return 2 * (2 * NS * (ldk / 16) * 512 * 2);
    case 1: return 2 * (2 * 2 * Bp * ldk * 2) + BAR_BYTES;      // + grid-barrier counters of the persistent kernel
    case 2: return 2 * (2 * NS * Hp * 32 * 2);
    case 3: return 2 * 2 * (Bp / 32) * NS * 32 * Hp * 4;
    case 4: return 2 * Bp * (int64_t)H * 4 + BAR_BYTES;
    default: return 0;
  }
}

// Forward recurrence of both directions.  G [B*T][8H] holds x W_ih^T + b_ih + b_hh (direction d in
// columns d*4H .., gate order i, f, g, o) and is overwritten by the gate activations; W_hh [2][4H][H];
// y, c [B*T][2H] outputs; wsplit / hsplit scratch per radmmm_lstm_scratch_bytes(.., 0 / 1).
extern "C" int radmmm_lstm_fwd(float* G, const float* W_hh, float* y, float* c, const int32_t* lens, void* wsplit,
                               void* hsplit, int B, int T, int H, radmmm_stream_t stream) {
  RADMMM_REQUIRE(G && W_hh && y && c && wsplit && hsplit, "lstm_fwd: null pointer");
  RADMMM_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 4 * MAXKB * 16, "lstm_fwd: bad dims (H <= %d)", 4 * MAXKB * 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  LstmArgs a = {};
  a.G = G; a.y = y; a.c = c; a.lens = lens; a.B = B; a.T = T; a.H = H;
  a.ldk = (H + 15) / 16 * 16; a.Bp = (B + 31) / 32 * 32; a.Hp = (H + 31) / 32 * 32; a.NS = (H + UPW - 1) / UPW;
  const long long wn = 2LL * a.NS * (a.ldk / 16) * 512;
  _Float16* Wh = static_cast<_Float16*>(wsplit);
  _Float16* Wl = Wh + wn;
  a.Wh = Wh; a.Wl = Wl;
  const long long hn = 2LL * 2 * a.Bp * a.ldk;
  a.hs_h = static_cast<_Float16*>(hsplit);
  a.hs_l = a.hs_h + hn;
  hipLaunchKernelGGL(lstm_pack_w_kernel, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, st, W_hh, Wh, Wl, H, a.NS, a.ldk / 16);
  if (hipMemsetAsync(hsplit, 0, (size_t)(2 * hn * 2), st) != hipSuccess) {
    radmmm::set_error("lstm_fwd: hipMemsetAsync failed");
    return -2;
  }
  const dim3 grid(a.NS, 2, a.Bp / 32);
  unsigned* bar = reinterpret_cast<unsigned*>(static_cast<char*>(hsplit) + 2 * hn * 2);
  if (use_persistent(grid)) {
    if (hipMemsetAsync(bar, 0, BAR_BYTES, st) != hipSuccess) {
      radmmm::set_error("lstm_fwd: hipMemsetAsync failed");
      return -2;
    }
    hipLaunchKernelGGL(lstm_fwd_kernel<true>, grid, dim3(256), 0, st, a, 0, bar);
  } else {
    for (int s = 0; s < T; ++s) hipLaunchKernelGGL(lstm_fwd_kernel<false>, grid, dim3(256), 0, st, a, s, bar);
  }
  return radmmm::check_launch("lstm_fwd");
}

// Backward recurrence.  G holds the saved gate activations and is overwritten by the pre-activation
// gradients dG [B*T][8H] (the caller forms dW_ih = dG^T x, dx = dG W_ih, db = colsum(dG),
// dW_hh[d] = dG_d^T h_prev with plain GEMMs).  gscale: DEVICE scalar, a power of two that brings dG
// into fp16 range (the caller derives it from max|dy| without a host sync).
extern "C" int radmmm_lstm_bwd(float* G, const float* c, const float* dy, const float* W_hh, const int32_t* lens,
                               void* wtpack, float* P, float* dcbuf, int B, int T, int H, const float* gscale,
                               radmmm_stream_t stream) {
  RADMMM_REQUIRE(G && c && dy && W_hh && wtpack && P && dcbuf && gscale, "lstm_bwd: null pointer");
  RADMMM_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 4 * MAXT * 32, "lstm_bwd: bad dims (H <= %d)", 4 * MAXT * 32);
  RADMMM_REQUIRE(radmmm_lstm_scratch_bytes(B, H, 3) < (1LL << 31), "lstm_bwd: batch too large for one call (partials >= 2 GiB)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  LstmArgs a = {};
  a.G = G; a.c = const_cast<float*>(c); a.dy = dy; a.lens = lens; a.B = B; a.T = T; a.H = H;
  a.ldk = (H + 15) / 16 * 16; a.Bp = (B + 31) / 32 * 32; a.Hp = (H + 31) / 32 * 32; a.NS = (H + UPW - 1) / UPW;
  a.P = P; a.dcbuf = dcbuf; a.gscale = gscale;
  const long long tn = 2LL * a.NS * a.Hp * 32;
  _Float16* Wth = static_cast<_Float16*>(wtpack);
  _Float16* Wtl = Wth + tn;
  a.Wth = Wth; a.Wtl = Wtl;
  hipLaunchKernelGGL(lstm_pack_wt_kernel, dim3((unsigned)((tn + 255) / 256)), dim3(256), 0, st, W_hh, Wth, Wtl, H, a.Hp, a.NS);
  const dim3 grid(a.NS, 2, a.Bp / 32);
  unsigned* bar = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(dcbuf) + 2LL * a.Bp * H * 4);
  if (use_persistent(grid)) {
    if (hipMemsetAsync(bar, 0, BAR_BYTES, st) != hipSuccess) {
      radmmm::set_error("lstm_bwd: hipMemsetAsync failed");
      return -2;
    }
    hipLaunchKernelGGL(lstm_bwd_kernel<true>, grid, dim3(256), 0, st, a, 0, bar);
  } else {
    for (int s = 0; s < T; ++s) hipLaunchKernelGGL(lstm_bwd_kernel<false>, grid, dim3(256), 0, st, a, s, bar);
  }
  return radmmm::check_launch("lstm_bwd");
}
