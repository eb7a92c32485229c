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
constexpr int MAXKB = 12;       // k blocks (16 wide) per wave in the forward GEMM: H <= 4*12*16 = 768 (size classes below)
constexpr int MAXNS = 96;       // slices whose partials one thread sums: H <= 8 * 96 = 768 (as MAXKB / MAXT)
constexpr int MAXT = 6;         // 32-column output tiles per wave in the backward GEMM: H <= 4*6*32 = 768

struct LstmArgs {
  // forward
  float* G;                     // [B*T][8H]: in  x W_ih^T + b  /  out gate activations (i, f, g, o per direction)
  const _Float16 *Wh, *Wl;      // [2][NS][ldk/16][64][8] split W_hh slices, fragment-major (zero padded)
  _Float16 *hs_h, *hs_l;        // [2 dirs][2 ping-pong][Bp/32][ldk/16][64][8] split h operand, fragment-major
  _Float16 *hq_h, *hq_l;        // persistent kernel: [T + 1 slots][2 dirs][Bp/32][ldk/16][64][8], slot s = the operand of step s
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
  int probe;                    // persistent kernels: poll ONE word until it is written before loading (and checking) everything
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

// ---- step-to-step exchange of the persistent variants -------------------------------------------------
// One launch runs all T steps; the workgroups of one (direction, batch block) group exchange h (forward) or the partial
// recurrent gradients (backward) through memory.  Round 1 put a grid barrier between the steps (counter + poll, operands
// with sc1): store -> wait for the write-through ack -> atomic add -> poll the counter -> load the operands is ~4 dependent
// trips through the fabric, the same 7 us as a kernel boundary.  Round 3: THE DATA IS ITS OWN FLAG.
//   forward:  every step has its own operand slot, pre-filled with 0xFFFFFFFF words by one memset before the launch (two
//             fp16 NaNs; a stored h is never NaN: split_h clamps through fminf / fmaxf, which drop NaNs).  A consumer
//             wave loads its fragments of slot s with sc1 (agent-coherent: reads around the non-coherent cache levels)
//             and repeats the loads until no word is the fill pattern.  Stores are 4-byte (two units of one batch row),
//             words are written once: a word that is not the fill pattern is final.
//   backward: the partials are fp32; the lowest mantissa bit carries a tag, (step >> 1) & 1, over a two-slot ping-pong
//             (slot step & 1): the previous occupant of a slot, two steps older, has the other tag; the slots start as
//             0xFFFFFFFF (tag 1, steps 0 and 1 write tag 0).  A consumer repeats its loads until every word carries
//             the expected tag, and clears the bit (<= 1 ulp of a partial, 6e-8 relative: far inside the 2e-6 of the
//             split products).  A slot is only rewritten by a producer that has consumed step s + 1 of EVERY workgroup,
//             i.e. after every reader of step s is done with it (data dependence; no fence needed).
// Per step the chain is now: store (write-through) -> visible -> the consumer's outstanding load returns.
// All workgroups of the grid must be resident at once: the host only takes this path when the grid has at most one
// workgroup per CU slot it may use (use_persistent below) and launches it cooperatively (the runtime checks the grid against
// the device; a refused launch falls back to one launch per step).  Workgroups that other kernels (RCCL channels, another
// stream) keep waiting for a CU only DELAY their consumers: a waiting wave polls fast at first, then sleeps between polls,
// and traps -- a loud launch failure instead of a hung GPU -- only after a minute without progress.
constexpr int AUX_SC1 = 16;      // cache-policy bit of the raw buffer builtins: sc1 (agent scope) on gfx94x/gfx950
constexpr unsigned FILL = 0xffffffffu;
constexpr unsigned SPIN_FAST = 512;                       // polls before the wave starts to sleep between them
constexpr unsigned long long SPIN_TIMEOUT = 60ull * 100000000ull;   // wall_clock64 ticks (100 MHz): one minute
// one failed poll: back off, and trap when nothing has arrived for SPIN_TIMEOUT (t0 = first slow poll of this wait)
__device__ __forceinline__ void spin_wait(unsigned& spins, unsigned long long& t0) {
  if (++spins < SPIN_FAST) return;
  __builtin_amdgcn_s_sleep(16);                          // ~1000 cycles: the fabric carries the producers' stores, not polls
  if ((spins & 255u) == 0) {
    const unsigned long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    else if (now - t0 > SPIN_TIMEOUT) __builtin_trap();
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned is_fill(const f16x8& f) {
  const u32x4 v = __builtin_bit_cast(u32x4, f);
  return (unsigned)(v.x == FILL) | (unsigned)(v.y == FILL) | (unsigned)(v.z == FILL) | (unsigned)(v.w == FILL);
}
__device__ __forceinline__ f16x8 load_frag(__amdgpu_buffer_rsrc_t r, int voff, int soff, bool sc1) {
  const u32x4 v = sc1 ? __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX_SC1)
                      : __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return __builtin_bit_cast(f16x8, v);
}

// ---- forward ------------------------------------------------------------------------------------------
// PERSIST = false: one launch per step (s0 = the step).  PERSIST = true: one launch runs all T steps,
// W_hh stays in registers and the cell state in a register of the thread that owns (batch row, unit).
// KPW: k blocks (16 wide) per wave, a compile-time capacity (the host picks the size class: H <= 64 KPW); the loads of
// blocks that do not exist go out of the buffer's range and return zeros -- no branch stands between the loads of a step.
constexpr int OOB = 0x7fffffff;

template <bool PERSIST, int KPW>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const LstmArgs a, const int s0) {
  __shared__ float red[PERSIST ? 2 : 1][4][32][33];             // (two copies: a fast wave may be one step ahead)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x, d = blockIdx.y, bz = blockIdx.z, bb = bz * 32;
  const int H = a.H, ldk = a.ldk;
  const int nkb = ldk >> 4, kpw = (nkb + 3) >> 2;
  const int nbz = a.Bp >> 5;
  const int pbl = tid >> 3, pju = tid & 7;
  const int pb = bb + pbl, pu = UPW * j + pju;
  const int pbc = pb < a.B ? pb : a.B - 1, puc = pu < H ? pu : H - 1;
  const int len = a.lens ? a.lens[pbc] : a.T;
  const bool live = pu < H && pb < a.B;
  // fragment offsets of this wave's k blocks (shared by the W slice and the h operand: both are [kb][64][8])
  int fv[KPW];
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int kb = wave * kpw + i;
    fv[i] = (i < kpw && kb < nkb) ? kb * 1024 + lane * 16 : OOB;
  }
  // W slice [dir][slice][kb][64][8]: every load is one coalesced 1 KiB request
  f16x8 bh[KPW], bl[KPW];
  {
    const long long wo = (((long long)d * a.NS + j) * nkb) * 512;
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Wh + wo), 0, nkb * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Wl + wo), 0, nkb * 1024, 0x00020000);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      bh[i] = load_frag(rwh, fv[i], 0, false);
      bl[i] = load_frag(rwl, fv[i], 0, false);
    }
  }
  const int slot_bytes = 2 * nbz * nkb * 1024;                   // persistent: one operand slot (both directions)
  const int hs_bytes = PERSIST ? (a.T + 1) * slot_bytes : 2 * 2 * a.Bp * ldk * 2;
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(PERSIST ? a.hq_h : a.hs_h, 0, hs_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(PERSIST ? a.hq_l : a.hs_l, 0, hs_bytes, 0x00020000);
  float c_carry = 0.f;
  const int s_end = PERSIST ? a.T : s0 + 1;
  for (int s = s0; s < s_end; ++s) {
    const int t = d == 0 ? s : a.T - 1 - s;
    // All loads of the gate stage are issued here, before the GEMM operands, with clamped (always valid) addresses; the
    // masks are applied to the values afterwards.
    const long long prow = (long long)pbc * a.T + t;
    float* Gp = a.G + prow * 8 * H + (long long)d * 4 * H + puc;
    const float gx0 = Gp[0], gx1 = Gp[H], gx2 = Gp[2 * H], gx3 = Gp[3 * H];
    const int tp = d == 0 ? t - 1 : t + 1;                       // time index of the previous step
    const int tpc = tp < 0 ? 0 : (tp >= a.T ? a.T - 1 : tp);
    float c_prev_ld = c_carry;
    if (!PERSIST) c_prev_ld = a.c[((long long)pbc * a.T + tpc) * 2 * H + (long long)d * H + puc];
    // h operand: [dir][ping-pong][batch block][kb][64][8], or slot s of [slot][dir][batch block][kb][64][8]
    const int hbase = __builtin_amdgcn_readfirstlane(PERSIST ? s * slot_bytes + (d * nbz + bz) * nkb * 1024
                                                             : ((d * 2 + (s & 1)) * nbz + bz) * nkb * 1024);
    f16x8 ah[KPW], al[KPW];
    unsigned spins = 0;
    unsigned long long spin_t0 = 0;
    if (PERSIST && s > 0 && a.probe) {
      // a waiting wave repeats ONE 4-byte load (the first word of its last k block) instead of its 18 KiB of fragments: the
      // full loads, which every word still has to pass, then mostly succeed at once and the fabric carries the producers'
      // stores instead of failed polls
      const int kl = (wave * kpw + kpw - 1) < nkb ? (wave * kpw + kpw - 1) : nkb - 1;
      for (;;) {
        const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rl, kl * 1024, hbase, AUX_SC1);
        asm volatile("" ::: "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)w) != FILL) break;
        spin_wait(spins, spin_t0);
      }
    }
    for (;;) {
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        ah[i] = load_frag(rh, fv[i], hbase, PERSIST);
        al[i] = load_frag(rl, fv[i], hbase, PERSIST);
      }
      if (!PERSIST || s == 0) break;                             // (slot 0 is zero-filled by the host)
      unsigned pending = 0;
#pragma unroll
      for (int i = 0; i < KPW; ++i) pending |= is_fill(ah[i]) | is_fill(al[i]);
      asm volatile("" ::: "memory");                             // the loads are repeated, not hoisted
      if (!__builtin_amdgcn_ballot_w64(pending != 0)) break;     // every word of this wave's fragments has been written
      spin_wait(spins, spin_t0);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      f32x16& acc = (i & 1) ? acc1 : acc0;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[i], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[i], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[i], acc, 0, 0, 0);
    }
    float (*rd)[32][33] = red[PERSIST ? (s & 1) : 0];
#pragma unroll
    for (int e = 0; e < 16; ++e) rd[wave][(e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)][lane & 31] = acc0[e] + acc1[e];
    __syncthreads();

    // gates and state update: thread -> (batch row, unit of the slice)
    const int bl_ = pbl, ju = pju, u = pu;
    float hn = 0.f, sv[5];                                        // (the exchange store goes first, the saved values after it)
    {
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pre[g] = rd[0][bl_][g * 8 + ju] + rd[1][bl_][g * 8 + ju] + rd[2][bl_][g * 8 + ju] + rd[3][bl_][g * 8 + ju];
      const bool valid = live && t < len;
      const float c_prev = (tp >= 0 && tp < a.T) ? c_prev_ld : 0.f;
      float ig = sigmoid_f(pre[0] + gx0);
      float fg = sigmoid_f(pre[1] + gx1);
      float gg = tanh_f(pre[2] + gx2);
      float og = sigmoid_f(pre[3] + gx3);
      float cn = fg * c_prev + ig * gg;
      hn = og * tanh_f(cn);
      if (!valid) { ig = 0.f; fg = 0.f; gg = 0.f; og = 0.f; cn = 0.f; hn = 0.f; }
      sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og; sv[4] = cn;
      c_carry = cn;
    }
    _Float16 hh, hl;
    split_h(hn, hh, hl);
    // element (row bl_, k = u) of the next step's operand: fragment (u >> 4, lane = ((u >> 3) & 1) * 32 + bl_), e = u & 7
    if (PERSIST) {
      // every (row < Bp, unit < 8 NS) is written, zeros where there is no batch row / hidden unit: the consumers wait
      // for every word.  Two units (lanes ju, ju + 1) share one 4-byte store.
      const unsigned ph = __builtin_bit_cast(unsigned short, hh), pl = __builtin_bit_cast(unsigned short, hl);
      const unsigned wh = ph | ((unsigned)__shfl_down((int)ph, 1) << 16), wl = pl | ((unsigned)__shfl_down((int)pl, 1) << 16);
      const int sbase = __builtin_amdgcn_readfirstlane((s + 1) * slot_bytes + (d * nbz + bz) * nkb * 1024);
      const int ho = (ju & 1) ? OOB : (((u >> 4) * 64 + ((u >> 3) & 1) * 32 + bl_) * 8 + (u & 7)) * 2;
      __builtin_amdgcn_raw_buffer_store_b32(wh, rh, ho, sbase, AUX_SC1);
      __builtin_amdgcn_raw_buffer_store_b32(wl, rl, ho, sbase, AUX_SC1);
      if (j == a.NS - 1 && UPW * a.NS < ldk) {                   // the k padding behind the last slice (H % 16 in 1..8)
        const int hz = (ju & 1) ? OOB : (((nkb - 1) * 64 + 32 + bl_) * 8 + (u & 7)) * 2;
        __builtin_amdgcn_raw_buffer_store_b32(0u, rh, hz, sbase, AUX_SC1);
        __builtin_amdgcn_raw_buffer_store_b32(0u, rl, hz, sbase, AUX_SC1);
      }
    } else if (live) {
      const int ho = ((((d * 2 + ((s + 1) & 1)) * nbz + bz) * nkb + (u >> 4)) * 64 + ((u >> 3) & 1) * 32 + bl_) * 8 + (u & 7);
      a.hs_h[ho] = hh;
      a.hs_l[ho] = hl;
    }
    if (live) {
      Gp[0] = sv[0]; Gp[H] = sv[1]; Gp[2 * H] = sv[2]; Gp[3 * H] = sv[3];
      a.c[prow * 2 * H + (long long)d * H + u] = sv[4];
      a.y[prow * 2 * H + (long long)d * H + u] = hn;
    }
  }
}

// ---- backward -----------------------------------------------------------------------------------------
// PERSIST as in the forward kernel: one launch for all steps, the W_hh^T fragments stay in registers
// and the carried cell gradient in a register of the owning thread (dcbuf unused).
// NSC: capacity for the producer slices one thread sums (NSC - 24 < NS <= NSC by the host's size class, except in the
// smallest); NT: 32-column output tiles per wave.  Missing slices / tiles are out-of-range buffer offsets.
template <bool PERSIST, int NSC, int NT>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const LstmArgs a, const int s0) {
  // [batch][k' (32) + pad]; two copies in the persistent kernel (a fast wave may be one step ahead)
  __shared__ __attribute__((aligned(16))) _Float16 sAh2[PERSIST ? 2 : 1][32][40], sAl2[PERSIST ? 2 : 1][32][40];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x, d = blockIdx.y, bz = blockIdx.z, bb = bz * 32;
  const int H = a.H, Hp = a.Hp, NS = a.NS, nbz = a.Bp >> 5;
  const int bl_ = tid >> 3, ju = tid & 7;
  const int b = bb + bl_, u = UPW * j + ju;
  const bool live = b < a.B && u < H;

  // this wave's W_hh^T fragments (tiles wave, wave + 4, ...) do not depend on anything computed
  // here: fetch them first so their latency hides behind the partial sums and the gate arithmetic
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  const int ntile = Hp >> 5;
  f16x8 wbh[NT][2], wbl[NT][2];
  {
    const long long wo = ((long long)d * NS + j) * Hp * 32;
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Wth + wo), 0, Hp * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Wtl + wo), 0, Hp * 64, 0x00020000);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int tile = wave + 4 * i;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int vo = tile < ntile ? (tile * 2 + kb) * 1024 + lane * 16 : OOB;
        wbh[i][kb] = load_frag(rwh, vo, 0, false);
        wbl[i][kb] = load_frag(rwl, vo, 0, false);
      }
    }
  }
  const float gsc = a.gscale[0];
  const float inv = 1.f / gsc;
  // partial recurrent gradients [dir][ping-pong][batch block] x [consumer slice][producer slice][32][8]
  const long long slot_floats = (long long)NS * 32 * Hp;
  const float* Pg = a.P + ((long long)d * 2 * nbz + bz) * slot_floats;    // slot 0 of this group; slot 1 is nbz slots further
  // consumer side: this slice's NS blocks of 1 KiB, nothing behind them (loads past NS return zeros)
  const __amdgpu_buffer_rsrc_t rc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Pg + (long long)j * NS * 256), 0, NS * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Pg + nbz * slot_floats + (long long)j * NS * 256), 0, NS * 1024, 0x00020000);
  // producer side: the whole slot
  const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Pg), 0, (int)(slot_floats * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Pg + nbz * slot_floats), 0, (int)(slot_floats * 4), 0x00020000);
  const int bc = b < a.B ? b : a.B - 1, uc = u < H ? u : H - 1;
  const int len = a.lens ? a.lens[bc] : a.T;
  float dc_carry = 0.f;
  const int s_end = PERSIST ? a.T : s0 + 1;
  for (int s = s0; s < s_end; ++s) {
    const int t = d == 0 ? a.T - 1 - s : s;
    // all loads of the gate stage, unconditionally and with clamped addresses; masks are applied afterwards
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
    float pv[NSC];
    {
      // consumer-major layout: the [32 x 8] blocks of all producer slices for this slice's 8 units are
      // contiguous (1 KiB each), so every load of the workgroup is one fully used, coalesced request.
      // The partials of step s - 1 are in slot (s - 1) & 1.  Persistent kernel: they carry the tag ((s - 1) >> 1) & 1 in
      // their lowest bit; whatever else is in the slot (step s - 3, or the host's fill before steps 0 / 1) the other one.
      const bool odd = (s + 1) & 1;
      const unsigned tag = (unsigned)((s - 1) >> 1) & 1u;
      unsigned spins = 0;
      unsigned long long spin_t0 = 0;
      if (PERSIST && s > 0 && a.probe) {                        // (as in the forward kernel: one word of the last producer's block)
        const __amdgpu_buffer_rsrc_t rc = odd ? rc1 : rc0;
        for (;;) {
          const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rc, (NS - 1) * 1024, 0, AUX_SC1);
          asm volatile("" ::: "memory");
          if ((((unsigned)__builtin_amdgcn_readfirstlane((int)w)) ^ tag) & 1u) {
            spin_wait(spins, spin_t0);
            continue;
          }
          break;
        }
      }
      for (;;) {
        const __amdgpu_buffer_rsrc_t rc = odd ? rc1 : rc0;
#pragma unroll
        for (int jj = 0; jj < NSC; ++jj) pv[jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, tid * 4 + jj * 1024, 0, PERSIST ? AUX_SC1 : 0));
        if (!PERSIST || s == 0) break;                           // (step 0 does not use them)
        unsigned pending = 0;
        int ns2 = NS;
        asm volatile("" : "+s"(ns2));                            // (the conditions are formed per pass, not kept in registers)
#pragma unroll
        for (int jj = 0; jj < NSC; ++jj) {
          const unsigned x = __builtin_bit_cast(unsigned, pv[jj]) ^ tag;
          pending |= (jj < NSC - 24 || jj < ns2) ? x : 0u;       // (blocks past NS read as zeros: tag 0 whatever is expected)
        }
        asm volatile("" ::: "memory");                           // the loads are repeated, not hoisted
        if (!__builtin_amdgcn_ballot_w64((pending & 1u) != 0)) break;
        spin_wait(spins, spin_t0);
      }
      if (PERSIST) {
#pragma unroll
        for (int jj = 0; jj < NSC; ++jj) pv[jj] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, pv[jj]) & ~1u);
      }
    }
#pragma unroll
    for (int n = NSC; n > 1; n = (n + 1) / 2)                     // pairwise tree
#pragma unroll
      for (int jj = 0; jj < n / 2; ++jj) pv[jj] += pv[jj + (n + 1) / 2];

    float dG[4] = {0.f, 0.f, 0.f, 0.f};
    {
      float dc_prev = 0.f;
      const float dh = dy_ld + (s > 0 ? pv[0] : 0.f);            // step 0: the partial buffer is uninitialised
      float dc = s > 0 ? dc_ld : 0.f;
      const float c_prev = (tp >= 0 && tp < a.T) ? c_prev_ld : 0.f;
      const float tc = tanh_f(cn);
      dc += dh * og * (1.f - tc * tc);
      if (live && t < len) {
        dG[0] = dc * gg * ig * (1.f - ig);
        dG[1] = dc * c_prev * fg * (1.f - fg);
        dG[2] = dc * ig * (1.f - gg * gg);
        dG[3] = dh * tc * og * (1.f - og);
        dc_prev = dc * fg;
      }
      if (live) {
        Gp[0] = dG[0]; Gp[H] = dG[1]; Gp[2 * H] = dG[2]; Gp[3 * H] = dG[3];
        if (!PERSIST) a.dcbuf[((long long)d * a.Bp + b) * H + u] = dc_prev;
      }
      dc_carry = dc_prev;
    }
    _Float16 (*sAh)[40] = sAh2[PERSIST ? (s & 1) : 0], (*sAl)[40] = sAl2[PERSIST ? (s & 1) : 0];
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
    const unsigned otag = (unsigned)(s >> 1) & 1u;
    const bool oodd = s & 1;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int tile = wave + 4 * i;
      const bool tok = tile < ntile;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if (PERSIST) {
        // W^T as the row operand: lane (batch row b = lane & 31, half) holds units 8 q + 4 half + 0..3 of the tile in
        // acc[4 q ..]: one 16-byte store per consumer slice q, 1 KiB contiguous per wave and store, tagged
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh[i][kb], al[kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbl[i][kb], ah[kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh[i][kb], ah[kb], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__builtin_bit_cast(unsigned, acc[4 * q + e] * inv) & ~1u) | otag;
          const int po = tok ? ((tile * 4 + q) * NS + j) * 1024 + (lane & 31) * 32 + (lane >> 5) * 16 : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(o, oodd ? rp1 : rp0, po, 0, AUX_SC1);
        }
      } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb], wbh[i][kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], wbl[i][kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], wbh[i][kb], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {                           // P[consumer slice][producer slice j][row][unit & 7]
          const int po = tok ? (((tile * 4 + ((lane & 31) >> 3)) * NS + j) * 256 + ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 8 + (lane & 7)) * 4 : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[e] * inv), oodd ? rp1 : rp0, po, 0, 0);
        }
      }
    }
  }
}

// One launch for all time steps (the workgroups exchange h / the partial gradients through memory, see above) when every
// workgroup of the grid can be resident at once: at most one per CU slot this process sizes its grids for (the device's
// CUs, or RADMMM_GEMM_CUS when a data-parallel run leaves CUs to RCCL's channel kernels, rowgemm_h3w.hip).  Otherwise, or
// with RADMMM_LSTM_PERSISTENT=0 (a supported switch, read per call), one launch per step.  The launch itself is
// cooperative (launch_coop): the runtime refuses a grid that cannot be co-resident, and the caller falls back.
}  // namespace
namespace radmmm { int gemm_cu_slots(); }
namespace {
bool use_persistent(const dim3& grid) {
  const char* e = getenv("RADMMM_LSTM_PERSISTENT");
  if (e && atoi(e) == 0) return false;
  return (long long)grid.x * grid.y * grid.z <= radmmm::gemm_cu_slots();
}
template <class K>
hipError_t launch_coop(K kernel, const dim3& grid, hipStream_t st, LstmArgs a, int s) {
  void* args[] = {&a, &s};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), grid, dim3(256), args, 0, st);
  if (e != hipSuccess) (void)hipGetLastError();           // (cleared: the caller falls back to one launch per step)
  return e;
}

// size classes: H <= 192 / 384 / 576 / 768 -> (k blocks per wave, producer-slice capacity, output tiles per wave)
template <bool PERSIST, class K>
hipError_t launch_one(K kernel, const LstmArgs& a, const dim3& grid, hipStream_t st, int s) {
  if constexpr (PERSIST) return launch_coop(kernel, grid, st, a, s);
  hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, a, s);
  return hipSuccess;
}
template <bool PERSIST>
hipError_t launch_fwd(const LstmArgs& a, const dim3& grid, hipStream_t st, int s) {
  if (a.H <= 192) return launch_one<PERSIST>(lstm_fwd_kernel<PERSIST, 3>, a, grid, st, s);
  if (a.H <= 384) return launch_one<PERSIST>(lstm_fwd_kernel<PERSIST, 6>, a, grid, st, s);
  if (a.H <= 576) return launch_one<PERSIST>(lstm_fwd_kernel<PERSIST, 9>, a, grid, st, s);
  return launch_one<PERSIST>(lstm_fwd_kernel<PERSIST, 12>, a, grid, st, s);
}
template <bool PERSIST>
hipError_t launch_bwd(const LstmArgs& a, const dim3& grid, hipStream_t st, int s) {
  if (a.H <= 192) return launch_one<PERSIST>(lstm_bwd_kernel<PERSIST, 24, 2>, a, grid, st, s);
  if (a.H <= 384) return launch_one<PERSIST>(lstm_bwd_kernel<PERSIST, 48, 3>, a, grid, st, s);
  if (a.H <= 576) return launch_one<PERSIST>(lstm_bwd_kernel<PERSIST, 72, 5>, a, grid, st, s);
  return launch_one<PERSIST>(lstm_bwd_kernel<PERSIST, 96, 6>, a, grid, st, s);
}


// LstmArgs.probe per direction of the pass: bit 0 forward, bit 1 backward.  Measured (tools/lstm_bench.py, B = 32, T' = 400,
// H = 524): the forward recurrence gains 5 % (2.20 -> 2.09 ms incl. projection), the backward one loses 2 %: default 1.
// RADMMM_DEBUG: RADMMM_LSTM_PROBE=0..3 overrides.
int lstm_probe(int bit) {
  const char* e = radmmm::debug_env("RADMMM_LSTM_PROBE");
  return ((e ? atoi(e) : 1) >> bit) & 1;
}

struct Dims {
  int64_t ldk, Bp, Hp, NS;
  Dims(int B, int H) : ldk((H + 15) / 16 * 16), Bp((B + 31) / 32 * 32), Hp((H + 31) / 32 * 32), NS((H + UPW - 1) / UPW) {}
};

}  // namespace

extern "C" int64_t radmmm_lstm_scratch_bytes(int B, int H, int which) {
  // which 0: split W_hh (hi + lo), 1: h operand ping-pong (hi + lo), 2: packed transposed slices (hi + lo),
  // 3: partial recurrent gradients P, 4: carried cell gradient
  const int64_t ldk = (H + 15) / 16 * 16, Bp = (B + 31) / 32 * 32, Hp = (H + 31) / 32 * 32, NS = (H + UPW - 1) / UPW;
  switch (which) {
    case 0: return 2 * (2 * NS * (ldk / 16) * 512 * 2);
    case 1: return 2 * (2 * 2 * Bp * ldk * 2);
    case 2: return 2 * (2 * NS * Hp * 32 * 2);
    case 3: return 2 * 2 * (Bp / 32) * NS * 32 * Hp * 4;
    case 4: return 2 * Bp * (int64_t)H * 4;
    default: return 0;
  }
}

// Bytes of the per-step operand slots of the single-launch forward recurrence ([T + 1][2][Bp/32][ldk/16] KiB, hi + lo), or
// 0 when these dimensions take the launch-per-step path (grid larger than the device, offsets beyond 2 GiB, switched off).
extern "C" int64_t radmmm_lstm_hseq_bytes(int B, int T, int H) {
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  const Dims q(B, H);
  const int64_t one = (int64_t)(T + 1) * 2 * q.Bp * q.ldk * 2;
  if (one >= (1LL << 31) || !use_persistent(dim3((unsigned)q.NS, 2, (unsigned)(q.Bp / 32)))) return 0;
  return 2 * one;
}

// Forward recurrence of both directions.  G [B*T][8H] holds x W_ih^T + b_ih + b_hh (direction d in
// columns d*4H .., gate order i, f, g, o) and is overwritten by the gate activations; W_hh [2][4H][H];
// y, c [B*T][2H] outputs; wsplit / hsplit scratch per radmmm_lstm_scratch_bytes(.., 0 / 1); hseq: scratch of
// radmmm_lstm_hseq_bytes(B, T, H) bytes, or null (-> one launch per step).
extern "C" int radmmm_lstm_fwd(float* G, const float* W_hh, float* y, float* c, const int32_t* lens, void* wsplit,
                               void* hsplit, void* hseq, int B, int T, int H, radmmm_stream_t stream) {
  RADMMM_REQUIRE(G && W_hh && y && c && wsplit && hsplit, "lstm_fwd: null pointer");
  RADMMM_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 4 * MAXKB * 16, "lstm_fwd: bad dims (H <= %d)", 4 * MAXKB * 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  LstmArgs a = {};
  a.G = G; a.y = y; a.c = c; a.lens = lens; a.B = B; a.T = T; a.H = H;
  a.ldk = (H + 15) / 16 * 16; a.Bp = (B + 31) / 32 * 32; a.Hp = (H + 31) / 32 * 32; a.NS = (H + UPW - 1) / UPW;
  a.probe = lstm_probe(0);
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
  const int64_t hq = radmmm_lstm_hseq_bytes(B, T, H);
  if (hseq && hq > 0) {
    // slot 0 (h = 0 before the first step) zeros, every other slot the fill pattern the consumers wait on
    const size_t slot = (size_t)2 * a.Bp * a.ldk * 2;
    char* q = static_cast<char*>(hseq);
    a.hq_h = reinterpret_cast<_Float16*>(q);
    a.hq_l = reinterpret_cast<_Float16*>(q + hq / 2);
    if (hipMemsetAsync(q, 0xff, (size_t)hq, st) != hipSuccess || hipMemsetAsync(q, 0, slot, st) != hipSuccess ||
        hipMemsetAsync(q + hq / 2, 0, slot, st) != hipSuccess) {
      radmmm::set_error("lstm_fwd: hipMemsetAsync failed");
      return -2;
    }
    if (launch_fwd<true>(a, grid, st, 0) == hipSuccess) return radmmm::check_launch("lstm_fwd");
    // the cooperative launch was refused (the grid cannot be co-resident right now): per-step launches on the ping-pong
    // operand buffers (hsplit, zeroed above)
  }
  for (int s = 0; s < T; ++s) launch_fwd<false>(a, grid, st, s);
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
  a.P = P; a.dcbuf = dcbuf; a.gscale = gscale; a.probe = lstm_probe(1);
  const long long tn = 2LL * a.NS * a.Hp * 32;
  _Float16* Wth = static_cast<_Float16*>(wtpack);
  _Float16* Wtl = Wth + tn;
  a.Wth = Wth; a.Wtl = Wtl;
  hipLaunchKernelGGL(lstm_pack_wt_kernel, dim3((unsigned)((tn + 255) / 256)), dim3(256), 0, st, W_hh, Wth, Wtl, H, a.Hp, a.NS);
  const dim3 grid(a.NS, 2, a.Bp / 32);
  if (use_persistent(grid)) {
    // both slots of the partials start with the tag that steps 0 and 1 do not write (see the exchange notes above)
    if (hipMemsetAsync(P, 0xff, (size_t)radmmm_lstm_scratch_bytes(B, H, 3), st) != hipSuccess) {
      radmmm::set_error("lstm_bwd: hipMemsetAsync failed");
      return -2;
    }
    if (launch_bwd<true>(a, grid, st, 0) == hipSuccess) return radmmm::check_launch("lstm_bwd");
  }
  for (int s = 0; s < T; ++s) launch_bwd<false>(a, grid, st, s);
  return radmmm::check_launch("lstm_bwd");
}
