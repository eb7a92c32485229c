// rowgemm_h3w: host side of the wide-tile split conv GEMM (kernel: rowgemm_h3w_kernel.h): tile height, epilogue kind and
// product scheme of a launch.
#include <stdlib.h>

#include "common.h"

namespace radmmm {
// rowgemm_h3w_pr{1,2,3}.hip: all (tile height, epilogue kind) instantiations of one product scheme
int launch_h3d_pr1(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes);
int launch_h3d_pr2(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes);
int launch_h3d_pr3(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes);
// rowgemm_win.hip: 5-tap convs with the A rows of a k slice fetched once for all taps (shared window), FP8-cross scheme
bool rowgemm_win_ok(int mb, int ek, const radmmm_rowgemm_h3_desc& d);
int launch_rowgemm_win(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes);
// rowgemm_one.hip: 1-tap convs (no extra K segment) on a three-stage A ring with wave-private B tiles, FP8-cross scheme
bool rowgemm_one_ok(int mb, int ek, const radmmm_rowgemm_h3_desc& d);
int launch_rowgemm_one(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes);
}  // namespace radmmm

namespace {
constexpr int BN = 256;
enum { EK_GENERIC = 0, EK_PLAIN, EK_SPLIT, EK_RES, EK_DGRAD };     // as in rowgemm_h3w_kernel.h
}  // namespace

namespace radmmm {

// Row-block count per workgroup.  Cost model fitted to measurements at M = 12 800 and 32 000
// (DESIGN.md §4.2): a workgroup costs (MB + 1.5) units (MFMA work ~ MB, the 256-row B tile and
// the fixed parts ~ 1.5); the grid runs `full` rounds of all CUs plus a tail round which, because the
// chip is power bound, runs faster when few CUs are busy: 0.65 + 0.35 * (tail workgroups / CUs).
int pick_h3w_mb(int M, int N, int slots) {
  const int ntn = (N + BN - 1) / BN;
  int best = 4;
  double best_cost = 1e300;
  // MB = 8 (256-row tiles, 20-36 bytes of scratch per lane) costs what this model says it should on the shape it was
  // measured on (12 800 x 1024: 405 us against 380 us at MB = 7, model 8.77 : 8.22) and is taken where it saves a round
  // of workgroups: N = 1152 (the start conv's data gradient) and N = 1052 (the LSTM's input gradient) fit 50 x 5 = 250
  // workgroups into ONE round instead of 290-500 in two.  RADMMM_H3W_MB8=0: candidates 4..7 only (A/B runs).
  static const int mb_max = (debug_env("RADMMM_H3W_MB8") && atoi(debug_env("RADMMM_H3W_MB8")) == 0) ? 7 : 8;
  for (int mb = 4; mb <= mb_max; ++mb) {
    const long long wg = (long long)((M + 32 * mb - 1) / (32 * mb)) * ntn;
    const long long full = wg / slots, tail = wg % slots;
    const double rounds = (double)full + (tail ? 0.65 + 0.35 * (double)tail / slots : 0.0);
    const double cost = rounds * (mb + 1.5);
    if (cost <= best_cost + 1e-9) {
      best_cost = cost;
      best = mb;
    }
  }
  return best;
}

// Workgroup slots a GEMM grid is sized for: the device's CUs, or RADMMM_GEMM_CUS when set (data-parallel runs
// leave a few CUs to RCCL's channel kernels so that an all-reduce landing during a GEMM launch does not push the
// grid into a second round: rad_mmm_amd/ddp.py reserve_collective_cus).  Read once per process.
int gemm_cu_slots() {
  static const int slots = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    if (const char* e = getenv("RADMMM_GEMM_CUS")) {
      const int v = atoi(e);
      if (v >= 32 && v <= n) n = v;
    }
    return n;
  }();
  return slots;
}

static int launch_rowgemm_h3w_inner(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes, int* mb_out,
                                    int* ek_out);

extern "C" int radmmm_colsum_final(const float* part, float* out, int nparts, int cols, radmmm_stream_t stream);
extern "C" int radmmm_colsum(const float* X, int ldx, float* out, float* scratch, int rows, int cols, int row_weight, int T,
                             const int32_t* lens, int taps, int dil, int square, radmmm_stream_t stream);

// the launch + (optional) the column sums of its pre-row-scale values: from the direct epilogue's per-tile partial rows, or,
// when the launch took the generic epilogue, by radmmm_colsum over C (weights 1 / ratio: the same sum)
int launch_rowgemm_h3w(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  int mb = 0, ek = 0;
  const int rc = launch_rowgemm_h3w_inner(d, stream, a_bytes, b_bytes, &mb, &ek);
  const radmmm_rowgemm_desc& p = d.base;
  if (rc || !p.colsum_out) return rc;
  if (ek != EK_GENERIC) return radmmm_colsum_final(p.colsum_scratch, p.colsum_out, (p.M + 32 * mb - 1) / (32 * mb), p.N, stream);
  if (!p.C) {
    set_error("rowgemm_h3: colsum_out with the generic epilogue sums C afterwards: C must not be NULL");
    return -1;
  }
  return radmmm_colsum(p.C, p.ldc, p.colsum_out, p.colsum_scratch, p.M, p.N, p.rowscale == 2 ? 2 : (p.rowscale == 1 ? 1 : 0), p.T,
                       p.lens, p.ratio_taps, p.ratio_dil, 0, stream);
}

// tile height and epilogue kind of a launch
static void decide_h3w(const radmmm_rowgemm_h3_desc& d, int* mb_out, int* ek_out) {
  int mb = pick_h3w_mb(d.base.M, d.base.N, gemm_cu_slots());
  if (const char* e = debug_env("RADMMM_H3W_MB")) {
    const int v = atoi(e);
    if (v >= 4 && v <= 8) mb = v;
  }
  // Epilogue kind.  The launches of the flow step (and everything else that fits: even N, no `add` input, operands that
  // 32-bit byte offsets can address) take the DIRECT epilogue specialised for the arrays they write (EK_*,
  // rowgemm_h3w_kernel.h); the rest keeps the fully general LDS-parking epilogue.
  // RADMMM_DEBUG_EPILOGUE=generic forces the latter (A/B runs, tests).
  const radmmm_rowgemm_desc& p = d.base;
  auto a8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
  auto a4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; };
  auto fits = [&](long long ld, int esz) { return ((long long)p.M + 64) * ld * esz < 0x7fffffffLL; };
  static const bool force_generic = [] {
    const char* e = debug_env("RADMMM_DEBUG_EPILOGUE");
    return e && e[0] == 'g';
  }();
  int ek = EK_GENERIC;
  // (the pair form of the dact input is decoded by the FP8-cross scheme's direct kernels and by the generic epilogue)
  const bool ok = !force_generic && p.N % 2 == 0 && !p.add && p.ldc % 2 == 0 && a8(p.C) && fits(p.ldc, 4) &&
                  (!p.dact_h || d.nprod == 2) &&
                  (!p.dact || (p.dact_h ? fits(p.lddact_h, 2) : (p.lddact % 2 == 0 && a8(p.dact_src) && fits(p.lddact, 4)))) &&
                  (!p.C2 || (p.ldc2 % 2 == 0 && a8(p.C2) && fits(p.ldc2, 4))) &&
                  (p.n_c2_src <= 0 || (p.ldc2 % 2 == 0 && fits(p.ldc2, 4) && a8(p.c2_src[0]) && a8(p.c2_src[p.n_c2_src > 1 ? 1 : 0]) &&
                                       a8(p.c2_src[p.n_c2_src > 2 ? 2 : 0]))) &&
                  (!p.Ch || (p.ldch % 2 == 0 && a4(p.Ch) && a4(p.Cl) && a4(p.Clo) && fits(p.ldch, 2))) &&
                  (!p.C2h || (p.ldc2h % 2 == 0 && a4(p.C2h) && a4(p.C2l) && fits(p.ldc2h, 2)));
  // (a direct kernel writes its split copies in its own scheme's format: 8-bit cross arrays under nprod 2, fp16 pairs else)
  const bool fmt_ok = (!p.Ch && !p.C2h) || (d.nprod == 2 ? (p.split_fmt == RADMMM_SPLIT_X8A || p.split_fmt == RADMMM_SPLIT_X8B)
                                                          : p.split_fmt == RADMMM_SPLIT_F16);
  if (ok && fmt_ok) {
    if (p.dact) {
      if (!p.C2 && p.Ch && !p.C2h && p.act == RADMMM_ACT_NONE) ek = EK_DGRAD;
    } else if (p.C2 || p.n_c2_src > 0) {
      if (!p.Ch) ek = EK_RES;
    } else if (!p.C2h) {
      ek = p.Ch ? EK_SPLIT : EK_PLAIN;
    }
  }
  *mb_out = mb;
  *ek_out = ek;
}

// rows of per-tile partial column sums a wide-kernel launch of `d` leaves in colsum_scratch (0: generic epilogue, none)
int h3w_colsum_rows(const radmmm_rowgemm_h3_desc& d) {
  int mb = 0, ek = 0;
  decide_h3w(d, &mb, &ek);
  return ek == EK_GENERIC ? 0 : (d.base.M + 32 * mb - 1) / (32 * mb);
}

static int launch_rowgemm_h3w_inner(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes, int* mb_out,
                                    int* ek_out) {
  int mb = 0, ek = 0;
  decide_h3w(d, &mb, &ek);
  *mb_out = mb;
  *ek_out = ek;
  if (d.nprod == 2) {                                                                 // FP8 cross terms
    const char* we = debug_env("RADMMM_WIN");         // RADMMM_DEBUG: RADMMM_WIN=0 keeps the per-tap A tiles (A/B runs, tests)
    const char* wx = debug_env("RADMMM_WIN_XT");      // RADMMM_DEBUG: 0 = launches with the extra K segment keep rowgemm_h3d (A/B runs)
    const bool xt_ok = !d.extra_tap || !(wx && atoi(wx) == 0);
    if (!(we && atoi(we) == 0) && xt_ok && rowgemm_win_ok(mb, ek, d)) return launch_rowgemm_win(mb, ek, d, stream, a_bytes, b_bytes);
    const char* oe = debug_env("RADMMM_ONE");         // RADMMM_DEBUG: RADMMM_ONE=0 keeps rowgemm_h3d for the 1-tap launches (A/B runs, tests)
    if (!(oe && atoi(oe) == 0) && rowgemm_one_ok(mb, ek, d)) return launch_rowgemm_one(mb, ek, d, stream, a_bytes, b_bytes);
    return launch_h3d_pr2(mb, ek, d, stream, a_bytes, b_bytes);
  }
  if (d.nprod == 1) return launch_h3d_pr1(mb, ek, d, stream, a_bytes, b_bytes);       // 16-bit throughput mode
  {                                                                                   // three f16 products
    const char* s3 = debug_env("RADMMM_SLOT3");       // RADMMM_DEBUG: RADMMM_SLOT3=0 keeps rowgemm_h3d for every three-product launch (A/B, tests)
    if (!(s3 && atoi(s3) == 0)) {
      if (rowgemm_win_ok(mb, ek, d)) return launch_rowgemm_win(mb, ek, d, stream, a_bytes, b_bytes);
      if (rowgemm_one_ok(mb, ek, d)) return launch_rowgemm_one(mb, ek, d, stream, a_bytes, b_bytes);
    }
  }
  return launch_h3d_pr3(mb, ek, d, stream, a_bytes, b_bytes);
}

}  // namespace radmmm

extern "C" int radmmm_gemm_cu_slots(void) { return radmmm::gemm_cu_slots(); }
