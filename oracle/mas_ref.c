/* CPU oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the reference's monotonic
 * alignment search, alignment.py:31-59 (`mas_width1`).  Integer/index path: must reproduce the
 * reference bit for bit given the same float32 log inputs.  Built by oracle/Makefile into
 * oracle/libmas_ref.so and loaded by oracle/radmmm_oracle.py (mas_width1_c); never linked into
 * the product.
 *
 *   logp [T1][T2] float32 = log(attn) (caller's log, as np.log in the reference)
 *   opt  [T1][T2] float32 receives the 0/1 alignment
 *   prev_ind scratch [T1][T2] int64
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int mas_width1_ref(const float* logp_in, int T1, int T2, float* opt) {
  if (T1 <= 0 || T2 <= 0) return -1;
  float* lp = (float*)malloc(sizeof(float) * (size_t)T1 * T2);
  float* acc = (float*)calloc((size_t)T1 * T2, sizeof(float));
  int64_t* prev = (int64_t*)calloc((size_t)T1 * T2, sizeof(int64_t));
  if (!lp || !acc || !prev) { free(lp); free(acc); free(prev); return -2; }
  memcpy(lp, logp_in, sizeof(float) * (size_t)T1 * T2);
  memset(opt, 0, sizeof(float) * (size_t)T1 * T2);
  for (int j = 1; j < T2; ++j) lp[j] = -INFINITY;          /* attn_map[0, 1:] = -inf        (:37) */
  for (int j = 0; j < T2; ++j) acc[j] = lp[j];              /* log_p[0, :] = attn_map[0, :]  (:39) */
  for (int i = 1; i < T1; ++i) {
    for (int j = 0; j < T2; ++j) {
      float prev_log = acc[(size_t)(i - 1) * T2 + j];
      int64_t prev_j = j;
      if (j - 1 >= 0 && acc[(size_t)(i - 1) * T2 + j - 1] >= acc[(size_t)(i - 1) * T2 + j]) {   /* (:46) */
        prev_log = acc[(size_t)(i - 1) * T2 + j - 1];
        prev_j = j - 1;
      }
      acc[(size_t)i * T2 + j] = lp[(size_t)i * T2 + j] + prev_log;
      prev[(size_t)i * T2 + j] = prev_j;
    }
  }
  int64_t cur = T2 - 1;                                      /* backtrack (:53-58) */
  for (int i = T1 - 1; i >= 0; --i) {
    opt[(size_t)i * T2 + cur] = 1.0f;
    cur = prev[(size_t)i * T2 + cur];
  }
  opt[cur] = 1.0f;
  free(lp); free(acc); free(prev);
  return 0;
}
