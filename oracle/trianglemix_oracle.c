/*
 * CPU oracle for the TriangleMix hot path, plain C.  TEST INFRASTRUCTURE ONLY:
 * linked or loaded by tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py, never by the product library.
 *
 * It restates the reference's Python algorithm literally: the O(N^2) double
 * loops of the mask generator and the dense matmul -> +mask -> softmax ->
 * matmul of the eager attention.  Parity is pinned through
 * tests/test_oracle_golden.py, which checks these functions against fixtures
 * generated from the reference's own files (oracle/gen_golden.py).
 *
 * Reference lines restated (relative to the reference checkout):
 *   vllm/attention/trianglemix_attention.py:82-89   streaming mask loop
 *   vllm/attention/trianglemix_attention.py:126-129 last mask loop
 *   vllm/attention/trianglemix_attention.py:166-169 middle mask loop
 *   vllm/attention/trianglemix_attention.py:192-196 triangle = clamp(s + l)
 *   vllm/attention/trianglemix_attention.py:254-258 0 / -inf additive form
 *   vllm/attention/ascend_trianglemix.py:88-106     eager attention
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* trianglemix_attention.py:82-89 */
void tmx_oracle_streaming_mask(float* out, int n, int si, int sl) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      float v = 0.f;
      if (i >= j && j <= si) v = 1.f;
      else if (i >= j && (i - j) <= sl) v = 1.f;
      out[(size_t)i * n + j] = v;
    }
}

/* trianglemix_attention.py:126-129 */
void tmx_oracle_last_mask(float* out, int n, int si, int sl, int last) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      out[(size_t)i * n + j] = (i >= j && (n - i) < last && j > si && (i - j) > sl) ? 1.f : 0.f;
}

/* trianglemix_attention.py:166-169 */
void tmx_oracle_middle_mask(float* out, int n, int si, int sl, int last) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      out[(size_t)i * n + j] = (i >= j && (n - i) >= last && j > si && (i - j) > sl) ? 1.f : 0.f;
}

/* trianglemix_attention.py:173-198: two mask passes, add, clamp to [0, 1]. */
int tmx_oracle_triangle_mask(float* out, int n, int si, int sl, int last) {
  float* tmp = (float*)malloc((size_t)n * n * sizeof(float));
  if (!tmp) return -1;
  tmx_oracle_streaming_mask(out, n, si, sl);
  tmx_oracle_last_mask(tmp, n, si, sl, last);
  for (size_t e = 0; e < (size_t)n * n; ++e) {
    float t = out[e] + tmp[e];
    out[e] = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
  }
  free(tmp);
  return 0;
}

/* trianglemix_attention.py:218-219 */
void tmx_oracle_dense_mask(float* out, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) out[(size_t)i * n + j] = (j <= i) ? 1.f : 0.f;
}

/* trianglemix_attention.py:254-258, in place: >0 -> 0, else -inf. */
void tmx_oracle_to_additive(float* m, size_t count) {
  for (size_t e = 0; e < count; ++e) m[e] = (m[e] > 0.f) ? 0.f : -INFINITY;
}

/*
 * ascend_trianglemix.py:88-106 on [B, N, H, D] float32 tensors; mask is an
 * additive [N, N] matrix or NULL (NULL = no mask at all, bidirectional).
 * One (batch, head) at a time, scores row by row: the same arithmetic as the
 * dense formulation without holding the [N, N] score matrix.
 * Returns 0, or -1 when the row buffer cannot be allocated.
 */
int tmx_oracle_eager_attention(const float* q, const float* k, const float* v, const float* mask,
                               float* out, int B, int N, int H, int D, float scale) {
  int rc = 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h) {
      float* s = (float*)malloc((size_t)N * sizeof(float));
      if (!s) { rc = -1; continue; }
      for (int i = 0; i < N; ++i) {
        const float* qi = q + (((size_t)b * N + i) * H + h) * D;
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
          const float* kj = k + (((size_t)b * N + j) * H + h) * D;
          float acc = 0.f;
          for (int d = 0; d < D; ++d) acc += qi[d] * kj[d];
          acc *= scale;
          if (mask) acc += mask[(size_t)i * N + j];
          s[j] = acc;
          if (acc > mx) mx = acc;
        }
        float den = 0.f;
        for (int j = 0; j < N; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
        float* oi = out + (((size_t)b * N + i) * H + h) * D;
        for (int d = 0; d < D; ++d) oi[d] = 0.f;
        for (int j = 0; j < N; ++j) {
          float p = s[j] / den;
          const float* vj = v + (((size_t)b * N + j) * H + h) * D;
          for (int d = 0; d < D; ++d) oi[d] += p * vj[d];
        }
      }
      free(s);
    }
  return rc;
}

/*
 * Row-range forms of the same loops, for timing a bounded sample of a large
 * problem: rows row0, row0 + row_step, ... (nrows of them) of the length-n
 * problem.  Identical arithmetic per row to the full-size functions above.
 */
void tmx_oracle_triangle_additive_rows(float* out, int n, int si, int sl, int last, int row0, int row_step,
                                       int nrows) {
  for (int r = 0; r < nrows; ++r) {
    const int i = row0 + r * row_step;
    float* o = out + (size_t)r * n;
    /* streaming pass, last pass, add + clamp, then 0 / -inf: trianglemix_attention.py:82-89,126-129,192-196,254-258 */
    for (int j = 0; j < n; ++j) {
      float v = 0.f;
      if (i >= j && j <= si) v = 1.f;
      else if (i >= j && (i - j) <= sl) v = 1.f;
      o[j] = v;
    }
    for (int j = 0; j < n; ++j) {
      float l = (i >= j && (n - i) < last && j > si && (i - j) > sl) ? 1.f : 0.f;
      float t = o[j] + l;
      t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
      o[j] = (t > 0.f) ? 0.f : -INFINITY;
    }
  }
}

/* q_rows: [nrows, H, D] (the sampled query rows), k / v: [n, H, D], mask_rows: [nrows, n] additive or NULL. */
int tmx_oracle_eager_attention_rows(const float* q_rows, const float* k, const float* v, const float* mask_rows,
                                    float* out_rows, int nrows, int n, int H, int D, float scale) {
  int rc = 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int h = 0; h < H; ++h)
    for (int r = 0; r < nrows; ++r) {
      float* s = (float*)malloc((size_t)n * sizeof(float));
      if (!s) { rc = -1; continue; }
      const float* qi = q_rows + ((size_t)r * H + h) * D;
      float mx = -INFINITY;
      for (int j = 0; j < n; ++j) {
        const float* kj = k + ((size_t)j * H + h) * D;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += qi[d] * kj[d];
        acc *= scale;
        if (mask_rows) acc += mask_rows[(size_t)r * n + j];
        s[j] = acc;
        if (acc > mx) mx = acc;
      }
      float den = 0.f;
      for (int j = 0; j < n; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
      float* oi = out_rows + ((size_t)r * H + h) * D;
      for (int d = 0; d < D; ++d) oi[d] = 0.f;
      for (int j = 0; j < n; ++j) {
        float p = s[j] / den;
        const float* vj = v + ((size_t)j * H + h) * D;
        for (int d = 0; d < D; ++d) oi[d] += p * vj[d];
      }
      free(s);
    }
  return rc;
}

/* Closed form used to cross-check the loops above (not a reference line). */
int tmx_oracle_allowed(int i, int j, int n, int si, int sl, int last) {
  return j <= i && (j <= si || (i - j) <= sl || (n - i) < last);
}
